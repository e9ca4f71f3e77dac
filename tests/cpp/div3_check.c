/* tests/test_div3.py: the Jaro table epilogue divides by 3.0 with three fma-class instructions (rf_jaro.hip div3) instead of the IEEE
 * division sequence.  This program checks, with the host's divide as the reference, (1) every value the epilogue can feed it -- the sum
 * ((c / len1) + (c / len2)) + (c - h) / c in the reference's order (jaro.rs:106-119) for every len1, len2 < 130, every count of common
 * characters and half-transpositions -- and (2) 50 million pseudo-random doubles over the whole exponent range without subnormals. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static double div3(double x)
{
    const double z = 0x1.5555555555555p-2;
    const double q = x * z;
    const double r = fma(-3.0, q, x);
    return fma(r, z, q);
}
static int same(double a, double b) { return memcmp(&a, &b, sizeof a) == 0; }

int main(void)
{
    unsigned long long checked = 0, bad = 0;
    for (int len1 = 1; len1 < 130; ++len1)
        for (int len2 = 1; len2 < 130; ++len2) {
            const int cmax = len1 < len2 ? (len1 < 64 ? len1 : 64) : (len2 < 64 ? len2 : 64);
            for (int c = 1; c <= cmax; ++c)
                for (int h = 0; h <= c / 2; ++h) {
                    double acc = 0.0;
                    acc += (double)c / (double)len1;
                    acc += (double)c / (double)len2;
                    acc += ((double)c - (double)h) / (double)c;
                    ++checked;
                    if (!same(div3(acc), acc / 3.0)) ++bad;
                }
        }
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 50000000; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t bits = s & 0x7FFFFFFFFFFFFFFFull;
        const unsigned e = (unsigned)(bits >> 52);
        if (e < 2 || e > 2044) continue;  /* no subnormal input or quotient, no overflow in 3 q */
        double x;
        memcpy(&x, &bits, sizeof x);
        ++checked;
        if (!same(div3(x), x / 3.0)) ++bad;
    }
    printf("checked %llu mismatches %llu\n", checked, bad);
    return bad != 0;
}
