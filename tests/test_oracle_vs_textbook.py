"""Differential tests: the C restatement (oracle/) against algorithm-independent textbook DPs
(tests/textbook.py).  Every kernel in the reference is exact, so for ANY input
  distance == textbook value, and with a cutoff: Some(d) iff d <= cutoff (src/common.rs:43-45).
Replaces the reference's crash-only fuzz targets (fuzz/fuzz_targets/*.rs) with value oracles.
"""
import numpy as np
import pytest

import textbook as tb
from oracle import oracle as o

RNG = np.random.default_rng(0xC0FFEE)


def _rand(n, alphabet):
    return bytes(RNG.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n).tolist())


def _mutate(s, edits, alphabet):
    s = bytearray(s)
    for _ in range(edits):
        op = RNG.integers(0, 3)
        pos = int(RNG.integers(0, len(s) + 1))
        ch = int(RNG.choice(np.frombuffer(alphabet, dtype=np.uint8)))
        if op == 0 and len(s):
            s[min(pos, len(s) - 1)] = ch
        elif op == 1:
            s.insert(pos, ch)
        elif len(s):
            del s[min(pos, len(s) - 1)]
    return bytes(s)


def _pairs(n, max_len, alphabet=b"abc"):
    out = []
    for i in range(n):
        l1 = int(RNG.integers(0, max_len + 1))
        a = _rand(l1, alphabet)
        if i % 2:
            b = _mutate(a, int(RNG.integers(0, 8)), alphabet)
        else:
            b = _rand(int(RNG.integers(0, max_len + 1)), alphabet)
        out.append((a, b))
    return out


SHORT = _pairs(300, 20) + _pairs(200, 70, b"ab") + _pairs(100, 64, b"abcdefghijklmnopqrstuvwxyz0123456789")
LONG = _pairs(60, 300, b"abcd") + _pairs(20, 700, b"ab")


@pytest.mark.parametrize("pairs", [SHORT, LONG], ids=["short", "long"])
def test_levenshtein_exact_and_cutoff(pairs):
    for a, b in pairs:
        d = tb.levenshtein_unit(a, b)
        bc = o.levenshtein.BatchComparator(a)
        assert o.levenshtein.distance(a, b) == d
        assert bc.distance(b) == d
        for k in sorted({0, 1, 2, 3, 4, 5, max(d - 1, 0), d, d + 1, 31, 32, 2 * d}):
            exp = d if d <= k else None
            assert o.levenshtein.distance(a, b, score_cutoff=k) == exp, (a, b, k)
            assert bc.distance(b, score_cutoff=k) == exp, (a, b, k, o.last_lev_path())
        for h in (0, 1, d, 100):
            # results never depend on the hint (levenshtein.rs:2153) -- except in the upstream corner
            # pinned by test_reference_quirk_small_hint_vs_length_difference below
            if len(a) <= 64 or abs(len(a) - len(b)) <= max(h, 31):
                assert bc.distance(b, score_hint=h) == d
        m = max(len(a), len(b))
        assert bc.similarity(b) == m - d
        if m:
            assert abs(bc.normalized_distance(b) - d / m) < 1e-15


def test_levenshtein_weighted():
    for a, b in SHORT[:200]:
        for w in [(1, 1, 2), (2, 2, 2), (1, 2, 3), (3, 1, 1), (2, 2, 5), (0, 0, 1)]:
            d = tb.levenshtein(a, b, w)
            assert o.levenshtein.distance(a, b, weights=w) == d, (a, b, w)
            assert o.levenshtein.BatchComparator(a).distance(b, weights=w) == d, (a, b, w)


@pytest.mark.parametrize("pairs", [SHORT, LONG], ids=["short", "long"])
def test_lcs_and_indel(pairs):
    for a, b in pairs:
        l = tb.lcs_len(a, b)
        ind = len(a) + len(b) - 2 * l
        lb, ib = o.lcs_seq.BatchComparator(a), o.indel.BatchComparator(a)
        assert o.lcs_seq.similarity(a, b) == l and lb.similarity(b) == l
        assert o.indel.distance(a, b) == ind and ib.distance(b) == ind
        assert lb.distance(b) == max(len(a), len(b)) - l
        for k in sorted({0, 1, 2, l - 1 if l else 0, l, l + 1}):
            exp = l if l >= k else None
            assert o.lcs_seq.similarity(a, b, score_cutoff=k) == exp
            assert lb.similarity(b, score_cutoff=k) == exp
        for k in sorted({0, 1, 2, 3, 4, 5, max(ind - 1, 0), ind, ind + 1}):
            exp = ind if ind <= k else None
            assert o.indel.distance(a, b, score_cutoff=k) == exp
            assert ib.distance(b, score_cutoff=k) == exp
        # levenshtein weights (1,1,>=2) is Indel (levenshtein.rs:1321-1327)
        assert o.levenshtein.BatchComparator(a).distance(b, weights=(1, 1, 2)) == ind


@pytest.mark.parametrize("pairs", [SHORT, LONG[:40]], ids=["short", "long"])
def test_jaro_and_winkler(pairs):
    for a, b in pairs:
        j, jw = tb.jaro(a, b), tb.jaro_winkler(a, b)
        jb, wb = o.jaro.BatchComparator(a), o.jaro_winkler.BatchComparator(a)
        assert abs(o.jaro.similarity(a, b) - j) < 1e-12, (a, b)
        assert abs(jb.similarity(b) - j) < 1e-12, (a, b)
        assert abs(o.jaro_winkler.similarity(a, b) - jw) < 1e-12
        assert abs(wb.similarity(b) - jw) < 1e-12
        for c in (0.0, 0.3, 0.7, 0.71, 0.9, 1.0):
            got = jb.similarity(b, score_cutoff=c)
            assert (got is not None) == (jb.similarity(b) >= c)
            got = wb.similarity(b, score_cutoff=c)
            assert (got is not None) == (wb.similarity(b) >= c), (a, b, c)


def test_reference_quirk_small_hint_vs_length_difference():
    """Quirk Q7 (found by this differential test, not documented upstream): the hint-doubling loop at
    levenshtein.rs:1069-1088 calls hyrroe2003_small_band_with_pm(score_hint) WITHOUT the
    `score_cutoff < |len1-len2|` guard that hyrroe2003_block has (:786); for len1 > 64 and an explicit
    score_hint with 2*max(hint,31) < len1 - len2 the band's break_score (:535-536) goes negative, wraps,
    and the function returns its start value -- so the reference's result depends on the hint there.
    The oracle restates that faithfully; the GPU path ignores hints and returns the exact distance."""
    a = b"ab" * 35
    b = b"baa"
    d = tb.levenshtein_unit(a, b)
    assert d == 67
    bc = o.levenshtein.BatchComparator(a)
    assert bc.distance(b) == d  # no hint: exact
    assert bc.distance(b, score_hint=100) == d
    assert bc.distance(b, score_hint=0) != d  # upstream bug reproduced (value 31..34)


def test_reference_quirk_levenshtein_similarity_cutoff_sentinel():
    """Quirk Q2: levenshtein `similarity_with_args` above the cutoff evaluates
    `maximum - usize::MAX` (details/distance.rs:209-210): a panic in debug builds; the oracle mirrors a
    release build's wrap (maximum + 1, which `score()` then keeps).  The GPU path returns None."""
    a, b = b"a" * 10, b"b" * 10  # distance 10, similarity 0
    assert o.levenshtein.BatchComparator(a).similarity(b) == 0
    assert o.levenshtein.BatchComparator(a).similarity(b, score_cutoff=0) == 0
    # cutoff 2 -> distance cutoff 8 -> hyrroe2003 returns usize::MAX -> 10 - MAX wraps to 11 >= 2 -> Some(11)
    assert o.levenshtein.BatchComparator(a).similarity(b, score_cutoff=2) == 11  # a sentinel, not a similarity
    # below 4 the mbleven path returns cutoff + 1 instead of usize::MAX and the value stays in range
    assert o.levenshtein.BatchComparator(b"aaaa").similarity(b"bbbb", score_cutoff=2) is None


def test_reference_quirk_lcs_band_leaves_out_a_block():
    """Quirk Q8 (found by tests/test_gpu_parity.py::test_randomized_differential, seed 9149; not documented upstream): lcs_blockwise
    (lcs_seq.rs:297-331) walks an Ukkonen band of blocks [first_block, last_block) and moves its right edge with
    `last_block = ceil_div(row + 1 + band_width_left, 64)` (:321-323).  The next row needs bit row + 1 + band_width_left, i.e. block
    (row + 1 + band_width_left) / 64 + 1 -- the two differ when that index is a multiple of 64, and for one row the block the band
    has just reached is left out.  A pair whose alignment runs along the band's edge at such a row loses one match: the reference
    then reports a similarity below the true LCS, or None.  Only queries of more than 64 symbols under a cutoff tight enough for
    `full_band_words < words` (:362-366) and loose enough for max_misses >= 5 (:469) get there.  The oracle restates the loop
    faithfully (this test); the device returns the exact value (tests/test_gpu_known_answers.py)."""
    import json
    import os

    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "q8_lcs_band_pair.json")))
    a, b = fx["s1"].encode(), fx["s2"].encode()
    l = tb.lcs_len(a, b)
    assert (len(a), len(b), l) == (300, 299, fx["lcs"])
    lb, ib = o.lcs_seq.BatchComparator(a), o.indel.BatchComparator(a)
    assert lb.similarity(b) == l and lb.distance(b) == 3 and ib.distance(b) == 5  # no cutoff: exact
    assert lb.distance(b, score_cutoff=3) is None      # the upstream defect, reproduced: the true distance is 3
    assert lb.similarity(b, score_cutoff=297) is None  # likewise
    assert o.last_lcs_q8_edges() > 0  # (instrumentation: the call above met the defect's precondition -- rfo_last_lcs_q8_edges ...)
    assert lb.distance(b, score_cutoff=4) == 3 and lb.similarity(b, score_cutoff=296) == 297  # a wider band is exact again
    # ... and a call that never walks a band of blocks reports none (70 x 66 symbols, LCS cutoff 60: full_band_words == words -> lcs_unroll)
    c, d = bytes(range(48, 118)), bytes(range(48, 114))
    assert o.lcs_seq.BatchComparator(c).similarity(d, score_cutoff=60) == 66 and o.last_lcs_q8_edges() == 0


def test_osa_exact_and_cutoff():
    rng = np.random.default_rng(77)
    pairs = SHORT[:250] + LONG[:12]
    # adjacent transpositions are what separates OSA from Levenshtein: make sure they occur
    for a, _ in SHORT[250:330]:
        b = bytearray(a)
        for _ in range(3):
            if len(b) > 1:
                i = int(rng.integers(0, len(b) - 1))
                b[i], b[i + 1] = b[i + 1], b[i]
        pairs.append((a, bytes(b)))
    seen_less = False
    for a, b in pairs:
        d = tb.osa(a, b)
        seen_less |= d < tb.levenshtein_unit(a, b)
        bc = o.osa.BatchComparator(a)
        assert o.osa.distance(a, b) == d and bc.distance(b) == d, (a, b)
        m = max(len(a), len(b))
        assert bc.similarity(b) == m - d
        for k in sorted({0, 1, max(d - 1, 0), d, d + 1}):
            exp = d if d <= k else None
            assert o.osa.distance(a, b, score_cutoff=k) == exp and bc.distance(b, score_cutoff=k) == exp
    assert seen_less
