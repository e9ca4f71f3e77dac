// rf_hint.hip -- the second half of a score_hint scan of a long query (rf_api_scan.hip run_many_hinted).
//
// reference: levenshtein.rs:1069-1088 -- for len1 > 64 the reference runs hyrroe2003_block under the band `max(score_hint, 31)` and doubles
// the band until the distance fits; results never depend on the hint (levenshtein.rs:2153-2160).  On the device one candidate is one
// lane, so "run it again with a wider band" would make every wavefront wait for its slowest lane.  Instead:
//   pass 1   the whole corpus under the cutoff k1 = max(hint, 31): the band kernel (k1 <= 31, one word sliding down the diagonal) or
//            the banded multi-word scans -- exact wherever the distance is <= k1, None elsewhere;
//   mark     hint_mark_kernel: per tile, the lanes pass 1 left unresolved (None, and not ruled out by the caller's own cutoff through
//            |len1 - len2| alone) as a 64-bit mask + their number; an exclusive sum over the tiles (hipcub) numbers them in slot order,
//            and the sums at the boundaries of the corpus' LENGTH RUNS (a run = the consecutive tiles of one length) go to the host,
//            which sizes the dense tiles: ceil(unresolved / 64) per run;
//   gather   hint_gather_kernel: dense tile i of a run takes the run's unresolved candidates 64 i .. 64 i + 63 (binary search in the
//            sums, then the n-th set bit of that tile's mask) and copies their payload, with a tile descriptor and an orig[] that
//            carries the ORIGINAL candidate index -- neighbours in a dense tile are neighbours in the corpus, so pass 2's stores
//            through orig[] stay in a few cache lines;
//   pass 2   the ordinary scan (any kernel: it is a general corpus of exact tiles) over the dense tiles under the caller's own
//            cutoff, storing through orig[] straight into the caller's result vector.
// The cost of a candidate within the hint is pass 1's; the others pay pass 1 + 32 bytes of list and payload traffic per 16 symbols +
// the full scan they would have paid anyway.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "rf_internal.hpp"

namespace rf {

namespace {

// largest i in [lo, hi) with a[i] <= x (a ascending, a[lo] <= x)
__device__ __forceinline__ uint32_t last_le(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t x)
{
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (a[mid] <= x)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// one wavefront per tile (grid-stride): which lanes did pass 1 leave unresolved?  mask[t] = those lanes, count[t] = how many.  No atomics:
// the first version appended to one list per length run behind an atomic counter, and a single-length corpus is ONE run -- 156 000
// wavefronts queueing on one address took as long as the scan (1.8 ms per 10 M candidates).
__global__ __launch_bounds__(kWave* kWavesPerBlock) void hint_mark_kernel(const TileDesc* __restrict__ tiles, const uint32_t* __restrict__ orig,
                                                                          uint32_t* __restrict__ out, uint32_t n, uint32_t n_tiles, uint32_t uniform_len, uint32_t len1,
                                                                          uint32_t raw_cutoff, uint32_t zero_value, uint64_t* __restrict__ mask,
                                                                          uint32_t* __restrict__ count)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    for (uint32_t t = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; t <= n_tiles; t += stride) {
        uint64_t m = 0;
        if (t < n_tiles) {
            const uint32_t len2 = tiles ? tiles[t].len : uniform_len;
            const uint32_t slot = (tiles ? tiles[t].slot0 : t * kWave) + lane;
            const uint32_t idx = orig ? orig[slot] : slot;
            const bool valid = orig ? idx != kPad : idx < n;
            const uint32_t gap = len1 > len2 ? len1 - len2 : len2 - len1;
            // (a gap beyond the caller's own cutoff rules the whole tile out: it stays None)
            const bool open = gap <= raw_cutoff && valid && out[idx] == RF_NONE_U32;
            if (len2 == 0) {  // nothing to scan: the distance is len1 (times the common weight factor; None beyond the caller's cutoff)
                if (open) out[idx] = zero_value;
            } else {
                m = __ballot(open);
            }
        }
        if (lane == 0) {
            if (t < n_tiles) mask[t] = m;
            count[t] = (uint32_t)__popcll(m);  // (count[n_tiles] = 0: the exclusive sum's last entry is the total)
        }
    }
}

// The sample taken BEFORE pass 1 (run_many_hinted): one wavefront per sampled tile -- tiles tile_begin, tile_begin + step, ... below tile_end, all inside pass 1's
// length window -- after pass 1's kernel has run over exactly those tiles.  acc[0] += real candidates seen, acc[1] += those of them the pass resolved.
__global__ __launch_bounds__(kWave* kWavesPerBlock) void hint_sample_kernel(const TileDesc* __restrict__ tiles, const uint32_t* __restrict__ orig,
                                                                            const uint32_t* __restrict__ out, uint32_t n, uint32_t uniform_len, uint32_t tile_begin,
                                                                            uint32_t tile_end, uint32_t step, uint32_t* __restrict__ acc)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t t64 = (uint64_t)tile_begin + (uint64_t)(blockIdx.x * kWavesPerBlock + threadIdx.x / kWave) * step;
    if (t64 >= tile_end) return;
    const uint32_t t = (uint32_t)t64;
    const uint32_t len2 = tiles ? tiles[t].len : uniform_len;
    if (len2 == 0) return;
    const uint32_t slot = (tiles ? tiles[t].slot0 : t * kWave) + lane;
    const uint32_t idx = orig ? orig[slot] : slot;
    const bool valid = orig ? idx != kPad : idx < n;
    const uint64_t seen = __ballot(valid);
    const uint64_t resolved = __ballot(valid && out[idx] != RF_NONE_U32);
    if (lane == 0) {
        atomicAdd(&acc[0], (uint32_t)__popcll(seen));
        atomicAdd(&acc[1], (uint32_t)__popcll(resolved));
    }
}

__global__ void hint_run_prefix_kernel(const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ run_first, uint32_t R, uint32_t* __restrict__ run_prefix)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= R) run_prefix[r] = prefix[run_first[r]];
}

// one wavefront per DENSE tile j: unresolved candidates 64 i .. 64 i + 63 of run r, in slot order (the last tile of a run may be partial)
__global__ __launch_bounds__(kWave* kWavesPerBlock) void hint_gather_kernel(const uint8_t* __restrict__ data, const TileDesc* __restrict__ tiles,
                                                                            const uint32_t* __restrict__ orig, uint32_t uniform_tile_bytes,
                                                                            const uint32_t* __restrict__ run_first, uint32_t R, const uint32_t* __restrict__ run_prefix,
                                                                            const uint32_t* __restrict__ prefix, const uint64_t* __restrict__ mask,
                                                                            const uint32_t* __restrict__ run_tile_base, const uint64_t* __restrict__ run_data_base,
                                                                            const uint32_t* __restrict__ run_len, uint32_t n_tiles2, uint8_t* __restrict__ data2,
                                                                            TileDesc* __restrict__ tiles2, uint32_t* __restrict__ orig2)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t stride = gridDim.x * kWavesPerBlock;
    for (uint32_t j = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; j < n_tiles2; j += stride) {
        const uint32_t r = last_le(run_tile_base, 0, R, j);  // (a run without unresolved candidates owns no dense tile and shares its base with the next: never the LAST such r)
        const uint32_t i = j - run_tile_base[r], len2 = run_len[r];
        const uint32_t have = min((uint32_t)kWave, run_prefix[r + 1] - run_prefix[r] - i * kWave);
        const uint32_t nch = (len2 + kChunk - 1) / kChunk;
        const uint64_t dst_off = run_data_base[r] + (uint64_t)i * nch * kWave * kChunk;
        // padding lanes copy the tile's first candidate (defined bytes, no result: orig2 = kPad)
        const uint32_t g = run_prefix[r] + i * kWave + (lane < have ? lane : 0);
        const uint32_t ts = last_le(prefix, run_first[r], run_first[r + 1], g);
        uint64_t m = mask[ts];
        for (uint32_t skip = g - prefix[ts]; skip; --skip) m &= m - 1;  // the (g - prefix[ts])-th unresolved lane of that tile
        const uint32_t ls = (uint32_t)__ffsll((long long)m) - 1u;
        const uint64_t src_off = tiles ? tiles[ts].data_off : (uint64_t)ts * uniform_tile_bytes;
        const uint32_t src_slot = (tiles ? tiles[ts].slot0 : ts * kWave) + ls;
        orig2[(size_t)j * kWave + lane] = lane < have ? (orig ? orig[src_slot] : src_slot) : kPad;
        if (lane == 0) {
            TileDesc d;
            d.data_off = dst_off;
            d.len = len2;
            d.slot0 = j * kWave;
            tiles2[j] = d;
        }
        const uint4* src = reinterpret_cast<const uint4*>(data + src_off) + ls;
        uint4* dst = reinterpret_cast<uint4*>(data2 + dst_off) + lane;
        for (uint32_t k = 0; k < nch; ++k) dst[(size_t)k * kWave] = src[(size_t)k * kWave];
    }
}

}  // namespace

hipError_t launch_hint_sample(const ScanParams& p, const uint32_t* out, uint32_t tile_begin, uint32_t tile_end, uint32_t step, uint32_t* acc, hipStream_t st)
{
    const uint32_t count = tile_end > tile_begin ? (tile_end - tile_begin + step - 1) / step : 0;
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(hint_sample_kernel, dim3((count + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kWave * kWavesPerBlock), 0, st, p.tiles, p.orig, out, p.n, p.uniform_len,
                       tile_begin, tile_end, step, acc);
    return hipGetLastError();
}

size_t hint_scan_temp_bytes(uint32_t n_tiles)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n_tiles + 1, nullptr);
    return std::max<size_t>(bytes, 256);
}

// mask / count per tile, prefix = exclusive sum of count over n_tiles + 1 entries, run_prefix[r] = prefix[run_first[r]] for r = 0 .. R
hipError_t launch_hint_mark(const ScanParams& p, uint32_t* out, uint32_t raw_cutoff, uint32_t zero_value, uint64_t* mask, uint32_t* count, uint32_t* prefix, void* temp,
                            size_t temp_bytes, const uint32_t* run_first, uint32_t R, uint32_t* run_prefix, hipStream_t st)
{
    const uint32_t units = p.n_tiles + 1;
    const uint32_t grid = (uint32_t)std::max(1, std::min(scan_grid(units), (int)((units + kWavesPerBlock - 1) / kWavesPerBlock)));
    hipLaunchKernelGGL(hint_mark_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0, st, p.tiles, p.orig, out, p.n, p.n_tiles, p.uniform_len, p.len1, raw_cutoff, zero_value,
                       mask, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, count, prefix, (int)units, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hint_run_prefix_kernel, dim3((R + 1 + 255) / 256), dim3(256), 0, st, prefix, run_first, R, run_prefix);
    return hipGetLastError();
}

hipError_t launch_hint_gather(const ScanParams& p, const uint32_t* run_first, uint32_t R, const uint32_t* run_prefix, const uint32_t* prefix, const uint64_t* mask,
                              const uint32_t* run_tile_base, const uint64_t* run_data_base, const uint32_t* run_len, uint32_t n_tiles2, uint8_t* data2, TileDesc* tiles2,
                              uint32_t* orig2, hipStream_t st)
{
    const uint32_t grid = (uint32_t)std::max(1, std::min(scan_grid(n_tiles2), (int)((n_tiles2 + kWavesPerBlock - 1) / kWavesPerBlock)));
    hipLaunchKernelGGL(hint_gather_kernel, dim3(grid), dim3(kWave * kWavesPerBlock), 0, st, p.data, p.tiles, p.orig, p.uniform_tile_bytes, run_first, R, run_prefix, prefix,
                       mask, run_tile_base, run_data_base, run_len, n_tiles2, data2, tiles2, orig2);
    return hipGetLastError();
}

}  // namespace rf
