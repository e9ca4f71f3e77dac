// rf_scratch.hip -- the library's own stream-ordered scratch allocator (round 4).
//
// Why not hipMallocAsync / hipFreeAsync: on this stack (ROCm 7.2, gfx950) the default memory pool hands a block that one host thread
// freed on ITS stream to another host thread's allocation on another stream while the first stream's kernels still use it --
// tools/mempool_repro.hip: 8 host threads, one stream each, fill / verify / free: tens of millions of corrupted words per run, with
// the three reuse policies of the pool switched off as well.  The library's per-call temporaries (top-k result lists, translated
// images of u32 corpora, slot-ordered result vectors, selection scratch) came from that pool, so two host threads sharing a corpus
// on two streams could read each other's scratch (tests/cpp/stress_threads.cpp found it).  A single-threaded caller never saw it.
//
// What this is: a caching allocator with the one rule that makes stream-ordered reuse safe.  scratch_free(p, st) records an event on
// `st` behind everything enqueued so far and parks the block; scratch_alloc(bytes, st) may take a parked block
//   * at once if it was parked on the SAME stream handle -- behind a hipStreamWaitEvent(st, done) that costs the host nothing: on the
//     stream the block was parked on it is a no-op (stream order already puts the new use behind the old one), and on a NEW stream
//     that got a destroyed stream's handle value it is what orders the two (round 5), or
//   * if its event has completed (hipEventQuery), whatever the stream;
// otherwise it calls hipMalloc.  The host never waits here.  Parked bytes are bounded (RF_SCRATCH_CACHE_MB, default 1024): beyond the
// bound the completed blocks are released, largest first; blocks larger than the bound are released as soon as their event is done.
// The pool is per process and device, never destroyed (the HIP runtime may be gone before static destructors run).
#include "rf_host.hpp"

namespace {

struct Block {
    void* ptr = nullptr;
    size_t bytes = 0;
    int device = 0;
    hipEvent_t done = nullptr;   // recorded by scratch_free behind the block's last use
    hipStream_t stream = nullptr;  // the stream it was parked on
    bool pending = false;        // `done` may not have completed yet
};

struct Pool {
    std::mutex mu;
    std::map<void*, Block> in_use;
    std::vector<Block> parked;
    size_t parked_bytes = 0;
    size_t cap = [] {
        const char* e = getenv("RF_SCRATCH_CACHE_MB");
        return (size_t)(e ? std::max(0ll, atoll(e)) : 1024ll) << 20;
    }();
};

Pool& pool()
{
    static Pool* p = new Pool();
    return *p;
}

bool completed(Block& b)
{
    if (!b.pending) return true;
    const hipError_t e = hipEventQuery(b.done);
    if (e == hipSuccess) {
        b.pending = false;
        return true;
    }
    if (e != hipErrorNotReady) (void)hipGetLastError();
    return false;
}

// parked blocks that may go: completed ones, while the parked bytes exceed `target` (0 = every completed block).  Returns them; the
// caller releases them OUTSIDE the lock (hipFree synchronizes the device).
std::vector<Block> collect(Pool& P, size_t target)
{
    std::vector<Block> out;
    if (P.parked_bytes <= target) return out;
    std::sort(P.parked.begin(), P.parked.end(), [](const Block& a, const Block& b) { return a.bytes > b.bytes; });
    for (size_t i = 0; i < P.parked.size() && P.parked_bytes > target;) {
        if (completed(P.parked[i])) {
            out.push_back(P.parked[i]);
            P.parked_bytes -= P.parked[i].bytes;
            P.parked.erase(P.parked.begin() + (long)i);
        } else {
            ++i;
        }
    }
    return out;
}

void release(std::vector<Block>& blocks)
{
    for (Block& b : blocks) {
        DeviceGuard g(b.device);
        (void)hipEventDestroy(b.done);
        (void)hipFree(b.ptr);
    }
    blocks.clear();
}

}  // namespace

extern "C" {

hipError_t scratch_alloc(void** out, size_t bytes, hipStream_t st)
{
    *out = nullptr;
    bytes = std::max<size_t>(256, (bytes + 255) / 256 * 256);
    int device = 0;
    if (const hipError_t e = hipGetDevice(&device); e != hipSuccess) return e;
    Pool& P = pool();
    std::vector<Block> drop;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        // best fit among the blocks this stream may take now; a block more than twice the request (+ 1 MiB) stays for a larger one
        long best = -1;
        for (size_t i = 0; i < P.parked.size(); ++i) {
            Block& b = P.parked[i];
            if (b.device != device || b.bytes < bytes || b.bytes > 2 * bytes + (1u << 20)) continue;
            if (b.stream != st && !completed(b)) continue;
            if (best < 0 || b.bytes < P.parked[(size_t)best].bytes) best = (long)i;
        }
        if (best >= 0 && !completed(P.parked[(size_t)best]) && hipStreamWaitEvent(st, P.parked[(size_t)best].done, 0) != hipSuccess) {
            (void)hipGetLastError();  // cannot order this stream behind the block's last use: leave it parked
            best = -1;
        }
        if (best >= 0) {
            Block b = P.parked[(size_t)best];
            P.parked.erase(P.parked.begin() + best);
            P.parked_bytes -= b.bytes;
            P.in_use[b.ptr] = b;
            *out = b.ptr;
            return hipSuccess;
        }
    }
    Block b;
    b.bytes = bytes;
    b.device = device;
    hipError_t e = hipMalloc(&b.ptr, bytes);
    if (e == hipErrorOutOfMemory) {  // give back what is parked and done, then try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(P.mu);
            drop = collect(P, 0);
        }
        release(drop);
        e = hipMalloc(&b.ptr, bytes);
    }
    if (e != hipSuccess) return e;
    e = hipEventCreateWithFlags(&b.done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipFree(b.ptr);
        return e;
    }
    {
        std::lock_guard<std::mutex> lock(P.mu);
        P.in_use[b.ptr] = b;
    }
    *out = b.ptr;
    return hipSuccess;
}

void scratch_free(void* p, hipStream_t st)
{
    if (!p) return;
    Pool& P = pool();
    std::vector<Block> drop;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.in_use.find(p);
        if (it == P.in_use.end()) return;  // (not ours: nothing sensible to do)
        Block b = it->second;
        P.in_use.erase(it);
        DeviceGuard g(b.device);
        b.stream = st;
        b.pending = hipEventRecord(b.done, st) == hipSuccess;
        if (!b.pending) {  // the stream is gone or broken: fall back to the one safe answer
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
        P.parked.push_back(b);
        P.parked_bytes += b.bytes;
        drop = collect(P, P.cap);
    }
    release(drop);
}

void scratch_trim(void)
{
    Pool& P = pool();
    std::vector<Block> drop;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        drop = collect(P, 0);
    }
    release(drop);
}

}  // extern "C"
