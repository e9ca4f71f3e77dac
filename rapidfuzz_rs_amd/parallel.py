"""Multi-GPU: one process per GPU, the corpus sharded by contiguous candidate ranges, NO collective on the
data path.  The only exchange is the one the north star names: every rank's k best entries under the total
order (score, global index) are all-gathered (RCCL over xGMI when the backend is "nccl") and merged by
rf_topk_merge_u32 on every rank.  k * 12 bytes per rank: latency-bound, link bandwidth is irrelevant.

The reference has no counterpart (single-threaded library, no top-k API); the oracle for this module is
"evaluate everything, stable-sort by (score, index), take k" (tests/test_parallel.py, tests/test_gpu_parity.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _native as N

_PAD_INDEX = np.uint64(0xFFFFFFFFFFFFFFFF)


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """rank r of R owns candidates [r*n/R, (r+1)*n/R) -- contiguous, balanced to within one candidate."""
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_ragged(offsets: np.ndarray, rank: int, world: int) -> np.ndarray:
    """The candidates of rank `rank` when a RAGGED corpus is sharded (every length bucket is dealt to all ranks, so that the
    ranks' work is balanced bucket by bucket): sort the candidates by length (stable: original order inside a bucket) and deal the
    sorted sequence round-robin, i.e. rank r takes every world-th candidate of every bucket and the buckets' remainders rotate over the
    ranks.  `shard_range` on a length-sorted input would hand rank 0 the short strings and rank R-1 the long ones, and a step takes as
    long as the slowest rank.  Returns the rank's ORIGINAL candidate indices, ascending (so (score, local index) order is (score, global
    index) order: `take_ragged` builds the shard, `sharded_topk(..., shard_index=idx)` reports global indices)."""
    lens = np.diff(np.asarray(offsets, dtype=np.uint64)).astype(np.int64)
    order = np.argsort(lens, kind="stable")
    return np.sort(order[rank::world]).astype(np.uint64)


def take_ragged(data: np.ndarray, offsets: np.ndarray, index: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(data, offsets) of the candidates `index` (e.g. shard_ragged's) of a ragged corpus, in that order."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    index = np.asarray(index, dtype=np.int64)
    lens = (offsets[index + 1] - offsets[index]).astype(np.int64)
    out_off = np.zeros(len(index) + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(lens)
    # source byte of every destination byte: start of its candidate + position inside it
    src = np.repeat(offsets[index].astype(np.int64) - out_off[:-1].astype(np.int64), lens) + np.arange(int(out_off[-1]), dtype=np.int64)
    return np.ascontiguousarray(np.asarray(data)[src]), out_off


def merge_topk(op: int, scores: np.ndarray, indices: np.ndarray, counts: np.ndarray, k: int):
    """Merge `len(counts)` lists (row-major [lists, k]) into the k best by (score, index)."""
    scores = np.ascontiguousarray(scores, dtype=np.uint32)
    indices = np.ascontiguousarray(indices, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    out_s, out_i = np.empty(k, dtype=np.uint32), np.empty(k, dtype=np.uint64)
    cnt = C.c_uint32()
    N.check(N.lib().rf_topk_merge_u32(op, scores.ctypes.data, indices.ctypes.data, counts.ctypes.data, len(counts), k,
                                      out_s.ctypes.data, out_i.ctypes.data, C.byref(cnt)))
    return out_s[: cnt.value], out_i[: cnt.value]


def allgather_topk(scores: np.ndarray, indices: np.ndarray, k: int, op: int = N.OP_DISTANCE, group=None, device=None):
    """All-gather every rank's local top-k (global indices!) and merge.  Works with any initialized torch.distributed backend.
    "nccl" (= RCCL): the lists travel as 16-byte rf_topk_entry rows in DEVICE memory -- one all_gather_into_tensor of k entries per
    rank and rf_topk_merge_entries_device on torch's current stream, one copy of the k merged entries back at the end (this function
    returns host arrays); "gloo": host tensors and the host merge."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    m = len(scores)
    scores = np.asarray(scores)
    if scores.dtype.kind == "f":
        #  both branches below carry u32 scores; an f64 score (Jaro / Jaro-Winkler / ratio, normalized_*) would be truncated, silently.  Those
        # lists travel as order-preserving 64-bit keys: sharded_topk_entries / rf_topk_entries_device.
        raise TypeError("allgather_topk merges u32 scores; f64-valued scorers exchange rf_topk_entry rows: use sharded_topk_entries")
    if dist.get_backend(group) == "nccl":
        # (the device the GROUP's collectives run on for this rank -- the caller's, else the one torch.distributed bound at init_process_group(device_id=...),
        # else the current one; never a guess that can differ from the shard's GPU)
        bound = getattr(dist.distributed_c10d._get_default_group() if group is None else group, "bound_device_id", None)
        dev = torch.device(device) if device is not None else (bound if bound is not None else torch.device("cuda", torch.cuda.current_device()))
        desc = op in (N.OP_SIMILARITY, N.OP_NORMALIZED_SIMILARITY)
        host = np.full((k, 2), -1, dtype=np.int64)  # (empty entry: key = index = UINT64_MAX)
        s64 = scores.astype(np.int64)
        host[:m, 0] = (0xFFFFFFFF - s64) if desc else s64
        host[:m, 1] = np.asarray(indices, dtype=np.uint64).view(np.int64)
        local = torch.from_numpy(host).to(dev)
        everyone = torch.empty((world * k, 2), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(everyone, local, group=group)
        merged = merge_entries_device(everyone, k, torch.empty((k, 2), dtype=torch.int64, device=dev)).cpu().numpy()
        keep = ~((merged[:, 0] == -1) & (merged[:, 1] == -1))
        key = merged[keep, 0]
        return ((0xFFFFFFFF - key) if desc else key).astype(np.uint32), merged[keep, 1].copy().view(np.uint64)
    payload = torch.zeros(2 * k + 1, dtype=torch.int64)
    payload[0] = m
    payload[1 : 1 + m] = torch.from_numpy(scores.astype(np.int64))
    payload[1 + k : 1 + k + m] = torch.from_numpy(np.asarray(indices, dtype=np.uint64).view(np.int64))
    if device is not None:
        payload = payload.to(device)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    g = torch.stack(gathered).cpu().numpy()
    counts = g[:, 0].astype(np.uint32)
    return merge_topk(op, g[:, 1 : 1 + k].astype(np.uint32), g[:, 1 + k : 1 + 2 * k].copy().view(np.uint64), counts, k)


def sharded_topk(scorer, shard_corpus, k: int, shard_start: int, op: int = N.OP_DISTANCE, args=None, out=None, group=None,
                 device=None, shard_index=None, **kw):
    """One rank's share of a distributed top-k: scan the local shard (global index = shard_start + local, or shard_index[local] for
    a shard that is not a contiguous range: shard_ragged), then the k-entry all-gather.  Every rank returns the same (scores, global
    indices).  score_hint=<expected k-th best distance> turns the scan into cutoff scans (see below); the result never depends on it."""
    if shard_index is not None:
        # an ascending index map keeps (score, local index) order = (score, global index) order: map after the local top-k
        shard_index = np.asarray(shard_index, dtype=np.uint64)
        inner = dict(kw)

        class _Mapped:  # the scorer with global indices on its top-k lists
            FLOAT = getattr(scorer, "FLOAT", False)
            _s1 = scorer._s1

            @staticmethod
            def topk(corpus, kk, *a, index_base=0, **k2):
                s_, i_ = scorer.topk(corpus, kk, *a, index_base=0, **k2)
                return s_, shard_index[np.asarray(i_, dtype=np.int64)]

        return sharded_topk(_Mapped, shard_corpus, k, 0, op, args, out, group, device, None, **inner)
    hint = kw.pop("score_hint", None)
    # (usize-valued metrics only: the rounds below are bounded by the query length, which means nothing for Jaro's f64 distances)
    if (hint is not None and op == N.OP_DISTANCE and args is None and out is None and kw.get("score_cutoff") is None
            and not getattr(scorer, "FLOAT", False)):
        kw.pop("score_cutoff", None)  # an explicit score_cutoff=None must not collide with the round's own cutoff below
        # score_hint across shards: every rank scans its shard under the cutoff `hint`; if the MERGED list holds k
        # entries they are the k best of the whole corpus, otherwise the hint doubles.  The merged list is the same on every rank, so
        # all ranks take the same branch; the bound on the rounds depends on the query alone for the same reason.
        hint, longest = int(hint), len(scorer._s1)
        while hint <= longest // 4:
            s, i = scorer.topk(shard_corpus, k, op, index_base=shard_start, score_cutoff=hint, **kw)
            ms, mi = allgather_topk(s, i, k, op, group=group, device=device)
            if len(ms) >= k:
                return ms, mi
            hint = max(1, 2 * hint)
    s, i = scorer.topk(shard_corpus, k, op, args, index_base=shard_start, out=out, **kw)
    return allgather_topk(s, i, k, op, group=group, device=device)


def allgather_filter(indices: np.ndarray, scores: np.ndarray, by_score: bool = False, descending: bool = False, group=None, device=None):
    """The exchange step of a sharded rf_filter: every rank's (global index, score) survivors, concatenated on every rank in the order one
    rf_filter call over the whole corpus gives -- ascending index (each global index lives on exactly one rank), or (score, index) with
    `by_score` (`descending` for the similarity ops).  Two collectives: the counts (8 bytes per rank), then the survivors padded to the
    longest list (16 bytes each): a thresholded scan's survivors are few by construction, so this is latency, like the top-k exchange."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    indices = np.ascontiguousarray(indices, dtype=np.uint64)
    is_f = np.asarray(scores).dtype.kind == "f"
    scores = np.ascontiguousarray(scores, dtype=np.float64 if is_f else np.uint32)
    assert len(indices) == len(scores)
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", device if device is not None else torch.cuda.current_device()) if nccl else torch.device("cpu")
    mine = torch.tensor([len(indices)], dtype=torch.int64, device=dev)
    counts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    if longest == 0:
        return indices[:0], scores[:0]
    payload = torch.zeros((longest, 2), dtype=torch.int64)
    payload[: len(indices), 0] = torch.from_numpy(indices.view(np.int64))
    payload[: len(indices), 1] = torch.from_numpy(scores.view(np.int64) if is_f else scores.astype(np.int64))  # (f64 scores travel as their bits)
    payload = payload.to(dev)
    everyone = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(everyone, payload, group=group)
    parts = [e[:c].cpu().numpy() for e, c in zip(everyone, counts)]
    g = np.concatenate(parts) if parts else np.zeros((0, 2), dtype=np.int64)
    gi = np.ascontiguousarray(g[:, 0]).view(np.uint64)
    gs = np.ascontiguousarray(g[:, 1]).view(np.float64) if is_f else g[:, 1].astype(np.uint32)
    if by_score:
        key = gs.astype(np.float64) if is_f else gs.astype(np.int64)
        order = np.lexsort((gi, -key if descending else key))
    else:
        order = np.argsort(gi, kind="stable")
    return gi[order], gs[order]


def sharded_filter(scorer, shard_corpus, op: int, shard_start: int = 0, args=None, shard_index=None, order: int = N.FILTER_BY_INDEX, group=None,
                   device=None, **kw):
    """One rank's share of a distributed thresholded scan (rf_filter_u32 / rf_filter_f64 per shard, then `allgather_filter`): every rank
    returns the same (global indices, scores) -- what `filter_many` over the unsharded corpus returns.  Global index = shard_start + local
    for a contiguous shard (`shard_range`), shard_index[local] for a dealt one (`shard_ragged`); FILTER_ANY comes back in index order."""
    if shard_index is not None:
        idx, sc = scorer.filter_many(op, shard_corpus, args, order=N.FILTER_ANY, index_base=0, **kw)
        idx = np.asarray(shard_index, dtype=np.uint64)[np.asarray(idx, dtype=np.int64)]
    else:
        idx, sc = scorer.filter_many(op, shard_corpus, args, order=N.FILTER_ANY, index_base=shard_start, **kw)
    desc = op in (N.OP_SIMILARITY, N.OP_NORMALIZED_SIMILARITY)
    if device is None and hasattr(shard_corpus, "device"):
        device = shard_corpus.device
    return allgather_filter(np.asarray(idx), np.asarray(sc), by_score=(order == N.FILTER_BY_SCORE), descending=desc, group=group, device=device)


def merge_keys_device(all_keys, k: int, out, stream=None):
    """Device-side merge of all-gathered rf_topk_keys_device lists (CUDA int64 tensors): the k smallest keys of
    `all_keys` into `out`, best first, -1 = empty; asynchronous on torch's current stream."""
    import torch

    assert all_keys.is_cuda and out.is_cuda and all_keys.dtype == torch.int64 and out.dtype == torch.int64 and out.numel() >= k
    st = stream if stream is not None else torch.cuda.current_stream(all_keys.device).cuda_stream
    N.check(N.lib().rf_topk_merge_keys_device(all_keys.data_ptr(), all_keys.numel(), k, out.data_ptr(), all_keys.device.index or 0, st))
    return out


# ---- 16-byte top-k entries (rf_topk_entry: order-preserving key, 64-bit global index): any metric, any k, any index space ----
def merge_entries_device(all_entries, k: int, out, stream=None):
    """Device-side merge of all-gathered topk_entries_device lists (CUDA int64 tensors [n, 2]) into the k best, best first,
    (-1, -1) = empty; asynchronous on torch's current stream."""
    import torch

    assert all_entries.is_cuda and out.is_cuda and all_entries.dtype == torch.int64 and out.dtype == torch.int64
    assert all_entries.is_contiguous() and out.is_contiguous() and out.numel() >= 2 * k
    st = stream if stream is not None else torch.cuda.current_stream(all_entries.device).cuda_stream
    N.check(N.lib().rf_topk_merge_entries_device(all_entries.data_ptr(), all_entries.numel() // 2, k, out.data_ptr(), all_entries.device.index or 0, st))
    return out


def merge_entries(entries: np.ndarray, k: int) -> np.ndarray:
    """Host merge of entries ([n, 2] uint64) into the k best (rf_topk_merge_entries): what a gloo exchange ends with."""
    e = np.ascontiguousarray(entries, dtype=np.uint64).reshape(-1, 2)
    out = np.empty((k, 2), dtype=np.uint64)
    N.check(N.lib().rf_topk_merge_entries(e.ctypes.data, len(e), k, out.ctypes.data))
    return out


def decode_entries(entries, op: int, is_float: bool):
    """[(score, global index)] of the non-empty entries of an [n, 2] array / tensor of rf_topk_entry."""
    e = entries.cpu().numpy() if hasattr(entries, "cpu") else np.asarray(entries)
    e = e.view(np.uint64).reshape(-1, 2)
    desc = 1 if op in (N.OP_SIMILARITY, N.OP_NORMALIZED_SIMILARITY) else 0
    L = N.lib()
    f = L.rf_topk_entry_score_f64 if is_float else L.rf_topk_entry_score_u32
    return [(f(int(key), desc), int(idx)) for key, idx in e if not (key == _PAD_INDEX and idx == _PAD_INDEX)]


def sharded_topk_entries(scorer, shard_corpus, k: int, shard_start: int, op: int, group=None, stream=None, **kw):
    """One rank's share of a distributed top-k over ANY metric: local entries (global indices), all-gather of k x 16 bytes per rank
    (RCCL when the backend is "nccl"; through host tensors for "gloo"), merge.  Every rank returns the same [k, 2] int64 tensor."""
    import torch
    import torch.distributed as dist

    import contextlib

    world = dist.get_world_size(group)
    dev = torch.device("cuda", shard_corpus.device)
    # The scan, the collective (or the host copy of the gloo path) and the merge must be ordered on ONE stream: torch's collectives
    # and .cpu() order against torch's CURRENT stream, so a caller-supplied raw stream is made current for the whole sequence
    # (with a non-current `stream` the all-gather could read `local` before the scan had written it).
    ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)) if stream is not None else contextlib.nullcontext()
    with ctx:
        st = torch.cuda.current_stream(dev).cuda_stream
        local = torch.empty((k, 2), dtype=torch.int64, device=dev)
        scorer.topk_entries_device(shard_corpus, k, local, op, index_base=shard_start, stream=st, **kw)
        merged = torch.empty((k, 2), dtype=torch.int64, device=dev)
        if dist.get_backend(group) == "nccl":
            everyone = torch.empty((world * k, 2), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(everyone, local, group=group)
            return merge_entries_device(everyone, k, merged, stream=st)
        host = [torch.empty((k, 2), dtype=torch.int64) for _ in range(world)]
        dist.all_gather(host, local.cpu(), group=group)
        merged.copy_(torch.from_numpy(merge_entries(torch.cat(host).numpy().view(np.uint64), k).view(np.int64)))
        return merged
