"""The N > 1 path on CPU: world_size-2 `gloo` processes exercise the shard arithmetic and the top-k
all-gather + merge (the only collective on the path).  No GPU compute happens here; the per-shard top-k lists
are produced by the oracle, exactly what a rank's rf_topk_u32 call returns on a GPU box."""
import os
import socket

import numpy as np
import pytest

from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd import parallel
from rapidfuzz_rs_amd.utils import synth
from oracle import oracle as o


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 64, 1000, 10**9 + 7):
        for world in (1, 2, 3, 8):
            edges = [parallel.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_topk(dist, k, cutoff, base=0, desc=False):
    idx = np.arange(len(dist), dtype=np.uint64) + np.uint64(base)
    keep = dist != np.uint64(2**64 - 1)
    d, i = dist[keep].astype(np.int64), idx[keep]
    order = np.lexsort((i, -d if desc else d))[:k]
    return d[order].astype(np.uint32), i[order]


def _worker(rank, world, port, q, rows, k, cutoff, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = parallel.shard_range(len(rows), rank, world)
        d = o.levenshtein.BatchComparator(q).rows(N.OP_DISTANCE, rows[a:b], score_cutoff=cutoff)
        s, i = _oracle_topk(d, k, cutoff, base=a)  # what rf_topk_u32(index_base=a) returns for this shard
        ms, mi = parallel.allgather_topk(s, i, k, N.OP_DISTANCE)
        ret[rank] = (ms.tolist(), mi.tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cutoff,k", [(None, 16), (3, 16), (40, 5), (0, 4)])
def test_topk_allgather_gloo_world2(cutoff, k):
    import torch.multiprocessing as mp

    q = synth.query(64, 0xC0FFEE05)
    rows = synth.rows_host(3001, 64, seed=0xC0FFEE05)
    synth.plant_near_duplicates(rows, q, every=211, seed=1)
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, rows, k, cutoff, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    d = o.levenshtein.BatchComparator(q).rows(N.OP_DISTANCE, rows, score_cutoff=cutoff)
    es, ei = _oracle_topk(d, k, cutoff)
    for r in range(world):  # every rank ends with the same, globally correct list
        assert ret[r][0] == es.tolist() and ret[r][1] == ei.tolist()
    if cutoff == 3:
        assert len(es) > 0  # the planted near-duplicates are found


def test_merge_topk_similarity_order():
    s = np.array([[9, 7, 7, 0], [8, 7, 1, 0]], dtype=np.uint32)  # lists are rows of k entries, `counts` valid each
    i = np.array([[4, 2, 9, 0], [5, 1, 3, 0]], dtype=np.uint64)
    ms, mi = parallel.merge_topk(N.OP_SIMILARITY, s, i, np.array([3, 3], dtype=np.uint32), 4)
    assert ms.tolist() == [9, 8, 7, 7] and mi.tolist() == [4, 5, 1, 2]


def _entries_worker(rank, world, port, lists, k, ret):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = torch.from_numpy(lists[rank].view(np.int64).copy())
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        ret[rank] = parallel.merge_entries(torch.cat(everyone).numpy().view(np.uint64), k).tolist()
    finally:
        dist.destroy_process_group()


def test_topk_entries_allgather_gloo_world2():
    """The 16-byte entry exchange (rf_topk_entry: any metric, any k, 64-bit global indices) at world size 2 over gloo: every rank's k
    entries are all-gathered and merged by (key, index) on the host (rf_topk_merge_entries); both ranks end with the same list,
    which is the sort of everything.  The entries are what rf_topk_entries_device returns for a Jaro-Winkler similarity scan of
    the rank's shard: f64 keys, descending, ties broken by global index, index bases beyond 2^32, one list padded with empties."""
    import torch.multiprocessing as mp

    q = synth.query(20, 7)
    rows = synth.rows_host(1501, 24, seed=8)
    synth.plant_near_duplicates(rows, q, every=97, seed=2)
    world, k, base = 2, 12, 2**35
    sims = o.jaro_winkler.BatchComparator(q).rows(N.OP_SIMILARITY, rows)

    def entries_of(lo, hi, kk):  # the definition in rfgpu.h: key = order-preserving image of the f64 score, complemented for similarities
        bits = sims[lo:hi].view(np.uint64) ^ np.uint64(0x8000000000000000)  # (similarities are >= 0)
        keys = ~bits
        order = np.lexsort((np.arange(lo, hi), keys))[:kk]
        e = np.full((k, 2), np.uint64(2**64 - 1), dtype=np.uint64)
        e[: len(order), 0] = keys[order]
        e[: len(order), 1] = np.uint64(base) + np.arange(lo, hi, dtype=np.uint64)[order]
        return e

    lists = [entries_of(0, 800, k), entries_of(800, 1501, 5)]  # the second shard found only 5 (a cutoff, say): 7 empty entries
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_entries_worker, args=(r, world, port, lists, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    exp = sorted([tuple(int(x) for x in e) for lst in lists for e in lst if int(e[0]) != 2**64 - 1])[:k]
    assert ret[0] == ret[1] == [list(e) for e in exp]
    # and the keys decode to the scores they came from
    got = parallel.decode_entries(np.array(ret[0], dtype=np.uint64), N.OP_SIMILARITY, True)
    assert all(v == float(sims[i - base]) for v, i in got) and got[0][0] == float(sims.max())


def test_shard_ragged_deals_every_length_bucket_and_balances_bytes():
    """on a LENGTH-SORTED input shard_range gives the ranks different mean lengths (the step is the slowest rank's);
    shard_ragged deals every length bucket to all ranks.  Payload bytes per rank within 1 %, every bucket within one candidate per
    rank, every candidate owned exactly once, indices ascending; take_ragged rebuilds exactly the owned candidates."""
    rng = np.random.default_rng(3)
    n = 200_000
    lens = np.sort(rng.integers(1, 300, size=n)).astype(np.uint64)  # length-sorted: the worst case for contiguous ranges
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    data = rng.integers(48, 123, size=int(offsets[-1]), dtype=np.uint8)
    for world in (2, 4, 8):
        shards = [parallel.shard_ragged(offsets, r, world) for r in range(world)]
        assert all((np.diff(s.astype(np.int64)) > 0).all() for s in shards)
        assert np.array_equal(np.sort(np.concatenate(shards)), np.arange(n, dtype=np.uint64))
        payload = [int(lens[s.astype(np.int64)].sum()) for s in shards]
        assert (max(payload) - min(payload)) / (sum(payload) / world) < 0.01
        contiguous = [int(lens[a:b].sum()) for a, b in (parallel.shard_range(n, r, world) for r in range(world))]
        assert (max(contiguous) - min(contiguous)) / (sum(contiguous) / world) > 0.5  # what it replaces on this input
        for L in (1, 57, 150, 299):
            per_rank = [int((lens[s.astype(np.int64)] == L).sum()) for s in shards]
            assert max(per_rank) - min(per_rank) <= 1
    # many buckets smaller than the world size: the remainders rotate instead of piling up on one rank
    lens2 = np.arange(1, 4001, dtype=np.uint64)
    off2 = np.zeros(4001, dtype=np.uint64)
    off2[1:] = np.cumsum(lens2)
    sizes = [len(parallel.shard_ragged(off2, r, 8)) for r in range(8)]
    assert max(sizes) - min(sizes) <= 1
    idx = parallel.shard_ragged(offsets, 1, 4)[:500]
    d2, o2 = parallel.take_ragged(data, offsets, idx)
    for j, i in enumerate(idx.astype(np.int64)):
        assert np.array_equal(d2[int(o2[j]) : int(o2[j + 1])], data[int(offsets[i]) : int(offsets[i + 1])])


def _ragged_worker(rank, world, port, q, data, offsets, k, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = parallel.shard_ragged(offsets, rank, world)
        d_r, o_r = parallel.take_ragged(data, offsets, idx)

        class _OracleScorer:  # what a rank's BatchComparator.topk returns on a GPU box: (scores, LOCAL indices) of its shard
            FLOAT = False
            _s1 = q

            @staticmethod
            def topk(corpus, kk, op=N.OP_DISTANCE, args=None, index_base=0, out=None, score_cutoff=None):
                d = o.levenshtein.BatchComparator(q).many(N.OP_DISTANCE, corpus[0], corpus[1], score_cutoff=score_cutoff)
                return _oracle_topk(d, kk, score_cutoff, base=index_base)

        ms, mi = parallel.sharded_topk(_OracleScorer, (d_r, o_r), k, 0, shard_index=idx)
        ret[rank] = (ms.tolist(), mi.tolist())
    finally:
        dist.destroy_process_group()


def test_sharded_topk_over_a_dealt_ragged_corpus_gloo_world2():
    """sharded_topk(shard_index=shard_ragged(...)): each rank scans its dealt share of a ragged corpus, local indices are mapped to
    original ones before the all-gather; both ranks end with the top-k of the whole corpus under (score, original index)."""
    import torch.multiprocessing as mp

    q = synth.query(48, 11)
    data, offsets = synth.ragged_host(4001, 64, seed=12)
    world, k = 2, 9
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q, data, offsets, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    d = o.levenshtein.BatchComparator(q).many(N.OP_DISTANCE, data, offsets)
    es, ei = _oracle_topk(d, k, None)
    for r in range(world):
        assert ret[r][0] == es.tolist() and ret[r][1] == ei.tolist()


def _filter_worker(rank, world, port, q, data, offsets, cutoff, dealt, by_score, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = len(offsets) - 1
        if dealt:
            idx = parallel.shard_ragged(offsets, rank, world)
            start = 0
        else:
            a, b = parallel.shard_range(n, rank, world)
            idx, start = np.arange(a, b, dtype=np.uint64), a
        d_r, o_r = parallel.take_ragged(data, offsets, idx)

        class _OracleScorer:  # what a rank's BatchComparator.filter_many returns on a GPU box: (index_base + LOCAL indices, scores), any order
            FLOAT = False
            _s1 = q

            @staticmethod
            def filter_many(op, corpus, args=None, order=N.FILTER_ANY, index_base=0, score_cutoff=None):
                d = o.levenshtein.BatchComparator(q).many(op, corpus[0], corpus[1], score_cutoff=score_cutoff)
                keep = np.nonzero(d != np.uint64(2**64 - 1))[0][::-1]  # (deliberately not in index order)
                return keep.astype(np.uint64) + np.uint64(index_base), d[keep].astype(np.uint32)

        gi, gs = parallel.sharded_filter(_OracleScorer, (d_r, o_r), N.OP_DISTANCE, shard_start=start, shard_index=idx if dealt else None,
                                         order=N.FILTER_BY_SCORE if by_score else N.FILTER_BY_INDEX, score_cutoff=cutoff)
        ret[rank] = (gi.tolist(), gs.tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dealt,by_score,cutoff", [(False, False, 35), (True, False, 35), (True, True, 36), (False, True, 0)])
def test_sharded_filter_gloo_world2(dealt, by_score, cutoff):
    """sharded_filter: both ranks end with filter_many's answer over the WHOLE corpus -- the candidates within the cutoff with their original
    indices, in index or (score, index) order -- whether the shards are contiguous ranges or shard_ragged's dealt index sets; a cutoff nothing
    passes gives two empty arrays (the padded exchange has nothing to send)."""
    import torch.multiprocessing as mp

    q = synth.query(40, 21)
    data, offsets = synth.ragged_host(3001, 64, seed=22)
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_filter_worker, args=(r, world, port, q, data, offsets, cutoff, dealt, by_score, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    d = o.levenshtein.BatchComparator(q).many(N.OP_DISTANCE, data, offsets, score_cutoff=cutoff)
    keep = np.nonzero(d != np.uint64(2**64 - 1))[0]
    if by_score:
        keep = keep[np.lexsort((keep, d[keep].astype(np.int64)))]
    if cutoff:
        assert len(keep) > 10
    for r in range(world):
        assert ret[r][0] == keep.tolist() and ret[r][1] == d[keep].astype(np.int64).tolist()
