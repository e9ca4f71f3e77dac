"""Run under `python -m torch.distributed.run --nproc-per-node R`: R ranks, one per GPU, over RCCL (backend "nccl") -- the
multi-GPU path of DESIGN.md 7 with REAL ranks.  Every rank scans its contiguous shard of ONE logical corpus and takes part in the
only exchange on the path, the k-entry all-gather + merge, through each of its forms:

  1. parallel.sharded_topk            host lists (scores, global indices), all_gather of 2k+1 words per rank
  2. parallel.sharded_topk_entries    16-byte device entries, all_gather_into_tensor + rf_topk_merge_entries_device (u32 and f64 scores)
  2b. parallel.sharded_filter         every rank's rf_filter survivors (global index, score), counts + padded all_gather (u32 and f64 scores)
  3. rf_topk_allgather_merge          raw ncclComm_t (ncclGetUniqueId on rank 0, broadcast, ncclCommInitRank on every rank)
  4. rf_topk_allgather_merge_entries  the same for 16-byte entries

and rank 0 checks all of them against ONE scan of the whole corpus in its own process.  Prints one JSON line on rank 0.

RF_TEST_BACKEND=gloo is the single-GPU stand-in (all ranks share GPU 0, forms 3 and 4 are skipped: RCCL refuses two ranks on one
device): it keeps this script exercised on the 1-GPU boxes; tests/test_gpu_parity.py runs the nccl form whenever the box has
>= 2 GPUs.
"""
import ctypes as C
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    import rapidfuzz_rs_amd as rf
    from rapidfuzz_rs_amd import _native as N
    from rapidfuzz_rs_amd import parallel
    from rapidfuzz_rs_amd.utils import synth

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    gloo = os.environ.get("RF_TEST_BACKEND") == "gloo"
    local = 0 if gloo else int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if gloo:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_total, ln, k, every = int(os.environ.get("RF_TEST_N", "1500000")), 64, 16, 50_000
    q = synth.query(64, 0xC0FFEE05)
    lo, hi = parallel.shard_range(n_total, rank, world)
    rows = synth.rows_device_range(lo, hi, ln, seed=0xC0FFEE05, device=dev, q=q, plant_every=every)
    shard = rf.Corpus.from_device_rows(rows)
    del rows
    lev = rf.distance.levenshtein.BatchComparator(q)
    jw = rf.distance.jaro_winkler.BatchComparator(q)
    got = {}
    # 1. host lists
    for name, kw in (("host_nocut", {}), ("host_cut3", {"score_cutoff": 3}), ("host_hint2", {"score_hint": 2})):
        s, i = parallel.sharded_topk(lev, shard, k, lo, device=None if gloo else dev, **kw)
        got[name] = [(int(a), int(b)) for a, b in zip(s, i)]
    # 2. device entries: u32 scores and f64 scores
    e = parallel.sharded_topk_entries(lev, shard, k, lo, N.OP_DISTANCE, score_cutoff=3)
    got["entries_lev_cut3"] = parallel.decode_entries(e, N.OP_DISTANCE, False)
    e = parallel.sharded_topk_entries(jw, shard, k, lo, N.OP_SIMILARITY)
    got["entries_jw"] = parallel.decode_entries(e, N.OP_SIMILARITY, True)
    # 2b. thresholded scans: every rank's rf_filter survivors, gathered (parallel.sharded_filter)
    fi, fs = parallel.sharded_filter(lev, shard, N.OP_DISTANCE, shard_start=lo, device=None if gloo else local, score_cutoff=3)
    got["filter_lev_cut3"] = [(int(b), int(a)) for a, b in zip(fi, fs)]
    fi, fs = parallel.sharded_filter(jw, shard, N.OP_SIMILARITY, shard_start=lo, order=N.FILTER_BY_SCORE, device=None if gloo else local, score_cutoff=0.8)
    got["filter_jw_08_by_score"] = [(float(b), int(a)) for a, b in zip(fi, fs)]
    if not gloo:
        # 3./4. the exchange below Python: a communicator made by hand on the RCCL torch has loaded
        libs = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
        rccl = C.CDLL(libs[0], mode=C.RTLD_GLOBAL)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        uid = UniqueId()
        if rank == 0:
            assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
        st = torch.cuda.current_stream().cuda_stream
        keys = torch.empty(k, dtype=torch.int64, device=dev)
        allk = torch.empty(k * world, dtype=torch.int64, device=dev)
        merged = torch.empty(k, dtype=torch.int64, device=dev)
        lev.topk_keys_device(shard, k, keys, N.OP_DISTANCE, rf.Args().score_cutoff(3), index_base=lo, stream=st)
        N.check(N.lib().rf_topk_allgather_merge(keys.data_ptr(), k, comm, world, allk.data_ptr(), merged.data_ptr(), local, st))
        torch.cuda.synchronize()
        got["raw_keys_cut3"] = [(int(x) >> 32, int(x) & 0xFFFFFFFF) for x in merged.cpu().tolist() if x != -1]
        ent = torch.empty((k, 2), dtype=torch.int64, device=dev)
        alle = torch.empty((k * world, 2), dtype=torch.int64, device=dev)
        me = torch.empty((k, 2), dtype=torch.int64, device=dev)
        jw.topk_entries_device(shard, k, ent, N.OP_SIMILARITY, index_base=lo, stream=st)
        N.check(N.lib().rf_topk_allgather_merge_entries(ent.data_ptr(), k, comm, world, alle.data_ptr(), me.data_ptr(), local, st))
        torch.cuda.synchronize()
        got["raw_entries_jw"] = parallel.decode_entries(me, N.OP_SIMILARITY, True)
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
    # every rank must hold the same merged lists
    box = [None] * world
    dist.all_gather_object(box, json.dumps(got, sort_keys=True))
    same = all(b == box[0] for b in box)
    ok, detail = same, {}
    if rank == 0:
        # the whole corpus in one process
        rows = synth.rows_device_range(0, n_total, ln, seed=0xC0FFEE05, device=dev, q=q, plant_every=every)
        whole = rf.Corpus.from_device_rows(rows)
        del rows
        d = lev.distance_many(whole).astype(np.int64)
        order = np.lexsort((np.arange(n_total), d))
        exp_nocut = [(int(d[i]), int(i)) for i in order[:k]]
        exp_cut3 = [(int(d[i]), int(i)) for i in order if d[i] <= 3][:k]
        s = jw.similarity_many(whole)
        order = np.lexsort((np.arange(n_total), -s))
        exp_jw = [(float(s[i]), int(i)) for i in order[:k]]
        keep = np.nonzero(d <= 3)[0]
        exp_filter = [(int(d[i]), int(i)) for i in keep]
        exp_filter_jw = [(float(s[i]), int(i)) for i in order if s[i] >= 0.8]
        want = {"filter_lev_cut3": exp_filter, "filter_jw_08_by_score": exp_filter_jw, "host_nocut": exp_nocut, "host_cut3": exp_cut3, "host_hint2": exp_nocut, "entries_lev_cut3": exp_cut3, "entries_jw": exp_jw,
                "raw_keys_cut3": exp_cut3, "raw_entries_jw": exp_jw}
        for name, val in got.items():
            good = [tuple(x) for x in val] == want[name]
            detail[name] = good
            ok = ok and good
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        C.CDLL(None).fflush(None)
        print(json.dumps({"ok": bool(ok), "ranks": world, "backend": "gloo (ranks share GPU 0)" if gloo else "nccl", "same_on_every_rank": same,
                          "gpus": 1 if gloo else world, "checks": detail}), flush=True)
    sys.exit(0 if ok or rank != 0 else 1)


if __name__ == "__main__":
    main()
