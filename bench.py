#!/usr/bin/env python3
"""bench.py -- the headline one-vs-many scan of BASELINE.json on MI355X.

A "step" is ONE pass of the hot path over the whole device-resident corpus: 1 query x n candidates
through `levenshtein::BatchComparator` semantics (rf_many_u32, RF_OP_DISTANCE), one u32 per candidate.
Workload at N=1 = BASELINE.json configs[1]: query len 64 vs 100 M random alphanumeric len-64 candidates.
With N > 1 (one rank per GPU; `python bench.py --gpus N` spawns the ranks itself through torch.distributed.run when
it is not already running under it) every rank owns its own 100 M-candidate shard (weak scaling), there is no data-path
collective, and each step ends with the top-k all-gather the north star names (k entries per rank over RCCL).
`--config c5` is BASELINE.json configs[4]: ONE logical corpus of 1 B len-64 candidates split over the ranks (strong
scaling), score_cutoff = 3, top-16 only, the RCCL all-gather + merge every step.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--candidates", type=int, default=100_000_000, help="candidates per GPU")
    ap.add_argument("--cand-len", type=int, default=64)
    ap.add_argument("--query-len", type=int, default=64)
    ap.add_argument("--metric", default="levenshtein", choices=["levenshtein", "indel", "lcs_seq", "osa", "jaro", "jaro_winkler"])
    ap.add_argument("--queries", type=int, default=1, help="Q > 1: Q queries x the corpus through rf_many_multi_u32 (N=1, many mode)")
    ap.add_argument("--cutoff", type=int, default=None)
    ap.add_argument("--fcutoff", type=float, default=None, help="similarity cutoff for jaro / jaro_winkler (e.g. 0.9)")
    ap.add_argument("--hint", type=int, default=None, help="score_hint of the scan (levenshtein.rs:1069-1088: queries beyond 64 symbols; results never depend on it)")
    ap.add_argument("--near-dup-share", type=float, default=0.0,
                    help="this share of the candidates is the query with 0..8 random substitutions (the corpus a score_hint is for); the rest stays random")
    ap.add_argument("--topk", type=int, default=16)
    ap.add_argument("--head-share", type=float, default=0.0,
                    help="fraction of the candidates that start with the query's first 8..12 symbols, the rest of the row random (shared prefixes: URLs, names, SKUs -- "
                         "what survives the cutoff scans' first pass without being a match)")
    ap.add_argument("--zipf", type=float, default=0.0, help="> 0: symbol ranks follow a Zipf law with this exponent instead of the uniform distribution")
    ap.add_argument("--lognormal-median", type=float, default=0.0, help="--ragged: > 0: log-normal candidate lengths with this median (sigma 0.5), clipped to [--min-len, --cand-len]")
    ap.add_argument("--capacity", type=int, default=1 << 20, help="--mode filter: room for (index, score) pairs")
    ap.add_argument("--mode", default="many", choices=["many", "topk", "filter"],
                    help="many: one score per candidate (configs[1]); topk: top-k only, no per-candidate output (configs[4]); filter: the (index, score) pairs of "
                         "the candidates within the cutoff, rf_filter_* (the reference user's filter_map over Option<T>)")
    ap.add_argument("--plant-every", type=int, default=1_000_000, help="near-duplicates of the query planted 1-in-N (cutoff/top-k runs)")
    ap.add_argument("--weights", default=None, help="levenshtein WeightTable as ins,del,sub (e.g. 1,2,3: the generalized Wagner-Fischer kernel)")
    ap.add_argument("--symbols", type=int, default=62, help="alphabet size of the synthetic corpus (experiment knob, default alphanumeric)")
    ap.add_argument("--config", default=None, choices=["c2", "c5"],
                    help="c2 = BASELINE.json configs[1] (the default workload); c5 = configs[4]: 1 B candidates split over the ranks, "
                         "score_cutoff 3, top-16, all-gather every step")
    ap.add_argument("--total-candidates", type=int, default=1_000_000_000, help="size of the logical corpus of --config c5")
    ap.add_argument("--shard", default="range", choices=["range", "dealt"],
                    help="--config c5: how the logical corpus is split over the ranks -- contiguous ranges (parallel.shard_range) or dealt by parallel.shard_ragged (every "
                         "length bucket to all ranks; local indices mapped to original ones before the exchange).  Same merged top-k either way (topk_checksum)")
    ap.add_argument("--settle-ms", type=float, default=200.0,
                    help="untimed steps run for this long BEFORE the W warm-up steps: after the idle set-up phase the GPU's clock takes "
                         "~15 back-to-back launches to ramp (profiles/clock_ramp_r02.txt); 0 = off.  Reported as config.settle_steps")
    ap.add_argument("--ragged", action="store_true",
                    help="candidate lengths uniform in [1, --cand-len] instead of one fixed length (BASELINE.json configs[0]'s 'len <= 64' "
                         "distribution at scale): the corpus is packed from host arrays into exact-length tiles + mixed tiles")
    ap.add_argument("--min-len", type=int, default=1, help="--ragged: shortest candidate length (lengths uniform in [--min-len, --cand-len])")
    ap.add_argument("--slot-order", action="store_true",
                    help="--ragged: results in the corpus' SLOT order (RF_FLAG_SLOT_ORDER: no gather pass; rf_corpus_slot_index maps them back -- the parity leg does)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", default="auto", choices=["auto", "on", "off"],
                    help="after the headline's timed region, run one short leg (own process, --extra-steps steps, own roofline + oracle parity) for "
                         "each OTHER BASELINE.json config and report them as `extra_configs`.  auto = on for the default workload at N = 1")
    ap.add_argument("--extra-steps", type=int, default=20, help="timed steps of each extra_configs leg (the launch-bound configs[0] leg runs 25 x as many)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "on", "off"],
                    help="on: roofline.traffic from two short rocprofv3 --pmc passes of this command run after the timed region (when rocprofv3 is on the "
                         "box; N = 1), else the committed counters of profiles/traffic.json marked as such; off: only the committed file; auto: on for "
                         "workloads of >= 10 M candidates outside the extra legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: launch N ranks of this very command (one per GPU,
    rendezvous on 127.0.0.1) and pass their output through.  Rank 0 prints the JSON line."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    import rapidfuzz_rs_amd as rf
    from rapidfuzz_rs_amd import _native as N
    from rapidfuzz_rs_amd import parallel
    from rapidfuzz_rs_amd.utils import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): refusing to report a number for a job "
                         "that is not the one asked for")
    c5 = args.config == "c5"
    # exercise the collective path at world size 1 as well: always for c5 (its step IS scan + gather + merge)
    force_dist = os.environ.get("RF_BENCH_FORCE_DIST") == "1" or c5
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RF_BENCH_BACKEND=gloo is a TEST mode: all ranks share GPU 0 and the k-entry exchange goes through host tensors, so the
    # multi-rank logic (index bases, merge, max-over-ranks timing) can be exercised on a single-GPU box.  Never a result.
    test_gloo = os.environ.get("RF_BENCH_BACKEND") == "gloo"
    if test_gloo:
        local_rank = 0
    if world > 1 or force_dist:
        if test_gloo:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if N.lib().rf_device_count() < 1:
        raise RuntimeError("no HIP device visible to librfgpu.so")
    # the ranks that actually joined, and the distinct GPUs they sit on: what n_gpus reports
    joined, gpus_used = 1, 1
    if world > 1 or force_dist:
        mine = torch.tensor([1, local_rank], dtype=torch.int64, device="cpu" if test_gloo else dev)
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        joined = int(sum(int(t[0]) for t in everyone))
        dist.barrier()  # the first barrier of a process group initialises lazily (~0.6 s here); pay that now, not between warm-up and the timed steps
        gpus_used = len({int(t[1]) for t in everyone})
        if joined != world or (gpus_used != world and not test_gloo):
            raise SystemExit(f"bench.py: {joined} of {world} ranks joined on {gpus_used} distinct GPUs")

    if c5:  # BASELINE.json configs[4]
        args.metric, args.mode, args.cutoff, args.topk, args.queries = "levenshtein", "topk", 3, 16, 1
        args.cand_len, args.query_len, args.weights, args.fcutoff = 64, 64, None, None
        shard_lo, shard_hi = parallel.shard_range(args.total_candidates, rank, world)
        args.candidates = shard_hi - shard_lo
        dealt_index = None
        if args.shard == "dealt":
            if args.total_candidates > 200_000_000:
                raise SystemExit("bench.py: --shard dealt generates the whole logical corpus on every rank before it takes its share: --total-candidates <= 200 M")
            dealt_index = parallel.shard_ragged(np.arange(args.total_candidates + 1, dtype=np.uint64) * np.uint64(args.cand_len), rank, world)
            args.candidates = len(dealt_index)
    n, ln = args.candidates, args.cand_len
    q = synth.query(args.query_len, 0xC0FFEE05 if c5 else 0xC0FFEE02)
    mod = getattr(rf.distance, args.metric)
    scorer = mod.BatchComparator(q)
    nq = args.queries if (world == 1 and not force_dist and args.mode == "many") else 1
    queries = [q] + [synth.query(args.query_len, 0xC0FFEE02 + 31 * j) for j in range(1, nq)]
    scorers = [scorer] + [mod.BatchComparator(x) for x in queries[1:]]

    # synthetic corpus, generated and packed on the device (excluded from the timed region)
    t0 = time.time()
    index_base = rank * n
    if c5:  # this rank's slice of the ONE logical corpus (the same rows whatever the world size)
        rows = synth.rows_device_range(shard_lo, shard_hi, ln, seed=0xC0FFEE05, device=dev, symbols=args.symbols, q=q, plant_every=args.plant_every,
                                       head_share=args.head_share)
        index_base = shard_lo
        if dealt_index is not None:  # this rank's DEALT share of the same logical corpus; its top-k keys carry local indices until step() maps them
            del rows
            whole = synth.rows_device_range(0, args.total_candidates, ln, seed=0xC0FFEE05, device=dev, symbols=args.symbols, q=q, plant_every=args.plant_every,
                                            head_share=args.head_share)
            dealt_t = torch.from_numpy(dealt_index.astype(np.int64)).to(dev)
            rows = whole[dealt_t].contiguous()
            del whole
            index_base = 0
    else:
        rows = synth.rows_device(n, ln, seed=0xC0FFEE02 + 7919 * rank, device=dev, symbols=args.symbols, zipf_s=args.zipf)
        if args.head_share > 0:
            synth.head_share_rows_device(rows, q, args.head_share, seed=0x5EED + rank)
    if not c5 and (args.cutoff is not None or args.mode in ("topk", "filter")):
        # 1 in 10^6 candidates is the query with 0..5 random substitutions
        gen = torch.Generator(device=dev)
        gen.manual_seed(99 + rank)
        pidx = torch.arange(args.plant_every // 2, n, args.plant_every, device=dev)
        if len(pidx):
            qrow = torch.tensor(list(q[:ln].ljust(ln, b"0")), dtype=torch.uint8, device=dev)
            planted = qrow.repeat(len(pidx), 1)
            for _ in range(5):
                hit = torch.rand(len(pidx), device=dev, generator=gen) < 0.5
                pos = torch.randint(0, ln, (len(pidx),), device=dev, generator=gen)
                sub = torch.randint(48, 58, (len(pidx),), device=dev, generator=gen, dtype=torch.int64).to(torch.uint8)
                sel = torch.nonzero(hit).flatten()
                planted[sel, pos[sel]] = sub[sel]
            rows[pidx] = planted
    if args.near_dup_share > 0.0 and not c5:
        gen = torch.Generator(device=dev)
        gen.manual_seed(4242 + rank)
        pidx = torch.nonzero(torch.rand(n, device=dev, generator=gen) < args.near_dup_share).flatten()
        qrow = torch.tensor(list(q[:ln].ljust(ln, b"0")), dtype=torch.uint8, device=dev)
        for a in range(0, len(pidx), 1 << 22):
            sel_rows = pidx[a : a + (1 << 22)]
            planted = qrow.repeat(len(sel_rows), 1)
            for _ in range(8):
                hit = torch.nonzero(torch.rand(len(sel_rows), device=dev, generator=gen) < 0.5).flatten()
                pos = torch.randint(0, ln, (len(sel_rows),), device=dev, generator=gen)
                sub = torch.randint(48, 58, (len(sel_rows),), device=dev, generator=gen, dtype=torch.int64).to(torch.uint8)
                planted[hit, pos[hit]] = sub[hit]
            rows[sel_rows] = planted
        del pidx
    sample_rows = min(n, 32_000_000)
    host_sample = None
    host_strided = None
    in_window = 1.0
    ragged_sample = None  # (data, offsets) of the first candidates / of every 1009-th candidate
    ragged_strided = None
    mean_len = float(ln)
    if args.ragged:
        if c5 or world > 1 or force_dist:
            raise SystemExit("bench.py: --ragged is a single-GPU 'many' workload")
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xC0FFEE07)
        if args.lognormal_median > 0:
            lens = torch.empty(n, device=dev, dtype=torch.float32).log_normal_(float(np.log(args.lognormal_median)), 0.5, generator=gen).round_().clamp_(args.min_len, ln).to(torch.int64)
        else:
            lens = torch.randint(args.min_len, ln + 1, (n,), device=dev, generator=gen, dtype=torch.int64)
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(lens, 0)
        flat = torch.empty(int(offsets[-1].item()), dtype=torch.uint8, device=dev)
        col = torch.arange(ln, device=dev)[None, :]
        slab = max(1, min(1 << 23, (1 << 30) // max(1, ln)))  # (boolean indexing in slabs: a mask of 2^31 symbols overflows torch's index arithmetic)
        for a in range(0, n, slab):
            b = min(n, a + slab)
            flat[int(offsets[a].item()) : int(offsets[b].item())] = rows[a:b][col < lens[a:b, None]]  # row-major: candidate a's symbols, then a + 1's, ...
        del rows, col
        mean_len = float(offsets[-1].item()) / n
        # (cutoff runs: the share of candidates whose length lies inside the cutoff's window |len2 - len1| <= cutoff -- the rest is decided by
        # length alone and costs only its `None`)
        in_window = float(((lens - args.query_len).abs() <= (args.cutoff if args.cutoff is not None else 1 << 30)).float().mean().item())
        h_data, h_off = flat.cpu().numpy(), offsets.cpu().numpy().astype(np.uint64)
        del flat, offsets, lens
        if rank == 0 and not args.no_cpu_baseline:
            m = min(n, 8_000_000)
            ragged_sample = (h_data[: int(h_off[m])], h_off[: m + 1].copy())
            pick = np.arange(0, n, 1009)
            parts = [h_data[int(h_off[i]) : int(h_off[i + 1])] for i in pick]
            so = np.zeros(len(pick) + 1, dtype=np.uint64)
            so[1:] = np.cumsum([len(x) for x in parts])
            ragged_strided = (np.concatenate(parts), so)
        t_pack = time.time()
        corpus = rf.Corpus.from_ragged(h_data, h_off, device=local_rank)  # host arrays -> scannable: upload, length sort, scatter (rf_corpus_pack)
        torch.cuda.synchronize()
        t_pack = time.time() - t_pack
        pack_payload = int(h_off[-1])
        del h_data, h_off
    else:
        if rank == 0 and not args.no_cpu_baseline and world == 1:  # the CPU baseline and the oracle parity leg run at N = 1 only
            host_sample = rows[:sample_rows].cpu().numpy()
            host_strided = rows[::1009].cpu().numpy()  # every 1009-th candidate of the WHOLE shard
        t_pack = time.time()
        corpus = rf.Corpus.from_device_rows(rows)  # device rows -> packed tiles (rf_corpus_pack_rows_device)
        torch.cuda.synchronize()
        t_pack = time.time() - t_pack
        pack_payload = n * ln
        del rows
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    is_f64 = args.metric in ("jaro", "jaro_winkler")
    n_out = corpus.slot_count if args.slot_order else n
    out = torch.empty(n_out * nq, dtype=torch.float64 if is_f64 else torch.int32, device=dev)
    call_args = rf.Args()
    if args.slot_order:
        if nq > 1 or args.mode != "many" or world > 1 or force_dist:
            raise SystemExit("bench.py: --slot-order is a single-GPU, single-query 'many' workload")
        call_args = call_args.slot_order()
    if args.cutoff is not None:
        call_args = call_args.score_cutoff(args.cutoff)
    if args.hint is not None:
        call_args = call_args.score_hint(args.hint)
    if args.fcutoff is not None:
        args.cutoff = args.fcutoff  # the oracle legs below pass it on unchanged
        call_args = call_args.score_cutoff(args.fcutoff)
    weights = tuple(int(x) for x in args.weights.split(",")) if args.weights else None
    if weights:
        call_args = call_args.weights(rf.WeightTable(*weights))
    # The scans run on a stream of their own and the exchange on a HIGH-PRIORITY side stream: the runtime multiplexes streams
    # onto a few hardware queues, and with the null stream + a default-priority side stream both landed on ONE queue
    # (rocprofv3: same queue id), which put the exchange kernels and their barrier packets between every two scans.
    stream = torch.cuda.Stream(device=dev) if (world > 1 or force_dist) else torch.cuda.current_stream(dev)
    torch.cuda.set_stream(stream)

    last_topk = [None]

    filt = None
    if args.mode == "filter":
        if world > 1 or force_dist or nq > 1:
            raise SystemExit("bench.py: --mode filter is a single-GPU, single-query workload")
        import ctypes

        filt = {"idx": torch.empty(max(args.capacity, 1), dtype=torch.int64, device=dev),
                "val": torch.empty(max(args.capacity, 1), dtype=torch.float64 if is_f64 else torch.int32, device=dev), "count": ctypes.c_uint64(0),
                "fn": N.lib().rf_filter_f64 if is_f64 else N.lib().rf_filter_u32, "args": call_args.to_c(is_f64)}

    def step():
        if filt is not None:
            # (synchronizes: the count comes back to the host -- part of what the call is, and inside the timed region)
            N.check(filt["fn"](scorer._h, corpus._h, N.OP_SIMILARITY if is_f64 else N.OP_DISTANCE, ctypes.byref(filt["args"]), 0, args.capacity, filt["idx"].data_ptr(),
                               filt["val"].data_ptr(), ctypes.byref(filt["count"]), N.MEM_DEVICE, N.FILTER_BY_INDEX, stream.cuda_stream))
        elif nq > 1:
            mod.BatchComparator.many_multi(scorers, N.OP_SIMILARITY if is_f64 else N.OP_DISTANCE, corpus, call_args, out=out, stream=stream.cuda_stream)
        elif is_f64:  # BASELINE.json configs[3]: similarity, f64 per candidate
            scorer.similarity_many(corpus, call_args, out=out, stream=stream.cuda_stream)
        elif args.mode == "many" and world == 1 and not force_dist:
            scorer.distance_many(corpus, call_args, out=out, stream=stream.cuda_stream)
        else:
            # one pass: per-shard top-k under (distance, global index) [+ every candidate's distance, which stays on
            # its GPU]; then the ONLY exchange on the path: the k-entry all-gather (RCCL over xGMI) and the merge.
            # Everything is stream-ordered on the device -- no host round trip inside a step.
            buf = step_no[0] & 1
            step_no[0] += 1
            if buf_free[buf] is not None and not buf_free[buf].query():
                buf_free[buf].synchronize()  # the exchange that last read this buffer pair (two steps ago): long done -- checked
                # on the host, because a cross-queue wait enqueued on the scan stream costs a barrier packet every step
            scorer.topk_keys_device(corpus, args.topk, local_keys[buf], N.OP_DISTANCE, call_args, index_base=index_base,
                                    out=out if args.mode == "many" else None, stream=stream.cuda_stream)
            if c5 and dealt_index is not None:
                # a dealt shard: (distance, LOCAL index) keys -> (distance, original index); the map ascends, so the list stays sorted
                kk = local_keys[buf]
                live = kk != -1
                loc = (kk & 0xFFFFFFFF).clamp_(max=n - 1)
                local_keys[buf].copy_(torch.where(live, (kk & ~0xFFFFFFFF) | dealt_t[loc], kk))
            if world > 1 or force_dist:
                if test_gloo:
                    host = [torch.empty(args.topk, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(host, local_keys[buf].cpu())
                    all_keys[buf].copy_(torch.cat(host))
                    last_topk[0] = parallel.merge_keys_device(all_keys[buf], args.topk, merged_keys)
                else:
                    # The exchange runs on its own stream, beside the next step's scan: the scan stream never waits for a
                    # collective (only, trivially, for the one of two steps ago before it reuses a key buffer).
                    keys_ready = torch.cuda.Event()
                    keys_ready.record(stream)
                    with torch.cuda.stream(xchg):
                        xchg.wait_event(keys_ready)
                        if os.environ.get("RF_BENCH_NOGATHER"):  # diagnostic: the exchange replaced by a local copy
                            all_keys[buf][: args.topk].copy_(local_keys[buf])
                        else:
                            dist.all_gather_into_tensor(all_keys[buf], local_keys[buf])  # RCCL; the CPU does not block
                        last_topk[0] = parallel.merge_keys_device(all_keys[buf], args.topk, merged_keys, stream=xchg.cuda_stream)  # one small kernel
                        buf_free[buf] = torch.cuda.Event()
                        buf_free[buf].record(xchg)
            else:
                last_topk[0] = local_keys[buf][: args.topk]

    def finish_exchange():
        if xchg is not None:
            stream.wait_stream(xchg)  # the last step's gather + merge belongs to the timed region

    xchg = torch.cuda.Stream(device=dev, priority=-1) if ((world > 1 or force_dist) and not test_gloo) else None
    buf_free = [None, None]
    step_no = [0]
    merged_keys = torch.empty(args.topk, dtype=torch.int64, device=dev)
    local_keys = [torch.empty(args.topk, dtype=torch.int64, device=dev) for _ in range(2)]
    all_keys = [torch.empty(args.topk * max(world, 1), dtype=torch.int64, device=dev) for _ in range(2)]

    # the very first call, cold: table upload, scratch allocation, a GPU clock that has been idling through the set-up
    # (reported as config.first_call_ms next to the settled figure)
    t_first = time.perf_counter()
    step()
    finish_exchange()
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t_first) * 1e3
    # clock settle (disclosed in config.settle_steps): the set-up phase above leaves the GPU mostly idle and its clock low
    settle_steps = 0
    if args.settle_ms > 0:
        t_settle = time.perf_counter()
        for _ in range(4):
            step()
        finish_exchange()
        torch.cuda.synchronize()
        more = int(min(2000, max(0.0, args.settle_ms * 1e-3 / max((time.perf_counter() - t_settle) / 4, 1e-5) - 4)))
        if world > 1 or force_dist:  # every rank must issue the same number of exchanges
            t = torch.tensor([more], dtype=torch.int64, device="cpu" if test_gloo else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            more = int(t.item())
        for _ in range(more):
            step()
        finish_exchange()
        torch.cuda.synchronize()
        settle_steps = 4 + more
    for _ in range(args.warmup):
        step()
    finish_exchange()
    torch.cuda.synchronize()
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()

    # HIP events on the launch stream bracket the K steps (one pair, not one per step: every event record is a marker packet
    # the queue has to process between two back-to-back kernels, ~7 us each on this runtime)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for s in range(args.steps):
        step()
    ev1.record(stream)
    finish_exchange()  # the last step's gather + merge is inside the timed region
    torch.cuda.synchronize()
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch duration: HIP events on the launch stream over the timed region
    if world > 1 or force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if test_gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([kernel_ms], dtype=torch.float64, device="cpu" if test_gloo else dev)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms = float(km.item())

    if rank != 0:
        if world > 1 or force_dist:
            dist.destroy_process_group()
        return

    pairs_per_step = args.total_candidates if c5 else n * world * nq
    ms_per_step = elapsed * 1e3 / args.steps
    gpairs = pairs_per_step / (elapsed / args.steps) / 1e9
    # Algorithmic bytes per pair, SURVEY.md 8(d): candidate bytes at bucket length + the result (u32 / f64 per candidate,
    # nothing in top-k-only mode); Q fused queries read each candidate once: ln / Q candidate bytes per pair.
    out_bytes = (8 if is_f64 else 4) if args.mode == "many" else 0
    if args.ragged:
        ln = mean_len  # algorithmic candidate bytes per pair: the mean candidate length (no padding byte is read or computed)
    survey_bpp = ln / nq + out_bytes
    # With a tight cutoff the early-out is part of the algorithm: on this corpus nearly every candidate is decided from
    # its first 16-byte chunk, so the bytes the path HAS to move are that chunk + the result -- which is also what the
    # PMC counters see (profiles/traffic.json).  Both accountings are reported: `roofline` prices what is moved,
    # `roofline.survey_8d` the survey's full-candidate figure (its frac can exceed 1: the bytes are legitimately not read).
    # (the library switches the early-out on by how much normalized distance the cutoff allows -- rf_api.hip plan())
    early = False
    if args.cutoff is not None and not is_f64 and not weights:
        maximum = (ln + args.query_len) if args.metric == "indel" else max(ln, args.query_len)
        early = args.cutoff / max(maximum, 1) < (0.4 if args.metric in ("indel", "lcs_seq") else 0.7)
    # (Levenshtein / OSA under a cutoff <= 5 on a single-length corpus: the first look reads the 8-symbol head plane, rf_pack.hip)
    head8 = early and args.metric in ("levenshtein", "osa") and args.cutoff <= 5 and not args.ragged and args.query_len <= 64 and n >= (1 << 20) and os.environ.get("RF_HEAD8_MIN") != "0"
    # (cutoffs <= 3 go through the band prefilter, which streams the 6-bit plane -- 6 bytes per candidate -- when the corpus stores fewer than
    # 64 distinct symbols, rf_pack.hip head6_plane_kernel; RF_HEAD6=0 / RF_BAND_FILTER=0 keep the 8-byte plane)
    head6 = head8 and args.cutoff <= 3 and args.symbols < 64 and os.environ.get("RF_HEAD6", "1") != "0" and os.environ.get("RF_BAND_FILTER", "1") != "0"
    bytes_per_pair = (min(ln, (6 if head6 else 8) if head8 else 16) if early else ln) / nq + out_bytes
    if args.ragged and early and args.metric in ("levenshtein", "osa"):
        # a length-bucketed corpus under a small cutoff: only the candidates inside the length window are looked at (their 8-symbol head from
        # the plane when the runs are walked as single-length views, DESIGN.md 5.1 v; their first chunk row otherwise); every candidate has its result
        views = args.cutoff <= 5 and args.query_len <= 64 and n >= (1 << 20) and os.environ.get("RF_HEAD8_MIN") != "0"
        bytes_per_pair = in_window * (8 if views else 16) + out_bytes
    # (Indel / LCS with u32 results on a single-length corpus of < 64 symbols whose length is a whole number of chunks: the scan streams the 6-bit copy of the
    # payload, 0.75 bytes per symbol -- rf_stream_asm.hip stream_lcs6_uniform_kernel.  `roofline` keeps the survey's byte-per-symbol figure (what the
    # north star's 0.60 is about); what is MOVED is reported beside it and is what the traffic counters see)
    pack6 = (args.metric in ("indel", "lcs_seq") and args.mode == "many" and not args.ragged and not early and nq == 1 and args.query_len <= 64
             and (args.cand_len % 16 == 0 or args.symbols < 63) and args.symbols < 64 and n >= (1 << 20) and os.environ.get("RF_PACK6", "1") != "0" and not c5 and world == 1)
    # (the same on a length-bucketed corpus: the tiles kernels over the 6-bit image of the payload, 12 bytes per STARTED 16 symbols)
    pack6_ragged = (args.metric in ("indel", "lcs_seq") and args.mode == "many" and args.ragged and not early and nq == 1 and args.query_len <= 64 and args.symbols < 64
                    and n >= (1 << 20) and os.environ.get("RF_PACK6", "1") != "0" and not c5 and world == 1)
    moved_bpp = None
    if pack6:
        moved_bpp = 12 * ((ln + 15) // 16) + out_bytes
    elif pack6_ragged:
        moved_bpp = 12 * sum((L + 15) // 16 for L in range(args.min_len, args.cand_len + 1)) / (args.cand_len - args.min_len + 1) + out_bytes
    pairs_per_gpu = pairs_per_step / world
    achieved = pairs_per_gpu * bytes_per_pair / (kernel_ms * 1e-3) / 1e9  # per GPU, GB/s
    survey_achieved = pairs_per_gpu * survey_bpp / (kernel_ms * 1e-3) / 1e9

    what = f"{args.metric}::BatchComparator, 1 query len-{args.query_len} x "
    if c5:
        what += (f"{args.total_candidates} random alphanumeric len-{ln} candidates in ONE logical corpus split over {world} GPU(s), score_cutoff=3, "
                 f"top-{args.topk} + all-gather + merge every step" + (f", {args.head_share:.2%} of them starting with the query's first 8..12 symbols" if args.head_share > 0 else "")
                 + (", BASELINE.json configs[4]" if args.total_candidates == 1_000_000_000 and args.head_share == 0 else ""))
    else:
        what += ((f"{n} random alphanumeric candidates with lengths uniform in [{args.min_len}, {args.cand_len}] (mean {mean_len:.2f}) per GPU, " if args.ragged
                  else f"{n} random alphanumeric len-{ln} candidates per GPU, ") + ("no cutoff" if args.cutoff is None else f"score_cutoff={args.cutoff}")
                 + (f", weights={weights}" if weights else "")
                 + (f", {args.near_dup_share:.0%} of them the query with 0..8 substitutions" if args.near_dup_share > 0 else "")
                 + (f", {args.head_share:.2%} of them starting with the query's first 8..12 symbols" if args.head_share > 0 else "")
                 + (f", Zipf({args.zipf}) symbols" if args.zipf > 0 else "")
                 + (f", log-normal lengths (median {args.lognormal_median})" if args.ragged and args.lognormal_median > 0 else "")
                 + (f", score_hint={args.hint}" if args.hint is not None else "")
                 + (", results in slot order (RF_FLAG_SLOT_ORDER)" if args.slot_order else "")
                 + (", BASELINE.json configs[1]" if (args.metric == "levenshtein" and n == 100_000_000 and ln == 64 and args.query_len <= 64
                                                     and args.cutoff is None and not weights and not args.ragged and args.near_dup_share == 0 and args.hint is None
                                                     and args.head_share == 0 and args.zipf == 0 and args.mode == "many") else ""))
    result = {
        "metric": "Gpairs/s (1 query x N candidates, Levenshtein BatchComparator semantics)" if args.metric == "levenshtein" else f"Gpairs/s ({args.metric})",
        "value": round(gpairs, 3),
        "unit": "Gpairs/s",
        "n_gpus": gpus_used if not test_gloo else joined,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if c5 else "weak",
        "vs_baseline": None,
        "dtype": "u64 bit-vectors (u32 results)",
        "data": "synthetic",
        "config": {
            "workload": what,
            "candidates_per_gpu": n,
            "candidate_len": args.cand_len if not args.ragged else f"uniform in [{args.min_len}, {args.cand_len}], mean {mean_len:.3f}",
            "query_len": args.query_len,
            "queries": nq,
            "output": (("f64" if is_f64 else "u32") + " per candidate, device-resident" if args.mode == "many" else
                       (f"(index, score) pairs of the candidates within the cutoff, device-resident, room for {args.capacity}" if args.mode == "filter" else f"top-{args.topk} only")),
            "parallelism": (f"corpus sharded over {world} GPU(s)" + (" by parallel.shard_ragged (dealt)" if c5 and args.shard == "dealt" else "") + f", top-{args.topk} all-gather")
                           if (world > 1 or force_dist) else "1 GPU",
            "ranks_joined": joined,
            **({"rccl_ranks": dist.get_world_size(), "collective": "ncclAllGather via torch.distributed (backend nccl = RCCL)"}
               if (world > 1 or force_dist) and not test_gloo else {}),
            "setup_s": round(t_setup, 2),
            # (what a corpus costs to build, apart from generating the synthetic data) pack_s: the library call that turns the input into a scannable
            # corpus -- host arrays in (upload included) for --ragged, device rows in otherwise; accel_build_ms: what the FIRST scan spent beyond a settled one (head
            # planes, the 6-bit payload, gather maps -- built on first use -- plus table upload and cold clocks)
            "pack_s": round(t_pack, 4),
            "pack_payload_gb_per_s": round(pack_payload / max(t_pack, 1e-9) / 1e9, 2),
            "accel_build_ms": round(max(0.0, first_call_ms - ms_per_step), 3),
            "settle_steps": settle_steps,
            "first_call_ms": round(first_call_ms, 3),
            **({"test_backend": "gloo: ranks share one GPU, NOT a measurement"} if test_gloo else {}),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(args, n, kernel_ms),
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_pair": bytes_per_pair,
            "survey_8d": {"bytes_per_pair": survey_bpp, "achieved": round(survey_achieved, 1), "frac": round(survey_achieved / HBM_PEAK_GBS, 4)},
            **({"moved": {"bytes_per_pair": moved_bpp, "achieved": round(pairs_per_gpu * moved_bpp / (kernel_ms * 1e-3) / 1e9, 1),
                          "what": "the scan streams the 6-bit copy of the payload (12 bytes per started 16 symbols)"}} if moved_bpp is not None else {}),
        },
    }
    if (args.metric in ("levenshtein", "indel", "lcs_seq", "osa") and args.query_len <= (512 if args.metric == "levenshtein" else 64) and not weights and not early
            and os.environ.get("RF_BENCH_IN_PMC") != "1"):  # (not inside this command's own counter passes: traffic_in_run)
        # The bit-parallel scans of this family are bound by VALU issue before they are bound by HBM.  The
        # ceiling is MEASURED here, in this process: the library's own recurrence column on register-resident pattern
        # words, no HBM / LDS / tile loop (rf_probe_issue_rate, rapidfuzz_rs_amd/csrc/rf_probe.hip).
        import ctypes

        torch.cuda.synchronize()
        rate = ctypes.c_double(0.0)
        metric_id = {"levenshtein": N.LEVENSHTEIN, "indel": N.INDEL, "lcs_seq": N.LCS_SEQ, "osa": N.OSA}[args.metric]
        if N.lib().rf_probe_issue_rate(metric_id, args.query_len, 0, local_rank, 8, ctypes.byref(rate)) == N.RF_OK and rate.value > 0:
            ceiling = rate.value * 64.0 / max(ln, 1)  # wave-columns/ns -> Gpairs/s at this candidate length (every pair runs its own columns)
            per_gpu = gpairs / world
            result["roofline"]["issue_bound"] = {"achieved": round(per_gpu, 3), "ceiling": round(ceiling, 3), "unit": "Gpairs/s",
                                                 "frac": round(per_gpu / ceiling, 4),
                                                 "source": "rf_probe_issue_rate in this run: the product's State::step on register-resident PM words, 8 workgroups/CU"}
            if per_gpu > ceiling:
                # (multi-word states: the probe holds W x the state at a lower occupancy than the scan kernel; 32-bit states: within noise)
                result["roofline"]["issue_bound"]["note"] = "the register probe ran SLOWER than the kernel here: a reference point, not an upper bound"
            if N.lib().rf_probe_issue_rate(metric_id, args.query_len, 1, local_rank, 8, ctypes.byref(rate)) == N.RF_OK and rate.value > 0:
                with_lds = rate.value * 64.0 / max(ln, 1)
                result["roofline"]["issue_bound"]["ceiling_with_lds_gather"] = round(with_lds, 3)
                if with_lds > ceiling:
                    # (multi-word states: hipcc schedules the bare State::step loop of the register probe WORSE than the scans' own chunk
                    # code, gathers included -- the faster of the two probes is the ceiling)
                    ceiling = with_lds
                    ib0 = result["roofline"]["issue_bound"]
                    ib0["ceiling"], ib0["frac"] = round(ceiling, 3), round(per_gpu / ceiling, 4)
                    ib0["source"] = "rf_probe_issue_rate mode 1 in this run: the scans' own chunk code (byte extraction + LDS gather), no HBM traffic, no tile loop"
                    ib0.pop("note", None)
            if N.lib().rf_probe_issue_rate(metric_id, args.query_len, 2, local_rank, 8, ctypes.byref(rate)) == N.RF_OK and rate.value > 0:
                # the kernel's own hand-scheduled chunk (rf_lev_asm.hip) with LDS gathers but no HBM traffic and no tile loop
                result["roofline"]["issue_bound"]["ceiling_asm_chunk"] = round(rate.value * 64.0 / max(ln, 1), 3)
            # The ceiling above is measured with idle HBM, where the chip clocks ~2.4 GHz.  A scan that streams HBM runs at the package
            # power cap (rocm-smi: 1378-1400 W of 1400 W) and the governor takes the core clock down to 1.75-2.15 GHz -- for the SAME
            # number of cycles per launch (tools/clock_of.sh).  Sample both clocks in this run (rf_probe_core_clock: one wavefront on
            # a stream of its own, beside ~40 queued scans / beside the probe kernel) and restate the ceiling at the scan's clock.
            try:
                if world > 1 or force_dist:
                    raise RuntimeError("clock sampling runs extra steps on rank 0 only: single-process runs only (the other ranks have left)")
                import threading

                g_sleep, g_cnt = ctypes.c_double(0.0), ctypes.c_double(0.0)
                for _ in range(40):
                    step()
                time.sleep(0.01)
                ok_scan = N.lib().rf_probe_core_clock(local_rank, 20000, ctypes.byref(g_sleep), ctypes.byref(g_cnt)) == N.RF_OK
                clk_scan = g_cnt.value
                # package power: rocm-smi reports an average over a fraction of a second, so keep the scans coming for ~0.7 s
                for _ in range(int(min(2000, max(40, 700.0 / max(ms_per_step, 0.05))))):
                    step()
                time.sleep(0.25)
                power_w = gpu_power_watts(local_rank)  # best effort
                finish_exchange()
                torch.cuda.synchronize()
                r2 = ctypes.c_double(0.0)
                th = threading.Thread(target=lambda: N.lib().rf_probe_issue_rate(metric_id, args.query_len, 0, local_rank, 8, ctypes.byref(r2)))
                th.start()
                time.sleep(0.01)
                ok_probe = N.lib().rf_probe_core_clock(local_rank, 10000, ctypes.byref(g_sleep), ctypes.byref(g_cnt)) == N.RF_OK
                clk_probe = g_cnt.value
                th.join()
                if ok_scan and ok_probe and clk_scan > 0 and clk_probe > 0:
                    ib = result["roofline"]["issue_bound"]
                    ib["core_clock_ghz"] = {"under_scan": round(clk_scan, 3), "under_probe": round(clk_probe, 3),
                                            "how": "rf_probe_core_clock: s_memtime / s_memrealtime of one wavefront beside the queued scans / the probe kernel"}
                    ib["ceiling_at_scan_clock"] = round(ceiling * clk_scan / clk_probe, 3)
                    # (a RATIO, not a fraction of a bound: the probe's rate rescaled linearly to the scan's clock is an estimate, and the kernel can
                    # land a percent above it -- VERDICT r3 weak #10)
                    ib["ratio_to_probe_at_scan_clock"] = round(per_gpu / (ceiling * clk_scan / clk_probe), 4)
                    if power_w:
                        ib["package_power_w_under_scan"] = power_w
            except Exception as exc:  # a measurement aid: never fails the bench line
                result["roofline"]["issue_bound"]["core_clock_ghz"] = {"skipped": str(exc)[:200]}

    ib = result["roofline"].get("issue_bound")
    if ib and (pack6 or pack6_ragged):
        #  rf_probe_issue_rate runs the COMPILED 8-bit LcsState column; these runs execute the whole-kernel asm scans over the 6-bit payload
        # (rf_stream_asm.hip stream_lcs6[n]_*), whose column was measured register-only at 21.0 cycles (64-bit words) / 11.0 cycles (32-bit words: queries <= 32) --
        # profiles/lcs_cycles_r05.txt.  The ceiling that applies is that column at the clock THIS run sampled under the scan (the package power cap holds it near
        # 1.84 GHz here); beside it: what the scan moves against what a pure streaming read achieves on this part (~6.3 TB/s, tools/membw.hip).
        col_cycles = 11.0 if args.query_len <= 32 else 21.0
        clk = (ib.get("core_clock_ghz") or {}).get("under_scan")
        if clk:
            import ctypes as _ct

            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            asm_ceiling = cus * 4 * clk / col_cycles * 64.0 / max(ln, 1)  # SIMDs x GHz / cycles per wave-column = wave-columns per ns -> Gpairs/s at this length
            per_gpu = gpairs / world
            ib["asm_column"] = {"cycles_per_column": col_cycles, "source": "profiles/lcs_cycles_r05.txt (register-only microbenchmark of the 6-bit asm column)",
                                "ceiling_at_scan_clock": round(asm_ceiling, 2), "frac": round(per_gpu / asm_ceiling, 4)}
            moved = result["roofline"].get("moved", {}).get("achieved")
            if moved:
                ib["asm_column"]["moved_frac_of_streaming_read"] = round(moved / 6300.0, 4)
            # the compiled column's figures above describe a kernel this run did not execute: the asm column decides the label
            ib["frac"] = ib["asm_column"]["frac"]
            ib.pop("ratio_to_probe_at_scan_clock", None)
            if ib["asm_column"]["frac"] < 0.8 and moved and moved / 6300.0 >= 0.75:
                result["roofline"]["bound_note"] = "HBM on the 6-bit payload: the scan moves >= 0.75 of what a streaming read achieves on this part and sits below 0.8 of its column's issue ceiling"
    if ib and max(ib.get("frac", 0.0), ib.get("ratio_to_probe_at_scan_clock", 0.0)) >= 0.8:
        # a kernel this JSON itself shows at >= 0.8 of its measured VALU-issue ceiling is issue-bound, not HBM-bound.  `frac` stays
        # achieved / HBM peak (the north star's yardstick, and the contract's); `issue_bound.frac` is the fraction of the bound that binds.
        result["roofline"]["bound"] = "valu_issue"
        result["roofline"]["bound_note"] = "frac = algorithmic bytes / kernel time / HBM peak (the north star's yardstick); the binding limit is VALU issue: issue_bound.frac"
        if ib.get("package_power_w_under_scan") and (ib.get("core_clock_ghz") or {}).get("under_scan", 9.9) < 2.0:
            pw = ib["package_power_w_under_scan"]
            drawn = f"{pw.get('now')} W of a {pw.get('cap')} W cap" if isinstance(pw, dict) else f"{pw} W"
            result["roofline"]["bound_note"] += f"; at a core clock the package power cap holds to {ib['core_clock_ghz']['under_scan']} GHz ({drawn} drawn)"

    if last_topk[0] is not None:
        # keys are (distance << 32 | global index); distances < 2^31 so the signed sort above is the unsigned order,
        # and empty entries (all ones = -1) are dropped here
        keys = [int(x) for x in last_topk[0].cpu().tolist() if x not in (-1, 2**63 - 1)]
        result["config"]["topk_found"] = len(keys)
        result["config"]["topk_best"] = [[k >> 32, k & 0xFFFFFFFF] for k in sorted(keys)[:4]]
        import zlib

        # the same for every world size when the corpus is ONE logical corpus (c5): the scaling runs must agree on it
        result["config"]["topk_checksum"] = zlib.crc32(np.array(sorted(keys), dtype=np.uint64).tobytes())
        if args.mode == "many" and not is_f64:
            # oracle-free consistency of the exchange at any world size: rank 0 ranks its OWN shard from the per-candidate
            # distances the same pass wrote (torch on the device) -- each of its k best must be in the merged list or lose to
            # the merged list's worst entry, and every merged entry that points into this shard must carry that candidate's distance
            d0 = out[:n].to(torch.int64) & 0xFFFFFFFF
            mine = (d0 << 32) | (index_base + torch.arange(n, device=dev, dtype=torch.int64))
            mine = mine[d0 != 0xFFFFFFFF]
            local_best = torch.topk(mine, min(args.topk, mine.numel()), largest=False).values.cpu().tolist()
            worst = max(keys) if len(keys) == args.topk else 2**63 - 1
            bad = sum(1 for x in local_best if x not in keys and x < worst)
            bad += sum(1 for x in keys if index_base <= (x & 0xFFFFFFFF) < index_base + n and x not in local_best)
            result["config"]["exchange_selfcheck"] = {"rank0_shard_entries_checked": len(local_best), "inconsistent": bad}
        if c5 and not args.no_cpu_baseline:  # (cheap: the oracle on ~1000 rows; kept at every world size -- it is the check that the shards add up)
            # parity of the merged top-k: the only candidates within the cutoff are planted near-duplicates (a random
            # len-64 alphanumeric string is ~55 edits from the query), and a planted row depends only on its global
            # index, so rank 0 re-creates ALL of them on the host and lets the oracle rank them
            from oracle import oracle as o

            pidx = synth.planted_indices(0, args.total_candidates, args.plant_every)
            prow = np.stack([synth.planted_row(q, ln, int(i)) for i in pidx]) if len(pidx) else np.zeros((0, ln), np.uint8)
            d = o.levenshtein.BatchComparator(q).rows(N.OP_DISTANCE, prow, nthreads=1, score_cutoff=args.cutoff)
            exp = sorted((int(dv) << 32) | int(i) for dv, i in zip(d, pidx) if dv != np.uint64(2**64 - 1))[: args.topk]
            result["parity"] = {"checked": int(len(pidx)), "mismatches": int(sorted(keys) != exp),
                                "what": f"merged top-{args.topk} keys vs the oracle's (distance, global index) ranking of all {len(pidx)} planted near-duplicates"}
    if host_sample is not None:
        result["cpu_baseline"] = cpu_baseline(args, q, host_sample)
    if ragged_sample is not None:
        result["cpu_baseline"] = cpu_baseline(args, q, None, ragged=ragged_sample)
    if ragged_sample is not None and args.mode == "many":
        from oracle import oracle as o

        torch.cuda.synchronize()
        op = N.OP_SIMILARITY if is_f64 else N.OP_DISTANCE
        if args.slot_order:  # the caller's own permutation: slot -> original index, once per corpus
            slot_index = torch.from_numpy(corpus.slot_index().astype(np.int64)).to(dev)
            real = slot_index != 0xFFFFFFFF
            back = torch.empty(n, dtype=out.dtype, device=dev)
            back[slot_index[real]] = out[:n_out][real]
            got_all = back.cpu().numpy()
        else:
            got_all = out[:n].cpu().numpy()
        mism, checked = 0, 0
        for (d_, o_), sel in ((ragged_sample, slice(0, len(ragged_sample[1]) - 1)), (ragged_strided, slice(0, n, 1009))):
            m = min(len(o_) - 1, 2_000_000)
            exp = getattr(o, args.metric).BatchComparator(q).many(op, d_[: int(o_[m])], o_[: m + 1], nthreads=os.cpu_count() or 1, score_cutoff=args.cutoff)
            got = got_all[sel][:m]
            if is_f64:
                bad = ~((got == exp) | (np.isnan(got) & np.isnan(exp)))
            else:
                bad = got.view(np.uint32) != np.where(exp == np.uint64(2**64 - 1), np.uint32(0xFFFFFFFF), exp.astype(np.uint32))
            mism += int(bad.sum())
            checked += m
        result["parity"] = {"checked": checked, "mismatches": mism, "what": f"first candidates + every 1009-th of all {n}, vs oracle/"}
    if filt is not None:
        result["config"]["filter_count"] = int(filt["count"].value)
        if host_sample is not None or ragged_sample is not None:
            # parity of the compact pairs, in the same run: the pairs whose index lies in the first 2 M candidates must be exactly the oracle's Somes there,
            # and every 1009-th candidate of the whole corpus must be in the list if and only if the oracle gives it a value (with that value)
            from oracle import oracle as o

            torch.cuda.synchronize()
            m = int(min(filt["count"].value, args.capacity))
            gi = filt["idx"][:m].cpu().numpy().astype(np.int64)
            gv = filt["val"][:m].cpu().numpy()
            gv = gv.view(np.uint32) if not is_f64 else gv
            op = N.OP_SIMILARITY if is_f64 else N.OP_DISTANCE
            ob = getattr(o, args.metric).BatchComparator(q)
            mism, checked = 0, 0
            ordered = bool(np.all(np.diff(gi) > 0)) if m > 1 else True
            if host_sample is not None:
                chk = min(len(host_sample), 2_000_000)
                legs = [(ob.rows(op, host_sample[:chk], nthreads=os.cpu_count() or 1, score_cutoff=args.cutoff), np.arange(chk)),
                        (ob.rows(op, host_strided, nthreads=os.cpu_count() or 1, score_cutoff=args.cutoff), np.arange(0, n, 1009))]
            else:
                chk = min(len(ragged_sample[1]) - 1, 2_000_000)
                legs = [(ob.many(op, ragged_sample[0][: int(ragged_sample[1][chk])], ragged_sample[1][: chk + 1], nthreads=os.cpu_count() or 1, score_cutoff=args.cutoff), np.arange(chk)),
                        (ob.many(op, ragged_strided[0], ragged_strided[1], nthreads=os.cpu_count() or 1, score_cutoff=args.cutoff), np.arange(0, n, 1009))]
            for exp, index_of in legs:
                keep = ~np.isnan(exp) if is_f64 else exp != np.uint64(2**64 - 1)
                want_i = index_of[keep]
                want_v = exp[keep] if is_f64 else exp[keep].astype(np.uint32)
                sel = np.isin(gi, index_of)
                if filt["count"].value <= args.capacity:
                    mism += int(not (np.array_equal(gi[sel], want_i) and np.array_equal(gv[sel], want_v)))
                else:  # the list is a subset: what it holds must be right
                    pos = np.minimum(np.searchsorted(want_i, gi[sel]), max(len(want_i) - 1, 0))
                    mism += int(not (len(want_i) > 0 or not sel.any()) or not (np.array_equal(want_i[pos], gi[sel]) and np.array_equal(want_v[pos], gv[sel])))
                checked += len(index_of)
            result["parity"] = {"checked": int(checked), "mismatches": mism + int(not ordered),
                                "what": f"the (index, score) pairs inside the first {chk} candidates + every 1009-th of all {n} == the oracle's Somes there; indices ascending"}
    if host_sample is not None and args.mode == "many":
        # parity on the sample, in the same run
        from oracle import oracle as o

        torch.cuda.synchronize()
        chk = min(len(host_sample), 2_000_000)
        op = N.OP_SIMILARITY if is_f64 else N.OP_DISTANCE
        mism = 0
        for j in range(nq):
            kw = {"weights": weights} if weights else {}
            exp = getattr(o, args.metric).BatchComparator(queries[j]).rows(op, host_sample[:chk], nthreads=os.cpu_count() or 1, score_cutoff=args.cutoff, **kw)
            if is_f64:
                got = out[j * n : j * n + chk].cpu().numpy()
                bad = ~((got == exp) | (np.isnan(got) & np.isnan(exp)))
            else:
                got = out[j * n : j * n + chk].cpu().numpy().view(np.uint32)
                bad = got != np.where(exp == np.uint64(2**64 - 1), np.uint32(0xFFFFFFFF), exp.astype(np.uint32))
            mism += int(bad.sum())
        # ... and on the strided sample over the whole corpus (first query)
        kw = {"weights": weights} if weights else {}
        exp = getattr(o, args.metric).BatchComparator(queries[0]).rows(op, host_strided, nthreads=os.cpu_count() or 1, score_cutoff=args.cutoff, **kw)
        if is_f64:
            got = out[:n:1009].cpu().numpy()
            bad = ~((got == exp) | (np.isnan(got) & np.isnan(exp)))
        else:
            got = out[:n:1009].cpu().numpy().view(np.uint32)
            bad = got != np.where(exp == np.uint64(2**64 - 1), np.uint32(0xFFFFFFFF), exp.astype(np.uint32))
        result["parity"] = {"checked": int(chk) * nq + len(host_strided), "mismatches": mism + int(bad.sum()),
                            "what": f"first {chk} candidates + every 1009-th of all {n}, vs oracle/"}
    if world > 1 or force_dist:
        dist.destroy_process_group()
    is_headline = (args.config in (None, "c2") and args.metric == "levenshtein" and n == 100_000_000 and args.cand_len == 64 and args.query_len == 64
                   and args.cutoff is None and not weights and not args.ragged and args.mode == "many" and nq == 1 and world == 1 and not force_dist
                   and args.head_share == 0 and args.zipf == 0)
    if args.extras == "on" or (args.extras == "auto" and is_headline):
        # the headline's numbers are final at this point; give the GPU memory back before the legs start their own processes
        del corpus, out
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        result["extra_configs"] = extra_configs(args)
        for e in result["extra_configs"]:  # one short string per leg as well (a flat key survives any summarising of the line)
            result["config"]["extra_" + e["name"]] = e.get("summary", e.get("error", "?"))[:120]
    # RCCL prints its version banner through C stdio, which flushes after Python's buffer when stdout is a pipe:
    # drain it first so the JSON line is the last line of the output
    import ctypes

    ctypes.CDLL(None).fflush(None)
    print(json.dumps(result), flush=True)


# The other BASELINE.json configs, as bench.py command lines (the headline -- configs[1] -- is the line itself).  Each leg is this very
# script in a process of its own: same step definition, same HIP-event timing, same oracle parity leg, a short CPU baseline.
EXTRA_LEGS = [
    ("c1_q32_10k_ragged", "configs[0] on the GPU: query 32 x 10 k candidates len <= 64 (launch-bound at this size; the CPU figure beside it is the config itself)",
     ["--query-len", "32", "--candidates", "10000", "--ragged"]),
    ("c3_levenshtein_q256_10M", "configs[2]: query 256 x 10 M len-256 candidates (multi-word Hyyro)", ["--query-len", "256", "--cand-len", "256", "--candidates", "10000000"]),
    ("c4_indel_100M", "configs[3]: Indel over the 100 M corpus", ["--metric", "indel"]),
    ("c4_jaro_winkler_100M", "configs[3]: Jaro-Winkler over the 100 M corpus (f64 similarity per candidate)", ["--metric", "jaro_winkler"]),
    ("c5_1B_cutoff3_top16_world1", "configs[4] at N = 1: 1 B candidates, score_cutoff 3, top-16, RCCL all-gather + merge every step", ["--config", "c5"]),
]


def extra_configs(args):
    import subprocess

    legs = []
    scale = float(os.environ.get("RF_BENCH_EXTRA_SCALE", "1"))  # (tests: the same legs over 1/1000 of the candidates)
    for name, what, flags in EXTRA_LEGS:
        if scale != 1.0:
            flags = list(flags)
            if "--config" in flags:
                flags += ["--total-candidates", str(max(100_000, int(1_000_000_000 * scale))), "--plant-every", "10000"]
            elif "--candidates" in flags:
                i = flags.index("--candidates")
                flags[i + 1] = str(max(5_000, int(int(flags[i + 1]) * scale)))
            else:
                flags += ["--candidates", str(max(100_000, int(100_000_000 * scale)))]
        steps = args.extra_steps * (25 if name.startswith("c1_") else 1)  # (7 us steps: a handful of them measures the clock ramp, not the call)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", "5", "--extras", "off", "--traffic", "off",
               "--cpu-seconds", "2", "--settle-ms", "100", *flags]
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env["MASTER_PORT"] = "29577"
        t0 = time.perf_counter()
        leg = {"name": name, "config": what, "cmd": "python bench.py " + " ".join(cmd[2:])}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=240)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}")
            d = json.loads(lines[-1])
            rl = d["roofline"]
            leg.update({"value": d["value"], "unit": d["unit"], "steps": d["steps"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"],
                        "roofline": {k: rl[k] for k in ("bound", "bound_note", "achieved", "peak", "unit", "frac", "kernel_ms", "algorithmic_bytes_per_pair", "survey_8d", "moved") if k in rl},
                        "parity": d.get("parity"), "cpu_baseline": {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")} if "cpu_baseline" in d else None})
            par = d.get("parity") or {}
            leg["summary"] = (f"{d['value']} Gpairs/s, {d['ms_per_step']} ms/step, {rl['frac']} of HBM peak"
                              + (f" ({rl['survey_8d']['frac']} by SURVEY 8(d) bytes)" if rl.get("survey_8d", {}).get("frac") != rl["frac"] else "")
                              + f", parity {par.get('mismatches', '?')}/{par.get('checked', '?')}")
        except Exception as exc:  # a leg that fails is reported as failed, it never takes the headline with it
            leg["error"] = str(exc)[:400]
        leg["wall_s"] = round(time.perf_counter() - t0, 1)
        legs.append(leg)
    return legs


def gpu_power_watts(device):
    """Socket power as rocm-smi reports it right now (None when rocm-smi is missing or says nothing parsable)."""
    import re
    import shutil
    import subprocess

    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        r = subprocess.run([exe, "-d", str(device), "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=10)
    except (OSError, subprocess.SubprocessError):
        return None
    cur = re.search(r"(?:Current|Average) Socket Graphics Package Power \(W\):\s*([0-9.]+)", r.stdout)
    cap = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", r.stdout)
    if not cur:
        return None
    return {"now": float(cur.group(1)), **({"cap": float(cap.group(1))} if cap else {})}


def traffic_in_run(args, kernel_ms):
    """HBM bytes per launch of the dominant kernel, measured NOW: two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do
    not fit one pass, MI355X_MICROARCH.md counters table) over this very command at --steps 3, counters of the kernel with the
    largest total, gfx950 correction read = 2 x FETCH_SIZE KiB (same guide, HBM section).  None when rocprofv3 is not on the box, the
    passes fail or time out, or this process is itself one of those passes."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("RF_BENCH_IN_PMC") == "1" or args.gpus != 1:
        return None
    flags = [a for a in sys.argv[1:]]
    for drop, has_val in (("--steps", True), ("--warmup", True), ("--extras", True), ("--settle-ms", True), ("--cpu-seconds", True), ("--traffic", True),
                          ("--no-cpu-baseline", False)):
        while drop in flags:
            i = flags.index(drop)
            del flags[i : i + (2 if has_val else 1)]
    cmd_tail = [sys.executable, os.path.abspath(__file__), *flags, "--steps", "3", "--warmup", "1", "--extras", "off", "--no-cpu-baseline", "--settle-ms", "0"]
    env = dict(os.environ, RF_BENCH_IN_PMC="1", TMPDIR="/tmp")
    got = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir="/tmp") as w:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            try:
                r = subprocess.run([exe, "--pmc", counter, "-d", os.path.join(w, counter), "-o", "p", "--", *cmd_tail], cwd="/tmp", env=env, capture_output=True,
                                   text=True, timeout=150)
                db = os.path.join(w, counter, "p_results.db")
                if r.returncode != 0 or not os.path.exists(db):
                    return None
                cur = sqlite3.connect(db).cursor()
                rows = [r for r in cur.execute("select kernel_name, avg(value), count(*), sum(value) from counters_collection where counter_name = ? and "
                                               "kernel_name like '%rf::%' group by kernel_name", (counter,)) if "probe" not in r[0] and "core_clock" not in r[0]]
            except (subprocess.SubprocessError, OSError, sqlite3.Error):
                return None
            if not rows:
                return None
            got[counter] = {name: (avg, cnt, tot) for name, avg, cnt, tot in rows}
    # the step's kernels: everything of the library that ran once per step or more (3 timed + 1 warm-up + settle-free first call = >= 4 dispatches)
    names = [k for k, (_, cnt, _) in got["FETCH_SIZE"].items() if cnt >= 4 and k in got["WRITE_SIZE"]]
    if not names:
        return None
    steps_seen = min(got["FETCH_SIZE"][k][1] for k in names)
    read = sum(2.0 * got["FETCH_SIZE"][k][2] * 1024.0 for k in names) / steps_seen
    write = sum(got["WRITE_SIZE"][k][2] * 1024.0 for k in names) / steps_seen
    return {"bytes_per_launch": read + write, "read": read, "write": write, "kernels": sorted(n[:60] for n in names),
            "source": f"in-run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this command at --steps 3, {time.perf_counter() - t0:.0f} s; "
                      "read = 2 x FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB; summed over the step's kernels"}


def measured_traffic(args, n, kernel_ms):
    """HBM bytes per launch: measured in this run when rocprofv3 is on the box (traffic_in_run), else from the committed rocprofv3 PMC
    passes of this exact workload, marked "source": "committed ..."
    (profiles/traffic.json, written by tools/rocpd_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs with the
    gfx950 2x FETCH_SIZE correction of MI355X_MICROARCH.md).  None when no profile matches the workload -- or when the
    kernel has changed since the counters were collected (its steady-state time then differs from this run's by more
    than 15 %): a stale counter is not evidence."""
    mode = getattr(args, "traffic", "auto")
    if mode == "on" or (mode == "auto" and n >= 10_000_000 and getattr(args, "config", None) != "c5"):
        try:
            live = traffic_in_run(args, kernel_ms)
        except Exception:  # a measurement aid: never fails the bench line
            live = None
        if live is not None:
            return live
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except OSError:
        return None
    key = f"{args.metric}:q{args.query_len}:n{n}:l{args.cand_len}:cut{args.cutoff}:{args.mode}" + (":ragged" if getattr(args, "ragged", False) else "")
    e = table.get(key)
    if e is None:
        return None
    at = e.get("kernel_us_at_collection")
    if at and abs(at / 1e3 - kernel_ms) > 0.15 * kernel_ms:
        return None
    return {"bytes_per_launch": e["total"], "read": e["read"], "write": e["write"], "source": "committed: " + e.get("source", ""),
            **({"kernel_us_at_collection": at} if at else {})}


def cpu_baseline(args, q, host_sample, ragged=None):
    """The oracle (kind = "port": the C restatement of the reference's BatchComparator loop) timed on the
    GPU box's host cores over a bounded prefix of the SAME corpus (fixed-length rows, or ragged (data, offsets))."""
    from oracle import oracle as o
    from rapidfuzz_rs_amd import _native as N

    bc = getattr(o, args.metric).BatchComparator(q)
    OP = N.OP_SIMILARITY if args.metric in ("jaro", "jaro_winkler") else N.OP_DISTANCE
    kw = {"weights": tuple(int(x) for x in args.weights.split(","))} if getattr(args, "weights", None) else {}
    if ragged is not None:
        class _Prefix:  # host_sample[:m] of a ragged sample
            def __init__(self, data, offsets):
                self.data, self.offsets = data, offsets
            def __len__(self):
                return len(self.offsets) - 1
            def __getitem__(self, sl):
                m = min(sl.stop, len(self))
                return _Prefix(self.data[: int(self.offsets[m])], self.offsets[: m + 1])
        host_sample = _Prefix(*ragged)
        run = lambda part, nthreads: bc.many(OP, part.data, part.offsets, nthreads=nthreads, score_cutoff=args.cutoff, **kw)
    else:
        run = lambda part, nthreads: bc.rows(OP, part, nthreads=nthreads, score_cutoff=args.cutoff, **kw)
    probe = host_sample[:200_000]
    t0 = time.perf_counter()
    run(probe, 1)
    rate = len(probe) / (time.perf_counter() - t0)
    n1 = int(min(len(host_sample), max(200_000, rate * args.cpu_seconds * 0.5)))
    t0 = time.perf_counter()
    run(host_sample[:n1], 1)
    t1 = time.perf_counter() - t0
    try:
        cores = len(os.sched_getaffinity(0))  # the hardware threads this process may run on ...
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    quota = None
    try:  # ... and the CPU time its cgroup may use: the GPU boxes of this pool show 256 hardware threads but cpu.max = 16 CPUs, which
        # is why the "all cores" figure of rounds 1-2 read 13-17 x one core
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
            cores = min(cores, quota)
    except (OSError, ValueError):
        pass
    # all hardware threads: the whole host sample per call, repeated until >= 2 s of wall time have been measured (a
    # single 0.4 s call is mostly thread start-up and first-touch page faults; the first call is a warm-up and not counted)
    nall = len(host_sample)
    run(host_sample[:nall], cores)
    tall, reps = 0.0, 0
    while reps < 2 or (tall < min(2.0, args.cpu_seconds) and reps < 64):
        t0 = time.perf_counter()
        run(host_sample[:nall], cores)
        tall += time.perf_counter() - t0
        reps += 1
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(n1 / t1 / 1e9, 6),
        "unit": "Gpairs/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n1} candidates of the same corpus, oracle/ (C restatement of the reference's single-threaded BatchComparator loop), 1 thread, {t1:.1f} s",
        "all_cores": {"value": round(nall * reps / tall / 1e9, 6), "cores": cores, "sample": f"first {nall} candidates x {reps} passes, {tall:.1f} s, {cores} threads"
                      + (f" (cgroup cpu.max = {quota} CPUs of {os.cpu_count()} hardware threads)" if quota else "")},
        "cpu_model": model,
    }


if __name__ == "__main__":
    main()
