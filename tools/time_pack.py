"""What a corpus costs to build: rf_corpus_pack -- ragged HOST bytes + offsets -> a device-resident, scannable corpus -- on the device packer
(rf_pack_ragged.hip, the default from 65536 candidates on) and on the host packer (RF_DEVICE_PACK_MIN=0: round 5's counting sort + byte scatter + one upload), with the
library's own phase timings (RF_PACK_TIMING=1, stderr).  Each packer runs in a process of its own (the switch is read once); the input is generated once and shared.
    python tools/time_pack.py [candidates]          (profiles/pack_r06.txt is this script's output for 100 M)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch  # noqa: F401  (pay the one-off import before anything is timed)

    import rapidfuzz_rs_amd as rf
    from rapidfuzz_rs_amd import _native as N

    for kind in ("ragged", "rows64"):
        data, offsets = np.load(os.path.join(sys.argv[3], kind + "_data.npy")), np.load(os.path.join(sys.argv[3], kind + "_off.npy"))
        torch.cuda.synchronize()
        for rep in range(2):  # (the first call also pays the pinned staging buffers / the page faults of the host image)
            t0 = time.perf_counter()
            c = rf.Corpus.from_ragged(data, offsets)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{kind:7s} n={len(offsets) - 1} payload={data.nbytes / 1e9:.2f} GB  rf_corpus_pack call {rep}: {dt:.3f} s  ({data.nbytes / dt / 1e9:.2f} GB/s of payload)", flush=True)
            if rep == 0:
                q = bytes(data[:40])
                t0 = time.perf_counter()
                rf.distance.levenshtein.BatchComparator(q).many(N.OP_DISTANCE, c, score_cutoff=3)
                t1 = time.perf_counter()
                rf.distance.levenshtein.BatchComparator(q).many(N.OP_DISTANCE, c, score_cutoff=3)
                t2 = time.perf_counter()
                print(f"        first cutoff-3 scan {1e3 * (t1 - t0):.1f} ms (builds the head plane, tile lists), second {1e3 * (t2 - t1):.1f} ms (host results both times)", flush=True)
            del c
    sys.exit(0)

from rapidfuzz_rs_amd.utils import synth  # noqa: E402

with tempfile.TemporaryDirectory(dir="/tmp") as d:
    data, offsets = synth.ragged_host(n, 64, seed=2, min_len=1)
    np.save(os.path.join(d, "ragged_data.npy"), data)
    np.save(os.path.join(d, "ragged_off.npy"), offsets)
    rows = synth.rows_host(n, 64, seed=1)
    np.save(os.path.join(d, "rows64_data.npy"), rows.reshape(-1))
    np.save(os.path.join(d, "rows64_off.npy"), np.arange(0, n * 64 + 1, 64, dtype=np.uint64))
    del data, offsets, rows
    for label, env in (("device packer (default)", {}), ("host packer (RF_DEVICE_PACK_MIN=0)", {"RF_DEVICE_PACK_MIN": "0"})):
        print(f"==== {label}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "child", d], env=dict(os.environ, RF_PACK_TIMING="1", **env), capture_output=True, text=True)
        print(r.stdout.rstrip())
        print("\n".join(ln for ln in r.stderr.splitlines() if ln.startswith("[rf_corpus_pack]")), flush=True)
