"""Times the HOST packing path (rf_corpus_pack: ragged host bytes -> device-resident corpus) and the u32 one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (pay the one-off import before anything is timed)
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd.utils import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
for kind in ("rows64", "ragged<=64"):
    if kind == "rows64":
        rows = synth.rows_host(n, 64, seed=1)
        data, offsets = rows.reshape(-1), np.arange(0, n * 64 + 1, 64, dtype=np.uint64)
    else:
        data, offsets = synth.ragged_host(n, 64, seed=2)
    t0 = time.time(); c = rf.Corpus.from_ragged(data, offsets); dt = time.time() - t0
    print(f"{kind}: n={n} payload={data.nbytes/1e9:.2f} GB  rf_corpus_pack {dt:.2f} s  ({data.nbytes/dt/1e9:.2f} GB/s)")
    del c
    if kind == "ragged<=64":
        w = data.astype(np.uint32) + 0x400
        t0 = time.time(); c = rf.Corpus.from_ragged_u32(w, offsets); dt = time.time() - t0
        print(f"{kind} as u32: rf_corpus_pack_u32 {dt:.2f} s  ({w.nbytes/4/dt/1e9:.2f} Gsym/s)")
