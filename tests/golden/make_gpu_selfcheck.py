#!/usr/bin/env python3
"""Writes tests/golden/gpu_selfcheck.json: one query, 256 ragged candidates and the values the ORACLE gives for them (the
oracle is pinned to the reference's own known answers, tests/test_oracle_known_answers.py).  With this file a GPU box can
check the HIP path without building or loading anything under oracle/: __graft_entry__.smoke() and
tests/test_gpu_parity.py::test_gpu_selfcheck_fixture compare against it.  f64 values are stored as IEEE-754 hex strings
(bit-exact).  Re-run after changing the generator; the committed file is data, not code."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402
from rapidfuzz_rs_amd import _native as N  # noqa: E402  (op ids only)
from rapidfuzz_rs_amd.utils import synth  # noqa: E402

q = synth.query(48, 20260928)
data, offsets = synth.ragged_host(256, 80, seed=20260929)
# a few structured candidates: the query itself, a prefix, a near-duplicate, an empty one
cands = [bytes(data[int(offsets[i]):int(offsets[i + 1])]) for i in range(256)]
cands[0], cands[1], cands[2], cands[3] = q, q[:20], q[:10] + b"!!" + q[12:], b""
data = np.frombuffer(b"".join(cands), dtype=np.uint8)
offsets = np.zeros(257, dtype=np.uint64)
offsets[1:] = np.cumsum([len(c) for c in cands])
out = {"query": q.decode("latin-1"), "candidates": [c.decode("latin-1") for c in cands], "expected": {}}
for name, cutoff in (("levenshtein", None), ("levenshtein", 30), ("indel", None), ("lcs_seq", None), ("osa", None)):
    v = getattr(o, name).BatchComparator(q).many(N.OP_DISTANCE, data, offsets, score_cutoff=cutoff)
    out["expected"][f"{name}:distance:{cutoff}"] = [None if x == np.uint64(2**64 - 1) else int(x) for x in v]
for name in ("jaro", "jaro_winkler"):
    v = getattr(o, name).BatchComparator(q).many(N.OP_SIMILARITY, data, offsets)
    out["expected"][f"{name}:similarity:None"] = [float(x).hex() for x in v]
v = o.levenshtein.BatchComparator(q).many(N.OP_NORMALIZED_SIMILARITY, data, offsets, score_cutoff=0.5)
out["expected"]["levenshtein:normalized_similarity:0.5"] = [None if np.isnan(x) else float(x).hex() for x in v]
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "gpu_selfcheck.json"), "w"), indent=0)
print("wrote gpu_selfcheck.json:", {k: len(v) for k, v in out["expected"].items()})
