"""Oracle-free self-check of the HIP path against a committed fixture (tests/golden/gpu_selfcheck.json: inputs and the
values the pinned oracle gave for them, written by tests/golden/make_gpu_selfcheck.py).  Needs a GPU and nothing under
oracle/."""
from __future__ import annotations

import json
import math
import os

import numpy as np

_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "gpu_selfcheck.json")


def run(device: int = 0, fixture: str = _FIXTURE) -> int:
    """Runs every expectation of the fixture through the device; returns the number of values checked, raises on a mismatch."""
    import rapidfuzz_rs_amd as rf
    from rapidfuzz_rs_amd import _native as N

    fx = json.load(open(fixture))
    q = fx["query"].encode("latin-1")
    corpus = rf.Corpus.from_list([c.encode("latin-1") for c in fx["candidates"]], device=device)
    ops = {"distance": N.OP_DISTANCE, "similarity": N.OP_SIMILARITY, "normalized_similarity": N.OP_NORMALIZED_SIMILARITY}
    checked = 0
    for key, exp in fx["expected"].items():
        metric, op, cutoff = key.split(":")
        kw = {} if cutoff == "None" else {"score_cutoff": float(cutoff) if "." in cutoff else int(cutoff)}
        got = getattr(rf.distance, metric).BatchComparator(q).many(ops[op], corpus, **kw)
        for i, (g, e) in enumerate(zip(got.tolist(), exp)):
            if got.dtype == np.uint32:
                ok = (e is None and g == N.NONE_U32) or (e is not None and g == e)
            else:
                ok = (e is None and math.isnan(g)) or (e is not None and g == float.fromhex(e))
            if not ok:
                raise AssertionError(f"{key}: candidate {i}: device {g!r}, fixture {e!r}")
            checked += 1
    return checked
