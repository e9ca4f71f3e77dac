"""A/B timing of one scan variant with the library named by RF_LIB (tools/ab.sh runs it alternately for two builds on the
same box: boxes differ by several percent, so only same-session pairs are comparable).  Not part of the product."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rapidfuzz_rs_amd as rf
from rapidfuzz_rs_amd import _native as N
from rapidfuzz_rs_amd.utils import synth

what = sys.argv[1] if len(sys.argv) > 1 else "lev64"
n = int(os.environ.get("AB_N", 100_000_000))
cfg = {
    "lev64": ("levenshtein", 64, 64, {}), "lev32": ("levenshtein", 32, 64, {}), "indel": ("indel", 64, 64, {}), "osa": ("osa", 64, 64, {}),
    "lev256": ("levenshtein", 256, 256, {}), "lev128": ("levenshtein", 128, 128, {}), "lev192": ("levenshtein", 192, 192, {}), "indel256": ("indel", 256, 256, {}), "levrag": ("levenshtein", 64, 64, {}), "levragc3": ("levenshtein", 64, 64, {"score_cutoff": 3}), "indelragc12": ("indel", 64, 64, {"score_cutoff": 12}), "indelrag": ("indel", 64, 64, {}), "osarag": ("osa", 64, 64, {}), "lev32rag": ("levenshtein", 32, 64, {}), "indel128": ("indel", 128, 64, {}), "lev64c3": ("levenshtein", 64, 64, {"score_cutoff": 3}), "jw": ("jaro_winkler", 64, 64, {}),
    "jaro": ("jaro", 64, 64, {}), "jwrag": ("jaro_winkler", 64, 64, {}), "jarorag": ("jaro", 64, 64, {}), "jwc9": ("jaro_winkler", 64, 64, {"score_cutoff": 0.9}),
    "jwragc9": ("jaro_winkler", 64, 64, {"score_cutoff": 0.9}), "lev256c8": ("levenshtein", 256, 256, {"score_cutoff": 8}),
    "indel57": ("indel", 60, 57, {}), "indel36": ("indel", 36, 36, {}), "indel32": ("indel", 32, 64, {}), "indel32rag": ("indel", 32, 64, {}), "lev320": ("levenshtein", 320, 320, {}), "lev512": ("levenshtein", 512, 512, {}), "lev256c200": ("levenshtein", 256, 256, {"score_cutoff": 200}),
}
import re as _re
_m = _re.fullmatch(r"lev64c(\d+)", what.split("+")[0])
_i = _re.fullmatch(r"indelc(\d+)", what.split("+")[0])
cfg = ("levenshtein", 64, 64, {"score_cutoff": int(_m.group(1))}) if _m else (("indel", 64, 64, {"score_cutoff": int(_i.group(1))}) if _i else cfg[what.split("+")[0]])
metric, qlen, clen, kw = cfg
if clen == 256 and qlen == 256:
    n //= 10
if clen > 256:
    n //= 40
q = synth.query(qlen, 0xC0FFEE02)
if "rag" in what.split("+")[0]:  # ragged corpus: lengths uniform in [20, 64] -> exact tiles of 45 lengths, most with a partial last chunk
    import numpy as np
    n = int(os.environ.get("AB_N", 20_000_000))
    rng = np.random.default_rng(5)
    lens = rng.integers(int(os.environ.get('AB_MINLEN', 20)), 65, size=n).astype(np.uint64)
    offsets = np.zeros(n + 1, dtype=np.uint64); offsets[1:] = np.cumsum(lens)
    data = synth.ALNUM[rng.integers(0, 62, size=int(offsets[-1]))]
    corpus = rf.Corpus.from_ragged(data, offsets)
else:
    rows = synth.rows_device(n, clen, seed=1)
    corpus = rf.Corpus.from_device_rows(rows)
    del rows
bc = getattr(rf.distance, metric).BatchComparator(q)
is_f = metric in ("jaro", "jaro_winkler") or "+norm" in what  # (+norm: normalized_similarity, f64 results)
out = torch.empty(n, dtype=torch.float64 if is_f else torch.int32, device="cuda")
keys = torch.empty(64, dtype=torch.int64, device="cuda")
if "+topk" in what:
    fn = lambda: bc.topk_keys_device(corpus, 16, keys, out=out if "+out" in what else None, **kw)
else:
    fn = lambda: bc.many(N.OP_NORMALIZED_SIMILARITY if "+norm" in what else (N.OP_SIMILARITY if is_f else N.OP_DISTANCE), corpus, out=out, **kw)
for _ in range(int(os.environ.get("AB_WARMUP", 30))):
    fn()
torch.cuda.synchronize()
reps = int(os.environ.get("AB_REPS", 20))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
chk = int(out.view(torch.int64).sum().item()) if is_f else int(out.to(torch.int64).sum().item())  # (bit pattern sum: equal builds agree exactly)
print(f"{os.path.basename(os.environ.get('RF_LIB', 'librfgpu.so')):18s} {what:18s} {ms:8.3f} ms  {n / ms / 1e6:8.2f} Gpairs/s  chk {chk & 0xFFFFFFFFFFFF:012x}")
