"""Synthetic workloads shaped like the reference's benchmarks: iid random ALPHANUMERIC strings
(`rand::distributions::Alphanumeric`, rapidfuzz-benches/benches/bench_levenshtein.rs:8-14).
Used by bench.py and the parity tests; not part of the scoring path."""
from __future__ import annotations

import numpy as np

ALNUM = np.frombuffer(b"0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz", dtype=np.uint8)


def query(length: int, seed: int) -> bytes:
    rng = np.random.default_rng(seed)
    return ALNUM[rng.integers(0, 62, size=length)].tobytes()


def rows_host(n: int, length: int, seed: int) -> np.ndarray:
    """uint8 [n, length] on the host."""
    rng = np.random.default_rng(seed)
    return ALNUM[rng.integers(0, 62, size=(n, length), dtype=np.uint8)]


def ragged_host(n: int, max_len: int, seed: int, min_len: int = 0, alphabet: np.ndarray = ALNUM):
    """n candidates with lengths uniform in [min_len, max_len] -> (data uint8, offsets uint64[n+1])."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_len, max_len + 1, size=n)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens, dtype=np.uint64)
    data = alphabet[rng.integers(0, len(alphabet), size=int(offsets[-1]))]
    return data, offsets


def plant_near_duplicates(rows: np.ndarray, q: bytes, every: int, seed: int, max_edits: int = 5) -> np.ndarray:
    """Overwrite every `every`-th row with the query after 0..max_edits random substitutions (so a cutoff /
    top-k search has something to find).  Returns the planted row indices."""
    rng = np.random.default_rng(seed)
    n, ln = rows.shape
    idx = np.arange(every // 2, n, every)
    qa = np.frombuffer(q, dtype=np.uint8)
    for i in idx:
        r = qa[:ln].copy() if len(qa) >= ln else np.concatenate([qa, ALNUM[rng.integers(0, 62, size=ln - len(qa))]])
        for _ in range(int(rng.integers(0, max_edits + 1))):
            r[int(rng.integers(0, ln))] = ALNUM[int(rng.integers(0, 62))]
        rows[i] = r
    return idx


def rows_device(n: int, length: int, seed: int, device=None, chunk: int = 1 << 24, symbols: int = 62, zipf_s: float = 0.0):
    """torch uint8 CUDA tensor [n, length] of alphanumerics, generated on the device in chunks (`symbols` < 62 draws
    from the first `symbols` of 0-9A-Za-z: an experiment knob for the LDS gather cost; zipf_s > 0: symbol ranks follow a Zipf law
    with that exponent instead of the uniform one)."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((n, length), dtype=torch.uint8, device=dev)
    flat = out.view(-1)
    total = n * length
    cdf = None
    if zipf_s > 0:
        w = 1.0 / torch.arange(1, symbols + 1, dtype=torch.float64, device=dev) ** zipf_s
        cdf = torch.cumsum(w / w.sum(), 0).to(torch.float32)
    for s in range(0, total, chunk * 16):
        e = min(total, s + chunk * 16)
        if cdf is not None:
            v = torch.searchsorted(cdf, torch.rand(e - s, device=dev, generator=g)).clamp_(max=symbols - 1).to(torch.uint8)
        else:
            v = torch.randint(0, symbols, (e - s,), dtype=torch.uint8, device=dev, generator=g)
        # 0-9 -> '0'.., 10-35 -> 'A'.., 36-61 -> 'a'..
        v += 48 + 7 * (v >= 10).to(torch.uint8) + 6 * (v >= 36).to(torch.uint8)
        flat[s:e] = v
    return out


# ---- one LOGICAL corpus, generated shard by shard (bench.py --config c5, the strong split of BASELINE configs[4]) ----
# Rows are produced in fixed blocks of `block` rows, each seeded by its GLOBAL block number, and a planted row depends
# only on its global index: rank r of R generating [start, end) gets exactly the rows a single rank would have there, so
# the merged top-k is the same for every world size (bench.py prints a checksum of it).
_BLOCK = 1 << 20


def planted_row(q: bytes, length: int, gidx: int) -> np.ndarray:
    """The near-duplicate of `q` planted at global index `gidx`: 0..5 substitutions by digits."""
    rng = np.random.default_rng(0xC0FFEE05 ^ (gidx * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFF))
    r = np.frombuffer(q[:length].ljust(length, b"0"), dtype=np.uint8).copy()
    for _ in range(int(rng.integers(0, 6))):
        r[int(rng.integers(0, length))] = 48 + int(rng.integers(0, 10))
    return r


def planted_indices(start: int, end: int, every: int) -> np.ndarray:
    first = every // 2
    k0 = max(0, -(-(start - first) // every))
    return np.arange(first + k0 * every, end, every, dtype=np.int64)


def rows_device_range(start: int, end: int, length: int, seed: int, device=None, symbols: int = 62, q: bytes = None,
                      plant_every: int = 0, block: int = _BLOCK, head_share: float = 0.0):
    """Rows [start, end) of the logical corpus (seed, length) as a torch uint8 CUDA tensor; with `q` and `plant_every`
    every plant_every-th global row is a near-duplicate of q (planted_row); with head_share > 0 that fraction of the rows carries
    the query's first 8..12 symbols (drawn block by block from the block's own seed: the same rows at every world size)."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty((end - start, length), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    for b in range(start // block, (end + block - 1) // block):
        g.manual_seed(seed + 1000003 * b)
        v = torch.randint(0, symbols, (block, length), dtype=torch.uint8, device=dev, generator=g)
        v += 48 + 7 * (v >= 10).to(torch.uint8) + 6 * (v >= 36).to(torch.uint8)
        if head_share > 0 and q is not None:
            hh = min(12, length, len(q))
            pick = torch.nonzero(torch.rand(block, device=dev, generator=g) < head_share).flatten()
            h = torch.randint(min(8, hh), hh + 1, (block,), device=dev, generator=g)[pick]
            qa = torch.frombuffer(bytearray(q[:hh]), dtype=torch.uint8).to(dev)
            cols = torch.arange(hh, device=dev)[None, :] < h[:, None]
            v[pick, :hh] = torch.where(cols, qa[None, :], v[pick, :hh])
        lo, hi = max(start, b * block), min(end, (b + 1) * block)
        out[lo - start : hi - start] = v[lo - b * block : hi - b * block]
    if q is not None and plant_every:
        idx = planted_indices(start, end, plant_every)
        if len(idx):
            rows = np.stack([planted_row(q, length, int(i)) for i in idx])
            out[torch.from_numpy(idx - start).to(dev)] = torch.from_numpy(rows).to(dev)
    return out


# ---- corpora that are NOT iid-uniform (round 6) --------------------------------------------------------
# The cutoff scans decide nearly every candidate of a random corpus from its first 8 symbols -- their best case.  Real dedup /
# record-linkage corpora share prefixes with the query (URLs, names, SKUs), follow a Zipf law over their symbols and a log-normal
# one over their lengths.  These models vary exactly what the plan layer looks at: head survivors, symbol counts, length histograms.
def head_share_rows_host(rows: np.ndarray, q: bytes, share: float, seed: int, head_lo: int = 8, head_hi: int = 12) -> np.ndarray:
    """In place: a fraction `share` of the rows get the query's first H symbols (H uniform in [head_lo, head_hi]), the rest of the
    row stays random.  Returns the indices of those rows."""
    rng = np.random.default_rng(seed)
    n, ln = rows.shape
    pick = np.nonzero(rng.random(n) < share)[0]
    qa = np.frombuffer(q, dtype=np.uint8)
    hi = min(head_hi, ln, len(qa))
    lo = min(head_lo, hi)
    h = rng.integers(lo, hi + 1, size=len(pick))
    cols = np.arange(hi)[None, :] < h[:, None]
    block = rows[pick, :hi]
    rows[pick, :hi] = np.where(cols, qa[None, :hi], block)
    return pick


def head_share_rows_device(rows, q: bytes, share: float, seed: int, head_lo: int = 8, head_hi: int = 12, chunk: int = 1 << 24):
    """The same on a torch uint8 CUDA tensor [n, len], chunk by chunk; returns the number of rows touched."""
    import torch

    n, ln = rows.shape
    hi = min(head_hi, ln, len(q))
    lo = min(head_lo, hi)
    g = torch.Generator(device=rows.device)
    g.manual_seed(seed)
    qa = torch.frombuffer(bytearray(q[:hi]), dtype=torch.uint8).to(rows.device)
    ar = torch.arange(hi, device=rows.device)
    touched = 0
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        pick = torch.nonzero(torch.rand(e - s, device=rows.device, generator=g) < share).flatten()
        if pick.numel() == 0:
            continue
        h = torch.randint(lo, hi + 1, (pick.numel(),), device=rows.device, generator=g)
        cols = ar[None, :] < h[:, None]
        block = rows[s:e][pick, :hi]
        rows[s:e][pick, :hi] = torch.where(cols, qa[None, :], block)
        touched += int(pick.numel())
    return touched


def zipf_alphabet_draw(rng: np.random.Generator, size, symbols: int = 62, s: float = 1.1) -> np.ndarray:
    """`size` alphanumerics whose ranks follow a Zipf law with exponent `s` (rank r with weight r^-s)."""
    w = 1.0 / np.arange(1, symbols + 1) ** s
    return ALNUM[rng.choice(symbols, size=size, p=w / w.sum())]


def lognormal_ragged_host(n: int, max_len: int, seed: int, median: float = 24.0, sigma: float = 0.5, min_len: int = 1, zipf_s: float = 0.0):
    """n candidates with log-normal lengths (median `median`, shape `sigma`, clipped to [min_len, max_len]); symbols uniform or, with
    zipf_s > 0, Zipf over the alphanumerics -> (data uint8, offsets uint64[n+1])."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.lognormal(np.log(median), sigma, size=n)), min_len, max_len).astype(np.int64)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens, dtype=np.uint64)
    total = int(offsets[-1])
    data = zipf_alphabet_draw(rng, total, s=zipf_s) if zipf_s > 0 else ALNUM[rng.integers(0, 62, size=total)]
    return data, offsets
