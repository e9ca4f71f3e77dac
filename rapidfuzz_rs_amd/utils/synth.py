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


def rows_device(n: int, length: int, seed: int, device=None, chunk: int = 1 << 24, symbols: int = 62):
    """torch uint8 CUDA tensor [n, length] of alphanumerics, generated on the device in chunks (`symbols` < 62 draws
    from the first `symbols` of 0-9A-Za-z: an experiment knob for the LDS gather cost)."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((n, length), dtype=torch.uint8, device=dev)
    flat = out.view(-1)
    total = n * length
    for s in range(0, total, chunk * 16):
        e = min(total, s + chunk * 16)
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
                      plant_every: int = 0, block: int = _BLOCK):
    """Rows [start, end) of the logical corpus (seed, length) as a torch uint8 CUDA tensor; with `q` and `plant_every`
    every plant_every-th global row is a near-duplicate of q (planted_row)."""
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty((end - start, length), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    for b in range(start // block, (end + block - 1) // block):
        g.manual_seed(seed + 1000003 * b)
        v = torch.randint(0, symbols, (block, length), dtype=torch.uint8, device=dev, generator=g)
        v += 48 + 7 * (v >= 10).to(torch.uint8) + 6 * (v >= 36).to(torch.uint8)
        lo, hi = max(start, b * block), min(end, (b + 1) * block)
        out[lo - start : hi - start] = v[lo - b * block : hi - b * block]
    if q is not None and plant_every:
        idx = planted_indices(start, end, plant_every)
        if len(idx):
            rows = np.stack([planted_row(q, length, int(i)) for i in idx])
            out[torch.from_numpy(idx - start).to(dev)] = torch.from_numpy(rows).to(dev)
    return out
