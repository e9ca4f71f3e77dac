"""Independent, algorithm-agnostic textbook implementations used as a SECOND oracle.

Nothing here follows the reference's bit-parallel code: plain two-row dynamic programs (numpy-vectorised
along the row where possible) and the classic Jaro definition.  They pin the C restatement in oracle/
against mathematics rather than against itself.
"""
from __future__ import annotations

import numpy as np


def _b(s) -> np.ndarray:
    if isinstance(s, np.ndarray):  # any integer element type (u32 "chars"): only equality of elements matters
        return s.reshape(-1)
    if isinstance(s, str):
        s = s.encode("latin-1")
    return np.frombuffer(bytes(s), dtype=np.uint8)


def levenshtein(a, b, weights=(1, 1, 1)) -> int:
    """Wagner-Fischer, weights = (insertion, deletion, substitution) turning a into b."""
    a, b = _b(a), _b(b)
    ins, dele, sub = weights
    prev = np.arange(len(a) + 1, dtype=np.int64) * dele  # D[i][0] = i deletions
    for j in range(1, len(b) + 1):
        cur = np.empty_like(prev)
        cur[0] = j * ins
        diag = prev[:-1] + np.where(a == b[j - 1], 0, sub)
        up = prev[1:] + ins
        best = np.minimum(diag, up)
        # left dependency cur[i-1] + dele is a running min-plus scan
        run = cur[0]
        for i in range(1, len(a) + 1):
            run = min(best[i - 1], run + dele)
            cur[i] = run
        prev = cur
    return int(prev[-1])


def levenshtein_unit(a, b) -> int:
    """Unit-cost Levenshtein, numpy min-plus scan (fast enough for a few hundred chars)."""
    a, b = _b(a), _b(b)
    n = len(a)
    prev = np.arange(n + 1, dtype=np.int64)
    idx = np.arange(n + 1, dtype=np.int64)
    for j in range(1, len(b) + 1):
        cand = np.empty(n + 1, dtype=np.int64)
        cand[0] = j
        cand[1:] = np.minimum(prev[:-1] + (a != b[j - 1]), prev[1:] + 1)
        # cur[i] = min_k<=i (cand[k] + (i - k))  ==  i + running_min(cand[k] - k)
        prev = idx + np.minimum.accumulate(cand - idx)
    return int(prev[-1])


def lcs_len(a, b) -> int:
    a, b = _b(a), _b(b)
    prev = np.zeros(len(a) + 1, dtype=np.int64)
    for j in range(1, len(b) + 1):
        match = prev[:-1] + (a == b[j - 1])
        cur = np.zeros_like(prev)
        cur[1:] = np.maximum(match, prev[1:])
        cur = np.maximum.accumulate(cur)  # left dependency
        prev = cur
    return int(prev[-1])


def indel(a, b) -> int:
    return len(_b(a)) + len(_b(b)) - 2 * lcs_len(a, b)


def jaro(a, b) -> float:
    a, b = _b(a), _b(b)
    la, lb = len(a), len(b)
    if la == 0 and lb == 0:
        return 1.0
    if la == 0 or lb == 0:
        return 0.0
    if la == 1 and lb == 1:
        return 1.0 if a[0] == b[0] else 0.0
    bound = max(la, lb) // 2 - 1
    fa = [False] * la
    fb = [False] * lb
    common = 0
    for j in range(lb):  # greedy, text-major like every standard implementation
        lo, hi = max(0, j - bound), min(la - 1, j + bound)
        for i in range(lo, hi + 1):
            if not fa[i] and a[i] == b[j]:
                fa[i] = fb[j] = True
                common += 1
                break
    if common == 0:
        return 0.0
    ai = [a[i] for i in range(la) if fa[i]]
    bj = [b[j] for j in range(lb) if fb[j]]
    trans = sum(1 for x, y in zip(ai, bj) if x != y) // 2
    return (common / la + common / lb + (common - trans) / common) / 3.0


def jaro_winkler(a, b, prefix_weight=0.1) -> float:
    sim = jaro(a, b)
    a, b = _b(a), _b(b)
    prefix = 0
    for x, y in zip(a[:4], b[:4]):
        if x != y:
            break
        prefix += 1
    if sim > 0.7:
        sim += prefix * prefix_weight * (1.0 - sim)
    return sim


def osa(a, b) -> int:
    """Optimal string alignment (restricted Damerau-Levenshtein): unit edits + adjacent transposition."""
    a, b = _b(a), _b(b)
    la, lb = len(a), len(b)
    d = np.zeros((la + 1, lb + 1), dtype=np.int64)
    d[:, 0] = np.arange(la + 1)
    d[0, :] = np.arange(lb + 1)
    for i in range(1, la + 1):
        for j in range(1, lb + 1):
            cost = 0 if a[i - 1] == b[j - 1] else 1
            v = min(d[i - 1, j] + 1, d[i, j - 1] + 1, d[i - 1, j - 1] + cost)
            if i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1]:
                v = min(v, d[i - 2, j - 2] + 1)
            d[i, j] = v
    return int(d[la, lb])
