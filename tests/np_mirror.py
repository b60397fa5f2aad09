"""Independent numpy restatement of the oracle's dot-scan spec (DESIGN.md 3 / SURVEY.md App. B).
Used only to pin oracle/dot_scan.c: a second implementation, written against the spec, not the C code."""
from __future__ import annotations

import math

import numpy as np

MASK = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def synth_i32(seed: int, row: int, e: int) -> int:
    key = splitmix64((seed + 0x632BE59BD9B4E019 * row) & MASK)
    h = splitmix64((key + e) & MASK)
    return (h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48) - 131070


def synth_row(seed: int, row: int, D: int, kind: int = 0, src: int = -1) -> np.ndarray:
    var = 1431655765.0
    c = np.float32(1.0 / math.sqrt(D * var))
    cp = np.float32(1.0 / math.sqrt(D * var * 26.0))
    if kind == 0:
        x = np.array([synth_i32(seed, row, e) for e in range(D)], dtype=np.int64)
        return x.astype(np.float32) * c
    if kind == 2:
        x = np.array([synth_i32(seed, src, e) for e in range(D)], dtype=np.int64)
        return x.astype(np.float32) * c
    x = np.array([5 * synth_i32(seed, src, e) + synth_i32(seed, row, e) for e in range(D)], dtype=np.int64)
    return x.astype(np.float32) * cp


def dot_tree(q: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """Fixed summation tree for every row of `rows` (n, D): lane L sums elements j*256+4L+c in (j, c) order,
    then acc[L] += acc[L ^ m] for m = 32..1."""
    q = np.asarray(q, dtype=np.float32)
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, q.size)
    n, D = rows.shape
    Dp = -(-D // 256) * 256
    qq = np.zeros(Dp, dtype=np.float64); qq[:D] = q
    rr = np.zeros((n, Dp), dtype=np.float64); rr[:, :D] = rows
    valid = np.zeros(Dp, dtype=bool); valid[:D] = True
    prod = rr * qq  # exact: products of two fp32 values fit fp64
    prod = prod.reshape(n, Dp // 256, 64, 4)
    val = valid.reshape(Dp // 256, 64, 4)
    acc = np.zeros((n, 64), dtype=np.float64)
    for j in range(Dp // 256):
        for c in range(4):
            m = val[j, :, c]
            acc[:, m] = acc[:, m] + prod[:, j, m, c]
    lanes = np.arange(64)
    for m in (32, 16, 8, 4, 2, 1):
        acc = acc + acc[:, lanes ^ m]
    return acc[:, 0]


def topk(scores: np.ndarray, K: int):
    """(score desc, index desc); pads with (-inf, -1)."""
    idx = np.arange(scores.size)
    order = sorted(idx[~np.isnan(scores)], key=lambda i: (-scores[i], -i))[:K]
    sc = np.full(K, -np.inf); ix = np.full(K, -1, dtype=np.int64)
    sc[:len(order)] = scores[order]; ix[:len(order)] = order
    return sc, ix


def loop_tick(state: dict, db: np.ndarray, l: int, locality=12, thresh=float(np.float32(0.85)), lag=50, min_new=3, min_k=5):
    """Cerebro.cpp:956-1100, one iteration."""
    out = dict(status=0, found=0, idx_curr=-1, idx_prev=-1, score=0.0, argmax=[-1, -1, -1], maxv=[-np.inf] * 3)
    if l - state["last_l"] < min_new:
        return out
    k = l - lag
    out["status"] = 1
    if k > min_k:
        out["status"] = 2
        for qi, row in enumerate((l - 1, l - 2, l - 3)):
            u = dot_tree(db[row], db[:k])
            mx = u.max()
            out["maxv"][qi] = float(mx)
            out["argmax"][qi] = int(np.flatnonzero(u == mx)[-1])
        a = out["argmax"]
        if abs(a[0] - a[1]) < locality and abs(a[0] - a[2]) < locality and out["maxv"][0] > thresh:
            out.update(found=1, idx_curr=l - 1, idx_prev=a[0], score=out["maxv"][0])
    state["last_l"] = l
    return out
