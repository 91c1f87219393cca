"""Counter-based synthetic inputs for the GN subproblem.

TEST INFRASTRUCTURE (see oracle/gn_oracle.py header).  Every implementation that needs the
same problem (oracle, tests, bench, a future Julia harness) can regenerate it from
``(problem_id, stream, linear index)`` alone:

    u64 = splitmix64(seed ^ (stream << 56) + idx),  seed = 20260101 + problem_id
    two consecutive draws (2*i, 2*i+1) -> Box-Muller -> N(0,1) for element i
    idx = column-major linear index of the element

streams: 0 -> J (m x n), 1 -> rx (m), 2 -> A^T (n x t), 3 -> cx (t).
"""
from __future__ import annotations

import numpy as np

SEED0 = 20260101
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(u: np.ndarray) -> np.ndarray:
    # 53-bit mantissa, in (0, 1]
    return ((u >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


def normal_stream(problem_id: int, stream: int, count: int, chunk: int = 1 << 22) -> np.ndarray:
    """``count`` N(0,1) draws for (problem, stream), element i from counters 2i, 2i+1."""
    out = np.empty(count, dtype=np.float64)
    seed = np.uint64(SEED0 + problem_id)
    with np.errstate(over="ignore"):
        base = seed ^ (np.uint64(stream) << np.uint64(56))
    for lo in range(0, count, chunk):
        hi = min(count, lo + chunk)
        idx = np.arange(lo, hi, dtype=np.uint64)
        with np.errstate(over="ignore"):
            u1 = _uniform01(splitmix64(base + np.uint64(2) * idx))
            u2 = _uniform01(splitmix64(base + np.uint64(2) * idx + np.uint64(1)))
        out[lo:hi] = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return out


def make_problem(problem_id: int, m: int, n: int, t: int):
    """Returns (J m x n, rx m, A t x n, cx t); J and A^T are filled column-major."""
    J = normal_stream(problem_id, 0, m * n).reshape((m, n), order="F")
    rx = normal_stream(problem_id, 1, m)
    At = normal_stream(problem_id, 2, n * t).reshape((n, t), order="F")
    cx = normal_stream(problem_id, 3, t)
    return J, rx, np.ascontiguousarray(At.T), cx


# ---- parity-only variants (SURVEY §8d "Parity-only extras") -----------------------------
def make_rank_deficient_A(problem_id: int, m: int, n: int, t: int):
    """Duplicate one row of A (row t-1 := row 0) => rankA = t - 1, code -1 path."""
    J, rx, A, cx = make_problem(problem_id, m, n, t)
    if t >= 2:
        A[t - 1, :] = A[0, :]
        cx[t - 1] = cx[0]          # keep the duplicated constraint consistent
    return J, rx, A, cx


def make_graded_J(problem_id: int, m: int, n: int, t: int, log10_cond: float = 8.0):
    """J with singular values 10^(-log10_cond * k / (n-1)), k = 0..n-1 (random orthogonal factors).  A wide J (m < n) has
    min(m, n) singular values: the first m of the same sequence (for m >= n nothing changes, bit for bit)."""
    J0, rx, A, cx = make_problem(problem_id, m, n, t)
    U, _ = np.linalg.qr(J0)                 # m x min(m, n)
    G = normal_stream(problem_id, 4, n * n).reshape((n, n), order="F")
    V, _ = np.linalg.qr(G)
    s = 10.0 ** (-log10_cond * np.arange(n) / max(n - 1, 1))
    k = min(m, n)
    if k < n:
        return (U * s[:k]) @ V[:, :k].T, rx, A, cx
    return (U * s) @ V.T, rx, A, cx


def make_rank_deficient_J(problem_id: int, m: int, n: int, t: int, defect: int = 0):
    """J whose last ``defect`` (default t + 3) columns are combinations of the first ones, so that
    rank(J) = n - t - 3 < n2 and the pivoted QR of J2 must truncate (rankJ2 = n2 - 3)."""
    J, rx, A, cx = make_problem(problem_id, m, n, t)
    defect = defect or (t + 3)
    W = normal_stream(problem_id, 5, (n - defect) * defect).reshape((n - defect, defect), order="F")
    J[:, n - defect:] = J[:, : n - defect] @ W
    return J, rx, A, cx
