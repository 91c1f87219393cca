"""CPU baseline leg of bench.py: the LAPACK call sequence Julia's LinearAlgebra issues for one
subproblem solve, INCLUDING the reference's redundant dense copies:
dgeqp3(A'), dgeqp3(R'), copy(J)+dormqr('R','N'), two slice copies, copy(J2)+dgeqp3, dgemv,
dormqr('L','T'), dtrtrs x2, dormqr('L','N')   (src/enlsip_functions.jl:700, 769, 219-225, 131-151).

TEST / MEASUREMENT INFRASTRUCTURE (see oracle/gn_oracle.py header): a reported baseline ("port":
scipy's OpenBLAS instead of Julia's), never the thing shipped.
"""
from __future__ import annotations

import os
import time

import numpy as np

from oracle import gn_oracle as go


def one_solve(J, rx, A_active, cx, eps_rank=go.SQRT_EPS):
    m, n = J.shape
    t = A_active.shape[0]
    F_A = go.qr_colnorm(A_active.T)                      # :700 (qr copies its argument)
    rankA = go.pseudo_rank(F_A.diagR(), eps_rank)
    F_L11 = go.qr_colnorm(F_A.R.T)                       # :769
    code = 1 if rankA == t else -1
    JQ1 = F_A.rmul_Q(J)                                  # :219  copy(J) + dormqr
    J1 = JQ1[:, :rankA].copy()                           # :220  slice copies
    J2 = JQ1[:, rankA:].copy()
    F_J2 = go.qr_colnorm(J2)                             # :223  copy + dgeqp3
    rankJ2 = go.pseudo_rank(F_J2.diagR(), eps_rank)
    return go.sub_search_direction(J1, rx, cx, F_A, F_L11, F_J2, n, t, rankA, rankA, rankJ2, code)


def blas_threads() -> int:
    try:
        from threadpoolctl import threadpool_info
        ths = [i["num_threads"] for i in threadpool_info() if i.get("user_api") == "blas"]
        return max(ths) if ths else 1
    except Exception:
        return 1


def _best_thread_count(J, rx, A_active, cx):
    """The BLAS thread count that solves fastest on THIS box (a container may see more cores than its CPU share allows:
    64 threads on a 16-core share run slower than 16).  Returns (threads, limiter) or (current, None) without threadpoolctl."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        return blas_threads(), None
    full = blas_threads()
    best, best_t = full, None
    for th in sorted({full, min(full, 32), min(full, 16), min(full, 8)}, reverse=True):
        with threadpool_limits(limits=th, user_api="blas"):
            one_solve(J, rx, A_active, cx)
            t0 = time.perf_counter()
            one_solve(J, rx, A_active, cx)
            dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = th, dt
    return best, threadpool_limits


def time_baseline(J, rx, A_active, cx, budget_s: float = 12.0, min_solves: int = 3):
    """Runs one_solve repeatedly for about budget_s seconds with the fastest BLAS thread count of this box.
    Returns (solves_per_s, solves, seconds, threads)."""
    one_solve(J, rx, A_active, cx)          # warm-up (thread pool, page faults)
    threads, limiter = _best_thread_count(J, rx, A_active, cx)

    def run():
        n = 0
        t0 = time.perf_counter()
        while True:
            one_solve(J, rx, A_active, cx)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s and n >= min_solves:
                break
            if n >= 10000:
                break
        return n / el, n, el

    if limiter is None:
        sps, n, el = run()
    else:
        with limiter(limits=threads, user_api="blas"):
            sps, n, el = run()
    return sps, n, el, threads


def host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1
