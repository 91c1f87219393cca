"""The synthetic workload of SURVEY.md §8(d), generated where it is consumed (torch tensors, any device).

Counter-based, so that every implementation regenerates the same problem from ``(problem_id, stream, linear index)``:

    u64   = splitmix64((seed ^ (stream << 56)) + counter),      seed = 20260101 + problem_id
    value = Box-Muller(counter 2 i, counter 2 i + 1) -> N(0, 1) for element i  (i = column-major linear index)

streams: 0 -> J (m x n), 1 -> rx (m), 2 -> A' (n x t), 3 -> cx (t).  The 64-bit streams are bit-identical to the NumPy
statement of the same generator used by the parity tests; the normal deviates agree to the last ulp or two (``log`` / ``cos`` of
the device's math library).  Used by ``bench.py`` and the full-size GPU tests: 65 536 problems of 256 x 32 are 4.3 GB of
deviates, generated in HBM in a fraction of a second instead of minutes on the host.

Layouts match the C ABI's device entry points: a batch of column-major ``m x n`` matrices is a C-order ``(batch, n, m)`` tensor.
"""
from __future__ import annotations

import math

import torch

SEED0 = 20260101
_GOLDEN = -7046029254386353131          # 0x9E3779B97F4A7C15 as int64
_MUL1 = -4658895280553007687            # 0xBF58476D1CE4E5B9
_MUL2 = -7723592293110705685            # 0x94D049BB133111EB


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    """Logical shift right of int64 bit patterns."""
    return (x >> k) & ((1 << (64 - k)) - 1)


def splitmix64(x: torch.Tensor) -> torch.Tensor:
    z = x + _GOLDEN
    z = (z ^ _lsr(z, 30)) * _MUL1
    z = (z ^ _lsr(z, 27)) * _MUL2
    return z ^ _lsr(z, 31)


def _uniform01(u: torch.Tensor) -> torch.Tensor:
    return (_lsr(u, 11).to(torch.float64) + 1.0) * (1.0 / 9007199254740992.0)      # (0, 1]


def _to_i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def normal_block(problem_ids: torch.Tensor, stream: int, count: int, start: int = 0) -> torch.Tensor:
    """(len(problem_ids), count) N(0, 1) deviates: row k = elements ``start .. start + count - 1`` of the stream of problem
    ``problem_ids[k]``."""
    dev = problem_ids.device
    base = (problem_ids.to(torch.int64) + SEED0) ^ _to_i64(stream << 56)
    ctr = base[:, None] + 2 * torch.arange(start, start + count, dtype=torch.int64, device=dev)[None, :]
    u1 = _uniform01(splitmix64(ctr))
    u2 = _uniform01(splitmix64(ctr + 1))
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)


def make_batch(first_problem: int, batch: int, m: int, n: int, t: int, device, chunk_elems: int = 1 << 26):
    """Problems ``first_problem .. first_problem + batch - 1``.  Returns ``(J (batch, n, m), rx (batch, m),
    At (batch, t, n), cx (batch, t))``: every matrix column-major per problem, as the C ABI takes them."""
    dev = torch.device(device)
    J = torch.empty((batch, n, m), dtype=torch.float64, device=dev)
    rx = torch.empty((batch, m), dtype=torch.float64, device=dev)
    At = torch.empty((batch, t, n), dtype=torch.float64, device=dev)
    cx = torch.empty((batch, t), dtype=torch.float64, device=dev)
    per = max(1, chunk_elems // max(m * n, 1))
    for lo in range(0, batch, per):
        hi = min(batch, lo + per)
        ids = torch.arange(first_problem + lo, first_problem + hi, dtype=torch.int64, device=dev)
        if m * n <= chunk_elems:
            J[lo:hi] = normal_block(ids, 0, m * n).view(hi - lo, n, m)
        else:                                            # one tall problem at a time, in pieces of whole columns
            cols = max(1, chunk_elems // m)
            for k in range(lo, hi):
                for c0 in range(0, n, cols):
                    c1 = min(n, c0 + cols)
                    J[k, c0:c1] = normal_block(ids[k - lo:k - lo + 1], 0, (c1 - c0) * m, start=c0 * m).view(c1 - c0, m)
        rx[lo:hi] = normal_block(ids, 1, m)
        if t:
            At[lo:hi] = normal_block(ids, 2, n * t).view(hi - lo, t, n)
            cx[lo:hi] = normal_block(ids, 3, t)
    return J, rx, At, cx


def make_row_block(problem_id: int, m: int, n: int, t: int, row_lo: int, row_hi: int, device, chunk_elems: int = 1 << 26):
    """Rows ``row_lo .. row_hi - 1`` of problem ``problem_id`` (config C4: the row block of one rank).  Returns
    ``(J_loc (n, m_loc) C-order == column-major m_loc x n, rx_loc (m_loc), At (t, n), cx (t))``; element (r, c) of J carries
    the counter of its position c m + r in the WHOLE matrix, so the blocks of all ranks tile the same problem."""
    dev = torch.device(device)
    m_loc = row_hi - row_lo
    ids = torch.tensor([problem_id], dtype=torch.int64, device=dev)
    base = (ids + SEED0) ^ _to_i64(0 << 56)
    J = torch.empty((n, m_loc), dtype=torch.float64, device=dev)
    rows = torch.arange(row_lo, row_hi, dtype=torch.int64, device=dev)
    cols_per = max(1, chunk_elems // max(m_loc, 1))
    for c0 in range(0, n, cols_per):
        c1 = min(n, c0 + cols_per)
        idx = torch.arange(c0, c1, dtype=torch.int64, device=dev)[:, None] * m + rows[None, :]
        ctr = base + 2 * idx
        u1 = _uniform01(splitmix64(ctr))
        u2 = _uniform01(splitmix64(ctr + 1))
        J[c0:c1] = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)
    rx = normal_block(ids, 1, m_loc, start=row_lo)[0]
    At = normal_block(ids, 2, n * t).view(t, n) if t else torch.empty((0, n), dtype=torch.float64, device=dev)
    cx = normal_block(ids, 3, t)[0] if t else torch.empty((0,), dtype=torch.float64, device=dev)
    return J, rx, At, cx


# the BASELINE.json configurations (shape per problem, problems per GPU)
CONFIGS = {
    "C2": dict(m=4096, n=512, t=64, batch=384,
               label="C2: batch of independent dense CNLS subproblems, m=4096 n=512 t=64 fp64, inputs resident in HBM"),
    "C3": dict(m=512, n=64, t=8, batch=1024,
               label="C3: batch of 1024 independent (m=512, n=64, t=8) subproblems fp64, inputs resident in HBM"),
    "C4": dict(m=262144, n=1024, t=0, batch=1,
               label="C4: one tall-skinny Jacobian m=262144 n=1024 t=0 fp64, rows sharded over the ranks (TSQR), inputs resident in HBM"),
    "C5": dict(m=256, n=32, t=4, batch=8192,
               label="C5: 65536 independent (m=256, n=32, t=4) subproblems fp64 in shards of 8192 per GPU, inputs resident in HBM"),
}
