"""Generates tests/golden/*.npz from the CPU oracle (committed with its outputs).

TEST INFRASTRUCTURE.  The reference itself (Julia) cannot run in the build container and ships
no vectors for this path, so these fixtures are *oracle* outputs — LAPACK
(scipy/OpenBLAS 0.3.28) driven exactly as the reference drives it — not reference outputs.
Inputs are regenerated from (generator, problem_id, m, n, t) by oracle/synth.py, so only the
expected outputs and the tiny HS65 case (data from test/problems/HS65.jl:7-17 of the reference,
evaluated at its starting point) are stored.

    python -m oracle.make_golden
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from oracle import gn_oracle as go, synth

OUT = Path(__file__).resolve().parents[1] / "tests" / "golden"

CASES = [
    # name, generator, problem_id, m, n, t
    ("c5_256x32_t4", "make_problem", 11, 256, 32, 4),
    ("c3_512x64_t8", "make_problem", 12, 512, 64, 8),
    ("c3_512x64_t0", "make_problem", 13, 512, 64, 0),
    ("t_eq_n_100x20", "make_problem", 14, 100, 20, 20),
    ("wide_10x30_t4", "make_problem", 15, 10, 30, 4),
    ("odd_777x45_t7", "make_problem", 16, 777, 45, 7),
    ("rankdefA_300x40_t6", "make_rank_deficient_A", 17, 300, 40, 6),
    ("rankdefJ_300x40_t6", "make_rank_deficient_J", 18, 300, 40, 6),
    ("graded_300x40_t5", "make_graded_J", 19, 300, 40, 5),
    ("mid_2048x128_t16", "make_problem", 20, 2048, 128, 16),
]


def hs65_start():
    """HS65 (reference test/problems/HS65.jl:7-17) at x0 = (-5, 5, 0): residual Jacobian, residuals,
    constraint values/Jacobian in the reference's ordering [c; x - x_low; x_upp - x]
    (src/cnls_model.jl:402-403, 449) and the initial working set {1, 2, 6}."""
    x = np.array([-5.0, 5.0, 0.0])
    rx = np.array([x[0] - x[1], (x[0] + x[1] - 10.0) / 3.0, x[2] - 5.0])
    J = np.array([[1.0, -1.0, 0.0], [1.0 / 3.0, 1.0 / 3.0, 0.0], [0.0, 0.0, 1.0]])
    low = np.array([-4.5, -4.5, -5.0])
    upp = np.array([4.5, 4.5, 5.0])
    c = 48.0 - x[0] ** 2 - x[1] ** 2 - x[2] ** 2
    cx = np.concatenate([[c], x - low, upp - x])
    A = np.vstack([[-2 * x[0], -2 * x[1], -2 * x[2]], np.eye(3), -np.eye(3)])
    active = np.array([1, 2, 6])  # 1-based ids: c, x1 >= -4.5, x2 <= 4.5
    return J, rx, A, cx, active


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    for name, gen, pid, m, n, t in CASES:
        J, rx, A, cx = getattr(synth, gen)(pid, m, n, t)
        S = go.gn_subproblem(J, rx, A, cx)
        np.savez_compressed(
            OUT / f"{name}.npz", gen=gen, pid=pid, m=m, n=n, t=t,
            p=S.p, b=S.b, d=S.d, rankA=S.rankA, rankJ2=S.rankJ2, code=S.code,
            jpvtA=S.jpvtA, jpvtL=S.jpvtL, jpvtJ2=S.jpvtJ2,
            diagA=S.F_A.diagR(), diagL=S.F_L11.diagR(), diagJ2=S.F_J2.diagR(),
            checksum_J=float(np.sum(J)), checksum_rx=float(np.sum(rx)))
        print(name, "rankA", S.rankA, "rankJ2", S.rankJ2, "code", S.code)
    J, rx, A, cx, active = hs65_start()
    Aact, cact = A[active - 1], cx[active - 1]
    S = go.gn_subproblem(J, rx, Aact, cact)
    np.savez_compressed(OUT / "hs65_iter0.npz", J=J, rx=rx, A=Aact, cx=cact, p=S.p, b=S.b, d=S.d,
                        rankA=S.rankA, rankJ2=S.rankJ2, code=S.code,
                        jpvtA=S.jpvtA, jpvtL=S.jpvtL, jpvtJ2=S.jpvtJ2)
    print("hs65_iter0 rankA", S.rankA, "rankJ2", S.rankJ2, "code", S.code, "p", S.p)


if __name__ == "__main__":
    main()
