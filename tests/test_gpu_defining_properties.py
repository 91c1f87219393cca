"""The search direction against its DEFINITION, without the oracle.

Every other parity test compares the HIP path with `oracle/gn_oracle.py`, a restatement of the reference's sequence of LAPACK calls
(src/enlsip_functions.jl:116-153, :206-234).  If that restatement shared a misreading of the Julia with the product, both would
agree and both would be wrong.  This file checks what the sequence is FOR: with consistent constraints and J restricted to the null
space of A of full column rank (rankJ2 = n - rankA; code 1, and code -1 with a constraint stated twice), `p` is the unique minimiser of ||J p + r|| subject to A p = -c
(docs/src/method.md: the linearised subproblem of an iteration).  The check value is computed by a different algorithm with
different library calls — null-space method on a singular value decomposition of A, least squares by `numpy.linalg.lstsq` (dgelsd),
no pivoted QR, no Householder bookkeeping — and by the optimality conditions themselves.  Only the input generator is shared.

Where the solution is not unique (rankJ2 < n - rankA) the reference's choice is specific to its algorithm and stays pinned
by the oracle comparisons; here only what holds for every choice is asserted: feasibility, and that no descent is left in the
subspace the solve reports to have used.
"""
import numpy as np
import pytest

from oracle import synth      # seeded input generator only

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    from enlsip_gn import GNSolver
    s = GNSolver(device=0)
    yield s
    s.close()


def direction_by_null_space(J, r, A, c):
    """argmin ||J p + r|| s.t. A p = -c: range / null space of A from its SVD, the reduced problem by dgelsd."""
    n = J.shape[1]
    t = A.shape[0]
    if t == 0:
        return np.linalg.lstsq(J, -r, rcond=None)[0], np.eye(n)
    U, s, Vt = np.linalg.svd(A, full_matrices=True)
    assert s[-1] > 1e-8 * s[0], "generator promised full row rank"
    Y, Z = Vt[:t].T, Vt[t:].T
    py = Y @ ((U.T @ (-c)) / s)
    if Z.shape[1] == 0:
        return py, Z
    pz = np.linalg.lstsq(J @ Z, -(r + J @ py), rcond=None)[0]
    return py + Z @ pz, Z


# measured on MI355X (tests/probes/property_margins.py): direction 1e-15 .. 3e-14, feasibility <= 3e-16, projected gradient <= 4e-16
TOL = 1e-12


def check_definition(p, J, r, A, c, tol):
    pd, Z = direction_by_null_space(J, r, A, c)
    scale = max(np.linalg.norm(pd), 1e-300)
    assert np.linalg.norm(p - pd) <= tol * scale
    if A.shape[0]:        # feasible for the linearised constraints
        assert np.linalg.norm(A @ p + c) <= 1e-13 * (np.linalg.norm(A, 2) * np.linalg.norm(p) + np.linalg.norm(c))
    if Z.shape[1]:        # no descent direction left inside the null space of A
        g = Z.T @ (J.T @ (J @ p + r))
        assert np.linalg.norm(g) <= 1e-13 * np.linalg.norm(J, 2) * (np.linalg.norm(J, 2) * np.linalg.norm(p) + np.linalg.norm(r))


SHAPES = [(40, 8, 0), (64, 16, 4), (33, 33, 1), (12, 12, 12), (300, 40, 40), (1500, 70, 5), (777, 45, 7), (2048, 128, 16),
          (4096, 96, 32), (20000, 40, 3), (1024, 320, 100), (600, 520, 70)]


@pytest.mark.parametrize("m,n,t", SHAPES)
def test_direction_is_the_constrained_least_squares_minimiser(m, n, t, solver):
    J, rx, A, cx = synth.make_problem(9100 + m + 3 * n + 7 * t, m, n, t)
    out = solver.solve(J, rx, A, cx)
    assert (out.rankA, out.rankJ2, out.code) == (t, n - t, 1)
    check_definition(out.p, J, rx, A, cx, TOL)


def test_c2_full_size_direction_by_its_definition(solver):
    """BASELINE configs[1] (m = 4096, n = 512, t = 64)."""
    m, n, t = 4096, 512, 64
    J, rx, A, cx = synth.make_problem(20260202, m, n, t)
    out = solver.solve(J, rx, A, cx)
    assert (out.rankA, out.rankJ2, out.code) == (t, n - t, 1)
    check_definition(out.p, J, rx, A, cx, TOL)


@pytest.mark.parametrize("batch,m,n,t", [(48, 256, 32, 4), (24, 1024, 56, 8), (6, 4096, 512, 64), (130, 200, 24, 0)])
def test_batched_directions_by_their_definition(batch, m, n, t, solver):
    """The batched entry point (C5 / C3 / C2 shapes; 130 problems: both halves of the two-stream split)."""
    probs = [synth.make_problem(7300 + 11 * k + m, m, n, t) for k in range(batch)]
    J = np.stack([np.ascontiguousarray(q[0].T) for q in probs])
    rx = np.stack([q[1] for q in probs])
    At = np.stack([q[2] for q in probs]) if t else np.zeros((batch, 0, n))
    cx = np.stack([q[3] for q in probs]) if t else np.zeros((batch, 0))
    p, b, d, infos, *_ = solver.solve_batched(J, rx, At, cx)
    for k in (range(batch) if m * n <= 1 << 16 else range(0, batch, max(1, batch // 3))):
        assert infos[k][:3] == (t, n - t, 1)
        check_definition(p[k], probs[k][0], probs[k][1], probs[k][2], probs[k][3], TOL)


def test_graded_jacobian_within_its_condition_number(solver):
    """cond(J) = 1e8: the forward error of any backward-stable method is eps * cond; the two methods may differ by that much."""
    m, n, t = 600, 60, 6
    J, rx, A, cx = synth.make_graded_J(31, m, n, t)
    out = solver.solve(J, rx, A, cx)
    assert (out.rankA, out.code) == (t, 1)
    pd, Z = direction_by_null_space(J, rx, A, cx)
    assert np.linalg.norm(out.p - pd) <= 1e-5 * np.linalg.norm(pd)
    # the optimality conditions are insensitive to the conditioning
    assert np.linalg.norm(A @ out.p + cx) <= 1e-11 * (np.linalg.norm(A, 2) * np.linalg.norm(out.p) + np.linalg.norm(cx))
    g = Z.T @ (J.T @ (J @ out.p + rx))
    assert np.linalg.norm(g) <= 1e-9 * np.linalg.norm(J, 2) * (np.linalg.norm(J, 2) * np.linalg.norm(out.p) + np.linalg.norm(rx))


@pytest.mark.parametrize("m,n,t", [(64, 16, 4), (300, 40, 6), (4096, 96, 32), (1024, 320, 100), (4096, 512, 64)])
def test_dependent_constraints_code_minus_one_is_still_the_minimiser(m, n, t, solver):
    """code -1 (src/enlsip_functions.jl:140-148: rankA < t, here one constraint stated twice, consistently).  The feasible set
    is that of the t - 1 independent constraints, the minimiser over it is unique, and the branch through F_L11 has to land
    on it: compared with the null-space method on A without the repeated row."""
    J, rx, A, cx = synth.make_rank_deficient_A(4400 + m + n, m, n, t)
    out = solver.solve(J, rx, A, cx)
    assert (out.rankA, out.rankJ2, out.code) == (t - 1, n - t + 1, -1)
    check_definition(out.p, J, rx, A[:-1], cx[:-1], TOL)
    assert np.linalg.norm(A @ out.p + cx) <= 1e-11 * (np.linalg.norm(A, 2) * np.linalg.norm(out.p) + np.linalg.norm(cx))


@pytest.mark.parametrize("kind,m,n,t", [("dependent", 200, 30, 4), ("wide", 50, 80, 10), ("dependent", 900, 140, 20)])
def test_rank_deficient_jacobian_what_holds_for_every_choice(kind, m, n, t, solver):
    """rankJ2 < n - rankA (dependent columns, or fewer rows than columns): the reference's basic solution (dimJ2 leading pivot
    columns, zeros elsewhere, src/enlsip_functions.jl:136-137) is one of many minimisers.  Every one of them is feasible; the
    residual of the basic solution is orthogonal to the rankJ2 pivot columns the solve reports, i.e. to the columns of
    J2 = (J Q1)[:, t:] it kept."""
    J, rx, A, cx = synth.make_rank_deficient_J(77 + m, m, n, t) if kind == "dependent" else synth.make_problem(77 + m, m, n, t)
    out = solver.solve(J, rx, A, cx)
    assert out.rankA == t and out.code == 1 and out.rankJ2 < n - t
    assert np.linalg.norm(A @ out.p + cx) <= 1e-11 * (np.linalg.norm(A, 2) * np.linalg.norm(out.p) + np.linalg.norm(cx))
    JQ1 = solver.JQ1(m, n)
    kept = JQ1[:, t:][:, out.jpvtJ2[:out.rankJ2] - 1]
    res = J @ out.p + rx
    assert np.linalg.norm(kept.T @ res) <= 1e-9 * np.linalg.norm(kept, 2) * (np.linalg.norm(J, 2) * np.linalg.norm(out.p) + np.linalg.norm(rx))
