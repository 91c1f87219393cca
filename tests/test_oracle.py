"""CPU tests of the oracle itself: LAPACK-semantics restatement vs real dgeqp3, golden-fixture
regression, mathematical invariants of the subproblem, and the arithmetic plan of the HIP path
(unpivoted QR + pivoted QR of R) against the reference-order LAPACK sequence."""
import math

import numpy as np
import pytest

from oracle import gn_oracle as go, lapack_semantics as ls, synth
from oracle.make_golden import CASES, hs65_start


# ---- pseudo_rank (src/enlsip_functions.jl:17-31) -------------------------------------------------
def test_pseudo_rank_cases():
    e = go.SQRT_EPS
    assert go.pseudo_rank(np.zeros(0), e) == 0
    assert go.pseudo_rank(np.array([1e-9, 1.0]), e) == 0              # absolute first test
    assert go.pseudo_rank(np.array([2.0, 1.0, 0.5]), e) == 3
    assert go.pseudo_rank(np.array([-2.0, 1.0, 1e-9]), e) == 2         # uses |.|
    # tolerance scales with sqrt(len): |d1| * sqrt(4) * eps = 2e * 1 -> 1.5*e*... boundary
    d = np.array([1.0, 1.0, 1.0, 1.9 * e])
    assert go.pseudo_rank(d, e) == 3                                   # 1.9e <= 2e
    d[3] = 2.1 * e
    assert go.pseudo_rank(d, e) == 4
    assert go.pseudo_rank(np.array([3.0]), e) == 1
    for diag in ([5.0, 1e-3, 1e-12, 1e-13], [1.0, 1e-20], [0.0, 1.0]):
        assert go.pseudo_rank(np.array(diag), e) == ls.pseudo_rank(diag, e)


# ---- LAPACK semantics restatement vs the real thing ------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(64, 8), (40, 40), (8, 20), (200, 50), (3, 3), (1, 5), (5, 1)])
def test_geqp2_matches_dgeqp3(rows, cols):
    rng = np.random.default_rng(rows * 1000 + cols)
    A = rng.standard_normal((rows, cols))
    F = go.qr_colnorm(A)
    f, tau, p = ls.geqp2(A)
    k = min(rows, cols)
    assert np.array_equal(F.jpvt, p)
    np.testing.assert_allclose(np.triu(f[:k]), np.triu(F.factors[:k]), rtol=0, atol=1e-13 * np.abs(F.factors).max())
    np.testing.assert_allclose(tau, F.tau, rtol=0, atol=1e-14)


def test_geqp2_tie_break_first_index():
    # identical column norms: LAPACK's idamax takes the FIRST maximum (bound-constraint Jacobians)
    A = np.eye(4)[:, [2, 0, 3]]
    F = go.qr_colnorm(A)
    _, _, p = ls.geqp2(A)
    assert np.array_equal(F.jpvt, p) and list(p) == [1, 2, 3]


def test_larft_and_wy():
    rng = np.random.default_rng(5)
    A = rng.standard_normal((30, 7))
    f, tau = ls.geqr2(A)
    V = ls.unit_lower(f, 7)
    T = ls.larft(V, tau)
    Q = np.eye(30) - V @ T @ V.T
    np.testing.assert_allclose(Q.T @ Q, np.eye(30), atol=1e-14)
    np.testing.assert_allclose((Q.T @ A)[:7], np.triu(f[:7]), atol=1e-13)
    x = rng.standard_normal(30)
    np.testing.assert_allclose(ls.apply_qt(f, tau, x), Q.T @ x, atol=1e-13)
    np.testing.assert_allclose(ls.apply_q(f, tau, x), Q @ x, atol=1e-13)


# ---- golden fixtures (oracle regression) ------------------------------------------------------------
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_golden_regression(case, golden_dir):
    name, gen, pid, m, n, t = case
    g = np.load(golden_dir / f"{name}.npz")
    J, rx, A, cx = getattr(synth, gen)(pid, m, n, t)
    assert float(np.sum(J)) == float(g["checksum_J"])          # generator is bit-stable
    S = go.gn_subproblem(J, rx, A, cx)
    assert (S.rankA, S.rankJ2, S.code) == (int(g["rankA"]), int(g["rankJ2"]), int(g["code"]))
    np.testing.assert_allclose(S.p, g["p"], rtol=1e-9, atol=1e-12)
    if S.code == 1:                                               # pivots are noise-free only then
        assert np.array_equal(S.jpvtJ2, g["jpvtJ2"])
    assert np.array_equal(S.jpvtA, g["jpvtA"])


def test_hs65_first_subproblem(golden_dir):
    """HS65 at its starting point: active set {c, x1>=-4.5, x2<=4.5} has rank 2 of 3 => stabilised
    path (code -1) on iteration 0."""
    J, rx, A, cx, active = hs65_start()
    np.testing.assert_allclose(cx, [-2, -0.5, 9.5, 5, 9.5, -0.5, 5])
    S = go.gn_subproblem(J, rx, A[active - 1], cx[active - 1])
    g = np.load(golden_dir / "hs65_iter0.npz")
    assert (S.rankA, S.code) == (2, -1) and int(g["rankA"]) == 2
    np.testing.assert_allclose(S.p, g["p"], atol=1e-13)
    # the three active constraints are inconsistent to first order (rank 2): the stabilised step
    # solves them in the least-squares sense, i.e. the residual is orthogonal to range(A_active)
    Aa, ca = A[active - 1], cx[active - 1]
    np.testing.assert_allclose(Aa.T @ (Aa @ S.p + ca), 0.0, atol=1e-12)


# ---- invariants of the solution (SURVEY §8c "Stated tolerance") --------------------------------------
@pytest.mark.parametrize("m,n,t", [(256, 32, 4), (512, 64, 8), (300, 40, 0), (120, 30, 30)])
def test_solution_invariants(m, n, t):
    J, rx, A, cx = synth.make_problem(77, m, n, t)
    S = go.gn_subproblem(J, rx, A, cx)
    p = S.p
    if t:
        assert np.abs(A @ p + cx).max() <= 1e-12 * (np.linalg.norm(A, 2) * np.linalg.norm(p) + np.linalg.norm(cx))
    # projected gradient: Z' J'(Jp + r) = 0 with Z = last n - t columns of Q1
    Q1 = S.F_A.Q_mul(np.eye(n)) if t else np.eye(n)
    Z = Q1[:, S.rankA:]
    g = Z.T @ (J.T @ (J @ p + rx))
    assert np.linalg.norm(g) <= 1e-11 * np.linalg.norm(J, 2) ** 2 * max(np.linalg.norm(p), 1.0)
    assert abs(np.linalg.norm(S.d) - np.linalg.norm(-J @ (Q1[:, :S.rankA] @ (Q1[:, :S.rankA].T @ p)) - rx)) < 1e-9


# ---- the HIP path's arithmetic plan reproduces dgeqp3's pivots / |diag| / p -----------------------------
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_factored_plan_matches_lapack_order(case):
    name, gen, pid, m, n, t = case
    if m > 1000:
        pytest.skip("numpy loops too slow; covered on the GPU")
    J, rx, A, cx = getattr(synth, gen)(pid, m, n, t)
    S = go.gn_subproblem(J, rx, A, cx)
    M = ls.gn_subproblem_factored(J, rx, A, cx, go.SQRT_EPS)
    assert (M["rankA"], M["rankJ2"], M["code"]) == (S.rankA, S.rankJ2, S.code)
    tol = 1e-11 if gen != "make_graded_J" else 1e-5      # forward error <= 1e-13 * cond
    assert np.linalg.norm(M["p"] - S.p) <= tol * np.linalg.norm(S.p)
    assert np.array_equal(M["jpvtA"], S.jpvtA) and np.array_equal(M["jpvtL"], S.jpvtL)
    kp = min(m, n - S.rankA)
    assert abs(np.linalg.norm(M["d"]) - np.linalg.norm(S.d)) <= 1e-12 * max(np.linalg.norm(S.d), 1)
    if S.code == 1:
        r = S.rankJ2
        assert np.array_equal(M["jpvtJ2"][:r], S.jpvtJ2[:r])
        assert sorted(M["jpvtJ2"][r:]) == sorted(S.jpvtJ2[r:])
        dref = np.abs(S.F_J2.diagR())
        np.testing.assert_allclose(np.abs(M["diagJ2"])[:r], dref[:r], rtol=1e-10)
        np.testing.assert_allclose(np.abs(M["d"][:r]), np.abs(S.d[:r]), rtol=0, atol=1e-9 * max(np.abs(S.d).max(), 1))


# ---- update_working_set restatement: sequencing on HS65 -------------------------------------------------
def test_update_working_set_hs65_sequence():
    J, rx, A, cx, active = hs65_start()
    W = go.new_working_set(0, 7)
    for idx in (1, 2, 6):
        go.add_constraint(W, int(np.where(W.inactive == idx)[0][0]) + 1)
    assert list(W.active[:3]) == [1, 2, 6] and W.t == 3
    C = go.Constraint(cx[active - 1].copy(), A[active - 1].copy(), False, np.zeros(3))
    go.evaluate_scaling(C)
    it = go.IterationRecord()
    p = np.zeros(3)
    count = []
    F_A, F_L11, F_J2 = go.update_working_set(W, rx, A, C, J.T @ rx, J, p, it, go.SQRT_EPS,
                                             solve=lambda: count.append(1))
    assert 1 <= len(count) <= 3                 # 1-3 subproblem solves per call
    assert it.rankA == min(W.t, 2) or it.rankA == W.t
    assert np.all(np.isfinite(p))
