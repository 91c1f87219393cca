"""Parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle
on the same seeded inputs, against the committed golden fixtures, and — at BASELINE sizes —
through size-independent properties.  Tolerances are the ones SURVEY.md §8(c) states:
well-conditioned: ||p - p_oracle|| / ||p_oracle|| <= 1e-11, ranks equal, |diag R| rel 1e-12-ish,
jpvt equal on tie-free inputs; ill-conditioned (cond up to 1e8): forward error <= 1e-13 * cond.
"""
import os

import numpy as np
import pytest

from oracle import gn_oracle as go, synth
from oracle.make_golden import CASES, hs65_start

pytestmark = pytest.mark.gpu

TOL_P = 1e-11


@pytest.fixture(scope="module", params=["auto", "pairs"])
def solver(request):
    """Every test of this module runs twice: with the library's own choice between the plain CAQR sweep and panel pairs (pairs
    only where the far update dominates: large batches, C4's row count) and with pairs forced for every shape with three panels
    or more (ENLSIP_GN_PAIR=1 is read when the handle is created), so that the pair geometry, the two-panel update kernel and the
    accessors on pair-shaped factors meet every shape of the suite."""
    from enlsip_gn import GNSolver
    old = os.environ.get("ENLSIP_GN_PAIR")
    if request.param == "pairs":
        os.environ["ENLSIP_GN_PAIR"] = "1"
    try:
        s = GNSolver(device=0)
    finally:
        if request.param == "pairs":
            if old is None:
                del os.environ["ENLSIP_GN_PAIR"]
            else:
                os.environ["ENLSIP_GN_PAIR"] = old
    yield s
    s.close()


@pytest.fixture(scope="module")
def solver_refl():
    from enlsip_gn import GNSolver, FLAG_UPDATE_REFLECTORS
    s = GNSolver(device=0, flags=FLAG_UPDATE_REFLECTORS)
    yield s
    s.close()


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def compare(out, ref, m, n, tol_p=TOL_P, pivots=True, tol_d=1e-10):
    assert (out.rankA, out.rankJ2, out.code) == (ref.rankA, ref.rankJ2, ref.code)
    assert out.dimA == ref.rankA and out.dimJ2 == ref.rankJ2
    assert rel(out.p, ref.p) <= tol_p
    if ref.b.size:
        assert rel(out.b, ref.b) <= 1e-12
    assert np.array_equal(out.jpvtA, ref.jpvtA)
    assert np.array_equal(out.jpvtL, ref.jpvtL)
    kp = min(m, n - ref.rankA)
    nd = max(np.linalg.norm(ref.d), 1e-300)
    assert abs(np.linalg.norm(out.d) - np.linalg.norm(ref.d)) <= 1e-12 * nd
    if pivots:
        r = ref.rankJ2
        assert np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r])
        assert sorted(out.jpvtJ2[r:]) == sorted(ref.jpvtJ2[r:])
        # leading components agree up to sign (thin Q unique up to column signs once pivots are fixed)
        if r:
            assert np.abs(np.abs(out.d[:r]) - np.abs(ref.d[:r])).max() <= tol_d * max(np.abs(ref.d).max(), 1.0)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_against_oracle_and_golden(case, solver, golden_dir):
    name, gen, pid, m, n, t = case
    J, rx, A, cx = getattr(synth, gen)(pid, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = solver.solve(J, rx, A, cx)
    tol = 1e-5 if gen == "make_graded_J" else TOL_P      # cond 1e8: 1e-13 * cond
    compare(out, ref, m, n, tol_p=tol, pivots=(ref.code == 1), tol_d=(1e-6 if gen == "make_graded_J" else 1e-10))
    g = np.load(golden_dir / f"{name}.npz")
    assert (out.rankA, out.rankJ2, out.code) == (int(g["rankA"]), int(g["rankJ2"]), int(g["code"]))
    assert rel(out.p, g["p"]) <= max(tol, 1e-9)


@pytest.mark.parametrize("m,n,t", [(40, 8, 0), (64, 16, 4), (33, 33, 1), (1, 1, 0), (2, 5, 1), (5, 2, 2),
                                   (1500, 70, 5), (20000, 40, 3), (4096, 96, 32)])
def test_shapes(m, n, t, solver):
    J, rx, A, cx = synth.make_problem(500 + m + n, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = solver.solve(J, rx, A, cx)
    compare(out, ref, m, n)


@pytest.mark.parametrize("m,n,t", [(300, 40, 5), (777, 45, 7), (1500, 70, 5), (2048, 128, 16), (4096, 96, 32)])
@pytest.mark.parametrize("flags", [0, 2])
def test_tile_rows_256(m, n, t, flags):
    """The 256-row CAQR tile (one more tree level, RPL = 4 kernel instantiations) against the oracle, with the MFMA
    and with the reflector-by-reflector trailing update."""
    from enlsip_gn import GNSolver
    s = GNSolver(device=0, flags=flags, tile_rows=256)
    try:
        J, rx, A, cx = synth.make_problem(700 + m + n, m, n, t)
        ref = go.gn_subproblem(J, rx, A, cx)
        compare(s.solve(J, rx, A, cx), ref, m, n)
    finally:
        s.close()


def test_mfma_and_reflector_updates_agree(solver, solver_refl):
    J, rx, A, cx = synth.make_problem(321, 3000, 200, 10)
    a = solver.solve(J, rx, A, cx)
    b = solver_refl.solve(J, rx, A, cx)
    assert {"sweep_reflectors", "jq1_plain"} <= solver_refl.route() and "sweep_reflectors" not in solver.route()
    assert rel(a.p, b.p) <= 1e-12 and np.array_equal(a.jpvtJ2, b.jpvtJ2)
    assert rel(np.abs(a.d[:190]), np.abs(b.d[:190])) <= 1e-11


def test_hs65_iter0(solver, golden_dir):
    J, rx, A, cx, active = hs65_start()
    ref = go.gn_subproblem(J, rx, A[active - 1], cx[active - 1])
    out = solver.solve(J, rx, A[active - 1], cx[active - 1])
    assert (out.rankA, out.code) == (2, -1)
    assert rel(out.p, ref.p) <= 1e-12
    assert rel(out.b, ref.b) <= 1e-12
    g = np.load(golden_dir / "hs65_iter0.npz")
    assert rel(out.p, g["p"]) <= 1e-12


def test_c2_full_size(solver):
    """BASELINE configs[1] (m=4096, n=512, t=64): oracle parity plus invariants."""
    m, n, t = 4096, 512, 64
    J, rx, A, cx = synth.make_problem(20260101 - synth.SEED0, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = solver.solve(J, rx, A, cx)
    compare(out, ref, m, n)
    p = out.p
    assert np.abs(A @ p + cx).max() <= 1e-12 * (np.linalg.norm(A, 2) * np.linalg.norm(p) + np.linalg.norm(cx))
    Q1 = ref.F_A.Q_mul(np.eye(n))
    Z = Q1[:, t:]
    g = Z.T @ (J.T @ (J @ p + rx))
    assert np.linalg.norm(g) <= 1e-11 * np.linalg.norm(J, 2) ** 2 * np.linalg.norm(p)


def test_accessors(solver):
    from enlsip_gn import FACTOR_A, FACTOR_L11, FACTOR_J2
    m, n, t = 700, 48, 6
    J, rx, A, cx = synth.make_problem(42, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    solver.solve(J, rx, A, cx)
    rng = np.random.default_rng(0)
    for which, F in ((FACTOR_A, ref.F_A), (FACTOR_L11, ref.F_L11), (FACTOR_J2, ref.F_J2)):
        fv = solver.factor(which)
        assert fv.R.shape == F.R.shape
        assert rel(np.abs(fv.R), np.abs(F.R)) <= 1e-11
        assert rel(np.abs(fv.diagR()), np.abs(F.diagR())) <= 1e-11
        assert np.array_equal(fv.p, F.p)
        v = rng.standard_normal(F.rows)
        qt = fv.Qt_mul(v)
        assert abs(np.linalg.norm(qt) - np.linalg.norm(v)) <= 1e-12 * np.linalg.norm(v)
        assert rel(np.abs(qt[:F.k]), np.abs(F.Qt_mul(v)[:F.k])) <= 1e-10
        assert rel(fv.Q_mul(qt), v) <= 1e-12
    assert rel(solver.JQ1(m, n), ref.F_A.rmul_Q(J)) <= 1e-12
    # R'R reproduces the Gram matrix of the permuted J2 (sign-free check of R itself)
    fv = solver.factor(FACTOR_J2)
    J2 = ref.F_A.rmul_Q(J)[:, ref.rankA:]
    R = fv.R
    G = J2[:, fv.p - 1].T @ J2[:, fv.p - 1]
    assert rel(R.T @ R, G) <= 1e-12


EXTREME_SHAPES = [(300, 40, 6), (256, 32, 4), (700, 130, 20), (900, 300, 40), (600, 520, 30), (300, 200, 90)]
EXTREME_SCALES = [(600, 0), (-600, 0), (0, 600), (0, -600), (600, -600), (-700, 700), (1000, 1000), (-990, 0)]


@pytest.mark.parametrize("m,n,t", EXTREME_SHAPES)
def test_extreme_magnitudes_match_lapack(m, n, t, solver):
    """VERDICT round 4, "missing" 2: qr(., ColumnNorm()) of the reference is dgeqp3, whose norms (dnrm2) and reflectors (dlarfg)
    scale internally — J * 2^600 or A * 2^-600 is factored like any other matrix, and pseudo_rank's FIRST test, which is absolute
    (src/enlsip_functions.jl:19), then decides the rank: a J of magnitude 2^-600 has rankJ2 = 0, an A of that magnitude rankA = 0
    and the stabilised path.  The kernels square plainly; the library detects the case on its result and solves again on copies
    scaled by a power of two (gn_rescale.hpp).  Here: J, rx and A, cx scaled by powers of two far outside the range of plain sums
    of squares, every combination, in every size class of the dispatch — ranks, code, pivots, p, b, |d|, |diag R| and R itself
    against the oracle (real LAPACK) on the SAME scaled inputs; then the consumers of the resident factors."""
    from enlsip_gn import FACTOR_A, FACTOR_L11, FACTOR_J2
    J0, rx0, A0, cx0 = synth.make_problem(61000 + m + n, m, n, t)
    for eJ, eA in EXTREME_SCALES:
        J, rx, A, cx = np.ldexp(J0, eJ), np.ldexp(rx0, eJ), np.ldexp(A0, eA), np.ldexp(cx0, eA)
        ref = go.gn_subproblem(J, rx, A, cx)
        out = solver.solve(J, rx, A, cx)
        tag = (m, n, t, eJ, eA)
        assert "rescaled" in solver.route(), tag
        assert (out.rankA, out.rankJ2, out.code, out.status) == (ref.rankA, ref.rankJ2, ref.code, 0), tag
        assert np.array_equal(out.jpvtA, ref.jpvtA) and np.array_equal(out.jpvtL, ref.jpvtL), tag
        r = ref.rankJ2
        assert np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r]), tag
        assert np.all(np.isfinite(out.p)) and rel(out.p, ref.p) <= 1e-11, (tag, rel(out.p, ref.p))
        assert rel(np.ldexp(out.b, -eA), np.ldexp(ref.b, -eA)) <= 1e-11, tag          # (norms of vectors of magnitude 2^600 overflow in NumPy too)
        assert abs(np.linalg.norm(np.ldexp(out.d, -eJ)) - np.linalg.norm(np.ldexp(ref.d, -eJ))) <= 1e-12 * np.linalg.norm(np.ldexp(ref.d, -eJ)), tag
        for which, F, e in ((FACTOR_A, ref.F_A, eA), (FACTOR_L11, ref.F_L11, eA), (FACTOR_J2, ref.F_J2, eJ)):
            fv = solver.factor(which)
            kk = r if which == FACTOR_J2 else min(F.R.shape)          # beyond rankJ2 the pivots (and so the rows of R) are rounding's
            if kk == 0:
                continue
            dg, dgr = np.ldexp(np.abs(fv.diagR()[:kk]), -e), np.ldexp(np.abs(F.diagR()[:kk]), -e)
            assert np.abs(dg - dgr).max() <= 1e-11 * dgr.max(), (tag, which)
            if which != FACTOR_J2 or r == min(F.R.shape):
                assert rel(np.ldexp(np.abs(fv.R), -e), np.ldexp(np.abs(F.R), -e)) <= 1e-10, (tag, which)
            v = np.linspace(-1.0, 1.0, F.rows)
            assert rel(fv.Q_mul(fv.Qt_mul(v)), v) <= 1e-12, (tag, which)
        # J * F_A.Q on the caller's (unscaled) J, and the re-solve with truncated dimensions on the resident factors
        if abs(eJ) <= 700:
            assert rel(np.ldexp(solver.JQ1(m, n), -eJ), np.ldexp(ref.F_A.rmul_Q(J), -eJ)) <= 1e-12, tag
        if ref.rankA and ref.rankJ2 > 1:
            dimA, dimJ2 = ref.rankA - 1, ref.rankJ2 // 2
            JQ1 = ref.F_A.rmul_Q(J)
            p_ref, b_ref, d_ref = go.sub_search_direction(JQ1[:, :ref.rankA], rx, cx, ref.F_A, ref.F_L11, ref.F_J2, n, t, ref.rankA, dimA, dimJ2, -1)
            p2, b2, d2 = solver.resolve(m, n, t, dimA, dimJ2, -1)
            assert rel(p2, p_ref) <= 1e-9 and rel(np.ldexp(b2, -eA), np.ldexp(b_ref, -eA)) <= 1e-10, tag
    # the next ordinary solve on the handle is an ordinary solve again
    out = solver.solve(J0, rx0, A0, cx0)
    assert "rescaled" not in solver.route() and rel(out.p, go.gn_subproblem(J0, rx0, A0, cx0).p) <= 1e-11


def test_extreme_magnitudes_in_a_batch_and_in_the_factored_flow(solver):
    """The same inside a batch — two of five problems far outside the range, each handed to a one-problem handle of its own, outputs
    in the caller's slots, accessors routed — with mixed-scale COLUMNS in a third (column norms spread over 2^+-250: no rescaling,
    plain arithmetic carries it), and in the factor_constraints / solve_factored flow of update_working_set's s == 0 branch."""
    from enlsip_gn import FACTOR_A, FACTOR_J2
    m, n, t, B = 300, 40, 6, 5
    probs = [list(synth.make_problem(62000 + k, m, n, t)) for k in range(B)]
    probs[1][0], probs[1][1] = np.ldexp(probs[1][0], 650), np.ldexp(probs[1][1], 650)              # J, rx huge
    probs[3][2], probs[3][3] = np.ldexp(probs[3][2], -620), np.ldexp(probs[3][3], -620)            # A, cx tiny: rankA = 0
    colscale = np.ldexp(1.0, np.linspace(-250, 250, n).round().astype(int))
    probs[4][0] = probs[4][0] * colscale                                                            # mixed-scale columns
    refs = [go.gn_subproblem(*P) for P in probs]
    p, b, d, infos, jA, jL, jJ = solver.solve_batched(np.stack([np.ascontiguousarray(P[0].T) for P in probs]), np.stack([P[1] for P in probs]),
                                                      np.stack([np.ascontiguousarray(P[2]) for P in probs]), np.stack([P[3] for P in probs]))
    assert "rescaled" in solver.route()
    for k, ref in enumerate(refs):
        assert (infos[k][0], infos[k][1], infos[k][2], infos[k][5]) == (ref.rankA, ref.rankJ2, ref.code, 0), k
        assert rel(p[k], ref.p) <= (1e-11 if k != 4 else 1e-9), (k, rel(p[k], ref.p))
        assert rel(b[k] / max(np.abs(ref.b).max(), 1e-300), ref.b / max(np.abs(ref.b).max(), 1e-300)) <= 1e-11, k
        r = ref.rankJ2
        assert np.array_equal(jJ[k][:r], ref.jpvtJ2[:r]), k
        fv = solver.factor(FACTOR_J2, prob=k)
        dg, dgr = np.abs(fv.diagR()[:r]), np.abs(ref.F_J2.diagR()[:r])
        assert np.abs(dg / dgr - 1.0).max() <= 1e-10, k
        assert np.array_equal(solver.factor(FACTOR_A, prob=k).p, ref.F_A.p), k
    # factored flow, A beyond the range
    J, rx, A, cx = synth.make_problem(62100, 400, 60, 9)
    for eJ, eA in ((0, 640), (0, -640), (610, -630)):
        Js, rxs, As, cxs = np.ldexp(J, eJ), np.ldexp(rx, eJ), np.ldexp(A, eA), np.ldexp(cx, eA)
        ref = go.gn_subproblem(Js, rxs, As, cxs)
        rankA, code, dimA = solver.factor_constraints(400, As, cxs)
        assert (rankA, code) == (ref.rankA, ref.code)
        fa = solver.factor(FACTOR_A)
        assert np.array_equal(fa.p, ref.F_A.p) and rel(np.ldexp(np.abs(fa.R), -eA), np.ldexp(np.abs(ref.F_A.R), -eA)) <= 1e-11
        out = solver.solve_factored(Js, rxs, 9)
        assert (out.rankA, out.rankJ2, out.code) == (ref.rankA, ref.rankJ2, ref.code)
        assert rel(out.p, ref.p) <= 1e-11 and rel(np.ldexp(out.b, -eA), np.ldexp(ref.b, -eA)) <= 1e-11


def test_extreme_magnitudes_in_both_halves_of_a_pipelined_batch(solver):
    """130 problems = two pipelined halves on two handles: one problem of EACH half far outside the range of plain sums
    of squares.  Each half hands its problem to a rescue handle of its own; outputs in the caller's slots, accessors routed through
    the half to the rescue handle (need_factors)."""
    from enlsip_gn import FACTOR_J2
    m, n, t, B = 300, 66, 5, 130
    probs = [list(synth.make_problem(63000 + k, m, n, t)) for k in range(B)]
    for k, e in ((3, 700), (100, -650)):
        probs[k][0], probs[k][1] = np.ldexp(probs[k][0], e), np.ldexp(probs[k][1], e)
    probs[101][2], probs[101][3] = np.ldexp(probs[101][2], 600), np.ldexp(probs[101][3], 600)
    p, b, d, infos, jA, jL, jJ = solver.solve_batched(np.stack([np.ascontiguousarray(P[0].T) for P in probs]), np.stack([P[1] for P in probs]),
                                                      np.stack([np.ascontiguousarray(P[2]) for P in probs]), np.stack([P[3] for P in probs]))
    assert solver.pipeline_split() == 65 and "rescaled" in solver.route()
    for k in (0, 3, 4, 64, 65, 100, 101, 102, 129):
        ref = go.gn_subproblem(*probs[k])
        assert (infos[k][0], infos[k][1], infos[k][2], infos[k][5]) == (ref.rankA, ref.rankJ2, ref.code, 0), k
        assert rel(p[k], ref.p) <= 1e-11, (k, rel(p[k], ref.p))
        r = ref.rankJ2
        assert np.array_equal(jJ[k][:r], ref.jpvtJ2[:r]), k
        if r:
            fv = solver.factor(FACTOR_J2, prob=k)
            dg, dgr = np.abs(fv.diagR()[:r]), np.abs(ref.F_J2.diagR()[:r])
            assert np.abs(dg / dgr - 1.0).max() <= 1e-10, k


@pytest.mark.parametrize("m,n,t", [(700, 48, 6), (900, 200, 70), (512, 256, 64), (300, 40, 0)])
def test_matrix_times_QA_and_the_full_constraint_product(m, n, t, solver):
    """`M * F_A.Q` for ANY matrix with the row count of the last solve (what the Julia glue's `*(::AbstractMatrix, ::DeviceQ)`
    calls: the consumers' `J * F_A.Q`, src/enlsip_functions.jl:384, :526, :1249, unmodified) against the oracle's dormqr, and
    `A * p` with the FULL constraint Jacobian — inactive rows included (:2227) — against NumPy."""
    from enlsip_gn.api import GNError
    J, rx, A, cx = synth.make_problem(4300 + m, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = solver.solve(J, rx, A, cx)
    assert rel(solver.matrix_times_QA(J), ref.F_A.rmul_Q(J)) <= 1e-12           # the resident J itself: get_JQ1's result
    assert rel(solver.matrix_times_QA(J), solver.JQ1(m, n)) <= 1e-14
    rng = np.random.default_rng(m)
    M = rng.standard_normal((m, n))
    assert rel(solver.matrix_times_QA(M), ref.F_A.rmul_Q(M)) <= 1e-12           # any other matrix of that row count
    with pytest.raises(GNError):
        solver.matrix_times_QA(M[: m - 1])
    # the full constraint Jacobian: l rows of which only the first t are active
    l = t + 9
    Afull = np.vstack([A, rng.standard_normal((l - t, n))])
    got = solver.full_constraints_times(Afull, out.p)
    assert np.abs(got - Afull @ out.p).max() <= 1e-13 * (np.abs(Afull).sum(axis=1).max() * np.abs(out.p).max() + 1.0)
    assert solver.full_constraints_times(np.zeros((0, n)), out.p).shape == (0,)


@pytest.mark.parametrize("m,n,t,dims", [(600, 40, 6, ((6, 34), (4, 20), (0, 0), (6, 10))),
                                        (700, 300, 100, ((100, 200), (60, 150), (0, 0), (100, 10)))])   # 2nd: distributed constraint stage
def test_resolve_truncated_dims(m, n, t, dims, solver):
    """sub_search_direction re-entry (src/enlsip_functions.jl:1253) with dimA/dimJ2 below the ranks."""
    J, rx, A, cx = synth.make_problem(43, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    solver.solve(J, rx, A, cx)
    JQ1 = ref.F_A.rmul_Q(J)
    for dimA, dimJ2 in dims:
        p_ref, b_ref, d_ref = go.sub_search_direction(JQ1[:, :ref.rankA], rx, cx, ref.F_A, ref.F_L11, ref.F_J2,
                                                      n, t, ref.rankA, dimA, dimJ2, -1)
        p, b, d = solver.resolve(m, n, t, dimA, dimJ2, -1)
        assert rel(p, p_ref) <= 1e-11
        assert rel(b, b_ref) <= 1e-12
        assert abs(np.linalg.norm(d) - np.linalg.norm(d_ref)) <= 1e-12 * np.linalg.norm(d_ref)
        assert rel(np.abs(d[:dimJ2]), np.abs(d_ref[:dimJ2])) <= 1e-10


@pytest.mark.parametrize("batch,m,n,t", [(5, 300, 24, 3), (4, 500, 200, 100)])     # 2nd: distributed constraint stage, batched
def test_batched_matches_single(batch, m, n, t, solver):
    Js, rxs, Ats, cxs, refs = [], [], [], [], []
    for k in range(batch):
        J, rx, A, cx = (synth.make_rank_deficient_A if k == 2 else synth.make_problem)(600 + k, m, n, t)
        Js.append(np.ascontiguousarray(J.T))           # (n, m) C-order == column-major m x n
        rxs.append(rx)
        Ats.append(np.ascontiguousarray(A))            # (t, n) C-order == column-major n x t
        cxs.append(cx)
        refs.append(go.gn_subproblem(J, rx, A, cx))
    p, b, d, infos, jA, jL, jJ = solver.solve_batched(np.stack(Js), np.stack(rxs), np.stack(Ats), np.stack(cxs))
    for k, ref in enumerate(refs):
        assert infos[k][0] == ref.rankA and infos[k][1] == ref.rankJ2 and infos[k][2] == ref.code
        assert rel(p[k], ref.p) <= (TOL_P if ref.code == 1 else 1e-9)
        if ref.code == 1:
            assert np.array_equal(jJ[k][:n - ref.rankA], ref.jpvtJ2)


@pytest.mark.parametrize("shape", [(131, 96, 12, 2), (130, 700, 70, 5)])
def test_pipelined_device_batch_matches_oracle(shape, monkeypatch):
    """A device-pointer batch >= 128 is split over two streams inside the library (second half on a child
    handle): every problem must still match the oracle, and accessors must reach both halves.  The first shape belongs to
    the small uniform ones that are not split by default (ENLSIP_GN_PIPELINE=1 forces it), the second one is split by default."""
    import torch
    from enlsip_gn import FACTOR_J2, GNSolver
    batch, m, n, t = shape
    monkeypatch.setenv("ENLSIP_GN_PIPELINE", "1")
    solver = GNSolver(device=0)
    dev = torch.device("cuda", 0)
    Js, rxs, Ats, cxs, refs = [], [], [], [], []
    for k in range(batch):
        J, rx, A, cx = synth.make_problem(9000 + k, m, n, t)
        Js.append(np.ascontiguousarray(J.T)); rxs.append(rx); Ats.append(np.ascontiguousarray(A)); cxs.append(cx)
        refs.append(go.gn_subproblem(J, rx, A, cx))
    dJ = torch.from_numpy(np.stack(Js)).to(dev)
    drx = torch.from_numpy(np.stack(rxs)).to(dev)
    dAt = torch.from_numpy(np.stack(Ats)).to(dev)
    dcx = torch.from_numpy(np.stack(cxs)).to(dev)
    dp = torch.zeros(batch, n, dtype=torch.float64, device=dev)
    djJ = torch.zeros(batch, n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    solver.solve_batched_dev(batch, m, n, t, dJ.data_ptr(), m, m * n, drx.data_ptr(), dAt.data_ptr(), n, n * t,
                             dcx.data_ptr(), dp=dp.data_ptr(), djJ=djJ.data_ptr())
    p = dp.cpu().numpy()
    jJ = djJ.cpu().numpy()
    for k, ref in enumerate(refs):
        assert rel(p[k], ref.p) <= TOL_P, k
        assert np.array_equal(jJ[k][:n - ref.rankA], ref.jpvtJ2), k
    for k in (0, 64, 65, 66, batch - 1):          # both sides of the split at (batch + 1) // 2
        F = solver.factor(FACTOR_J2, prob=k)
        assert np.array_equal(F.p, refs[k].jpvtJ2)
        assert np.allclose(np.abs(F.diagR()), np.abs(np.diag(refs[k].F_J2.R)[: len(F.diagR())]), rtol=1e-9, atol=1e-12)
    solver.close()


@pytest.mark.parametrize("kind,m,n,t", [
    ("full", 600, 300, 40),          # F_A in registers (n t > 8192, RPL 8), blocked pivoted QR with RPL 8
    ("full", 1000, 512, 20),         # register F_A with few columns, n2 = 492
    ("full", 700, 257, 33),          # odd sizes, register F_A (RPL 8), kp = 224
    ("full", 900, 400, 64),          # t = 64 = the widest register F_A
    ("full", 350, 200, 50),          # register F_A with RPL 4, blocked QR with RPL 4
    ("rankdefA", 800, 300, 40),      # code -1 through the register F_A
    ("rankdefJ", 900, 260, 35),      # rank-deficient J2 through the blocked pivoted QR
    ("graded", 1200, 280, 30),       # graded singular values: norm-downdate recomputations
    ("full", 300, 330, 40),          # m < n2: kp = m
    ("full", 450, 520, 8),           # n2 = 512: the register form of the ranking sort at its limit (one record per thread)
    ("full", 450, 521, 8),           # n2 = 513: the LDS form of the ranking sort (P = 1024)
    ("full", 400, 700, 20),          # n2 = 680, kp = 400: the LDS form well inside its range
    ("full", 513, 140, 0),           # no constraints, one row past a tile boundary
    ("full", 1300, 600, 8),          # kp = 592 > 512 rows: launch-per-step pivoted QR (k_qd_*)
    ("rankdefJ", 1100, 560, 6),      # the same path with a rank-deficient J2
    ("full", 900, 400, 120),         # many constraints: F_A and F_L11 through the distributed pivoted QR
    ("rankdefA", 700, 300, 100),     # ... with a rank-deficient A (code -1: F_L11.Q' b carried through the factorisation)
    ("full", 600, 150, 200),         # ... with t > n (kA = n)
    ("full", 1100, 1000, 700),       # ... with more than 512 rows (16 rows per lane)
])
def test_shape_sweep_register_paths(kind, m, n, t, solver):
    """Shapes chosen to run the register-resident factorisations (k_geqp3_reg, k_sb_factor_reg), the gathered block
    update and the one-tile / partial-unit variants of the trailing update away from the C2 benchmark shape."""
    gen = {"full": synth.make_problem, "rankdefA": synth.make_rank_deficient_A, "rankdefJ": synth.make_rank_deficient_J,
           "graded": synth.make_graded_J}[kind]
    J, rx, A, cx = gen(31000 + m + n + t, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = solver.solve(J, rx, A, cx)
    if kind in ("full",):
        compare(out, ref, m, n)
    else:
        assert out.rankA == ref.rankA and out.rankJ2 == ref.rankJ2 and out.code == ref.code
        assert np.array_equal(out.jpvtA, ref.jpvtA)
        assert rel(out.p, ref.p) <= (1e-6 if kind == "graded" else 1e-9)
        if ref.code == 1:       # full-rank A: J2 is well defined, so are the leading pivots and |d| of the truncated factorisation
            r = ref.rankJ2
            assert np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r])
            assert sorted(out.jpvtJ2[r:]) == sorted(ref.jpvtJ2[r:])
            assert abs(np.linalg.norm(out.d) - np.linalg.norm(ref.d)) <= 1e-12 * np.linalg.norm(ref.d)
            assert np.abs(np.abs(out.d[:r]) - np.abs(ref.d[:r])).max() <= 1e-6 * np.abs(ref.d).max()


@pytest.mark.parametrize("kind,m,n,t", [
    ("full", 50, 20, 30),            # t > n: kA = n, code -1, carried b longer than the F_L11 steps, n2 = 0
    ("full", 100, 63, 0),            # n2 + 1 = 64: the widest wave-per-problem pivot stage
    ("full", 100, 64, 0),            # n2 + 1 = 65: falls back to the workgroup pivot stage
    ("full", 64, 64, 63),            # 63 reflectors of length 64, n2 = 1
    ("full", 30, 40, 5),             # m < n2: kp = m, columns that never pivot keep LAPACK's swap order
    ("full", 9, 64, 60),             # tiny m
    ("full", 200, 33, 33),           # t = n: J2 is empty
    ("rankdefA", 200, 30, 12),       # code -1, retry with a wider J2
    ("rankdefA", 40, 12, 20),        # rank-deficient and t > n
    ("rankdefJ", 256, 32, 4),        # rank-deficient J2 (C5 shape)
    ("graded", 512, 64, 8),          # graded singular values (C3 shape): norm-downdate recomputations
])
def test_small_wave_kernels(kind, m, n, t, solver):
    """Edge shapes of the one-wave-per-problem stages (k_constraint_small, k_jq1_rows, k_pivot_small, the passenger
    column of the CAQR factor kernel)."""
    gen = {"full": synth.make_problem, "rankdefA": synth.make_rank_deficient_A, "rankdefJ": synth.make_rank_deficient_J,
           "graded": synth.make_graded_J}[kind]
    J, rx, A, cx = gen(41000 + m + n + t, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = solver.solve(J, rx, A, cx)
    if kind == "full":
        compare(out, ref, m, n, pivots=(ref.code == 1))
    else:
        assert out.rankA == ref.rankA and out.rankJ2 == ref.rankJ2 and out.code == ref.code
        assert np.array_equal(out.jpvtA, ref.jpvtA)
        assert rel(out.p, ref.p) <= (1e-5 if kind == "graded" else 1e-9)
        if ref.code == 1:       # full-rank A: J2 is well defined, so are the leading pivots and |d| of the truncated factorisation
            r = ref.rankJ2
            assert np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r])
            assert abs(np.linalg.norm(out.d) - np.linalg.norm(ref.d)) <= 1e-12 * np.linalg.norm(ref.d)
            assert np.abs(np.abs(out.d[:r]) - np.abs(ref.d[:r])).max() <= 1e-6 * np.abs(ref.d).max()


def test_pipelined_batch_on_caller_stream():
    """Inputs produced on the caller's stream right before the call: the second pipeline half runs on the library's own
    stream and must be ordered after them (event fork in enlsip_gn_solve_batched_dev)."""
    import torch
    from enlsip_gn import GNSolver
    batch, m, n, t = 140, 128, 16, 3
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    Js, rxs, Ats, cxs, refs = [], [], [], [], []
    for k in range(batch):
        J, rx, A, cx = synth.make_problem(12000 + k, m, n, t)
        Js.append(np.ascontiguousarray(J.T)); rxs.append(rx); Ats.append(np.ascontiguousarray(A)); cxs.append(cx)
        refs.append(go.gn_subproblem(J, rx, A, cx))
    hJ = torch.from_numpy(np.stack(Js)).pin_memory()
    hrx = torch.from_numpy(np.stack(rxs)).pin_memory()
    hAt = torch.from_numpy(np.stack(Ats)).pin_memory()
    hcx = torch.from_numpy(np.stack(cxs)).pin_memory()
    os.environ["ENLSIP_GN_PIPELINE"] = "1"                          # this small shape is not split by default
    try:
        s = GNSolver(device=0, stream=st.cuda_stream)
    finally:
        del os.environ["ENLSIP_GN_PIPELINE"]
    try:
        with torch.cuda.stream(st):
            big = torch.randn(64, 1024, 1024, device=dev)          # keeps the stream busy for a while
            big = (big @ big).sum()
            dJ = hJ.to(dev, non_blocking=True) * 1.0               # inputs become valid only when the stream gets here
            drx = hrx.to(dev, non_blocking=True) * 1.0
            dAt = hAt.to(dev, non_blocking=True) * 1.0
            dcx = hcx.to(dev, non_blocking=True) * 1.0
            dp = torch.zeros(batch, n, dtype=torch.float64, device=dev)
            s.solve_batched_dev(batch, m, n, t, dJ.data_ptr(), m, m * n, drx.data_ptr(), dAt.data_ptr(), n, n * t,
                                dcx.data_ptr(), dp=dp.data_ptr())
        st.synchronize()
        p = dp.cpu().numpy()
        for k, ref in enumerate(refs):
            assert rel(p[k], ref.p) <= TOL_P, k
    finally:
        s.close()


@pytest.mark.parametrize("kind,m,n,t,scaling", [("full", 600, 40, 6, False), ("full", 4096, 512, 64, True),
                                                 ("rankdefA", 300, 40, 6, False), ("full", 200, 30, 30, True)])
def test_multiplier_estimates_on_device(kind, m, n, t, scaling, solver):
    """enlsip_gn_gradient / _first_lagrange / _second_lagrange against the oracle's restatement of
    first/second_lagrange_mult_estimate! (src/enlsip_functions.jl:461-537) on the same problem."""
    gen = synth.make_rank_deficient_A if kind == "rankdefA" else synth.make_problem
    J, rx, A, cx = gen(4242 + m + n, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = solver.solve(J, rx, A, cx)
    assert out.rankA == ref.rankA
    diag_scale = 1.0 + 0.25 * np.abs(synth.normal_stream(77, 5, t)) if scaling else np.ones(t)
    grad = J.T @ rx
    g_dev = solver.gradient(n)
    assert rel(g_dev, grad) <= 1e-13
    it = go.IterationRecord()
    lam_ref = go.first_lagrange_mult_estimate(A, grad, cx, scaling, diag_scale, ref.F_A, it, go.SQRT_EPS)
    for gfx in (grad, None):                       # host gradient and the device-computed one
        lam, gres = solver.first_lagrange(t, gfx, diag_scale if scaling else None)
        assert rel(lam, lam_ref) <= 1e-10
        assert abs(gres - it.grad_res) <= 1e-10 * max(1.0, abs(it.grad_res))
    lam2_ref = go.second_lagrange_mult_estimate(J, ref.F_A, rx, ref.p, t, scaling, diag_scale)
    lam2 = solver.second_lagrange(t, out.p, diag_scale if scaling else None)
    assert rel(lam2, lam2_ref) <= 1e-9
    Jp, Ap = solver.jacobian_times(m, t, out.p)          # line-search products (:2226-2229) on the resident J, A
    assert rel(Jp, J @ out.p) <= 1e-13
    assert np.abs(Ap - A @ out.p).max() <= 1e-13 * max(1.0, np.abs(A @ out.p).max())


@pytest.mark.parametrize("kind,m,n,t", [("full", 600, 40, 6), ("rankdefA", 300, 30, 9), ("full", 4096, 512, 64), ("full", 50, 20, 30)])
def test_factor_constraints_alone(kind, m, n, t, solver):
    """enlsip_gn_factor_constraints: the first factorisation of update_working_set (src/enlsip_functions.jl:700) and the
    multiplier estimate that follows it (:704), before any solve; J-related entries must refuse."""
    from enlsip_gn import FACTOR_A, FACTOR_L11, FACTOR_J2, GNError
    gen = synth.make_rank_deficient_A if kind == "rankdefA" else synth.make_problem
    J, rx, A, cx = gen(6100 + m + n, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    rankA, code, dimA = solver.factor_constraints(m, A, cx)
    assert (rankA, code, dimA) == (ref.rankA, ref.code, ref.rankA)
    for which, F in ((FACTOR_A, ref.F_A), (FACTOR_L11, ref.F_L11)):
        fv = solver.factor(which)
        assert rel(np.abs(fv.R), np.abs(F.R)) <= 1e-10 and np.array_equal(fv.p, F.p)
    grad = J.T @ rx
    it = go.IterationRecord()
    lam_ref = go.first_lagrange_mult_estimate(A, grad, cx, False, np.ones(t), ref.F_A, it, go.SQRT_EPS)
    lam, gres = solver.first_lagrange(t, grad, None)
    assert rel(lam, lam_ref) <= 1e-9 and abs(gres - it.grad_res) <= 1e-10 * max(1.0, abs(it.grad_res))
    with pytest.raises(GNError):
        solver.factor(FACTOR_J2).R
    with pytest.raises(GNError):
        solver.second_lagrange(t, ref.p, None)
    with pytest.raises(GNError):
        solver.resolve(m, n, t, ref.rankA, ref.rankJ2, -1)
    out = solver.solve_factored(J, rx, t)                 # :768-771: the solve goes on with the SAME factorisation
    assert (out.rankA, out.rankJ2, out.code) == (ref.rankA, ref.rankJ2, ref.code)
    assert rel(out.p, ref.p) <= 1e-9 and np.array_equal(out.jpvtA, ref.jpvtA)
    assert solver.factor(FACTOR_J2).R.shape == ref.F_J2.R.shape
    with pytest.raises(GNError):
        solver.solve_factored(J, rx, t)                   # no factor_constraints right before
    out = solver.solve(J, rx, A, cx)                      # the ordinary solve is unaffected
    assert rel(out.p, ref.p) <= 1e-9


@pytest.mark.parametrize("m,n,t", [(50, 10, 0), (1, 1, 0), (5, 1, 1), (2000, 600, 0), (3, 3, 3), (2, 4, 4), (700, 300, 300)])
def test_factored_flow_edge_shapes(m, n, t, solver):
    """factor_constraints -> first_lagrange -> solve_factored at the corners of the shape space: no constraints (t = 0),
    a single unknown, and a fully determined direction (t = n, rankJ2 = 0)."""
    J, rx, A, cx = synth.make_problem(123 + m + n + t, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    rankA, code, dimA = solver.factor_constraints(m, A, cx)
    assert (rankA, code) == (ref.rankA, ref.code)
    lam, gres = solver.first_lagrange(t, J.T @ rx, None)
    assert lam.shape == (t,) and np.isfinite(gres)
    out = solver.solve_factored(J, rx, t)
    assert (out.rankA, out.rankJ2) == (ref.rankA, ref.rankJ2)
    assert rel(out.p, ref.p) <= 1e-10


@pytest.mark.parametrize("m,n,t", [(256, 32, 4), (900, 300, 20), (4096, 512, 64), (700, 300, 100), (1300, 600, 8)])
@pytest.mark.parametrize("kind", ["nanJ", "nanA", "zeroA", "allzero"])
def test_nonfinite_and_degenerate_inputs_return(kind, m, n, t, solver):
    """NaN / all-zero inputs: whatever IEEE arithmetic makes of them, every pivot search must still return one of its
    candidates (a NaN norm ranks as +inf) — an empty search used to index with -1 and fault in k_geqp3_reg."""
    J, rx, A, cx = synth.make_problem(5, m, n, t)
    if kind == "nanJ":
        J[m // 2, n // 3] = np.nan
    elif kind == "nanA":
        A[0, 0] = np.nan
    elif kind == "zeroA":
        A[:] = 0.0
    else:
        J[:] = 0.0; A[:] = 0.0; rx[:] = 0.0; cx[:] = 0.0
    out = solver.solve(J, rx, A, cx)
    assert 0 <= out.rankA <= min(n, t) and 0 <= out.rankJ2 <= min(m, n)
    assert out.p.shape == (n,)
    if kind == "zeroA":
        assert out.rankA == 0 and out.code == -1 and np.all(np.isfinite(out.p))
    if kind == "allzero":
        assert out.rankA == 0 and out.rankJ2 == 0 and np.all(out.p == 0.0)
    J, rx, A, cx = synth.make_problem(6, m, n, t)            # the handle is fine afterwards
    ref = go.gn_subproblem(J, rx, A, cx)
    assert rel(solver.solve(J, rx, A, cx).p, ref.p) <= TOL_P


def _plateau_problem(pid, m, n, t):
    """J whose J2 = (J * F_A.Q)[:, t+1:] has a spectrum of three plateaus and an exactly rank-deficient rest, every plateau at
    least 10x away from the pseudo_rank threshold |R11| sqrt(len) eps_rank of the eps_rank it is meant for (R11 lies between
    0.3 and 1 for this spectrum): eps_rank = 1e-3 must count plateau 1 only, 1e-6 plateaus 1-2, 1e-12 plateaus 1-3."""
    _, rx, A, cx = synth.make_problem(pid, m, n, t)
    n2 = n - t
    F_A = go.qr_colnorm(A.T)
    Q1 = F_A.Q_mul(np.eye(n))
    r1, r2, r3 = n2 // 3, n2 // 4, n2 // 5
    L = np.sqrt(n2)
    s = np.zeros(n2)
    s[:r1] = np.linspace(1.0, 0.5, r1)
    s[r1:r1 + r2] = L * np.linspace(2e-5, 1e-5, r2)
    s[r1 + r2:r1 + r2 + r3] = L * np.linspace(2e-8, 1e-8, r3)
    U, _ = np.linalg.qr(synth.normal_stream(pid, 6, m * n2).reshape((m, n2), order="F"))
    V, _ = np.linalg.qr(synth.normal_stream(pid, 7, n2 * n2).reshape((n2, n2), order="F"))
    J2 = (U * s) @ V.T
    J1 = synth.normal_stream(pid, 8, m * t).reshape((m, t), order="F")
    return np.hstack([J1, J2]) @ Q1.T, rx, A, cx, (r1, r1 + r2, r1 + r2 + r3)


@pytest.mark.parametrize("m,n,t", [(300, 40, 5), (1200, 280, 30), (512, 64, 8), (700, 300, 100)])
def test_eps_rank_drives_the_ranks(m, n, t, solver):
    """pseudo_rank (src/enlsip_functions.jl:17-31) under three thresholds on a spectrum with a >= 10x gap either side of every
    threshold: the ranks are EXACTLY the oracle's (and the designed ones), and p follows the oracle."""
    J, rx, A, cx, designed = _plateau_problem(7100 + m + n, m, n, t)
    got = []
    for eps_rank, want in zip((1e-3, 1e-6, 1e-12), designed):
        ref = go.gn_subproblem(J, rx, A, cx, eps_rank)
        out = solver.solve(J, rx, A, cx, eps_rank=eps_rank)
        assert (out.rankA, out.code) == (ref.rankA, ref.code) == (t, 1)
        assert out.rankJ2 == ref.rankJ2 == want, (eps_rank, out.rankJ2, ref.rankJ2, want)
        assert np.array_equal(out.jpvtJ2[:want], ref.jpvtJ2[:want])
        assert rel(out.p, ref.p) <= 1e-6              # cond up to 1e8 at the lowest threshold
        got.append(out.rankJ2)
    assert got[0] < got[1] < got[2] < n - t


@pytest.mark.parametrize("m,n,t", [(600, 40, 6), (900, 300, 20), (700, 300, 100)])
def test_solve_with_dimension_overrides(m, n, t, solver):
    """enlsip_gn_solve with dimA / dimJ2 below the ranks (the subspace-minimisation dimensions of
    src/enlsip_functions.jl:1118-1176) against sub_search_direction of the oracle with the same dimensions."""
    J, rx, A, cx = synth.make_problem(7300 + m + n, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    JQ1 = ref.F_A.rmul_Q(J)
    for dimA, dimJ2 in ((t, ref.rankJ2 - 3), (t - 2, ref.rankJ2 // 2), (0, 1)):
        p_ref, b_ref, d_ref = go.sub_search_direction(JQ1[:, :ref.rankA], rx, cx, ref.F_A, ref.F_L11, ref.F_J2,
                                                      n, t, ref.rankA, dimA, dimJ2, ref.code)
        out = solver.solve(J, rx, A, cx, dimA=dimA, dimJ2=dimJ2)
        assert (out.rankA, out.rankJ2, out.dimA, out.dimJ2) == (ref.rankA, ref.rankJ2, dimA, dimJ2)
        assert rel(out.p, p_ref) <= 1e-10             # (code 1 ignores dimA in both: full triangular solve, :131-133)


def _newton_reference(J, rx, A, cx, Gam):
    """The lines of newton_search_direction after its Hessian sums (src/enlsip_functions.jl:398-421) on the oracle's factors."""
    m, n = J.shape
    t = A.shape[0]
    ref = go.gn_subproblem(J, rx, A, cx)
    r = ref.rankA
    Q1 = ref.F_A.Q_mul(np.eye(n))
    if r == t:
        b = -cx[ref.F_A.p - 1]
        p1 = np.linalg.solve(np.triu(ref.F_A.R)[:t, :t].T, b) if t else np.zeros(0)
    else:                           # :371-373
        b = ref.F_L11.Qt_mul(-cx[ref.F_A.p - 1])
        dp1 = np.linalg.solve(np.triu(ref.F_L11.R)[:r, :r], b[:r])
        p1 = ref.F_L11.P[:r, :r] @ dp1
    if r == n:
        return p1, False            # :374-376: with rankA == n the reference returns p1 as it is (no F_A.Q applied)
    JQ1 = J @ Q1
    J1, J2 = JQ1[:, :r], JQ1[:, r:]
    E = Q1.T @ Gam @ Q1
    if t > r:                       # :396-399 (needs t >= n: F_L11.p has min(n, t) entries)
        vp = ref.F_L11.p - 1
        E = E[np.ix_(vp, vp)]
    t = r
    W22 = E[t:, t:] + J2.T @ J2
    W21 = E[t:, :t] + J2.T @ J1
    d = -W21 @ p1 - J2.T @ rx
    sW = (W22 + W22.T) * 0.5
    try:
        L = np.linalg.cholesky(sW)
    except np.linalg.LinAlgError:
        return np.zeros(n), True
    p2 = np.linalg.solve(L.T, np.linalg.solve(L, d))
    return Q1 @ np.concatenate([p1, p2]), False


@pytest.mark.parametrize("m,n,t", [(60, 12, 3), (300, 40, 0), (500, 130, 20), (54, 20, 13), (40, 9, 9)])
def test_newton_direction_on_device(m, n, t, solver):
    """enlsip_gn_newton_direction (src/enlsip_functions.jl:348-423 after the Hessian sums) against the same lines in NumPy on the
    oracle's factors: a positive definite case, an indefinite one (error = true, p = 0), and after a truncated re-solve."""
    J, rx, A, cx = synth.make_problem(8800 + m + n + t, m, n, t)
    rng = np.random.default_rng(m + n)
    S = rng.standard_normal((n, n))
    Gam = 0.3 * (S + S.T) + 0.1 * rng.standard_normal((n, n))          # not exactly symmetric, as r_mat - c_mat need not be
    solver.solve(J, rx, A, cx)
    p, err = solver.newton_direction(Gam)
    pr, errr = _newton_reference(J, rx, A, cx, Gam)
    assert err == errr
    assert rel(p, pr) <= 1e-10 if not errr else not p.any()
    if n - t > 0:
        bad = Gam - 1e4 * np.eye(n)                                      # W22 indefinite
        p, err = solver.newton_direction(bad)
        assert err and not p.any() and _newton_reference(J, rx, A, cx, bad)[1]
        if t > 1:                                                        # a truncated re-solve must not leak its p1 into the Newton step
            solver.resolve(m, n, t, t - 1, max(n - t - 1, 0), code=-1)
            p, err = solver.newton_direction(Gam)
            assert err == errr and (errr or rel(p, pr) <= 1e-10)
    # the resident factors are still those of the solve
    assert np.array_equal(solver.factor(0).p, go.gn_subproblem(J, rx, A, cx).jpvtA)


def test_newton_direction_rank_deficient_working_set(solver):
    """t >= n > rankA: the E[F_L11.p, F_L11.p] branch of newton_search_direction (:371-373, :396-399) on the device against the same
    lines in NumPy; with n > t > rankA the reference runs out of bounds and the entry point refuses."""
    from enlsip_gn import GNError
    rng = np.random.default_rng(77)
    m, n, t, r = 50, 7, 9, 4
    J, rx, _, _ = synth.make_problem(515, m, n, 0)
    A = rng.standard_normal((t, r)) @ rng.standard_normal((r, n))       # rank 4 < n <= t
    cx = rng.standard_normal(t)
    S = rng.standard_normal((n, n))
    Gam = 0.3 * (S + S.T) + 5.0 * np.eye(n)
    out = solver.solve(J, rx, A, cx)
    assert (out.rankA, out.code) == (r, -1)
    p, err = solver.newton_direction(Gam)
    pr, errr = _newton_reference(J, rx, A, cx, Gam)
    assert err == errr and not errr and rel(p, pr) <= 1e-9
    # n > t > rankA: refused
    A2 = rng.standard_normal((5, 3)) @ rng.standard_normal((3, n))
    out = solver.solve(J, rx, A2, rng.standard_normal(5))
    assert out.rankA == 3
    with pytest.raises(GNError):
        solver.newton_direction(Gam)


def test_argument_errors(solver):
    from enlsip_gn import GNError
    J, rx, A, cx = synth.make_problem(1, 50, 10, 2)
    with pytest.raises(GNError):
        solver.solve(np.zeros((50, 2000)), rx, np.zeros((0, 2000)), np.zeros(0))   # n > limit


def test_linearity_in_rhs(solver):
    """Size-independent property at C2 scale: p is linear in (rx, cx) for fixed J, A."""
    m, n, t = 4096, 512, 64
    J, rx, A, cx = synth.make_problem(7, m, n, t)
    p1 = solver.solve(J, rx, A, cx).p
    p2 = solver.solve(J, 2.0 * rx, A, 2.0 * cx).p
    assert rel(p2, 2.0 * p1) <= 1e-12
    rx2 = synth.normal_stream(8, 1, m)
    p3 = solver.solve(J, rx2, A, np.zeros(t)).p
    p4 = solver.solve(J, rx + rx2, A, cx).p
    assert rel(p4, p1 + p3) <= 1e-11


@pytest.mark.parametrize("m,n,t,G", [(4000, 64, 0, 4), (3001, 48, 5, 3), (20000, 96, 0, 8), (300, 40, 6, 2),
                                     (6000, 300, 10, 3), (5000, 600, 4, 2),      # combine stage: register blocks / > 512 rows
                                     (8800, 1008, 0, 8), (16384, 144, 16, 8)])    # stacks whose tiles are rotated out EXACTLY (below)
def test_tsqr_row_shards_match_single_solve(m, n, t, G, solver):
    """Row-sharded TSQR (config C4 structure) rehearsed on one GPU: G local stages + combine must
    reproduce the oracle's p, ranks, pivots and ||d|| of the unsharded problem.
    (8800, 1008, 0, 8): the combine stage factors a stack of eight 1008-row triangles; with panel pairs a tile of the stack holds
    nothing but the first 32 rows of the next triangle, the first panel rotates them out exactly, and the second panel of the pair
    meets columns of rounding dust whose norms fall by 1e-16 per column down to 1e-159 — reflectors built from underflowing sums
    of squares were not orthogonal (p off by 1e-10); make_reflector now treats columns below 1e-140 as zero."""
    from enlsip_gn.tsqr import tsqr_solve_shards
    J, rx, A, cx = synth.make_problem(11 if m == 8800 else 900 + m, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    res = tsqr_solve_shards(solver, J, rx, A, cx, G)
    assert res.n2 == n - ref.rankA and res.rankA == ref.rankA and res.rankJ2 == ref.rankJ2
    assert rel(res.p, ref.p) <= TOL_P
    assert abs(res.d_norm - np.linalg.norm(ref.d)) <= 1e-11 * np.linalg.norm(ref.d)
    assert np.array_equal(res.jpvtJ2, ref.jpvtJ2)
    r = ref.rankJ2
    assert np.abs(np.abs(res.dlead[:r]) - np.abs(ref.d[:r])).max() <= 1e-10 * max(np.abs(ref.d).max(), 1.0)


def _ws_case(kind):
    """(J, rx, A_all, cx_all, q, active ids 1-based) for update_working_set comparisons."""
    if kind == "hs65":
        J, rx, A, cx, active = hs65_start()
        return J, rx, A, cx, 0, active
    m, n, l = 200, 12, 7
    J, rx, A, cx = synth.make_problem(4242, m, n, l)
    if kind == "eq_only":
        return J, rx, A, cx, l, np.arange(1, l + 1)
    # 2 equalities + 3 active inequalities out of 5
    return J, rx, A, cx, 2, np.array([1, 2, 3, 5, 6])


@pytest.mark.parametrize("kind", ["hs65", "eq_only", "mixed"])
@pytest.mark.parametrize("scaling", [False, True])
def test_update_working_set_mirror(kind, scaling, solver):
    """Host mirror of update_working_set over the HIP library vs the oracle's restatement of
    src/enlsip_functions.jl:686-795: same direction, same working-set mutations, same multipliers."""
    from enlsip_gn import working_set as ws
    J, rx, A, cx, q, active = _ws_case(kind)
    m, n = J.shape
    l = A.shape[0]

    def build(mod):
        W = mod.new_working_set(q, l) if hasattr(mod, "new_working_set") else mod.WorkingSet.create(q, l)
        add = (lambda s: mod.add_constraint(W, s)) if hasattr(mod, "add_constraint") else W.add_constraint
        for idx in active:
            if idx > q:
                add(int(np.where(W.inactive == idx)[0][0]) + 1)
        C = mod.Constraint(cx[active - 1].copy(), A[active - 1].copy(), scaling, np.zeros(len(active)))
        return W, C

    Wr, Cr = build(go)
    go.evaluate_scaling(Cr)
    itr = go.IterationRecord()
    pr = np.zeros(n)
    nref = []
    go.update_working_set(Wr, rx, A, Cr, J.T @ rx, J, pr, itr, go.SQRT_EPS, solve=lambda: nref.append(1))

    Wg, Cg = build(ws)
    # same scaling step as the reference performs before the call (src/enlsip_functions.jl:2696)
    Cg.A, Cg.cx, Cg.diag_scale = Cr_scaled_inputs(A, cx, active, scaling)
    itg = ws.IterationRecord()
    pg = np.zeros(n)
    ngpu = []
    F_A, F_L11, F_J2 = ws.update_working_set(solver, Wg, rx, A, Cg, J.T @ rx, J, pg, itg, go.SQRT_EPS,
                                             on_solve=lambda: ngpu.append(1))
    assert Wg.t == Wr.t and np.array_equal(Wg.active, Wr.active) and np.array_equal(Wg.inactive, Wr.inactive)
    assert itg.delete == itr.delete and itg.index_del == itr.index_del
    assert (itg.rankA, itg.rankJ2) == (itr.rankA, itr.rankJ2)
    assert rel(pg, pr) <= 1e-10
    np.testing.assert_allclose(itg.lam, itr.lam, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(itr.lam).max()))
    assert len(ngpu) in (len(nref), len(nref) + 1)       # the device form pays one extra solve only when s != 0
    assert F_J2.R.shape[1] == n - itg.rankA


def Cr_scaled_inputs(A, cx, active, scaling):
    """evaluate_scaling! (src/structures.jl:160-178) applied to fresh copies."""
    Aa, ca = A[active - 1].copy(), cx[active - 1].copy()
    diag = np.zeros(len(active))
    for i in range(len(active)):
        row = float(np.linalg.norm(Aa[i]))
        diag[i] = row
        if scaling:
            if abs(row) < np.finfo(float).eps:
                row = 1.0
            Aa[i] /= row
            ca[i] /= row
            diag[i] = 1.0 / row
    return Aa, ca, diag
