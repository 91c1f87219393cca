"""Randomised shape sweep of the HIP path against the oracle (diagnostic; the regular suite holds the fixed cases).
usage: python3 tests/probes/fuzz_gpu.py [cases=120] [seed=1]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    s = GNSolver(device=0)
    bad = 0
    t0 = time.time()
    for k in range(cases):
        cls = rng.integers(0, 5)
        if cls == 0:      # tiny
            n = int(rng.integers(1, 12)); m = int(rng.integers(1, 30)); t = int(rng.integers(0, n + 3))
        elif cls == 1:    # small wave kernels
            n = int(rng.integers(8, 65)); m = int(rng.integers(n // 2 + 1, 700)); t = int(rng.integers(0, min(n, 63) + 1))
        elif cls == 2:    # mid
            n = int(rng.integers(65, 300)); m = int(rng.integers(n, 2500)); t = int(rng.integers(0, min(n, 80) if rng.integers(0, 2) else n + 20))
        elif cls == 3:    # register paths / wide / many constraints
            n = int(rng.integers(300, 640)); m = int(rng.integers(200, 1800))
            t = int(rng.integers(0, 70)) if rng.integers(0, 3) else int(rng.integers(65, 400))
        else:             # m < n
            n = int(rng.integers(20, 200)); m = int(rng.integers(1, n)); t = int(rng.integers(0, min(n, 40)))
        kind = ["full", "full", "full", "rankdefA", "rankdefJ", "graded"][int(rng.integers(0, 6))]
        if kind == "rankdefA" and t < 2: kind = "full"
        if kind in ("rankdefJ", "graded") and (n - t < 3 or m < 4): kind = "full"
        gen = {"full": synth.make_problem, "rankdefA": synth.make_rank_deficient_A, "rankdefJ": synth.make_rank_deficient_J,
               "graded": synth.make_graded_J}[kind]
        try:
            J, rx, A, cx = gen(90000 + k, m, n, t)
            ref = go.gn_subproblem(J, rx, A, cx)
        except Exception as e:          # generator / oracle limits, not ours
            print(f"skip {kind} m={m} n={n} t={t}: {type(e).__name__}")
            continue
        out = s.solve(J, rx, A, cx)
        tol = 1e-5 if kind == "graded" else (1e-8 if kind != "full" else 1e-10)
        ok = (out.rankA, out.code) == (ref.rankA, ref.code) and np.all(np.isfinite(out.p))
        if kind != "graded":
            ok = ok and out.rankJ2 == ref.rankJ2
        if kind == "rankdefA" and m < n - ref.rankA:
            # Q1's null-space columns are rounding-determined for a rank-deficient A, and an underdetermined J2 (m < n2) takes a
            # BASIC solution in that basis: p itself is not unique across implementations, its residuals are
            rj, ra = np.linalg.norm(J @ out.p + rx), np.linalg.norm(A @ out.p + cx)
            rj0, ra0 = np.linalg.norm(J @ ref.p + rx), np.linalg.norm(A @ ref.p + cx)
            ok = ok and abs(rj - rj0) <= 1e-8 * max(1.0, rj0) and abs(ra - ra0) <= 1e-8 * max(1.0, ra0)
        else:
            ok = ok and rel(out.p, ref.p) <= tol
        if ok and kind == "full" and n <= 160 and m >= n and (k % 3 == 0):
            # resident factors behind the accessors, re-solve with truncated dimensions, multiplier estimates
            from enlsip_gn import FACTOR_A, FACTOR_L11, FACTOR_J2
            for which, F in ((FACTOR_A, ref.F_A), (FACTOR_L11, ref.F_L11), (FACTOR_J2, ref.F_J2)):
                fv = s.factor(which)
                ok = ok and fv.R.shape == F.R.shape and (F.R.size == 0 or rel(np.abs(fv.R), np.abs(F.R)) <= 1e-9)
                ok = ok and np.array_equal(fv.p, F.p)
                if F.rows:
                    v = rng.standard_normal(F.rows)
                    ok = ok and rel(fv.Q_mul(fv.Qt_mul(v)), v) <= 1e-11
            if ref.rankA and n - ref.rankA > 1:
                dimA = int(rng.integers(0, ref.rankA + 1)); dimJ2 = int(rng.integers(0, ref.rankJ2 + 1))
                JQ1 = ref.F_A.rmul_Q(J)
                p_ref, b_ref, d_ref = go.sub_search_direction(JQ1[:, :ref.rankA], rx, cx, ref.F_A, ref.F_L11, ref.F_J2,
                                                              n, t, ref.rankA, dimA, dimJ2, -1)
                p2, b2, d2 = s.resolve(m, n, t, dimA, dimJ2, -1)
                ok = ok and rel(p2, p_ref) <= 1e-9 and rel(b2, b_ref) <= 1e-10
            if t:
                lam_ref = go.first_lagrange_mult_estimate(A, J.T @ rx, cx, False, np.ones(t), ref.F_A, go.IterationRecord(), go.SQRT_EPS)
                lam, _ = s.first_lagrange(t, None, None)
                ok = ok and rel(lam, lam_ref) <= 1e-8
            if not ok:
                print("  (accessor / resolve / multiplier stage)")
        if ok and (k % 4 == 1) and not (kind == "rankdefA" and m < n - ref.rankA):
            # the update_working_set sequence: constraint stage alone, first estimate, the solve that goes on with it;
            # then the second estimate and the line-search products on the resident data (any kind, any size)
            try:
                rA, code, dimA = s.factor_constraints(m, A, cx)
                ok2 = (rA, code) == (ref.rankA, ref.code)
                g = J.T @ rx
                if t:
                    it = go.IterationRecord()
                    lam_ref = go.first_lagrange_mult_estimate(A, g, cx, False, np.ones(t), ref.F_A, it, go.SQRT_EPS)
                    lam, gres = s.first_lagrange(t, g, None)
                    if kind == "full":
                        ok2 = ok2 and rel(lam, lam_ref) <= 1e-8 and abs(gres - it.grad_res) <= 1e-8 * max(1.0, abs(it.grad_res))
                    else:
                        ok2 = ok2 and np.all(np.isfinite(lam))
                o2 = s.solve_factored(J, rx, t)
                if not ok2: print("   first estimate / ranks")
                ok2 = ok2 and (o2.rankA, o2.code) == (ref.rankA, ref.code) and rel(o2.p, out.p) <= 1e-12
                if not ok2: print(f"   solve_factored vs solve: rel {rel(o2.p, out.p):.2e}")
                Jp, Ap = s.jacobian_times(m, t, o2.p)
                jp_bound = 1e-13 * max(n, 8) * float((np.abs(J) @ np.abs(o2.p)).max() + 1e-300)
                ok2 = ok2 and np.abs(Jp - J @ o2.p).max() <= jp_bound and (t == 0 or np.abs(Ap - A @ o2.p).max() <= 1e-13 * max(n, 8) * float((np.abs(A) @ np.abs(o2.p)).max() + 1e-300))
                if not ok2: print("   jacobian_times")
                if t and kind == "full":
                    lam2_ref = go.second_lagrange_mult_estimate(J, ref.F_A, rx, ref.p, t, False, np.ones(t))
                    lam2 = s.second_lagrange(t, o2.p, None)
                    # J1'(rx + J p) is pure rounding when J2 is underdetermined (zero residual): absolute scale of the estimate
                    dR = np.abs(np.diag(ref.F_A.R))
                    scale2 = np.linalg.norm(lam2_ref) + np.linalg.norm(J) * (np.linalg.norm(rx) + np.linalg.norm(J) * np.linalg.norm(ref.p)) / dR.min()
                    e2 = float(np.linalg.norm(lam2 - lam2_ref))
                    if not (e2 <= 1e-9 * scale2): print(f"   second estimate abs {e2:.2e} scale {scale2:.2e}")
                    ok2 = ok2 and e2 <= 1e-9 * scale2
            except Exception as e:
                ok2 = False
                print(f"  exception in the factored flow: {type(e).__name__}: {e}")
            if not ok2:
                ok = False
                print("  (factor_constraints / solve_factored / estimates stage)")
        if not ok:
            bad += 1
            print(f"FAIL {kind} m={m} n={n} t={t}: ranks hip ({out.rankA},{out.rankJ2},{out.code}) ref ({ref.rankA},{ref.rankJ2},{ref.code}) "
                  f"rel p {rel(out.p, ref.p):.2e}")
    print(f"{cases} cases, {bad} failures, {time.time() - t0:.1f} s")
    s.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
