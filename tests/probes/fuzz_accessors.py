"""Randomised shapes through the ACCESSORS of a finished solve (factor views R / diagR / jpvt / Q'v / Qv of F_A, F_L11, F_J2, J*Q1,
the re-solve with truncated dimensions) against the oracle — the consumers of the stored reflectors and T factors.
usage: python3 tests/probes/fuzz_accessors.py [cases=60] [seed=1]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver, FACTOR_A, FACTOR_L11, FACTOR_J2


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    s = GNSolver(device=0)
    bad = 0
    t0 = time.time()
    for k in range(cases):
        cls = int(rng.integers(0, 4))
        if cls == 0:      # one tile, wave-per-problem kernels, passenger panels
            n = int(rng.integers(8, 65)); m = int(rng.integers(n + 2, 520)); t = int(rng.integers(1, min(n - 2, 40) + 1))
        elif cls == 1:    # several tiles, last panel narrow
            n = int(rng.integers(40, 200)); m = int(rng.integers(n + 10, 2600)); t = int(rng.integers(1, min(n - 4, 60) + 1))
        elif cls == 2:    # register paths of the pivoted QR
            n = int(rng.integers(200, 560)); m = int(rng.integers(n + 10, 1800)); t = int(rng.integers(1, 64))
        else:             # many constraints
            n = int(rng.integers(80, 300)); m = int(rng.integers(n + 10, 900)); t = int(rng.integers(65, n - 4))
        J, rx, A, cx = synth.make_problem(int(rng.integers(1, 1 << 30)), m, n, t)
        ref = go.gn_subproblem(J, rx, A, cx)
        out = s.solve(J, rx, A, cx)
        msgs = []
        if (out.rankA, out.rankJ2, out.code) != (ref.rankA, ref.rankJ2, ref.code):
            msgs.append("ranks")
        for name, which, F in (("A", FACTOR_A, ref.F_A), ("L11", FACTOR_L11, ref.F_L11), ("J2", FACTOR_J2, ref.F_J2)):
            fv = s.factor(which)
            if fv.R.shape != F.R.shape or rel(np.abs(fv.R), np.abs(F.R)) > 1e-10:
                msgs.append(f"R({name})")
            if not np.array_equal(fv.p, F.p):
                msgs.append(f"p({name})")
            v = rng.standard_normal(F.rows)
            qt = fv.Qt_mul(v)
            if abs(np.linalg.norm(qt) - np.linalg.norm(v)) > 1e-11 * np.linalg.norm(v):
                msgs.append(f"|Q'v|({name})")
            if rel(np.abs(qt[:F.k]), np.abs(F.Qt_mul(v)[:F.k])) > 1e-9:
                msgs.append(f"Q'v({name})")
            if rel(fv.Q_mul(qt), v) > 1e-11:
                msgs.append(f"QQ'v({name})")
        if rel(s.JQ1(m, n), ref.F_A.rmul_Q(J)) > 1e-11:
            msgs.append("JQ1")
        dimA = int(rng.integers(0, ref.rankA + 1)); dimJ2 = int(rng.integers(0, ref.rankJ2 + 1))
        JQ1 = ref.F_A.rmul_Q(J)
        p_ref, b_ref, d_ref = go.sub_search_direction(JQ1[:, :ref.rankA], rx, cx, ref.F_A, ref.F_L11, ref.F_J2, n, t, ref.rankA, dimA, dimJ2, -1)
        p, b, d = s.resolve(m, n, t, dimA, dimJ2, -1)
        if rel(p, p_ref) > 1e-9 or rel(b, b_ref) > 1e-10:
            msgs.append(f"resolve({dimA},{dimJ2}) {rel(p, p_ref):.1e}")
        if msgs:
            bad += 1
            print(f"FAIL m={m} n={n} t={t}: {' '.join(msgs)}", flush=True)
    s.close()
    print(f"{cases} cases, {bad} failures, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
