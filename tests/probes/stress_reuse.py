"""Handle-reuse stress (diagnostic): one handle, random shapes in random order with a high share of rank-deficient / zero /
NaN constraint matrices, so that speculative first passes meet workspaces full of other problems' data.  Checks that every
call returns and that full-rank problems still match the oracle.  usage: python3 tests/probes/stress_reuse.py [calls=300] [seed=1]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
s = GNSolver(device=0)
bad = 0
t0 = time.time()
for k in range(calls):
    cls = rng.integers(0, 4)
    if cls == 0: n = int(rng.integers(4, 65)); m = int(rng.integers(n, 600)); t = int(rng.integers(1, min(n, 60) + 1))
    elif cls == 1: n = int(rng.integers(65, 320)); m = int(rng.integers(n, 1500)); t = int(rng.integers(1, 64))
    elif cls == 2: n = int(rng.integers(320, 700)); m = int(rng.integers(300, 1600)); t = int(rng.integers(1, 64))
    else: n = int(rng.integers(100, 500)); m = int(rng.integers(n, 1200)); t = int(rng.integers(65, min(n, 300)))
    kind = ["full", "rankdefA", "zeroA", "nanA", "dupA", "nanJ", "infJ", "zeroJ"][int(rng.integers(0, 8))]
    J, rx, A, cx = (synth.make_rank_deficient_A if (kind == "rankdefA" and t >= 2) else synth.make_problem)(80000 + k, m, n, t)
    if kind == "zeroA": A[:] = 0.0
    if kind == "nanA": A[int(rng.integers(0, t)), int(rng.integers(0, n))] = np.nan
    if kind == "dupA" and t >= 3: A[1:] = A[0]                 # rank 1
    if kind == "nanJ": J[int(rng.integers(0, m)), int(rng.integers(0, n))] = np.nan
    if kind == "infJ": J[int(rng.integers(0, m)), int(rng.integers(0, n))] = np.inf
    if kind == "zeroJ": J[:] = 0.0
    if os.environ.get("STRESS_VERBOSE"): print(k, kind, m, n, t, flush=True)
    out = s.solve(J, rx, A, cx)
    ok = out.p.shape == (n,) and 0 <= out.rankA <= min(n, t)
    if kind == "full" and k % 4 == 0:
        ref = go.gn_subproblem(J, rx, A, cx)
        nb = np.linalg.norm(ref.p)
        ok = ok and np.linalg.norm(out.p - ref.p) <= 1e-9 * (nb if nb else 1.0)
    if k % 3 == 0:          # the consumers of the resident factors after whatever the solve left behind (errors are fine, crashes not)
        from enlsip_gn import FACTOR_A, FACTOR_L11, FACTOR_J2, GNError
        try:
            for which, ln in ((FACTOR_A, n), (FACTOR_L11, t), (FACTOR_J2, m)):
                fv = s.factor(which)
                R = fv.R
                if R.shape[0]:
                    fv.Qt_mul(np.ones(ln)); fv.Q_mul(np.ones(ln))
            s.JQ1(m, n)
            s.gradient(n); s.jacobian_times(m, t, out.p)
            s.first_lagrange(t, None, None); s.second_lagrange(t, out.p, None)
            kp = min(m, n - out.rankA)
            s.resolve(m, n, t, int(rng.integers(0, out.rankA + 1)), int(rng.integers(0, max(min(out.rankJ2, kp), 0) + 1)), -1)
        except GNError:
            pass
    if not ok:
        bad += 1
        print("FAIL", kind, m, n, t, out.rankA, out.rankJ2, flush=True)
# batched calls with degenerate members
for k in range(calls // 20):
    n = int(rng.integers(8, 200)); m = int(rng.integers(n, 800)); t = int(rng.integers(1, min(n, 80))); B = int(rng.integers(2, 9))
    Js, rxs, Ats, cxs = [], [], [], []
    for b in range(B):
        J, rx, A, cx = synth.make_problem(90000 + 16 * k + b, m, n, t)
        kind = int(rng.integers(0, 5))
        if kind == 1: A[:] = 0.0
        if kind == 2: A[0, 0] = np.nan
        if kind == 3 and t >= 2: A[1:] = A[0]
        if kind == 4: J[0, 0] = np.nan
        Js.append(np.ascontiguousarray(J.T)); rxs.append(rx); Ats.append(np.ascontiguousarray(A)); cxs.append(cx)
    p, b_, d, infos, jA, jL, jJ = s.solve_batched(np.stack(Js), np.stack(rxs), np.stack(Ats), np.stack(cxs))
    if p.shape != (B, n):
        bad += 1
print(f"{calls} calls (+ {calls // 20} batched), {bad} failures, {time.time() - t0:.1f} s")
s.close()
sys.exit(1 if bad else 0)
