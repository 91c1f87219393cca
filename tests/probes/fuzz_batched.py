"""Randomised BATCHES of mixed problems (full rank, rank-deficient A with different defects, rank-deficient / graded J2) against
the oracle: problems of one launch then differ in rankA, n2, kp and code (diagnostic; the regular suite holds fixed cases).
usage: python3 tests/probes/fuzz_batched.py [batches=40] [seed=1]"""
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
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    s = GNSolver(device=0)
    bad = 0
    solved = 0
    t0 = time.time()
    for it in range(nb):
        cls = int(rng.integers(0, 5))
        if cls == 0:      # wave-per-problem kernels
            n = int(rng.integers(4, 65)); m = int(rng.integers(n, 600)); t = int(rng.integers(2, min(n, 40) + 1)); batch = int(rng.integers(2, 200))
        elif cls == 1:    # mid
            n = int(rng.integers(65, 260)); m = int(rng.integers(n, 1500)); t = int(rng.integers(2, 70)); batch = int(rng.integers(2, 24))
        elif cls == 2:    # register / blocked pivoted QR, several panels
            n = int(rng.integers(260, 600)); m = int(rng.integers(n, 1800)); t = int(rng.integers(2, 64)); batch = int(rng.integers(2, 10))
        elif cls == 3:    # many constraints (distributed constraint stage)
            n = int(rng.integers(100, 400)); m = int(rng.integers(n, 900)); t = int(rng.integers(65, n)); batch = int(rng.integers(2, 8))
        else:             # m < n
            n = int(rng.integers(20, 200)); m = int(rng.integers(2, n)); t = int(rng.integers(2, min(n, 30))); batch = int(rng.integers(2, 30))
        t = min(t, n)
        Js, rxs, Ats, cxs, refs, kinds = [], [], [], [], [], []
        for k in range(batch):
            kind = ["full", "full", "defA1", "defA2", "defA3", "defJ", "graded", "zeroA"][int(rng.integers(0, 8))]
            if kind in ("defJ", "graded") and (n - t < 4 or m < n):
                kind = "full"
            pid = 500000 + 1000 * it + k
            if kind == "defJ":
                J, rx, A, cx = synth.make_rank_deficient_J(pid, m, n, t)
            elif kind == "graded":
                J, rx, A, cx = synth.make_graded_J(pid, m, n, t)
            else:
                J, rx, A, cx = synth.make_problem(pid, m, n, t)
            if kind.startswith("defA"):
                for q in range(min(int(kind[-1]), t - 1)):        # duplicate q + 1 rows: rankA = t - (q + 1)
                    A[t - 1 - q, :] = A[q, :]; cx[t - 1 - q] = cx[q]
            if kind == "zeroA":
                A[:] = 0.0; cx[:] = 0.0
            Js.append(np.ascontiguousarray(J.T)); rxs.append(rx); Ats.append(np.ascontiguousarray(A)); cxs.append(cx)
            refs.append(go.gn_subproblem(J, rx, A, cx)); kinds.append(kind)
        p, b, d, infos, jA, jL, jJ = s.solve_batched(np.stack(Js), np.stack(rxs), np.stack(Ats), np.stack(cxs))
        for k, ref in enumerate(refs):
            kind = kinds[k]
            J = Js[k].T; A = Ats[k]; rx = rxs[k]; cx = cxs[k]
            ok = infos[k][0] == ref.rankA and infos[k][2] == ref.code and np.all(np.isfinite(p[k]))
            if kind != "graded":
                ok = ok and infos[k][1] == ref.rankJ2
            if ref.rankA < t and m < n - ref.rankA:       # non-unique basic solution (see fuzz_gpu.py): compare residuals
                rj, ra = np.linalg.norm(J @ p[k] + rx), np.linalg.norm(A @ p[k] + cx)
                rj0, ra0 = np.linalg.norm(J @ ref.p + rx), np.linalg.norm(A @ ref.p + cx)
                ok = ok and abs(rj - rj0) <= 1e-8 * max(1.0, rj0) and abs(ra - ra0) <= 1e-8 * max(1.0, ra0)
            else:
                tol = 1e-5 if kind == "graded" else (1e-10 if kind == "full" else 1e-8)
                ok = ok and rel(p[k], ref.p) <= tol
            solved += 1
            if not ok:
                bad += 1
                print(f"FAIL batch {it} prob {k}/{batch} {kind} m={m} n={n} t={t}: hip {infos[k][:3]} ref ({ref.rankA},{ref.rankJ2},{ref.code}) "
                      f"rel p {rel(p[k], ref.p):.2e}", flush=True)
        if it % 10 == 9:
            print(f"  {it + 1} batches, {solved} problems, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"{nb} batches, {solved} problems, {bad} failures, {time.time() - t0:.1f} s")
    s.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
