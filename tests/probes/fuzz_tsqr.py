"""Randomised row-sharded TSQR solves (the C4 structure rehearsed on one GPU: G local stages + combine) against the oracle's
unsharded solve: p, ranks, pivots, |d|.  usage: python3 tests/probes/fuzz_tsqr.py [cases=30] [seed=1]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.zeros(1, device="cuda:0")          # torch's HIP context first (as the test fixtures do)
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver
from enlsip_gn.tsqr import tsqr_solve_shards


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    s = GNSolver(device=0)
    bad = 0
    t0 = time.time()
    for k in range(cases):
        G = int(rng.integers(2, 9))
        cls = int(rng.integers(0, 3))
        if cls == 0:
            n = int(rng.integers(8, 100)); t = int(rng.integers(0, min(n - 2, 20)))
        elif cls == 1:
            n = int(rng.integers(100, 520)); t = int(rng.integers(0, 30))
        else:
            n = int(rng.integers(520, 1025)); t = int(rng.integers(0, 12))
        m = int(rng.integers(G * (n + 8), G * (n + 8) + 12000))
        J, rx, A, cx = synth.make_problem(int(rng.integers(1, 1 << 30)), m, n, t)
        ref = go.gn_subproblem(J, rx, A, cx)
        res = tsqr_solve_shards(s, J, rx, A, cx, G)
        ok = res.n2 == n - ref.rankA and res.rankA == ref.rankA and res.rankJ2 == ref.rankJ2 and rel(res.p, ref.p) <= 1e-9 \
            and abs(res.d_norm - np.linalg.norm(ref.d)) <= 1e-10 * np.linalg.norm(ref.d) and np.array_equal(res.jpvtJ2, ref.jpvtJ2)
        if not ok:
            bad += 1
            print(f"FAIL m={m} n={n} t={t} G={G}: rel p {rel(res.p, ref.p):.2e} ranks {(res.rankA, res.rankJ2)} vs {(ref.rankA, ref.rankJ2)}", flush=True)
    s.close()
    print(f"{cases} cases, {bad} failures, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
