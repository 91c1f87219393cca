"""Randomised shapes with MORE THAN 512 rows of R0 (kp = min(m, n - rankA) in 513..1024): the pivoted QR stage opens with its
launch-per-step head and hands over to the register blocks (run_qrcp_block, jhead).  Against the oracle.
usage: python3 tests/probes/fuzz_tall_r0.py [cases=24] [seed=1]"""
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
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    s = GNSolver(device=0)
    bad = 0
    t0 = time.time()
    for k in range(cases):
        n = int(rng.integers(530, 1025)); t = int(rng.integers(0, 12)); m = int(rng.integers(max(n - t, 514), 2200))
        kind = ["full", "full", "rankdefJ", "graded"][int(rng.integers(0, 4))]
        gen = {"full": synth.make_problem, "rankdefJ": synth.make_rank_deficient_J, "graded": synth.make_graded_J}[kind]
        try:
            J, rx, A, cx = gen(int(rng.integers(1, 1 << 30)), m, n, t)
        except ValueError:
            continue
        ref = go.gn_subproblem(J, rx, A, cx)
        out = s.solve(J, rx, A, cx)
        r = ref.rankJ2
        ok = (out.rankA, out.rankJ2, out.code) == (ref.rankA, ref.rankJ2, ref.code) and np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r]) \
            and rel(out.p, ref.p) <= (1e-8 if kind == "full" else 1e-4)
        if not ok:
            bad += 1
            print(f"FAIL {kind} m={m} n={n} t={t}: ranks {(out.rankA, out.rankJ2, out.code)} vs {(ref.rankA, ref.rankJ2, ref.code)} rel p {rel(out.p, ref.p):.2e}")
    s.close()
    print(f"{cases} cases, {bad} failures, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
