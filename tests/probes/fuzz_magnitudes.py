"""Randomised shapes x randomised power-of-two scales of (J, rx) and (A, cx), far outside the range of plain sums of squares,
against the oracle (real LAPACK) on the SAME scaled inputs: ranks, code, pivots, p, b, |d| — the rescale path of the library
(gn_rescale.hpp) on every size class, single problems and small batches.  usage: python3 tests/probes/fuzz_magnitudes.py [cases=80] [seed=1]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def pick_exp(rng):
    k = int(rng.integers(0, 5))
    if k == 0:
        return 0
    e = int(rng.integers(450, 960))
    return e if k in (1, 2) else -e


def check(out_p, out_b, out_d, rankA, rankJ2, code, jpvtJ2, ref, eJ, eA, kind):
    ok = (rankA, rankJ2, code) == (ref.rankA, ref.rankJ2, ref.code) and bool(np.all(np.isfinite(out_p)))
    tol = 1e-9 if kind == "full" else 1e-7
    ok = ok and rel(out_p, ref.p) <= tol
    if ref.b.size:
        ok = ok and rel(np.ldexp(out_b, -eA), np.ldexp(ref.b, -eA)) <= 1e-10
    nd = np.linalg.norm(np.ldexp(ref.d, -eJ))
    ok = ok and abs(np.linalg.norm(np.ldexp(out_d, -eJ)) - nd) <= 1e-11 * max(nd, 1e-300)
    r = ref.rankJ2
    if kind == "full":
        ok = ok and np.array_equal(np.asarray(jpvtJ2)[:r], ref.jpvtJ2[:r])
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    s = GNSolver(device=0)
    bad = 0
    rescaled = 0
    t0 = time.time()
    for k in range(cases):
        cls = int(rng.integers(0, 5))
        if cls == 0:
            n = int(rng.integers(2, 33)); m = int(rng.integers(n, 300)); t = int(rng.integers(0, min(n, 16) + 1))
        elif cls == 1:
            n = int(rng.integers(8, 65)); m = int(rng.integers(n // 2 + 1, 700)); t = int(rng.integers(0, min(n, 63) + 1))
        elif cls == 2:
            n = int(rng.integers(65, 300)); m = int(rng.integers(n, 1800)); t = int(rng.integers(0, min(n, 80)))
        elif cls == 3:
            n = int(rng.integers(300, 640)); m = int(rng.integers(200, 1500)); t = int(rng.integers(0, 70)) if rng.integers(0, 3) else int(rng.integers(65, 300))
        else:
            n = int(rng.integers(20, 160)); m = int(rng.integers(1, n)); t = int(rng.integers(0, min(n, 40)))
        kind = ["full", "full", "rankdefA"][int(rng.integers(0, 3))]
        if kind == "rankdefA" and (t < 2 or m < n):
            kind = "full"
        batch = 1 if rng.integers(0, 3) else int(rng.integers(2, 7))
        gen = synth.make_rank_deficient_A if kind == "rankdefA" else synth.make_problem
        probs, refs, exps = [], [], []
        for q in range(batch):
            J, rx, A, cx = gen(int(rng.integers(1, 1 << 30)), m, n, t)
            eJ, eA = pick_exp(rng), (pick_exp(rng) if t else 0)
            J, rx, A, cx = np.ldexp(J, eJ), np.ldexp(rx, eJ), np.ldexp(A, eA), np.ldexp(cx, eA)
            probs.append((J, rx, A, cx)); exps.append((eJ, eA)); refs.append(go.gn_subproblem(J, rx, A, cx))
        ok = True
        if batch == 1:
            out = s.solve(*probs[0])
            ok = check(out.p, out.b, out.d, out.rankA, out.rankJ2, out.code, out.jpvtJ2, refs[0], exps[0][0], exps[0][1], kind) and out.status == 0
        else:
            p, b, d, infos, jA, jL, jJ = s.solve_batched(np.stack([np.ascontiguousarray(P[0].T) for P in probs]), np.stack([P[1] for P in probs]),
                                                         np.stack([np.ascontiguousarray(P[2]) for P in probs]) if t else None,
                                                         np.stack([P[3] for P in probs]) if t else None)
            for q in range(batch):
                n2 = n - infos[q][0]
                ok = ok and check(p[q], b[q], d[q], infos[q][0], infos[q][1], infos[q][2], jJ[q][:n2], refs[q], exps[q][0], exps[q][1], kind)
        rescaled += int("rescaled" in s.route())
        if not ok:
            bad += 1
            print(f"FAIL {kind} batch={batch} m={m} n={n} t={t} exps={exps}", flush=True)
    s.close()
    print(f"{cases} cases ({rescaled} through the rescale path), {bad} failures, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
