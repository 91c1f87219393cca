"""GPU bring-up diagnostics (not a pytest file): run a ladder of shapes through the C ABI and
print, per output, how far it is from the CPU oracle.  Usage on the GPU box:
    python tests/probes/gpu_diag.py [quick|full] > gpurun_out/diag.log
"""
from __future__ import annotations

import sys
import time
import traceback
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))
sys.path.insert(0, str(ROOT))

from enlsip_gn import GNSolver, FACTOR_A, FACTOR_L11, FACTOR_J2, FLAG_UPDATE_REFLECTORS  # noqa: E402
from oracle import gn_oracle as go, synth  # noqa: E402


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / (nb if nb > 0 else 1.0))


def check(solver, name, J, rx, A, cx, accessors=False):
    m, n = J.shape
    t = A.shape[0]
    ref = go.gn_subproblem(J, rx, A, cx)
    t0 = time.time()
    out = solver.solve(J, rx, A, cx)
    dt = time.time() - t0
    n2 = n - ref.rankA
    kp = min(m, n2)
    line = (f"{name:28s} m={m:5d} n={n:4d} t={t:3d} | rankA {out.rankA}/{ref.rankA} rankJ2 {out.rankJ2}/{ref.rankJ2} "
            f"code {out.code}/{ref.code} st {out.status} | p {rel(out.p, ref.p):.2e} b {rel(out.b, ref.b):.2e} "
            f"|d|lead {rel(np.abs(out.d[:kp]), np.abs(ref.d[:kp])):.2e} dnorm {abs(np.linalg.norm(out.d) - np.linalg.norm(ref.d)) / max(np.linalg.norm(ref.d), 1e-300):.2e} "
            f"pA {np.array_equal(out.jpvtA, ref.jpvtA)} pL {np.array_equal(out.jpvtL, ref.jpvtL)} "
            f"pJ {np.array_equal(out.jpvtJ2, ref.jpvtJ2)} | {dt * 1e3:.1f} ms")
    print(line, flush=True)
    if accessors:
        for which, F, nm in ((FACTOR_A, ref.F_A, "A"), (FACTOR_L11, ref.F_L11, "L11"), (FACTOR_J2, ref.F_J2, "J2")):
            fv = solver.factor(which)
            R = fv.R
            Rr = F.R
            if R.shape != Rr.shape:
                print(f"    {nm}: R shape {R.shape} vs {Rr.shape}")
                continue
            e_abs = rel(np.abs(R), np.abs(Rr)) if R.size else 0.0
            rows = F.rows
            v = np.random.default_rng(3).standard_normal(rows)
            qt = fv.Qt_mul(v)
            qtr = F.Qt_mul(v)
            back = fv.Q_mul(qt)
            lead = min(F.k, len(v)) if nm != "J2" else kp
            print(f"    {nm}: |R| {e_abs:.2e} jpvt {np.array_equal(fv.p, F.p)} |Qt v|lead {rel(np.abs(qt[:lead]), np.abs(qtr[:lead])):.2e} "
                  f"norm {abs(np.linalg.norm(qt) - np.linalg.norm(v)):.2e} Q(Qt v)-v {rel(back, v):.2e}", flush=True)
        JQ1 = solver.JQ1(m, n)
        print(f"    JQ1 vs oracle {rel(JQ1, ref.F_A.rmul_Q(J)):.2e}", flush=True)
    return out, ref


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    cases = [
        ("tiny_t0", synth.make_problem, (40, 8, 0)),
        ("tiny", synth.make_problem, (64, 16, 4)),
        ("c5", synth.make_problem, (256, 32, 4)),
        ("c3", synth.make_problem, (512, 64, 8)),
        ("t_eq_n", synth.make_problem, (100, 20, 20)),
        ("wide_m_lt_n2", synth.make_problem, (10, 30, 4)),
        ("odd", synth.make_problem, (777, 45, 7)),
        ("rankdefA", synth.make_rank_deficient_A, (300, 40, 6)),
        ("rankdefJ", synth.make_rank_deficient_J, (300, 40, 6)),
        ("graded", synth.make_graded_J, (300, 40, 5)),
        ("two_level", synth.make_problem, (1500, 70, 5)),
        ("mid", synth.make_problem, (2048, 128, 16)),
    ]
    if mode == "c2":
        s = GNSolver(device=0)
        J, rx, A, cx = synth.make_problem(900, 4096, 512, 64)
        check(s, "c2", J, rx, A, cx)
        s.set_profiling(True)
        for rep in range(3):
            s.solve(J, rx, A, cx)
            print("    stage ms", {k: round(v, 3) for k, v in s.stage_ms().items()}, "update", s.update_stats(), flush=True)
        return 0
    if mode == "full":
        cases += [("c2", synth.make_problem, (4096, 512, 64)), ("three_level", synth.make_problem, (20000, 40, 3))]
    for flags, label in ((0, "mfma"), (FLAG_UPDATE_REFLECTORS, "refl")):
        for tile in (512, 256):
            print(f"===== update={label} tile_rows={tile} =====", flush=True)
            try:
                s = GNSolver(device=0, flags=flags, tile_rows=tile)
            except Exception:
                traceback.print_exc()
                return 1
            for i, (name, gen, (m, n, t)) in enumerate(cases):
                try:
                    J, rx, A, cx = gen(100 + i, m, n, t)
                    check(s, name, J, rx, A, cx, accessors=(flags == 0 and tile == 512))
                except Exception:
                    print(f"{name}: EXCEPTION")
                    traceback.print_exc()
            if mode == "full" and flags == 0:
                s.set_profiling(True)
                J, rx, A, cx = synth.make_problem(900, 4096, 512, 64)
                for rep in range(3):
                    s.solve(J, rx, A, cx)
                    print("    stage ms", {k: round(v, 3) for k, v in s.stage_ms().items()}, "update", s.update_stats(), flush=True)
                s.set_profiling(False)
            s.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
