"""Single-solve timing over unusual shapes beside the LAPACK port (diagnostic: looks for performance cliffs).
usage: python3 tests/probes/shape_timing.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver

SHAPES = [(4096, 512, 64), (2000, 1000, 0), (2000, 1000, 500), (5000, 700, 300), (300, 1000, 100), (1024, 1024, 1024), (4096, 512, 200),
          (20000, 1000, 0), (8192, 256, 128), (600, 600, 599), (3000, 64, 63), (100000, 32, 4)]
s = GNSolver(device=0)
probs, hip_ms, outs = [], [], []
# all GPU timings first: the BLAS threads of the LAPACK port keep spinning after a call and slow the launch-heavy solves down
for m, n, t in SHAPES:
    J, rx, A, cx = synth.make_problem(70000 + m + n + t, m, n, t)
    J = np.asfortranarray(J)            # column-major like a Julia Matrix: the wrapper then passes the buffer as it is
    out = s.solve(J, rx, A, cx)
    t0 = time.perf_counter()
    for _ in range(3): out = s.solve(J, rx, A, cx)
    hip_ms.append((time.perf_counter() - t0) / 3 * 1e3)
    probs.append((J, rx, A, cx)); outs.append(out)
for (m, n, t), (J, rx, A, cx), tg, out in zip(SHAPES, probs, hip_ms, outs):
    t0 = time.perf_counter(); ref = go.gn_subproblem(J, rx, A, cx); tc = time.perf_counter() - t0
    nb = np.linalg.norm(ref.p)
    print(f"m={m:6d} n={n:5d} t={t:5d}: hip {tg:8.2f} ms (host buffers, PCIe included)   lapack {tc * 1e3:8.1f} ms   rel p {np.linalg.norm(out.p - ref.p) / (nb if nb else 1):.1e}"
          f"  ranks {out.rankA},{out.rankJ2} vs {ref.rankA},{ref.rankJ2}", flush=True)
s.close()
