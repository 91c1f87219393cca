"""Probe (diagnostic): one solve that takes the launch-per-step pivoted QR (kp > 512), for a kernel trace of k_qd_step."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import synth
from enlsip_gn import GNSolver
s = GNSolver(device=0)
J, rx, A, cx = synth.make_problem(77, 2048, 1024, 0)
for rep in range(4):
    t0 = time.perf_counter(); out = s.solve(J, rx, A, cx); t1 = time.perf_counter()
    print(f"solve 2048x1024: {(t1 - t0) * 1e3:.2f} ms rankJ2 {out.rankJ2}", flush=True)
s.close()
