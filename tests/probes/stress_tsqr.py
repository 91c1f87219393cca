"""Row-sharded TSQR building blocks with degenerate inputs (diagnostic): must return."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
from oracle import synth
from enlsip_gn import GNSolver
from enlsip_gn.tsqr import tsqr_solve_shards
s = GNSolver(device=0)
for (m, n, t, G) in [(3000, 48, 5, 3), (4000, 64, 0, 4), (6000, 300, 10, 3), (5000, 600, 4, 2), (2000, 200, 100, 2)]:
    for kind in ("ok", "zeroA", "nanA", "nanJ", "zeroJ"):
        J, rx, A, cx = synth.make_problem(97000 + m, m, n, t)
        if kind == "zeroA" and t: A[:] = 0.0
        if kind == "nanA" and t: A[0, 0] = np.nan
        if kind == "nanJ": J[5, 3] = np.nan
        if kind == "zeroJ": J[:] = 0.0
        res = tsqr_solve_shards(s, J, rx, A, cx, G)
        print(m, n, t, G, kind, "rankA", res.rankA, "rankJ2", res.rankJ2, flush=True)
s.close(); print("done")
