import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from enlsip_gn import GNSolver
rng = np.random.default_rng(0)
s = GNSolver(device=0)
for (m, n, t) in [(2_000_000, 32, 4), (5_000_000, 8, 0), (1_000_000, 100, 10)]:
    J = np.asfortranarray(rng.standard_normal((m, n))); rx = rng.standard_normal(m)
    A = rng.standard_normal((t, n)); cx = rng.standard_normal(t)
    t0 = time.perf_counter(); out = s.solve(J, rx, A, cx); dt = time.perf_counter() - t0
    # normal equations check in the null space of A
    r = J @ out.p + rx
    if t:
        Q, _ = np.linalg.qr(A.T, mode="complete"); Z = Q[:, t:]
        g = Z.T @ (J.T @ r); cons = np.abs(A @ out.p + cx).max()
    else:
        g = J.T @ r; cons = 0.0
    print(m, n, t, f"{dt*1e3:.0f} ms ranks {out.rankA},{out.rankJ2} |Z'J'r| {np.linalg.norm(g):.2e} (|J'rx| {np.linalg.norm(J.T @ rx):.2e}) cons {cons:.1e}", flush=True)
s.close()
