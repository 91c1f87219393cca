"""NaN / Inf / zero inputs must come back (with whatever IEEE arithmetic makes of them), never hang (diagnostic)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import synth
from enlsip_gn import GNSolver
s = GNSolver(device=0)
for (m, n, t) in [(256, 32, 4), (512, 64, 8), (900, 300, 20), (4096, 512, 64), (700, 300, 100), (1300, 600, 8)]:
    for kind in ("nanJ", "infJ", "zeroJ", "nanA", "zeroA", "allzero", "hugeJ", "tinyJ", "hugeA"):
        J, rx, A, cx = synth.make_problem(5, m, n, t)
        if kind == "nanJ": J[m // 2, n // 3] = np.nan
        if kind == "infJ": J[1, 1] = np.inf
        if kind == "zeroJ": J[:] = 0.0
        if kind == "nanA": A[0, 0] = np.nan
        if kind == "zeroA": A[:] = 0.0
        if kind == "hugeJ": J *= 1e200; rx *= 1e200
        if kind == "tinyJ": J *= 1e-200; rx *= 1e-200
        if kind == "hugeA": A *= 1e200; cx *= 1e200
        if kind == "allzero": J[:] = 0.0; A[:] = 0.0; rx[:] = 0.0; cx[:] = 0.0
        t0 = time.perf_counter()
        try:
            out = s.solve(J, rx, A, cx)
            msg = f"ranks ({out.rankA},{out.rankJ2}) code {out.code} status {out.status} finite p: {bool(np.all(np.isfinite(out.p)))}"
        except Exception as e:
            msg = f"error {type(e).__name__}: {str(e)[:80]}"
        print(f"{m}x{n} t={t} {kind:8s}: {(time.perf_counter() - t0) * 1e3:7.1f} ms  {msg}", flush=True)
s.close()
print("done")
