"""Leading-dimension / stride padding on the device entry (diagnostic): ldj > m (odd and even), ldat > n, strides with gaps."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver, SQRT_EPS
dev = torch.device("cuda", 0)
s = GNSolver(device=0)
bad = 0
for (B, m, n, t, pj, pa) in [(3, 4096, 512, 64, 0, 0), (3, 4096, 512, 64, 7, 3), (3, 4096, 512, 64, 8, 0), (4, 256, 32, 4, 5, 1),
                             (2, 700, 300, 100, 9, 2), (5, 600, 40, 6, 1, 1), (2, 1300, 600, 8, 3, 0), (130, 300, 40, 6, 3, 1)]:
    ldj, ldat = m + pj, n + pa
    sJ, sAt = ldj * n + 11, ldat * max(t, 1) + 5
    probs = [synth.make_problem(98000 + k, m, n, t) for k in range(B)]
    Jbuf = torch.full((B * sJ,), float("nan"), dtype=torch.float64, device=dev)       # NaN in the gaps: must never be read as data
    Abuf = torch.full((B * sAt,), float("nan"), dtype=torch.float64, device=dev)
    for k, (J, rx, A, cx) in enumerate(probs):
        Jp = np.full((n, ldj), np.nan); Jp[:, :m] = J.T
        Jbuf[k * sJ: k * sJ + ldj * n] = torch.tensor(Jp.ravel(), device=dev)
        Ap = np.full((t, ldat), np.nan); Ap[:, :n] = A
        Abuf[k * sAt: k * sAt + ldat * t] = torch.tensor(Ap.ravel(), device=dev)
    rx = torch.tensor(np.stack([p[1] for p in probs]), dtype=torch.float64, device=dev)
    cx = torch.tensor(np.stack([p[3] for p in probs]), dtype=torch.float64, device=dev)
    pout = torch.empty((B, n), dtype=torch.float64, device=dev)
    s.solve_batched_dev(B, m, n, t, Jbuf.data_ptr(), ldj, sJ, rx.data_ptr(), Abuf.data_ptr(), ldat, sAt, cx.data_ptr(), SQRT_EPS, dp=pout.data_ptr())
    torch.cuda.synchronize()
    P = pout.cpu().numpy()
    worst = 0.0
    for k in (0, B - 1):
        ref = go.gn_subproblem(*probs[k])
        worst = max(worst, np.linalg.norm(P[k] - ref.p) / np.linalg.norm(ref.p))
    ok = worst <= 1e-10
    bad += (not ok)
    print(B, m, n, t, "ldj", ldj, "ldat", ldat, "rel", f"{worst:.1e}", "ok" if ok else "FAIL", flush=True)
s.close()
print("failures", bad); sys.exit(1 if bad else 0)
