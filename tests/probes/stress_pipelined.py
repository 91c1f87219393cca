"""Pipelined device-batch entry with degenerate members in both halves (diagnostic): returns, and the regular members still match
the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver, SQRT_EPS
dev = torch.device("cuda", 0)
import os
os.environ.setdefault("ENLSIP_GN_PIPELINE", "1")     # split the small shapes too
s = GNSolver(device=0)
bad = 0
for (B, m, n, t) in [(256, 300, 40, 6), (192, 700, 200, 30), (160, 600, 300, 80), (256, 256, 32, 4)]:
    probs = [synth.make_problem(95000 + k, m, n, t) for k in range(B)]
    for k in (3, B // 2 + 5): probs[k][2][:] = 0.0                    # zero A in both halves
    probs[B // 3][2][0, 0] = np.nan
    probs[B - 2][0][0, 0] = np.nan
    if t >= 2: probs[7][2][1:] = probs[7][2][0]
    J = torch.tensor(np.stack([p[0].T for p in probs]), dtype=torch.float64, device=dev).contiguous()      # (B, n, m)
    rx = torch.tensor(np.stack([p[1] for p in probs]), dtype=torch.float64, device=dev)
    At = torch.tensor(np.stack([p[2] for p in probs]), dtype=torch.float64, device=dev).contiguous()       # (B, t, n)
    cx = torch.tensor(np.stack([p[3] for p in probs]), dtype=torch.float64, device=dev)
    pout = torch.empty((B, n), dtype=torch.float64, device=dev)
    for rep in range(2):
        s.solve_batched_dev(B, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr(), n, n * t, cx.data_ptr(), SQRT_EPS,
                            dp=pout.data_ptr())
    torch.cuda.synchronize()
    P = pout.cpu().numpy()
    for k in (0, 1, B // 2, B // 2 + 1, B - 1):
        ref = go.gn_subproblem(*probs[k])
        e = np.linalg.norm(P[k] - ref.p) / np.linalg.norm(ref.p)
        if not e <= 1e-10:
            bad += 1; print("FAIL", B, m, n, t, k, e)
    print("batch", B, m, n, t, "ok", flush=True)
s.close()
print("failures", bad)
sys.exit(1 if bad else 0)
