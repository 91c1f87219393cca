"""Probe (diagnostic): batches around the 65535 grid-dimension limit."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
from enlsip_gn import GNSolver, GNError
torch.cuda.init()
dev = torch.device("cuda", 0)
s = GNSolver(device=0)
for (B, m, n, t) in [(65535, 16, 4, 1), (65536, 16, 4, 1), (100000, 16, 4, 1), (70000, 256, 32, 4)]:
    g = torch.Generator(device=dev); g.manual_seed(B)
    J = torch.randn((B, n, m), dtype=torch.float64, device=dev, generator=g)
    rx = torch.randn((B, m), dtype=torch.float64, device=dev, generator=g)
    At = torch.randn((B, t, n), dtype=torch.float64, device=dev, generator=g)
    cx = torch.randn((B, t), dtype=torch.float64, device=dev, generator=g)
    p = torch.zeros((B, n), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    try:
        s.solve_batched_dev(B, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr(), n, n * t, cx.data_ptr(), dp=p.data_ptr())
        torch.cuda.synchronize()
        res = (torch.einsum("btn,bn->bt", At, p) + cx).abs().max().item()
        print(f"batch {B} ({m}x{n}, t={t}): ok, worst constraint residual {res:.2e}, finite {bool(torch.isfinite(p).all())}", flush=True)
    except GNError as e:
        print(f"batch {B}: GNError: {e}", flush=True)
s.close()
