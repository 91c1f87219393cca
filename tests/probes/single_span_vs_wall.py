"""Single C2 solve (batch 1) on a caller's stream: GPU-side span between two events around the call against the wall clock of the
call, and where the host is when — the share of a solve that is not kernels.  Diagnostic probe (GPU)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "enlsip.jl_amd", "python"))
from enlsip_gn import GNSolver
m, n, t = 4096, 512, 64
EPS = float(np.sqrt(np.finfo(float).eps))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(9)
J = torch.randn(1, n, m, device=dev, dtype=torch.float64, generator=g)
rx = torch.randn(1, m, device=dev, dtype=torch.float64, generator=g)
At = torch.randn(1, t, n, device=dev, dtype=torch.float64, generator=g)
cx = torch.randn(1, t, device=dev, dtype=torch.float64, generator=g)
p = torch.zeros(1, n, device=dev, dtype=torch.float64)
st = torch.cuda.Stream()
s = GNSolver(device=0, stream=st.cuda_stream)
def one():
    s.solve_batched_dev(1, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr(), n, n * t, cx.data_ptr(), EPS, dp=p.data_ptr())
for _ in range(5): one()
torch.cuda.synchronize()
spans, walls = [], []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    t0 = time.perf_counter()
    one()
    w = time.perf_counter() - t0
    e1.record(st)
    torch.cuda.synchronize()
    spans.append(e0.elapsed_time(e1)); walls.append(w * 1e3)
print(f"GPU span between events, median {np.median(spans):.3f} ms (min {min(spans):.3f}); wall clock of the call, median {np.median(walls):.3f} ms (min {min(walls):.3f})")
