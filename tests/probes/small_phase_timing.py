"""Phase timing of the wave-per-problem pivot stage (diagnostic, not a test).  Needs the timing-only build
`bash enlsip.jl_amd/build.sh -DENLSIP_GN_LAB -DGN_PS_STAMPS`, which makes k_pivot_small overwrite the last entries of d_out of problem 0
with 100 MHz wall-clock stamps.  usage: python3 tests/probes/small_phase_timing.py m n t batch"""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "enlsip.jl_amd", "python")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
from enlsip_gn import GNSolver, SQRT_EPS
m, n, t, B = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda", 0)
J = torch.randn((B, n, m), dtype=torch.float64, device=dev); rx = torch.randn((B, m), dtype=torch.float64, device=dev)
At = torch.randn((B, t, n), dtype=torch.float64, device=dev); cx = torch.randn((B, t), dtype=torch.float64, device=dev)
p = torch.empty((B, n), dtype=torch.float64, device=dev); d = torch.empty((B, m), dtype=torch.float64, device=dev)
s = GNSolver(device=0)
for rep in range(3):
    s.solve_batched_dev(B, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr(), n, n * t, cx.data_ptr(), SQRT_EPS, dp=p.data_ptr(), dd=d.data_ptr())
    torch.cuda.synchronize()
    st = d[0, m - 8:m - 1].cpu().numpy()
    print("phases (us): extract %.1f qrcp %.1f store+rank %.1f backsub %.1f applyQ1 %.1f outputs %.1f total %.1f" % tuple(list(np.diff(st) / 100.0) + [(st[-1] - st[0]) / 100.0]))
