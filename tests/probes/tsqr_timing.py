"""Timing of the C4 building blocks on ONE GPU (diagnostic, not a test): the local stage of a rank that owns
m/G rows of a tall Jacobian (n = 1024, t = 0) and the combine stage on G stacked triangles."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))
import numpy as np
import torch
from enlsip_gn import GNSolver, SQRT_EPS
from enlsip_gn.tsqr import hip_local_stage, hip_combine_stage

def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    m, n, t = 262144, 1024, 0
    m_loc = m // G
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    J = torch.randn((n, m_loc), dtype=torch.float64, device=dev, generator=gen)
    rx = torch.randn((m_loc,), dtype=torch.float64, device=dev, generator=gen)
    R = torch.zeros((n * n,), dtype=torch.float64, device=dev)
    z = torch.zeros((n,), dtype=torch.float64, device=dev)
    s = GNSolver(device=0)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n2, tail = hip_local_stage(s, m_loc, n, t, J.data_ptr(), m_loc, rx.data_ptr(), 0, 0, R.data_ptr(), z.data_ptr(), SQRT_EPS)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        flops = 2.0 * n * n * (m_loc - n / 3.0)
        print(f"local stage m_loc={m_loc} n={n}: {(t1 - t0) * 1e3:.2f} ms  ({flops / (t1 - t0) * 1e-12:.1f} TFLOP/s of the unpivoted QR)  n2={n2}")
    Rs = R.repeat(G); zs = z.repeat(G)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p, dlead, ct, info, jp = hip_combine_stage(s, G, n, n2, Rs.data_ptr(), zs.data_ptr(), SQRT_EPS)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"combine stage G={G} (stacked {G * n2} x {n2}): {(t1 - t0) * 1e3:.2f} ms  rankJ2={info.rankJ2}")
    s.close()

if __name__ == "__main__":
    main()
