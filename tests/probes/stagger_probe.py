"""Would a STAGGERED pipeline pay?  The library splits a batch into two halves on two streams; both halves start together and stay in
lockstep (both in J*Q1, both in the sweep, both in the pivoted QR).  Here the same 384 C2 problems go through two handles whose
sub-batches are offset by half a sub-solve (A: 96 + 96, B: 48 + 96 + 48, and finer), so that one stream's matrix-pipe kernels meet
the other's latency-bound ones.  Diagnostic probe (GPU): prints solves/s per schedule, alternating, three rounds."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "enlsip.jl_amd", "python"))
from enlsip_gn import GNSolver

m, n, t, B = 4096, 512, 64, 384
SQRT_EPS = float(np.sqrt(np.finfo(float).eps))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
J = torch.randn(B, n, m, device=dev, dtype=torch.float64, generator=g)
rx = torch.randn(B, m, device=dev, dtype=torch.float64, generator=g)
At = torch.randn(B, t, n, device=dev, dtype=torch.float64, generator=g)
cx = torch.randn(B, t, device=dev, dtype=torch.float64, generator=g)
p = torch.zeros(B, n, device=dev, dtype=torch.float64)
torch.cuda.synchronize()


def solve(s, lo, hi):
    s.solve_batched_dev(hi - lo, m, n, t, J[lo].data_ptr(), m, m * n, rx[lo].data_ptr(), At[lo].data_ptr(), n, n * t,
                        cx[lo].data_ptr(), SQRT_EPS, dp=p[lo].data_ptr())


def make(pipeline):
    if pipeline: os.environ.pop("ENLSIP_GN_PIPELINE", None)
    else: os.environ["ENLSIP_GN_PIPELINE"] = "0"
    return GNSolver(device=0)

lib = make(True)                 # the library's own two halves
a, b = make(False), make(False)  # one stream each
pool = ThreadPoolExecutor(max_workers=2)


def cuts(sizes, lo):
    out = []
    for s in sizes:
        out.append((lo, lo + s)); lo += s
    return out


def schedule(sa, sb):
    ca, cb = cuts(sa, 0), cuts(sb, sum(sa))
    assert ca[-1][1] == sum(sa) and cb[-1][1] == B
    def run():
        fa = pool.submit(lambda: [solve(a, lo, hi) for lo, hi in ca])
        fb = pool.submit(lambda: [solve(b, lo, hi) for lo, hi in cb])
        fa.result(); fb.result()
    return run

SCHEDULES = {
    "library (192 | 192, lockstep)": lambda: solve(lib, 0, B),
    "two handles 192 | 192": schedule([192], [192]),
    "A 96+96 | B 48+96+48": schedule([96, 96], [48, 96, 48]),
    "A 64+64+64 | B 32+64+64+32": schedule([64, 64, 64], [32, 64, 64, 32]),
    "A 128+64 | B 64+128": schedule([128, 64], [64, 128]),
    "A 192 | B 96+96": schedule([192], [96, 96]),
}
for f in SCHEDULES.values():
    f(); f()
torch.cuda.synchronize()
for rnd in range(3):
    for name, f in SCHEDULES.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        print(f"round {rnd}  {name:34s} {dt * 1e3:7.2f} ms/step  {B / dt:8.0f} solves/s", flush=True)
