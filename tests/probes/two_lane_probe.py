"""Probe (GPU box): two free-running lanes of half batches, each a one-stream solve on its own handle, against the joined
two-stream step of bench.py.  Does step k+1's first half hide the pivoted-QR stage of step k's second half?"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "enlsip.jl_amd", "python"))
import torch
torch.zeros(1, device="cuda:0")
from enlsip_gn import GNSolver, SQRT_EPS
from enlsip_gn import workload as wl

m, n, t = 4096, 512, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lag_ms = float(sys.argv[4]) if len(sys.argv) > 4 else -1.0
dev = torch.device("cuda:0")
J, rx, At, cx = wl.make_batch(0, B, m, n, t, dev)
p = torch.empty((B, n), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
solvers = [GNSolver(device=0) for _ in range(L)]
bounds = [(B * i) // L for i in range(L + 1)]

def part(i):
    lo, hi = bounds[i], bounds[i + 1]
    solvers[i].solve_batched_dev(hi - lo, m, n, t, J[lo].data_ptr(), m, m * n, rx[lo].data_ptr(), At[lo].data_ptr(), n, n * t,
                                 cx[lo].data_ptr(), SQRT_EPS, dp=p[lo].data_ptr())

for i in range(L):
    part(i); part(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); part(0); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e3

def lane(i, delay):
    if delay > 0:
        time.sleep(delay * 1e-3)
    for _ in range(K):
        part(i)

lag = lag_ms if lag_ms >= 0 else one / L
ths = [threading.Thread(target=lane, args=(i, lag * i)) for i in range(L)]
t0 = time.perf_counter()
for th in ths: th.start()
for th in ths: th.join()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"lanes {L} batch {B} K {K} lag_ms {lag:.2f} one_lane_alone_ms {one:.2f} | ms/step {el / K * 1e3:.3f} solves/s {B * K / el:.1f}", flush=True)
