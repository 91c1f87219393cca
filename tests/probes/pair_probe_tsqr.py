"""Probe (not a pytest file): row-sharded TSQR on one GPU against the oracle on the whole matrix."""
import sys
from pathlib import Path
import numpy as np
import torch
torch.zeros(1, device='cuda:0')
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver
from enlsip_gn.tsqr import tsqr_solve_shards

shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (16384, 144, 16, 8), (16384, 128, 16, 8), (16384, 112, 0, 8), (16384, 128, 0, 8), (16384, 304, 16, 8), (16384, 288, 16, 8),
    (8192, 112, 0, 2), (8192, 112, 0, 4), (40000, 250, 10, 5)]
s = GNSolver(device=0)
for (m, n, t, G) in shapes:
    J, rx, A, cx = synth.make_problem(11, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = tsqr_solve_shards(s, J, rx, A, cx, G)
    ep = np.linalg.norm(out.p - ref.p) / np.linalg.norm(ref.p)
    ed = abs(out.d_norm - np.linalg.norm(ref.d)) / np.linalg.norm(ref.d)
    r = ref.rankJ2
    piv = bool(np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r]))
    dl = np.abs(np.abs(out.dlead[:r]) - np.abs(ref.d[:r]))
    print(f"m={m} n={n} t={t} G={G} n2={out.n2}: rel p {ep:.2e}  d-norm {ed:.2e}  pivots {piv}  max|d| diff {dl.max():.2e} at {int(dl.argmax())}", flush=True)
s.close()
