"""Probe of the panel-pair path (not a pytest file): single solves of multi-level-tree shapes against the oracle.
python tests/probes/pair_probe.py  -> one line per shape: rel. error of p, | ||d|| - ||d_ref|| |, pivots equal"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))
from oracle import gn_oracle as go, synth
from enlsip_gn import GNSolver

shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (20000, 128, 0), (20000, 112, 0), (20000, 128, 16), (20000, 144, 16), (9000, 112, 0), (8192, 112, 0), (4096, 112, 0),
    (20000, 80, 0), (20000, 96, 16), (40000, 240, 0), (40000, 250, 10)]
s = GNSolver(device=0)
for (m, n, t) in shapes:
    J, rx, A, cx = synth.make_problem(7, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    out = s.solve(J, rx, A, cx)
    ep = np.linalg.norm(out.p - ref.p) / np.linalg.norm(ref.p)
    ed = abs(np.linalg.norm(out.d) - np.linalg.norm(ref.d)) / np.linalg.norm(ref.d)
    r = ref.rankJ2
    piv = bool(np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r]))
    dl = np.abs(np.abs(out.d[:r]) - np.abs(ref.d[:r]))
    print(f"m={m} n={n} t={t} n2={n - ref.rankA}: rel p {ep:.2e}  d-norm {ed:.2e}  pivots {piv}  max|d| diff {dl.max():.2e} at {int(dl.argmax())}", flush=True)
s.close()
