"""Probe (not a pytest file): plain solve of a stack of G QR-derived upper triangles against the oracle; args: n2 G [extra_rows]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))
from oracle import gn_oracle as go
from enlsip_gn import GNSolver
n2 = int(sys.argv[1]); G = int(sys.argv[2]); extra = int(sys.argv[3]) if len(sys.argv) > 3 else 92
rng = np.random.default_rng(5)
J = np.vstack([np.linalg.qr(rng.standard_normal((n2 + extra, n2)), mode="r") for _ in range(G)])
rx = rng.standard_normal(G * n2)
A = np.zeros((0, n2)); cx = np.zeros(0)
ref = go.gn_subproblem(J, rx, A, cx)
s = GNSolver(device=0)
out = s.solve(J, rx, A, cx)
r = ref.rankJ2
dl = np.abs(np.abs(out.d[:r]) - np.abs(ref.d[:r]))
print(f"stack n2={n2} G={G}: rel p {np.linalg.norm(out.p - ref.p) / np.linalg.norm(ref.p):.2e} pivots {bool(np.array_equal(out.jpvtJ2[:r], ref.jpvtJ2[:r]))} max|d| diff {dl.max():.2e} at {int(dl.argmax())}; cond(J) {np.linalg.cond(J):.1f}")
s.close()
