"""Probe (not a pytest file): unpivoted R0 / z of a stack of G upper triangles, pairs on vs off (two child processes)."""
import os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))

def child(tag, n2, G, mode):
    import torch
    torch.zeros(1, device="cuda:0")
    from enlsip_gn import GNSolver, SQRT_EPS
    from enlsip_gn.tsqr import hip_local_stage
    rng = np.random.default_rng(5)
    blocks = []
    for g in range(G):
        if mode == "tri":
            blocks.append(np.triu(rng.standard_normal((n2, n2)) + 30 * np.eye(n2)))
        elif mode == "qr":
            blocks.append(np.linalg.qr(rng.standard_normal((n2 + 92, n2)), mode="r"))
        else:
            blocks.append(rng.standard_normal((n2, n2)))
    J = np.vstack(blocks)
    rx = rng.standard_normal(G * n2)
    m = G * n2
    Jd = torch.tensor(np.ascontiguousarray(J.T), dtype=torch.float64, device="cuda:0")
    rxd = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
    R = torch.empty((n2 * n2,), dtype=torch.float64, device="cuda:0")
    z = torch.empty((n2,), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    s = GNSolver(device=0)
    n2o, tail = hip_local_stage(s, m, n2, 0, Jd.data_ptr(), m, rxd.data_ptr(), 0, 0, R.data_ptr(), z.data_ptr(), SQRT_EPS)
    np.savez(f"/tmp/r0_{tag}.npz", R=R.cpu().numpy().reshape(n2, n2).T, z=z.cpu().numpy(), tail=tail)
    s.close()

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    sys.exit(0)
n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 1008
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "tri"
for tag, env in (("pair", {}), ("plain", {"ENLSIP_GN_PAIR": "0"})):
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, __file__, "child", tag, str(n2), str(G), mode], env=e)
a, b = np.load("/tmp/r0_pair.npz"), np.load("/tmp/r0_plain.npz")
Ra, Rb = np.abs(a["R"]), np.abs(b["R"])
dz = np.abs(np.abs(a["z"]) - np.abs(b["z"]))
rowerr = np.abs(Ra - Rb).max(axis=1)
bad = np.nonzero(rowerr > 1e-11)[0]
badz = np.nonzero(dz > 1e-11)[0]
print(f"n2={n2} G={G} {mode}: max |R| row diff {rowerr.max():.2e}; first bad R row {bad[:5]}; count {bad.size}; |z| diff max {dz.max():.2e} first bad {badz[:8]} count {badz.size}; tails {float(a['tail']):.6e} {float(b['tail']):.6e}")
import scipy.linalg as sl
pa = sl.solve_triangular(a["R"], a["z"]); pb = sl.solve_triangular(b["R"], b["z"])
print(f"  host solve R0 p = z: rel diff {np.linalg.norm(pa - pb) / np.linalg.norm(pb):.2e}; relative row diffs max {(np.abs(Ra - Rb).max(axis=1) / np.abs(Rb).max(axis=1)).max():.2e} at row {int((np.abs(Ra - Rb).max(axis=1) / np.abs(Rb).max(axis=1)).argmax())}; rel |z| diff max {(dz / np.abs(b['z'])).max():.2e} at {int((dz / np.abs(b['z'])).argmax())}")
for r in bad[:3]:
    cols = np.nonzero(np.abs(Ra[r] - Rb[r]) > 1e-11)[0]
    print("  row", r, "bad cols", cols[:10], "...", cols[-3:], "n", cols.size)
