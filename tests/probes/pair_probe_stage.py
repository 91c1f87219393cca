"""Probe (not a pytest file): which TSQR stage deviates with panel pairs on?  m n t G"""
import os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))

def child(tag, m, n, t, G):
    import torch
    torch.zeros(1, device="cuda:0")
    from oracle import synth
    from enlsip_gn import GNSolver, SQRT_EPS
    from enlsip_gn.tsqr import hip_local_stage, hip_combine_stage, row_range
    J, rx, A, cx = synth.make_problem(11, m, n, t)
    dev = "cuda:0"
    Jd = torch.tensor(np.ascontiguousarray(J.T), dtype=torch.float64, device=dev)
    rxd = torch.tensor(rx, dtype=torch.float64, device=dev)
    Atd = torch.tensor(np.ascontiguousarray(A), dtype=torch.float64, device=dev) if t else None
    cxd = torch.tensor(cx, dtype=torch.float64, device=dev) if t else None
    s = GNSolver(device=0)
    Rs, zs, tails = [], [], []
    for g in range(G):
        lo, hi = row_range(m, G, g)
        Jl = Jd[:, lo:hi].contiguous(); rl = rxd[lo:hi].contiguous()
        R = torch.empty((n * n,), dtype=torch.float64, device=dev); z = torch.empty((n,), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        n2, tail = hip_local_stage(s, hi - lo, n, t, Jl.data_ptr(), hi - lo, rl.data_ptr(), Atd.data_ptr() if t else 0,
                                   cxd.data_ptr() if t else 0, R.data_ptr(), z.data_ptr(), SQRT_EPS)
        Rs.append(R[: n2 * n2].clone()); zs.append(z[:n2].clone()); tails.append(tail)
    np.savez(f"/tmp/st_{tag}.npz", R=torch.stack(Rs).cpu().numpy(), z=torch.stack(zs).cpu().numpy(), tails=np.array(tails), n2=n2)
    # combine on a stack that is the same for both children: the plain child's when it exists
    src = np.load("/tmp/st_plain.npz") if os.path.exists("/tmp/st_plain.npz") else np.load(f"/tmp/st_{tag}.npz")
    Rst = torch.tensor(src["R"].reshape(-1), dtype=torch.float64, device=dev); zst = torch.tensor(src["z"].reshape(-1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    p, dlead, ctail, info, jp = hip_combine_stage(s, G, n, n2, Rst.data_ptr(), zst.data_ptr(), SQRT_EPS)
    np.savez(f"/tmp/cb_{tag}.npz", p=p, dlead=dlead, ctail=ctail, jp=jp)
    s.close()

if sys.argv[1] == "child":
    child(sys.argv[2], *[int(x) for x in sys.argv[3:7]])
    sys.exit(0)
m, n, t, G = [int(x) for x in sys.argv[1:5]]
for f in ("/tmp/st_plain.npz", "/tmp/st_pair.npz"):
    if os.path.exists(f): os.remove(f)
for tag, env in (("plain", {"ENLSIP_GN_PAIR": "0"}), ("pair", {})):
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, __file__, "child", tag, str(m), str(n), str(t), str(G)], env=e)
a, b = np.load("/tmp/st_pair.npz"), np.load("/tmp/st_plain.npz")
n2 = int(a["n2"])
for g in range(G):
    Ra, Rb = np.abs(a["R"][g].reshape(n2, n2).T), np.abs(b["R"][g].reshape(n2, n2).T)
    rowerr = np.abs(Ra - Rb).max(axis=1)
    dz = np.abs(np.abs(a["z"][g]) - np.abs(b["z"][g]))
    print(f"shard {g}: |R| max row diff {rowerr.max():.2e} (row {int(rowerr.argmax())})  |z| max diff {dz.max():.2e} (at {int(dz.argmax())})  tail {a['tails'][g]:.9e} vs {b['tails'][g]:.9e}")
ca, cb = np.load("/tmp/cb_pair.npz"), np.load("/tmp/cb_plain.npz")
print(f"combine on the SAME stack: rel p diff {np.linalg.norm(ca['p'] - cb['p']) / np.linalg.norm(cb['p']):.2e}  |dlead| diff {np.abs(np.abs(ca['dlead']) - np.abs(cb['dlead'])).max():.2e}  pivots equal {bool(np.array_equal(ca['jp'], cb['jp']))} ctail {float(ca['ctail']):.9e} {float(cb['ctail']):.9e}")
# third child: the plain child's stack through an ordinary solve on a FRESH handle, pairs on
code = r'''
import sys, numpy as np
sys.path.insert(0, "%s"); sys.path.insert(0, "%s")
import torch; torch.zeros(1, device="cuda:0")
from enlsip_gn import GNSolver
src = np.load("/tmp/st_plain.npz"); n2 = int(src["n2"]); G = src["R"].shape[0]
J = np.vstack([src["R"][g].reshape(n2, n2).T for g in range(G)])
rx = -src["z"].reshape(-1)
s = GNSolver(device=0)
out = s.solve(J, rx, np.zeros((0, n2)), np.zeros(0))
np.savez("/tmp/cb_fresh.npz", p=out.p)
s.close()
''' % (str(ROOT), str(ROOT / "enlsip.jl_amd" / "python"))
subprocess.check_call([sys.executable, "-c", code])
cf = np.load("/tmp/cb_fresh.npz")
print(f"fresh handle, pairs on, same stack: rel p diff to the plain combine {np.linalg.norm(cf['p'] - cb['p']) / np.linalg.norm(cb['p']):.2e}; to the pair combine {np.linalg.norm(cf['p'] - ca['p']) / np.linalg.norm(cb['p']):.2e}")
# fourth: unpivoted R0 / z of that same stack, pairs on vs off
code2 = r'''
import sys, os, numpy as np
sys.path.insert(0, "%s"); sys.path.insert(0, "%s")
import torch; torch.zeros(1, device="cuda:0")
from enlsip_gn import GNSolver, SQRT_EPS
from enlsip_gn.tsqr import hip_local_stage
src = np.load("/tmp/st_plain.npz"); n2 = int(src["n2"]); G = src["R"].shape[0]
J = np.vstack([src["R"][g].reshape(n2, n2).T for g in range(G)])
rx = -src["z"].reshape(-1)
m = G * n2
Jd = torch.tensor(np.ascontiguousarray(J.T), dtype=torch.float64, device="cuda:0"); rxd = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
R = torch.empty((n2 * n2,), dtype=torch.float64, device="cuda:0"); z = torch.empty((n2,), dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
s = GNSolver(device=0)
hip_local_stage(s, m, n2, 0, Jd.data_ptr(), m, rxd.data_ptr(), 0, 0, R.data_ptr(), z.data_ptr(), SQRT_EPS)
np.savez("/tmp/r0x_" + sys.argv[1] + ".npz", R=R.cpu().numpy().reshape(n2, n2).T, z=z.cpu().numpy())
s.close()
''' % (str(ROOT), str(ROOT / "enlsip.jl_amd" / "python"))
for tag, env in (("plain", {"ENLSIP_GN_PAIR": "0"}), ("pair", {})):
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, "-c", code2, tag], env=e)
xa, xb = np.load("/tmp/r0x_pair.npz"), np.load("/tmp/r0x_plain.npz")
Ra, Rb = np.abs(xa["R"]), np.abs(xb["R"])
rowerr = np.abs(Ra - Rb).max(axis=1)
dz = np.abs(np.abs(xa["z"]) - np.abs(xb["z"]))
print(f"R0 of that stack: max |R| row diff {rowerr.max():.2e} at row {int(rowerr.argmax())}; rows with diff > 1e-11: {np.nonzero(rowerr > 1e-11)[0][:12]}; |z| diff max {dz.max():.2e} at {int(dz.argmax())}; z entries > 1e-11: {np.nonzero(dz > 1e-11)[0][:16]} count {(dz > 1e-11).sum()}")
r = int(rowerr.argmax())
cols = np.nonzero(np.abs(Ra[r] - Rb[r]) > 1e-11)[0]
print("  worst row", r, "bad cols", cols[:12], "count", cols.size, " z there", xa["z"][r], xb["z"][r])
