"""Probe (not a pytest file): smallest sub-stack of the failing C4-like stack whose unpivoted R0 differs between pairs on / off."""
import os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))

def child(tag):
    import torch
    torch.zeros(1, device="cuda:0")
    from enlsip_gn import GNSolver, SQRT_EPS
    from enlsip_gn.tsqr import hip_local_stage
    src = np.load("/tmp/min_stack.npz")
    out = {}
    s = GNSolver(device=0)
    for key in src.files:
        if not key.startswith("J_"): continue
        J = src[key]; rx = src["rx_" + key[2:]]
        m, n = J.shape
        Jd = torch.tensor(np.ascontiguousarray(J.T), dtype=torch.float64, device="cuda:0"); rxd = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
        R = torch.empty((n * n,), dtype=torch.float64, device="cuda:0"); z = torch.empty((n,), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        hip_local_stage(s, m, n, 0, Jd.data_ptr(), m, rxd.data_ptr(), 0, 0, R.data_ptr(), z.data_ptr(), SQRT_EPS)
        out["R_" + key[2:]] = R.cpu().numpy().reshape(n, n).T; out["z_" + key[2:]] = z.cpu().numpy()
    np.savez(f"/tmp/min_{tag}.npz", **out)
    s.close()

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(sys.argv[2]); sys.exit(0)
rng = np.random.default_rng(5)
n2 = 1008
tri = [np.linalg.qr(rng.standard_normal((n2 + 92, n2)), mode="r") for _ in range(8)]
zreal = None
if os.path.exists("/tmp/st_plain.npz"):      # the library's own triangles of the failing case (tests/probes/pair_probe_stage.py ran before)
    src = np.load("/tmp/st_plain.npz")
    tri = [src["R"][g].reshape(n2, n2).T.copy() for g in range(8)]
    zreal = src["z"].reshape(-1)
    print("using the library's triangles")
# the failing data came from the library's own local stage; numpy's triangles did not fail, so also try sign patterns like ours:
# dlarfg leaves R_jj = -sign(alpha) * norm; numpy (LAPACK dgeqrf) does the same, so the data are alike; what differed was z
cases = {}
def add(name, G, nc, rhs):
    J = np.vstack([tri[g][:, :nc] for g in range(G)])
    cases["J_" + name] = J; cases["rx_" + name] = rhs(G * n2)
add("g8_n1008_rand", 8, 1008, lambda k: rng.standard_normal(k))
add("g8_n1008_big", 8, 1008, lambda k: 1e3 * rng.standard_normal(k))
add("g8_n640_rand", 8, 640, lambda k: rng.standard_normal(k))
add("g8_n608_rand", 8, 608, lambda k: rng.standard_normal(k))
add("g4_n640_rand", 4, 640, lambda k: rng.standard_normal(k))
add("g2_n640_rand", 2, 640, lambda k: rng.standard_normal(k))
add("g3_n640_rand", 3, 640, lambda k: rng.standard_normal(k))
if zreal is not None: add("g8_n1008_zreal", 8, 1008, lambda k: -zreal)
lib = tri
print("library diag signs (fraction negative):", [float((np.diag(t) < 0).mean()) for t in lib[:3]])
npt = [np.linalg.qr(np.random.default_rng(100 + g).standard_normal((n2 + 92, n2)), mode="r") for g in range(3)]
print("numpy diag signs (fraction negative):", [float((np.diag(t) < 0).mean()) for t in npt])
def addt(name, tris, nc):
    J = np.vstack([t[:, :nc] for t in tris]); cases["J_" + name] = J; cases["rx_" + name] = rng.standard_normal(J.shape[0])
addt("np3_n640", npt, 640)
addt("np3_neg_n640", [t * -np.sign(np.diag(t))[:, None] for t in npt], 640)
addt("np3_pos_n640", [t * np.sign(np.diag(t))[:, None] for t in npt], 640)
sg = np.random.default_rng(3)
addt("lib3_randsign_n640", [t * sg.choice([-1.0, 1.0], size=n2)[:, None] for t in lib[:3]], 640)
addt("lib3_pos_n640", [t * np.sign(np.diag(t))[:, None] for t in lib[:3]], 640)
from oracle import synth
Jsyn = synth.make_problem(11, 8800, 1008, 0)[0]
npsame = [np.linalg.qr(Jsyn[1100 * g:1100 * (g + 1)], mode="r") for g in range(3)]
print("numpy triangles of the same shards vs the library's: max | |R_np| - |R_lib| | =", [float(np.abs(np.abs(x) - np.abs(y)).max()) for x, y in zip(npsame, lib[:3])])
addt("npsame3_n640", npsame, 640)
pert = np.random.default_rng(9)
addt("lib3_pert1e-13_n640", [t * (1 + 1e-13 * pert.standard_normal(t.shape)) for t in lib[:3]], 640)
addt("lib3_x2_n640", [2.0 * t for t in lib[:3]], 640)
addt("lib3_x1.7_n640", [1.7 * t for t in lib[:3]], 640)
np.savez("/tmp/min_stack.npz", **cases)
for tag, env in (("plain", {"ENLSIP_GN_PAIR": "0"}), ("pair", {"ENLSIP_GN_PAIR": os.environ.get("PAIR_MODE", "1")})):
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, __file__, "child", tag], env=e)
a, b = np.load("/tmp/min_pair.npz"), np.load("/tmp/min_plain.npz")
for key in a.files:
    if not key.startswith("R_"): continue
    Ra, Rb = np.abs(a[key]), np.abs(b[key]); za, zb = np.abs(a["z_" + key[2:]]), np.abs(b["z_" + key[2:]])
    rowerr = np.abs(Ra - Rb).max(axis=1); dz = np.abs(za - zb)
    bad = np.nonzero(rowerr > 1e-11)[0]
    print(f"{key[2:]}: max |R| row diff {rowerr.max():.2e}; first bad rows {bad[:4]}; |z| diff {dz.max():.2e}")
J = cases["J_g3_n640_rand"]
Rnp = np.abs(np.linalg.qr(J, mode="r"))
for tag, x in (("pair", a), ("plain", b)):
    e = np.abs(np.abs(x["R_g3_n640_rand"]) - Rnp).max(axis=1)
    print(f"  g3_n640 {tag} vs numpy QR: max row err {e.max():.2e} first rows > 1e-11: {np.nonzero(e > 1e-11)[0][:6]}")
for g in range(3):
    d = np.abs(np.diag(tri[g]))
    print(f"  triangle {g}: min |diag| {d.min():.3e} at {int(d.argmin())}; |diag[580:590]| {np.array2string(d[580:590], precision=3)}; nan {np.isnan(tri[g]).any()} max|entry| {np.abs(tri[g]).max():.3e}; exact zeros in upper triangle {(np.triu(tri[g]) == 0).sum() - 1008 * 1007 // 2}")
    sub = tri[g][576:600, 576:600]
    print("   tiny entries (<1e-8, nonzero) in upper part near 584:", np.argwhere((np.abs(np.triu(sub)) < 1e-8) & (np.triu(sub) != 0))[:6].tolist())
print("---- detail g3_n640: pair vs numpy |R| ----")
Rp = np.abs(a["R_g3_n640_rand"])
np.set_printoptions(linewidth=200, precision=3)
for r in range(580, 590):
    cols = np.arange(r, min(r + 8, 640))
    print(r, "rel diff:", (Rp[r, cols] - Rnp[r, cols]) / Rnp[r, cols])
colerr = np.abs(Rp - Rnp)[:584, :].max(axis=0)
print("rows < 584: max abs err per column block of 32:", [float(colerr[c:c + 32].max()) for c in range(0, 640, 32)])
cn_p = np.sqrt((Rp ** 2).sum(axis=0)); cn_n = np.sqrt((Rnp ** 2).sum(axis=0))
bad = np.nonzero(np.abs(cn_p - cn_n) > 1e-10)[0]
print("columns whose norm |R[:, c]| differs from numpy's (invariant = column norm of J):", bad[:12], "count", bad.size, "max", np.abs(cn_p - cn_n).max())
