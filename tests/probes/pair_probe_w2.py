"""Probe (not a pytest file): first pair only, far columns after the two level-0 passes (A/B form, stage by stage):
per-tile column norms must equal those of the input tile (tile-local orthogonal transforms)."""
import os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
src = np.load("/tmp/st_plain.npz"); n2 = int(src["n2"])
tri = [src["R"][g].reshape(n2, n2).T.copy() for g in range(3)]
nc = 640
J = np.vstack([t[:, :nc] for t in tri]); rx = np.random.default_rng(1).standard_normal(J.shape[0])
np.savez("/tmp/w_stack.npz", J=J, rx=rx)
m = J.shape[0]
probe = str(ROOT / "tests" / "pair_probe_w.py")
for stage, what in ((4, "L0(a) on far"), (5, "L0(a), L0(b) on far")):
    e = dict(os.environ); e["ENLSIP_GN_PAIR"] = "2"; e["ENLSIP_GN_DEBUG_MAXPAN"] = "2"; e["ENLSIP_GN_DEBUG_STAGE"] = str(stage)
    subprocess.check_call([sys.executable, probe, "child", f"st{stage}", "2"], env=e)
    W = np.load(f"/tmp/w_st{stage}_2.npy")
    print(f"stage {stage} ({what}):")
    for ti, r0 in enumerate(range(0, m, 512)):
        r1 = min(r0 + 512, m)
        ref = np.sqrt((J[r0:r1, 64:] ** 2).sum(axis=0)); got = np.sqrt((W[r0:r1, 64:nc] ** 2).sum(axis=0))
        err = np.abs(got - ref); bad = np.nonzero(err > 1e-11)[0] + 64
        # rows 0..31 of the tile are not touched by L0(b): check them separately after stage 4
        print(f"  tile {ti} rows {r0}..{r1 - 1}: max col-norm err {err.max():.2e} bad cols {bad[:8]} count {bad.size}")
# ---- tile 3: is what was applied to the far columns the product of the STORED reflectors of the second panel? ----
W4 = np.load("/tmp/w_st4_2.npy"); W5 = np.load("/tmp/w_st5_2.npy")
r0 = 1536; rows = np.arange(r0 + 32, r0 + 512)
V = np.zeros((rows.size, 32))
for j in range(32):
    V[j, j] = 1.0
    V[j + 1:, j] = W5[rows[j + 1:], 32 + j]
print("tile 3, second panel: stored V nonzero pattern: rows with any nonzero below the diagonal:", np.nonzero(np.abs(np.tril(V, -1)).sum(axis=1))[0][:40])
print("  V'V - 2 I diag:", np.array2string(np.diag(V.T @ V) - 2.0, precision=2))
print("  max |off-diagonal of V'V|:", np.abs(V.T @ V - np.diag(np.diag(V.T @ V))).max())
C4 = W4[np.ix_(rows, np.arange(64, nc))].copy(); C5 = W5[np.ix_(rows, np.arange(64, nc))]
X = C4.copy()
for j in range(32):
    v = V[:, j]; tau = 2.0 / (v @ v)
    X -= tau * np.outer(v, v @ X)
d = np.abs(X - C5)
print("  host product of stored reflectors (tau = 2 / v'v) vs kernel result: max diff", d.max(), "worst col", 64 + int(d.max(axis=0).argmax()), "worst row (tile slot)", 32 + int(d.max(axis=1).argmax()))
for jj in range(32):
    X = C4.copy()
    for j in range(jj + 1):
        v = V[:, j]; X -= (2.0 / (v @ v)) * np.outer(v, v @ X)
    # rows 32+j (diag slots) are final after reflector j
    dd = np.abs(X[: jj + 1] - C5[: jj + 1]).max()
    if dd > 1e-11: print(f"   first diag-slot rows wrong once reflectors 0..{jj} are in: {dd:.2e}"); break
Rb = np.triu(W5[np.ix_(rows[:32], np.arange(32, 64))])
print("  diag of the tile's R (second panel):", np.array2string(np.diag(Rb), precision=4))
