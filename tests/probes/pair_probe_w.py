"""Probe (not a pytest file): the working matrix W after k panels of the CAQR sweep, pairs on vs off, on the library's own triangles:
per-aggregate check of the invariants (column norms of the trailing rows = those of numpy's R[r:, :])."""
import os, subprocess, sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))

def child(tag, maxpan):
    import torch
    torch.zeros(1, device="cuda:0")
    from enlsip_gn import GNSolver, SQRT_EPS
    from enlsip_gn.tsqr import hip_local_stage
    src = np.load("/tmp/w_stack.npz")
    J, rx = src["J"], src["rx"]
    m, n = J.shape
    Jd = torch.tensor(np.ascontiguousarray(J.T), dtype=torch.float64, device="cuda:0"); rxd = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
    R = torch.empty((n * n,), dtype=torch.float64, device="cuda:0"); z = torch.empty((n,), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    s = GNSolver(device=0)
    hip_local_stage(s, m, n, 0, Jd.data_ptr(), m, rxd.data_ptr(), 0, 0, R.data_ptr(), z.data_ptr(), SQRT_EPS)
    ldw = C.c_int64(0)
    cap = (m + 64) * (n + 1)
    buf = np.zeros(cap)
    rc = s._lib.enlsip_gn_debug_copy_W(s._h, 0, buf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ldw), cap)
    assert rc == 0, rc
    W = buf[: ldw.value * (n + 1)].reshape(n + 1, ldw.value).T[:m, :]
    np.save(f"/tmp/w_{tag}_{maxpan}.npy", W)
    s.close()

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(sys.argv[2], int(sys.argv[3])); sys.exit(0)
src = np.load("/tmp/st_plain.npz"); n2 = int(src["n2"])
tri = [src["R"][g].reshape(n2, n2).T.copy() for g in range(3)]
nc = 640
J = np.vstack([t[:, :nc] for t in tri]); rx = np.random.default_rng(1).standard_normal(J.shape[0])
np.savez("/tmp/w_stack.npz", J=J, rx=rx)
Rnp = np.linalg.qr(J, mode="r")
pans = [int(x) for x in sys.argv[1:]] or [16, 18, 20]
for mp in pans:
    for tag, mode in (("plain", "0"), ("pair", "1")):
        e = dict(os.environ); e["ENLSIP_GN_PAIR"] = mode; e["ENLSIP_GN_DEBUG_MAXPAN"] = str(mp)
        subprocess.check_call([sys.executable, __file__, "child", tag, str(mp)], env=e)
    r = 32 * mp
    ref = np.sqrt((Rnp[r:, :] ** 2).sum(axis=0))               # column norms of the true trailing matrix
    for tag in ("plain", "pair"):
        W = np.load(f"/tmp/w_{tag}_{mp}.npy")
        got = np.sqrt((W[r:, :nc] ** 2).sum(axis=0))
        err = np.abs(got - ref)[r:]
        bad = np.nonzero(err > 1e-10)[0] + r
        print(f"after {mp} panels, {tag}: trailing column norms (rows >= {r}) max err {err.max():.2e}; bad columns {bad[:10]} count {bad.size}")
