"""Diagnostic (GPU box, stamps build): phase times of a pivot step of the blocked pivoted QR, single C2 problem and batch.
   bash enlsip.jl_amd/build.sh -DENLSIP_GN_LAB -DENLSIP_SB_STEP_STAMPS -o enlsip.jl_amd/lib/libenlsip_gn_stamps.so
   ENLSIP_GN_LIB=enlsip.jl_amd/lib/libenlsip_gn_stamps.so python tests/probes/sb_step_probe.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "enlsip.jl_amd", "python"))
import torch
torch.zeros(1, device="cuda:0")
from enlsip_gn import GNSolver, SQRT_EPS, _lib
from enlsip_gn import workload as wl
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m, n, t = 4096, 512, 64
dev = torch.device("cuda:0")
J, rx, At, cx = wl.make_batch(0, B, m, n, t, dev)
p = torch.empty((B, n), dtype=torch.float64, device=dev)
s = GNSolver(device=0)
lib = C.CDLL(str(_lib.LIB_PATH))
out = (C.c_longlong * 24)()
names = ["owner's part (1/8 of the steps)", "wait for the owner", "LDS reads + dots + reduction", "column loop", "norm downdate",
         "bookkeeping", "wait for the slowest wave", "next pivot search"]
for rep in range(3):
    s.solve_batched_dev(B, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr(), n, n * t, cx.data_ptr(), SQRT_EPS, dp=p.data_ptr())
    torch.cuda.synchronize()
    assert lib.enlsip_gn_debug_sb_phase(out) == 0
steps = out[8]
tot = sum(out[i] for i in range(8))
print(f"batch {B}: {steps} steps, {tot / steps * 10:.0f} ns per step (wave 0 of problem 0)")
for i, nm in enumerate(names):
    print(f"  {out[i] / steps * 10:7.0f} ns  {nm}")
if out[12]:
    print(f"k_sb_update_blk, workgroup 1 of problem 0, mean over {out[12]} launches: block update {out[10] / out[12] * 10:.0f} ns, norm downdate tail {out[11] / out[12] * 10:.0f} ns")
nb = out[23]
if nb:
    bn = ["early exits + keys", "ranking sort + candidate list", "candidates -> registers", "pivot steps", "T factor", "write back + active list"]
    print(f"k_sb_factor_reg per block (mean over {nb} blocks of problem 0, all forms):")
    for i, nm in enumerate(bn):
        print(f"  {out[16 + i] / nb * 10:8.0f} ns  {nm}")
