"""One rank of the row-sharded TSQR (the library's collective enlsip_gn_solve_tsqr, and the two-stage form with the exchange done
by torch.distributed: enlsip_gn.tsqr.tsqr_solve_lib / tsqr_solve) on a real device, launched by
tests/test_gpu_robustness.py under torch.distributed.run: every rank builds the same seeded problem, keeps its row block on the
GPU, joins the collective solve and compares the result with the oracle's single solve.  On this pool's one-GPU boxes the ranks
share device 0, the process group is gloo and the library's exchange runs over its caller-supplied transport (RCCL refuses two
ranks on one device); on a multi-GPU node the same script runs with TSQR_BACKEND=nccl, one device per rank, and the library
creates its own RCCL communicator."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enlsip.jl_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("TSQR_BACKEND", "gloo")
    devidx = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(devidx)
    dev = torch.device("cuda", devidx)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    from oracle import gn_oracle as go, synth
    from enlsip_gn import GNSolver
    from enlsip_gn.tsqr import tsqr_solve, tsqr_solve_lib, tsqr_attach, row_range
    bad = 0
    s = GNSolver(device=devidx)
    tsqr_attach(s, transport="rccl" if backend == "nccl" else "host")
    for (m, n, t) in [(3001, 48, 5), (6000, 300, 0), (2500, 130, 40)]:
        J, rx, A, cx = synth.make_problem(4400 + m, m, n, t)
        ref = go.gn_subproblem(J, rx, A, cx)
        lo, hi = row_range(m, world, rank)
        Jl = torch.tensor(np.ascontiguousarray(J[lo:hi].T), dtype=torch.float64, device=dev)
        rl = torch.tensor(rx[lo:hi], dtype=torch.float64, device=dev)
        At = torch.tensor(np.ascontiguousarray(A), dtype=torch.float64, device=dev) if t else None
        cd = torch.tensor(cx, dtype=torch.float64, device=dev) if t else None
        torch.cuda.synchronize()
        for name, fn in (("lib", tsqr_solve_lib), ("two-stage", tsqr_solve)):
            out = fn(s, Jl, rl, At, cd)
            err = float(np.linalg.norm(out.p - ref.p) / np.linalg.norm(ref.p))
            dn = abs(out.d_norm - float(np.linalg.norm(ref.d))) / float(np.linalg.norm(ref.d))
            n2 = n - ref.rankA
            dl = float(np.abs(np.abs(out.dlead) - np.abs(ref.d[:n2])).max() / max(np.abs(ref.d).max(), 1e-300)) if n2 else 0.0
            ok = (err <= 1e-10 and dn <= 1e-10 and dl <= 1e-9 and (out.rankA, out.rankJ2) == (ref.rankA, ref.rankJ2)
                  and np.array_equal(out.jpvtJ2, ref.jpvtJ2))
            print(f"rank {rank}/{world} {name} m={m} n={n} t={t}: rel p {err:.1e} rel |d| {dn:.1e} lead {dl:.1e} "
                  f"ranks ({out.rankA},{out.rankJ2}) {'ok' if ok else 'FAIL'}", flush=True)
            bad += 0 if ok else 1
    s.close()
    dist.barrier()
    dist.destroy_process_group()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
