#!/usr/bin/env python3
"""bench.py — GN-subproblem solves/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE configs[1] "C2" — dense synthetic CNLS subproblems with
m=4096 residuals, n=512 parameters, t=64 active (equality) constraints, fp64.  One *step* is one
pass of the hot path (src/enlsip_functions.jl:700, 768-771, 206-234, 116-153 of the reference)
over one batch of `--batch` independent subproblems that are already resident in HBM; the batch
(default 384 problems = 6.4 GB of Jacobians) is larger than the 256 MB Infinity Cache so the
traffic is real HBM traffic.  value = problems solved by all ranks / wall time of the K timed
steps (barrier + synchronize on both sides, MAX over ranks).  Ranks shard independent
subproblems: no collective on the data path (weak scaling).

Extra objects on the JSON line:
  roofline      dominant kernel = level-0 CAQR trailing update (k_caqr_update_v4): algorithmic
                bytes 8(2 m_k n_k + m_k b + b^2) per launch / launch time measured
                with HIP events on the library's stream (enlsip_gn_get_update_stats).
                Shapes without a level-0 update (--m 256 --n 32 ..., the C3 / C5 small-problem
                configs) report the whole solve against the HBM stream of its operands instead.
  whole_solve   SURVEY §8d's per-solve algorithmic bytes and flops x this GPU's solve rate
                (GB/s and f64 TFLOP/s of the whole pipeline, context for `value`).
  cpu_baseline  the same LAPACK call sequence the Julia reference dispatches to (scipy/OpenBLAS
                "port", oracle/cpu_baseline.py) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "enlsip.jl_amd" / "python"))
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MEASURED_COPY_GBS = 5785.0   # in-place read+write stream with non-temporal loads and stores, update-kernel grid (tests/microbench/copy_variants.hip; 5.3 TB/s with plain accesses)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=384, help="independent C2 subproblems per GPU per step")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the batch over this many handles/HIP streams driven by host threads")
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--t", type=int, default=64)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-baseline sampling (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tile-rows", type=int, default=0, help="CAQR tile rows (0 = library default 512; 256)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    # rehearsal switch (tests only): ENLSIP_BENCH_DIST_BACKEND=gloo runs the N > 1 control flow with every rank on the
    # GPUs that exist (local_rank modulo device count) — RCCL refuses two ranks on one device, the one-GPU boxes have one
    backend = os.environ.get("ENLSIP_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from enlsip_gn import GNSolver, SQRT_EPS

    m, n, t, B = args.m, args.n, args.t, args.batch
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20260101 + rank)
    # column-major m x n per problem == (B, n, m) C-order
    J = torch.randn((B, n, m), dtype=torch.float64, device=dev, generator=gen)
    rx = torch.randn((B, m), dtype=torch.float64, device=dev, generator=gen)
    At = torch.randn((B, t, n), dtype=torch.float64, device=dev, generator=gen)   # column-major n x t
    cx = torch.randn((B, t), dtype=torch.float64, device=dev, generator=gen)
    p = torch.empty((B, n), dtype=torch.float64, device=dev)
    b = torch.empty((B, max(t, 1)), dtype=torch.float64, device=dev)
    d = torch.empty((B, m), dtype=torch.float64, device=dev)
    jJ = torch.empty((B, n), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    solver = GNSolver(device=local_rank, tile_rows=args.tile_rows)
    S = max(1, min(args.streams, B))
    solvers = [solver] + [GNSolver(device=local_rank, tile_rows=args.tile_rows) for _ in range(S - 1)]
    bounds = [(B * i) // S for i in range(S + 1)]

    def part(i):
        lo, hi = bounds[i], bounds[i + 1]
        solvers[i].solve_batched_dev(hi - lo, m, n, t, J[lo].data_ptr(), m, m * n, rx[lo].data_ptr(),
                                     At[lo].data_ptr(), n, n * t, cx[lo].data_ptr(), SQRT_EPS,
                                     dp=p[lo].data_ptr(), db=b[lo].data_ptr(), dd=d[lo].data_ptr(),
                                     djJ=jJ[lo].data_ptr())

    if S > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=S)

    def step():
        if S == 1:
            part(0)
        else:
            list(pool.map(part, range(S)))     # ctypes releases the GIL: the S streams run concurrently

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step()                  # set-up, not a warm-up step: allocates the handles' workspaces and the second pipeline stream
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()              # returns after the stream has drained (info check), see include/enlsip_gn.h
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: the step solved what it claims (constraint residual of problem 0, finite outputs)
    A0 = At[0].cpu().numpy()                   # (t, n) = A_active
    p0 = p[0].cpu().numpy()
    c0 = cx[0].cpu().numpy()
    cons = float(np.abs(A0 @ p0 + c0).max()) if t else 0.0
    ok = bool(torch.isfinite(p).all().item()) and cons < 1e-9

    # single-problem latency (batch = 1) for context
    lat_ms = None
    s1 = GNSolver(device=local_rank)
    for _ in range(2):
        s1.solve_batched_dev(1, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr(), n, n * t,
                             cx.data_ptr(), SQRT_EPS, dp=p.data_ptr())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        s1.solve_batched_dev(1, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr(), n, n * t,
                             cx.data_ptr(), SQRT_EPS, dp=p.data_ptr())
    torch.cuda.synchronize()
    lat_ms = (time.perf_counter() - t1) / reps * 1e3
    s1.close()

    roofline = None
    stage_ms = None
    if rank == 0 and not args.no_roofline:
        solver.set_profiling(True)
        step()
        step()
        avg_ms, launches, bytes_total = solver.update_stats()
        stage_ms = solver.stage_ms()
        solver.set_profiling(False)
        if launches:
            per_launch_bytes = bytes_total / launches
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "k_caqr_update_v4 (level-0 trailing update)",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "launches_per_step": launches, "avg_launch_ms": round(avg_ms, 5),
                        "algorithmic_bytes_per_launch": per_launch_bytes,
                        # measured ceiling of an in-place read+write stream with this kernel's grid on this chip
                        # (tests/microbench/access_patterns.hip, profiles/r1_notes.md); `frac` stays against `peak`
                        "measured_copy_peak": MEASURED_COPY_GBS,
                        "frac_of_measured_copy": round(achieved / MEASURED_COPY_GBS, 4)}
            # HBM bytes per launch from the committed PMC passes of this same command and workload
            # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 correction applied; profiles/): PMC counters
            # cannot be collected from inside the run, so the figure is carried only for a matching config.
            # (two separate --pmc passes post-processed by tests/pmc_update_traffic.py)
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            for name in ("r1h_update_traffic_pmc.json", "r1e_update_traffic_pmc.json"):      # newest first
                pmc = os.path.join(pdir, name)
                if not os.path.exists(pmc):
                    continue
                with open(pmc) as fh:
                    rec = json.load(fh)
                if rec.get("config") == {"m": m, "n": n, "t": t, "batch": B}:
                    roofline["traffic"] = rec["hbm_bytes_per_launch_avg"]
                    roofline["traffic_source"] = "profiles/" + name
                    break

    # whole-solve figures from SURVEY section 8(d)'s per-solve formulas (full rank, n2 = n - t): every operand crosses HBM
    # once per direction; flops of the LAPACK sequence the path replaces
    n2 = n - min(n, t)
    solve_bytes = 8.0 * (2 * m * n + 2 * m + 2 * n * t + t + n)
    solve_flops = (2.0 * t * t * (n - t / 3.0) + (4.0 / 3.0) * t ** 3 + 4.0 * m * t * (n - t / 2.0)
                   + 2.0 * n2 * n2 * (m - n2 / 3.0) + 2.0 * m * t + (4.0 * m * n2 - 2.0 * n2 * n2) + n2 * n2 + t * t + 4.0 * n * t)
    sps_rank = B * args.steps / elapsed                    # this GPU's rate (weak scaling: same work per rank)
    whole = {"bytes_per_solve": solve_bytes, "flops_per_solve": solve_flops,
             "hbm_GBps": round(sps_rank * solve_bytes / 1e9, 1), "hbm_frac": round(sps_rank * solve_bytes / 1e9 / HBM_PEAK_GBS, 4),
             "f64_TFLOPs": round(sps_rank * solve_flops / 1e12, 2), "f64_frac_of_78.6": round(sps_rank * solve_flops / 78.6e12, 4)}
    if rank == 0 and not args.no_roofline and roofline is None:
        # no level-0 trailing update at this shape (n2 <= 32: the single-panel / wave-per-problem pipeline): the bound of
        # the whole solve is the HBM stream of its operands (batched small-problem configs)
        roofline = {"bound": "hbm", "kernel": "whole solve (constraint + J Q1 + panel + pivoted solve kernels)",
                    "achieved": whole["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": whole["hbm_frac"],
                    "traffic": None, "algorithmic_bytes_per_solve": solve_bytes}

    if (m, n, t) == (4096, 512, 64):
        wl = "C2: batch of independent dense CNLS subproblems, m=4096 n=512 t=64 fp64, inputs resident in HBM"
    elif (m, n) == (512, 64):
        wl = f"C3: batch of independent (m=512, n=64, t={t}) subproblems fp64, inputs resident in HBM"
    elif (m, n) == (256, 32):
        wl = f"C5: batch of independent (m=256, n=32, t={t}) subproblems fp64, inputs resident in HBM"
    else:
        wl = f"custom: batch of independent (m={m}, n={n}, t={t}) subproblems fp64, inputs resident in HBM"

    cpu = None
    if rank == 0 and world == 1 and args.cpu_budget > 0:      # the CPU leg runs at N = 1 only
        from oracle import cpu_baseline as cb      # measurement leg only
        Jh = np.asfortranarray(J[0].cpu().numpy().T)
        sps, nsolves, secs, threads = cb.time_baseline(Jh, rx[0].cpu().numpy(), A0, c0, budget_s=args.cpu_budget)
        cpu = {"value": round(sps, 3), "unit": "solves/s", "cores": threads, "kind": "port",
               "sample": f"{nsolves} solves of problem 0 of the same batch in {secs:.1f} s, scipy/OpenBLAS "
                         f"dgeqp3+dormqr+dtrtrs sequence with the reference's copies, {threads} BLAS threads "
                         f"on {cb.host_cores()} host cores"}

    if rank == 0:
        total = world * B * args.steps
        out = {
            "metric": "GN subproblem solves/sec at (m=4096,n=512); achieved fraction of HBM roofline",
            "value": round(total / elapsed, 2), "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl, "m": m, "n": n, "t": t,
                       "batch_per_gpu": B, "streams_per_gpu": S, "parallelism": f"independent subproblems x{world}"},
            "single_problem_latency_ms": round(lat_ms, 3),
            "results_check": {"finite": ok, "max_constraint_residual_problem0": cons},
            "roofline": roofline, "cpu_baseline": cpu, "stage_ms_per_step": stage_ms,
            "whole_solve": whole,
        }
        print(json.dumps(out), flush=True)
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
