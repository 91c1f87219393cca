"""Row-sharded TSQR of one tall residual Jacobian across the GPUs of a node (config C4).

One process per GPU under ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI).  Rank g holds
rows ``[g*m/G, (g+1)*m/G)`` of ``J`` and ``rx``; the small constraint data are replicated.  There is
exactly ONE exchange step on the path: an all-gather of the ``n2 x n2`` triangles (8*n2^2 bytes per
rank, 8.4 MB at n2 = 1024) and of the ``n2``-vectors, plus a scalar all-reduce for ``||d||``.  xGMI
is point-to-point (7 links per GPU), so the all-gather of G-1 messages uses all links at once
and costs tens of microseconds against ~1 ms of local factorisation.

The local and combine stages are the HIP library's ``enlsip_gn_tsqr_local_dev`` /
``enlsip_gn_tsqr_combine_dev``; they are injectable only so that the exchange / stacking logic can be
exercised with the ``gloo`` backend on machines without a GPU (tests/test_tsqr_host.py).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import _lib as L
from .api import GNSolver, SQRT_EPS


@dataclass
class TSQRResult:
    p: np.ndarray          # search direction (n)
    dlead: np.ndarray      # leading n2 entries of F_J2.Q' d
    d_norm: float          # ||d||_2 over all ranks
    rankA: int
    rankJ2: int
    code: int
    jpvtJ2: np.ndarray
    n2: int


def row_range(m: int, G: int, g: int):
    """Rows of rank g: contiguous blocks, the first m % G ranks get one extra row."""
    base, extra = divmod(m, G)
    lo = g * base + min(g, extra)
    return lo, lo + base + (1 if g < extra else 0)


def hip_local_stage(solver: GNSolver, m_loc, n, t, dJ, ldj, drx, dAt, dcx, dR, dz, eps_rank):
    """enlsip_gn_tsqr_local_dev on raw device pointers; returns (n2, tail_sq)."""
    tail = C.c_double(0.0)
    n2 = C.c_int64(0)
    v = lambda x: C.c_void_p(x) if x else None
    solver._chk(solver._lib.enlsip_gn_tsqr_local_dev(solver._h, m_loc, n, t, v(dJ), ldj, v(drx), v(dAt), max(n, 1),
                                                     v(dcx), eps_rank, v(dR), v(dz), C.byref(tail), C.byref(n2)))
    return int(n2.value), float(tail.value)


def hip_combine_stage(solver: GNSolver, G, n, n2, dRstack, dzstack, eps_rank):
    p = np.zeros(n)
    dlead = np.zeros(n2)
    jp = np.zeros(n2, dtype=np.int64)
    ct = C.c_double(0.0)
    info = L.Info()
    v = lambda x: C.c_void_p(x) if x else None
    solver._chk(solver._lib.enlsip_gn_tsqr_combine_dev(solver._h, G, n2, v(dRstack), v(dzstack), eps_rank,
                                                       p.ctypes.data_as(C.c_void_p), dlead.ctypes.data_as(C.c_void_p),
                                                       C.byref(ct), C.byref(info), jp.ctypes.data_as(C.c_void_p)))
    return p, dlead, float(ct.value), info, jp


def tsqr_solve(solver: Optional[GNSolver], J_loc, rx_loc, At, cx, eps_rank: float = SQRT_EPS, group=None,
               local_stage: Callable = None, combine_stage: Callable = None) -> TSQRResult:
    """Collective over ``group``.  ``J_loc``: torch tensor (n, m_loc) C-order == column-major
    m_loc x n on this rank's device; ``rx_loc`` (m_loc); ``At`` (t, n) C-order == column-major n x t
    (replicated); ``cx`` (t).  Every rank returns the same result."""
    import torch
    import torch.distributed as dist

    G = dist.get_world_size(group) if dist.is_initialized() else 1
    n, m_loc = J_loc.shape
    t = 0 if At is None else At.shape[0]
    dev = J_loc.device
    # no fill: k_tsqr_extract writes every entry it hands back, and a fill on torch's stream would not be ordered against the
    # library's own stream
    R = torch.empty((n * n,), dtype=torch.float64, device=dev)
    z = torch.empty((n,), dtype=torch.float64, device=dev)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()      # inputs produced on torch's stream are complete before the library reads them
    if local_stage is None:
        n2, tail = hip_local_stage(solver, m_loc, n, t, J_loc.data_ptr(), m_loc, rx_loc.data_ptr(),
                                   At.data_ptr() if t else 0, cx.data_ptr() if t else 0, R.data_ptr(), z.data_ptr(),
                                   eps_rank)
    else:
        n2, tail = local_stage(J_loc, rx_loc, At, cx, R, z, eps_rank)
    # ---- the one exchange step --------------------------------------------------------------------
    Rstack = torch.empty((G * n2 * n2,), dtype=torch.float64, device=dev)
    zstack = torch.empty((G * n2,), dtype=torch.float64, device=dev)
    tails = torch.tensor([tail], dtype=torch.float64, device=dev)
    if G > 1:
        if dev.type == "cuda" and dist.get_backend(group) != "nccl":
            # rehearsal only (several ranks on one GPU under gloo, tests/tsqr_rank_worker.py): RCCL refuses two ranks on one
            # device, gloo moves host tensors
            Rc, zc, tc = torch.empty(Rstack.shape, dtype=torch.float64), torch.empty(zstack.shape, dtype=torch.float64), tails.cpu()
            dist.all_gather_into_tensor(Rc, R[: n2 * n2].cpu().contiguous(), group=group)
            dist.all_gather_into_tensor(zc, z[:n2].cpu().contiguous(), group=group)
            dist.all_reduce(tc, op=dist.ReduceOp.SUM, group=group)
            Rstack.copy_(Rc); zstack.copy_(zc); tails.copy_(tc)
        else:
            dist.all_gather_into_tensor(Rstack, R[: n2 * n2].contiguous(), group=group)
            dist.all_gather_into_tensor(zstack, z[:n2].contiguous(), group=group)
            dist.all_reduce(tails, op=dist.ReduceOp.SUM, group=group)
    else:
        Rstack.copy_(R[: n2 * n2])
        zstack.copy_(z[:n2])
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()      # the library runs on its own stream
    if combine_stage is None:
        p, dlead, ctail, info, jp = hip_combine_stage(solver, G, n, n2, Rstack.data_ptr(), zstack.data_ptr(), eps_rank)
        rankA, rankJ2, code = int(info.rankA), int(info.rankJ2), int(info.code)
    else:
        p, dlead, ctail, rankA, rankJ2, code, jp = combine_stage(G, n, n2, Rstack, zstack, eps_rank)
    d_norm = float(np.sqrt(float(tails.item()) + ctail + float(np.dot(dlead, dlead))))
    return TSQRResult(p=p, dlead=dlead, d_norm=d_norm, rankA=rankA, rankJ2=rankJ2, code=code, jpvtJ2=jp, n2=n2)


# ---- the collective inside the library (enlsip_gn_solve_tsqr) --------------------------------------------------------------
def tsqr_attach(solver: GNSolver, group=None, transport: str = "rccl"):
    """Give the handle its communicator for ``tsqr_solve_lib`` (collective over ``group``).

    ``rccl``: rank 0 asks the library for an RCCL unique id, ``torch.distributed`` only carries those 128 bytes to the other
    ranks, and every rank lets the library create its own RCCL communicator (ncclCommInitRank) — the data path never touches
    torch.  ``host``: an all-gather callback that stages through host memory and the group's (gloo) backend — the rehearsal
    transport for several ranks on ONE GPU, which RCCL refuses."""
    import torch
    import torch.distributed as dist
    G = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lib, h = solver._lib, solver._h
    if G == 1 and transport != "rccl":
        solver._chk(lib.enlsip_gn_tsqr_set_exchange(h, None, None, 1, 0))
        return
    if transport == "rccl":
        # also with ONE rank: the library then runs its exchange as a self-gather through RCCL (transport "rccl" in
        # tsqr_transport), so the RCCL leg is exercised on a one-GPU box
        ident = C.create_string_buffer(128)
        if rank == 0:
            rc = lib.enlsip_gn_tsqr_unique_id(ident)
            if rc != 0:
                raise RuntimeError(f"enlsip_gn_tsqr_unique_id failed with code {rc} (RCCL not loadable?)")
        if G > 1:
            box = [ident.raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        solver._chk(lib.enlsip_gn_tsqr_init_rccl(h, ident, G, rank))
        return
    if transport != "host":
        raise ValueError(transport)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int

    def _gather(ctx, dsend, drecv, nbytes, stream):
        try:
            mine = torch.empty(nbytes, dtype=torch.uint8)
            if hip.hipMemcpy(mine.data_ptr(), dsend, nbytes, 2) != 0:          # hipMemcpyDeviceToHost
                return 1
            every = torch.empty(G * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(every, mine, group=group)
            return 2 if hip.hipMemcpy(drecv, every.data_ptr(), G * nbytes, 1) != 0 else 0      # hipMemcpyHostToDevice
        except Exception:                                                           # never unwind through the C frames
            return 3

    cb = L.ALLGATHER_FN(_gather)
    solver._tsqr_cb = cb                                                            # keep the thunk alive with the handle
    solver._chk(lib.enlsip_gn_tsqr_set_exchange(h, C.cast(cb, C.c_void_p), None, G, rank))


def tsqr_solve_lib(solver: GNSolver, J_loc, rx_loc, At, cx, eps_rank: float = SQRT_EPS) -> TSQRResult:
    """``enlsip_gn_solve_tsqr``: collective over the communicator attached with ``tsqr_attach``.  Arguments as ``tsqr_solve``;
    the inputs must be complete (synchronise the producing stream first: the library runs on its own)."""
    n, m_loc = J_loc.shape
    t = 0 if At is None else At.shape[0]
    p, dlead, jp = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int64)
    dn = C.c_double(0.0)
    info = L.Info()
    v = lambda x: C.c_void_p(x) if x else None
    solver._chk(solver._lib.enlsip_gn_solve_tsqr(
        solver._h, m_loc, n, t, v(J_loc.data_ptr()), m_loc, v(rx_loc.data_ptr()), v(At.data_ptr()) if t else None, max(n, 1),
        v(cx.data_ptr()) if t else None, eps_rank, p.ctypes.data_as(C.c_void_p), dlead.ctypes.data_as(C.c_void_p),
        C.byref(dn), C.byref(info), jp.ctypes.data_as(C.c_void_p)))
    n2 = n - int(info.rankA)
    return TSQRResult(p=p, dlead=dlead[:n2].copy(), d_norm=float(dn.value), rankA=int(info.rankA), rankJ2=int(info.rankJ2),
                      code=int(info.code), jpvtJ2=jp[:n2].copy(), n2=n2)


TRANSPORT_NAMES = {0: "none", 1: "rccl", 2: "callback"}


def tsqr_transport(solver: GNSolver) -> str:
    """What moved the messages in the handle's last ``enlsip_gn_solve_tsqr``: "none" (one rank, device copy), "rccl", "callback"."""
    code = C.c_int(-1)
    solver._chk(solver._lib.enlsip_gn_tsqr_get_transport(solver._h, C.byref(code)))
    return TRANSPORT_NAMES.get(int(code.value), str(code.value))


def tsqr_exchange(solver: GNSolver) -> dict:
    """Transport, rank count and rank tags seen in the handle's last ``enlsip_gn_solve_tsqr`` (enlsip_gn_tsqr_get_exchange):
    ``rank_tags_seen == ranks`` means one message from each distinct rank arrived in its slot."""
    tr, rk, seen = C.c_int(-1), C.c_int(0), C.c_int(-1)
    solver._chk(solver._lib.enlsip_gn_tsqr_get_exchange(solver._h, C.byref(tr), C.byref(rk), C.byref(seen)))
    return {"transport": TRANSPORT_NAMES.get(int(tr.value), str(tr.value)), "ranks": int(rk.value), "rank_tags_seen": int(seen.value)}


def tsqr_stage_ms(solver: GNSolver):
    arr = (C.c_float * 3)()
    solver._chk(solver._lib.enlsip_gn_tsqr_get_stage_ms(solver._h, arr))
    return {"local": float(arr[0]), "exchange": float(arr[1]), "combine": float(arr[2])}


def tsqr_solve_shards_dev(solver: GNSolver, Jd, rxd, Atd, cxd, G: int, eps_rank: float = SQRT_EPS) -> TSQRResult:
    """Single-process rehearsal on ONE GPU with the data already in HBM: ``Jd`` (n, m) C-order == column-major m x n,
    ``rxd`` (m), ``Atd`` (t, n) C-order == column-major n x t or None, ``cxd`` (t) or None.  The G row blocks are factored one after
    the other on the same handle and stacked exactly as the all-gather would (the handle's resident F_A / p1 come from the
    last local stage, identical on every 'rank')."""
    import torch
    n, m = Jd.shape
    t = 0 if Atd is None else Atd.shape[0]
    dev = Jd.device
    Rs, zs, tail_total, n2 = [], [], 0.0, None
    for g in range(G):
        lo, hi = row_range(m, G, g)
        Jl = Jd[:, lo:hi].contiguous()                 # column-major (hi - lo) x n
        rl = rxd[lo:hi].contiguous()
        R = torch.empty((n * n,), dtype=torch.float64, device=dev)      # k_tsqr_extract writes all n2 * n2 / n2 entries it hands back
        z = torch.empty((n,), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)                    # the library runs on its own stream
        n2, tail = hip_local_stage(solver, hi - lo, n, t, Jl.data_ptr(), hi - lo, rl.data_ptr(),
                                   Atd.data_ptr() if t else 0, cxd.data_ptr() if t else 0, R.data_ptr(), z.data_ptr(),
                                   eps_rank)
        Rs.append(R[: n2 * n2].clone())
        zs.append(z[:n2].clone())
        tail_total += tail
        del Jl, rl
    Rstack = torch.cat(Rs).contiguous()
    zstack = torch.cat(zs).contiguous()
    torch.cuda.synchronize(dev)
    p, dlead, ctail, info, jp = hip_combine_stage(solver, G, n, n2, Rstack.data_ptr(), zstack.data_ptr(), eps_rank)
    d_norm = float(np.sqrt(tail_total + ctail + float(np.dot(dlead, dlead))))
    return TSQRResult(p=p, dlead=dlead, d_norm=d_norm, rankA=int(info.rankA), rankJ2=int(info.rankJ2),
                      code=int(info.code), jpvtJ2=jp, n2=n2)


def tsqr_solve_shards(solver: GNSolver, J, rx, A_active, cx, G: int, eps_rank: float = SQRT_EPS) -> TSQRResult:
    """Same, host arrays in: ``J`` (m, n), ``rx`` (m), ``A_active`` (t, n), ``cx`` (t)."""
    import torch
    t = A_active.shape[0]
    dev = torch.device("cuda", 0)
    Jd = torch.tensor(np.ascontiguousarray(np.asarray(J).T), dtype=torch.float64, device=dev)
    rxd = torch.tensor(np.asarray(rx), dtype=torch.float64, device=dev)
    Atd = torch.tensor(np.ascontiguousarray(A_active), dtype=torch.float64, device=dev) if t else None
    cxd = torch.tensor(np.asarray(cx), dtype=torch.float64, device=dev) if t else None
    return tsqr_solve_shards_dev(solver, Jd, rxd, Atd, cxd, G, eps_rank)
