"""CPU tests (gloo, world_size 2) of the host logic of the multi-GPU paths: row partitioning, the
single exchange step of the TSQR driver (all-gather of triangles + vectors, all-reduce of the tail
norm), stacking order, and the problem-range sharding used by bench.py.  The HIP stages are
replaced by NumPy stand-ins built from the oracle (test infrastructure only): what is under test
is enlsip_gn.tsqr.tsqr_solve's communication and bookkeeping, which is device-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gn_oracle as go, lapack_semantics as ls, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _np_local_stage(J_loc, rx_loc, At, cx, R, z, eps_rank):
    """Stand-in for enlsip_gn_tsqr_local_dev: F_A (replicated), J*Q1, unpivoted QR of the local rows."""
    Jl = J_loc.numpy().T.copy()
    n = Jl.shape[1]
    t = 0 if At is None else At.shape[0]
    A = At.numpy() if t else np.zeros((0, n))
    c = cx.numpy() if t else np.zeros(0)
    F_A = go.qr_colnorm(A.T)
    rankA = go.pseudo_rank(F_A.diagR(), eps_rank)
    JQ1 = F_A.rmul_Q(Jl)
    b = -c[F_A.p - 1] if t else np.zeros(0)
    p1 = np.linalg.solve(F_A.R[:t, :t].T, b) if t else np.zeros(0)      # full-rank A in these tests
    d = -JQ1[:, :rankA] @ p1 - rx_loc.numpy()
    n2 = n - rankA
    f, tau = ls.geqr2(JQ1[:, rankA:])
    dq = ls.apply_qt(f, tau, d)
    kp = min(Jl.shape[0], n2)
    Rl = np.zeros((n2, n2))
    Rl[:kp] = np.triu(f[:kp, :n2])
    R[: n2 * n2] = torch.from_numpy(Rl.reshape(-1, order="F").copy())
    zz = np.zeros(n2)
    zz[:kp] = dq[:kp]
    z[:n2] = torch.from_numpy(zz)
    return n2, float(dq[kp:] @ dq[kp:])


def _make_combine(A, cx, eps_rank_unused=None):
    def combine(G, n, n2, Rstack, zstack, eps_rank):
        Rs = Rstack.numpy().reshape(G, n2, n2).transpose(0, 2, 1)        # each block column-major
        stack = np.concatenate(list(Rs), axis=0)                          # (G n2) x n2, rank order
        zs = zstack.numpy().copy()
        F = go.qr_colnorm(stack)
        rankJ2 = go.pseudo_rank(F.diagR(), eps_rank)
        dq = F.Qt_mul(zs)
        dp2 = np.linalg.solve(F.R[:rankJ2, :rankJ2], dq[:rankJ2])
        p2 = np.concatenate([dp2, np.zeros(n2 - rankJ2)])[go.invperm(F.p)]
        t = A.shape[0]
        F_A = go.qr_colnorm(A.T)
        b = -cx[F_A.p - 1] if t else np.zeros(0)
        p1 = np.linalg.solve(F_A.R[:t, :t].T, b) if t else np.zeros(0)
        p = F_A.Q_mul(np.concatenate([p1, p2])) if t else p2
        return p, dq[:n2].copy(), float(dq[n2:] @ dq[n2:]), t, rankJ2, 1, F.p.copy()
    return combine


def _worker(rank, world, port, m, n, t, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from enlsip_gn.tsqr import tsqr_solve, row_range
    J, rx, A, cx = synth.make_problem(77, m, n, t)
    lo, hi = row_range(m, world, rank)
    J_loc = torch.from_numpy(np.ascontiguousarray(J[lo:hi].T))
    rx_loc = torch.from_numpy(rx[lo:hi].copy())
    At = torch.from_numpy(np.ascontiguousarray(A)) if t else None
    cxt = torch.from_numpy(cx.copy()) if t else None
    res = tsqr_solve(None, J_loc, rx_loc, At, cxt, go.SQRT_EPS, local_stage=_np_local_stage,
                     combine_stage=_make_combine(A, cx))
    np.savez(out + f".{rank}.npz", p=res.p, d_norm=res.d_norm, rankJ2=res.rankJ2, n2=res.n2, jp=res.jpvtJ2)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("m,n,t", [(600, 24, 0), (601, 20, 3)])
def test_tsqr_driver_gloo_world2(tmp_path, m, n, t):
    port = _free_port()
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(2, port, m, n, t, out), nprocs=2, join=True)
    J, rx, A, cx = synth.make_problem(77, m, n, t)
    ref = go.gn_subproblem(J, rx, A, cx)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    for r in (r0, r1):                                   # every rank returns the same, correct answer
        assert np.linalg.norm(r["p"] - ref.p) <= 1e-11 * np.linalg.norm(ref.p)
        assert int(r["rankJ2"]) == ref.rankJ2 and int(r["n2"]) == n - ref.rankA
        assert abs(float(r["d_norm"]) - np.linalg.norm(ref.d)) <= 1e-11 * np.linalg.norm(ref.d)
        assert np.array_equal(r["jp"], ref.jpvtJ2)
    assert np.array_equal(r0["p"], r1["p"])


def test_row_range_partition():
    from enlsip_gn.tsqr import row_range
    for m, G in ((10, 3), (262144, 8), (7, 8), (4096, 2)):
        edges = [row_range(m, G, g) for g in range(G)]
        assert edges[0][0] == 0 and edges[-1][1] == m
        assert all(edges[i][1] == edges[i + 1][0] for i in range(G - 1))
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1
