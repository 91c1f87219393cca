"""BASELINE.json configurations C3, C5 and C4 at FULL size on the GPU (C2: tests/test_gpu_parity.py::test_c2_full_size).

Every problem of a batch is checked on the device through the size-independent properties of the subproblem
(src/enlsip_functions.jl:55-115, :158-205 of the reference: `min 1/2 ||J p + r||^2 s.t. A p + c = 0`):
    constraint residual     |A p + c|_inf <= 1e-12 (||A||_F ||p|| + ||c||)
    projected gradient      || (I - A'(A A')^-1 A) J'(J p + r) || <= 1e-11 ||J||_F^2 ||p||
    ranks                   rankA = t, rankJ2 = n - t, code = 1 (the synthetic Jacobians are well conditioned)
and a sample of problems — including the first and last one of every launch chunk — against the LAPACK oracle on the
same data (p to 1e-11, ranks, jpvtJ2 equal).  Inputs: the counter-based generator of SURVEY §8(d), evaluated in HBM
(enlsip_gn/workload.py); the sampled problems are copied back so that the oracle sees the very same numbers.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import gn_oracle as go

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _solve_batch_dev(solver, J, rx, At, cx):
    from enlsip_gn import SQRT_EPS
    B, n, m = J.shape
    t = At.shape[1]
    dev = J.device
    p = torch.empty((B, n), dtype=torch.float64, device=dev)
    b = torch.empty((B, max(t, 1)), dtype=torch.float64, device=dev)
    d = torch.empty((B, m), dtype=torch.float64, device=dev)
    info = torch.empty((B, 6), dtype=torch.int64, device=dev)
    jJ = torch.empty((B, n), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    solver.solve_batched_dev(B, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr() if t else 0, max(n, 1), n * t,
                             cx.data_ptr() if t else 0, SQRT_EPS, dp=p.data_ptr(), db=b.data_ptr(), dd=d.data_ptr(),
                             dinfo=info.data_ptr(), djJ=jJ.data_ptr())
    solver.synchronize()
    return p, b, d, info, jJ


def _check_all_problems(J, rx, At, cx, p, d, info, chunk=2048):
    """Properties of every problem; the m x n products run on the device in chunks, the t x t systems on the host."""
    B, n, m = J.shape
    t = At.shape[1]
    assert bool(torch.isfinite(p).all().item())
    inf = info.cpu().numpy()
    assert (inf[:, 0] == t).all() and (inf[:, 1] == n - t).all() and (inf[:, 2] == 1).all(), "ranks / code"
    assert (inf[:, 3] == t).all() and (inf[:, 4] == n - t).all() and (inf[:, 5] == 0).all(), "dims / status"
    worst_c, worst_g, worst_d = 0.0, 0.0, 0.0
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        Jc, pc = J[lo:hi], p[lo:hi]
        res = torch.bmm(pc[:, None, :], Jc)[:, 0, :] + rx[lo:hi]                 # J p + r      (b, m)
        g = torch.bmm(Jc, res[:, :, None])[:, :, 0]                                # J'(J p + r)  (b, n)
        nJ2 = (Jc * Jc).sum(dim=(1, 2))
        npn = pc.norm(dim=1)
        if t:
            Ac = At[lo:hi]                                                        # (b, t, n) = A
            cons = (torch.bmm(Ac, pc[:, :, None])[:, :, 0] + cx[lo:hi]).abs().amax(dim=1)
            scale = Ac.flatten(1).norm(dim=1) * npn + cx[lo:hi].norm(dim=1)
            worst_c = max(worst_c, float((cons / scale).max().item()))
            Ah, gh = Ac.cpu().numpy(), g.cpu().numpy()
            lam = np.linalg.solve(Ah @ Ah.transpose(0, 2, 1), (Ah @ gh[:, :, None]))     # (A A') lam = A g
            gproj = gh - (Ah.transpose(0, 2, 1) @ lam)[:, :, 0]
        else:
            gproj = g.cpu().numpy()
        ratio = np.linalg.norm(gproj, axis=1) / (nJ2.cpu().numpy() * npn.cpu().numpy())
        worst_g = max(worst_g, float(ratio.max()))
        # ||d|| = ||-J1 p1 - rx|| (Q3 is orthogonal): with p = Q1 [p1; p2], J1 p1 = J p - J2 p2 is not available here, but
        # ||d||^2 = ||J p + r||^2 + ||d[1:rankJ2]||^2 (the solved part) holds for the full-rank case
        lead = d[lo:hi, : n - t]
        lhs = (d[lo:hi] ** 2).sum(dim=1)
        rhs = (res ** 2).sum(dim=1) + (lead ** 2).sum(dim=1)
        worst_d = max(worst_d, float(((lhs - rhs).abs() / rhs).max().item()))
    assert worst_c <= 1e-12, worst_c
    assert worst_g <= 1e-11, worst_g
    assert worst_d <= 1e-10, worst_d


def _check_sample_against_oracle(J, rx, At, cx, p, d, info, jJ, sample):
    B, n, m = J.shape
    t = At.shape[1]
    for k in sample:
        Jh = np.asfortranarray(J[k].cpu().numpy().T)
        ref = go.gn_subproblem(Jh, rx[k].cpu().numpy(), At[k].cpu().numpy().reshape(t, n), cx[k].cpu().numpy())
        pk = p[k].cpu().numpy()
        assert np.linalg.norm(pk - ref.p) <= 1e-11 * np.linalg.norm(ref.p), (k, np.linalg.norm(pk - ref.p))
        ik = info[k].cpu().numpy()
        assert (ik[0], ik[1], ik[2]) == (ref.rankA, ref.rankJ2, ref.code), k
        n2 = n - ref.rankA
        assert np.array_equal(jJ[k, :n2].cpu().numpy(), ref.jpvtJ2), k
        dk = d[k].cpu().numpy()
        assert abs(np.linalg.norm(dk) - np.linalg.norm(ref.d)) <= 1e-12 * np.linalg.norm(ref.d)
        assert np.abs(np.abs(dk[:n2]) - np.abs(ref.d[:n2])).max() <= 1e-10 * np.abs(ref.d).max()


def _sample(B, chunk_edges=()):
    rng = np.random.default_rng(B)
    s = {0, B - 1, *chunk_edges}
    while len(s) < 16:
        s.add(int(rng.integers(0, B)))
    return sorted(k for k in s if 0 <= k < B)


@pytest.fixture(scope="module")
def solver():
    from enlsip_gn import GNSolver
    s = GNSolver(device=0)
    yield s
    s.close()


def test_c3_full_batch(solver):
    """BASELINE configs[2]: 1024 independent (m=512, n=64, t=8) subproblems in one call."""
    from enlsip_gn import workload as wl
    B, m, n, t = 1024, 512, 64, 8
    J, rx, At, cx = wl.make_batch(0, B, m, n, t, "cuda:0")
    p, b, d, info, jJ = _solve_batch_dev(solver, J, rx, At, cx)
    _check_all_problems(J, rx, At, cx, p, d, info)
    _check_sample_against_oracle(J, rx, At, cx, p, d, info, jJ, _sample(B))


def test_c2_bench_path_batch_256(solver):
    """The path `bench.py` itself times (BASELINE configs[1] as a resident batch): 256 x (m=4096, n=512, t=64) in ONE call on the
    device, with the library's own choices — automatic panel pairs (far update of the first pair >= 8192 workgroups) AND the two
    pipelined halves (batch >= 128: second half on the child handle's stream, driven by a host thread).  Every problem by its
    optimality conditions; the first and last problem of EACH half and a random sample against the LAPACK oracle on the same
    numbers (p, ranks, code, jpvtJ2, |d|: src/enlsip_functions.jl:206-234 at the bench's own batch)."""
    from enlsip_gn import workload as wl
    B, m, n, t = 256, 4096, 512, 64
    J, rx, At, cx = wl.make_batch(0, B, m, n, t, "cuda:0")
    p, b, d, info, jJ = _solve_batch_dev(solver, J, rx, At, cx)
    assert solver.pipeline_split() == B // 2, "the batch was not split over the two pipelined halves"
    assert solver.plan_uses_pairs(), "the sweep did not pair its panels"
    _check_all_problems(J, rx, At, cx, p, d, info, chunk=32)
    half = B // 2
    rng = np.random.default_rng(4096)
    sample = sorted({0, half - 1, half, B - 1, *[int(k) for k in rng.integers(0, B, size=4)]})
    _check_sample_against_oracle(J, rx, At, cx, p, d, info, jJ, sample)


def test_c3_unconstrained_variant(solver):
    """SURVEY 8d lists a t = 0 variant of C3."""
    from enlsip_gn import workload as wl
    B, m, n, t = 1024, 512, 64, 0
    J, rx, At, cx = wl.make_batch(7000, B, m, n, t, "cuda:0")
    p, b, d, info, jJ = _solve_batch_dev(solver, J, rx, At, cx)
    _check_all_problems(J, rx, At, cx, p, d, info)
    _check_sample_against_oracle(J, rx, At, cx, p, d, info, jJ, _sample(B)[:6])


def test_c5_one_gpu_shard(solver):
    """BASELINE configs[4], one GPU's shard: 8192 independent (m=256, n=32, t=4) subproblems (problems 8192 g ... of GPU g;
    here g = 3)."""
    from enlsip_gn import workload as wl
    B, m, n, t = 8192, 256, 32, 4
    J, rx, At, cx = wl.make_batch(3 * B, B, m, n, t, "cuda:0")
    p, b, d, info, jJ = _solve_batch_dev(solver, J, rx, At, cx)
    _check_all_problems(J, rx, At, cx, p, d, info)
    _check_sample_against_oracle(J, rx, At, cx, p, d, info, jJ, _sample(B))


def test_c5_all_65536_in_one_call(solver):
    """The whole of C5 on one GPU in ONE call: above the launch limit (the problem index is a grid dimension), so the
    library cuts it into chunks; the sample takes the problems either side of the cut.  Accessors then address the resident
    (last) chunk by the index in the caller's batch and refuse the earlier ones."""
    from enlsip_gn import workload as wl, FACTOR_J2
    B, m, n, t = 65536, 256, 32, 4
    J, rx, At, cx = wl.make_batch(0, B, m, n, t, "cuda:0")
    p, b, d, info, jJ = _solve_batch_dev(solver, J, rx, At, cx)
    _check_all_problems(J, rx, At, cx, p, d, info, chunk=8192)
    _check_sample_against_oracle(J, rx, At, cx, p, d, info, jJ, _sample(B, (32767, 32768)))
    # resident factors: last chunk only
    k = B - 5
    Jh = np.asfortranarray(J[k].cpu().numpy().T)
    ref = go.gn_subproblem(Jh, rx[k].cpu().numpy(), At[k].cpu().numpy().reshape(t, n), cx[k].cpu().numpy())
    R = solver.factor(FACTOR_J2, prob=k).R
    assert np.abs(np.abs(np.diag(R)) - np.abs(np.diag(ref.F_J2.R))).max() <= 1e-12 * np.abs(ref.F_J2.R[0, 0])
    from enlsip_gn.api import GNError
    with pytest.raises(GNError):
        solver.factor(FACTOR_J2, prob=5).R


def test_batch_above_launch_limit_small_shape(solver):
    """Chunking with an uneven tail: 40000 problems of (m=64, n=8, t=2) = two chunks of 20000, against the oracle at the cut."""
    from enlsip_gn import workload as wl
    B, m, n, t = 40000, 64, 8, 2
    J, rx, At, cx = wl.make_batch(100, B, m, n, t, "cuda:0")
    p, b, d, info, jJ = _solve_batch_dev(solver, J, rx, At, cx)
    assert "chunked" in solver.route()
    _check_all_problems(J, rx, At, cx, p, d, info, chunk=20000)
    _check_sample_against_oracle(J, rx, At, cx, p, d, info, jJ, [0, 19999, 20000, 20001, B - 1])


@pytest.fixture(scope="module")
def solver_pairs():
    """Panel pairs forced (ENLSIP_GN_PAIR=1 at handle creation): a 32768-row shard is below the library's own threshold for them."""
    import os
    from enlsip_gn import GNSolver
    old = os.environ.get("ENLSIP_GN_PAIR")
    os.environ["ENLSIP_GN_PAIR"] = "1"
    try:
        s = GNSolver(device=0)
    finally:
        if old is None:
            del os.environ["ENLSIP_GN_PAIR"]
        else:
            os.environ["ENLSIP_GN_PAIR"] = old
    yield s
    s.close()


@pytest.mark.parametrize("t,with_oracle,pairs", [(0, False, False), (16, True, False), (0, False, True), (16, False, True)])
def test_c4_full_size_eight_row_shards(solver, solver_pairs, t, with_oracle, pairs):
    """BASELINE configs[3]: ONE tall Jacobian, m = 262144, n = 1024, rows sharded 8 ways (the shards run one after the other on
    this GPU and are stacked exactly as the all-gather stacks them).  Properties for both variants; the t = 16 variant also
    against LAPACK's dgeqp3 on the whole 262144 x 1024 matrix (about half a minute of host time).  `pairs`: the same with the
    CAQR's panel pairs forced for the shards and the stacked problem (the t = 16 stack is the one whose tiles are rotated out
    exactly: the rounding-dust columns of tests/test_gpu_parity.py::test_tsqr_row_shards_match_single_solve at full size)."""
    from enlsip_gn import workload as wl
    from enlsip_gn.tsqr import tsqr_solve_shards_dev
    if pairs:
        solver = solver_pairs
    m, n, G = 262144, 1024, 8
    J, rx, At, cx = wl.make_batch(424242 + t, 1, m, n, t, "cuda:0")
    Jd, rxd = J[0], rx[0]
    Atd, cxd = (At[0], cx[0]) if t else (None, None)
    out = tsqr_solve_shards_dev(solver, Jd, rxd, Atd, cxd, G)
    assert (out.rankA, out.rankJ2, out.code, out.n2) == (t, n - t, 1, n - t)
    p = torch.tensor(out.p, dtype=torch.float64, device=Jd.device)
    res = p @ Jd + rxd                                          # J p + r
    g = Jd @ res                                                # J'(J p + r)
    nJ2 = float((Jd * Jd).sum().item())
    if t:
        Ah = Atd.cpu().numpy()
        gh = g.cpu().numpy()
        cons = np.abs(Ah @ out.p + cxd.cpu().numpy()).max()
        assert cons <= 1e-12 * (np.linalg.norm(Ah) * np.linalg.norm(out.p) + np.linalg.norm(cxd.cpu().numpy()))
        gproj = gh - Ah.T @ np.linalg.solve(Ah @ Ah.T, Ah @ gh)
    else:
        gproj = g.cpu().numpy()
    assert np.linalg.norm(gproj) <= 1e-11 * nJ2 * np.linalg.norm(out.p)
    # ||d||^2 = ||J p + r||^2 + ||d[1:n2]||^2
    lhs, rhs = out.d_norm ** 2, float((res * res).sum().item()) + float(np.dot(out.dlead, out.dlead))
    assert abs(lhs - rhs) <= 1e-10 * rhs
    if t == 0:
        assert abs(out.d_norm - float(rxd.norm().item())) <= 1e-12 * out.d_norm       # d = Q'(-rx)
    assert sorted(out.jpvtJ2.tolist()) == list(range(1, n - t + 1))
    if with_oracle:
        Jh = J[0].cpu().numpy().T                                # (m, n) Fortran-ordered view of the same numbers
        ref = go.gn_subproblem(Jh, rxd.cpu().numpy(), At[0].cpu().numpy().reshape(t, n), cxd.cpu().numpy())
        assert np.linalg.norm(out.p - ref.p) <= 1e-11 * np.linalg.norm(ref.p)
        assert (ref.rankA, ref.rankJ2) == (out.rankA, out.rankJ2)
        assert np.array_equal(out.jpvtJ2, ref.jpvtJ2)
        assert abs(out.d_norm - np.linalg.norm(ref.d)) <= 1e-12 * np.linalg.norm(ref.d)
        assert np.abs(np.abs(out.dlead) - np.abs(ref.d[: n - t])).max() <= 1e-10 * np.abs(ref.d).max()
