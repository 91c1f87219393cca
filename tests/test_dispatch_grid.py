"""The dispatch-derived shape grid (tests/dispatch_grid.py): on the CPU, that its cases reach every kernel-selection bit the
header declares; on the GPU, every case through the library against the oracle, with the route the library REPORTS containing the
route the grid expects — so neither the grid nor the library can grow a branch the other does not know."""
import numpy as np
import pytest

import dispatch_grid as dg
from oracle import gn_oracle as go, synth

GENS = {"full": synth.make_problem, "rankdefA": synth.make_rank_deficient_A, "rankdefJ": synth.make_rank_deficient_J,
        "graded": synth.make_graded_J}


def rel(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def test_grid_covers_every_route_bit():
    names = dg.header_route_names()
    assert len(names) == len(set(names)) >= 50
    covered = {}
    for c in dg.grid():
        for bit in dg.expected_route(c["batch"], c["m"], c["n"], c["t"], c["kind"] == "rankdefA"):
            covered.setdefault(bit, c)
    assert set(covered) <= set(names), sorted(set(covered) - set(names))          # the grid speaks the header's vocabulary
    missing = set(names) - set(covered) - set(dg.COVERED_ELSEWHERE)
    assert not missing, f"route bits without a grid case: {sorted(missing)}"
    assert not (set(dg.COVERED_ELSEWHERE) & set(covered)) - {"jq1_plain"}, "a bit listed as covered elsewhere is reachable: drop it from the list"


def test_grid_walks_the_boundaries_the_sources_name():
    """The thresholds come from the C++ sources; the grid must put a case on either side of each."""
    g = dg.grid()
    n2p1 = {(c["n"] - min(c["n"], c["t"])) + 1 for c in g}
    assert any(x <= 32 for x in n2p1) and any(32 < x <= 64 for x in n2p1) and any(x > 64 for x in n2p1)
    ts = {c["t"] for c in g}
    assert 0 in ts and dg.Q1R_MAXK in ts and dg.Q1R_MAXK + 1 in ts and 63 in ts and 64 in ts and any(t > 64 for t in ts)
    kps = {min(c["m"], c["n"] - min(c["n"], c["t"])) for c in g}
    for lo, hi in ((1, 32), (33, 64), (65, 128), (129, 256), (257, 448), (449, 512), (513, 1024)):
        assert any(lo <= k <= hi for k in kps), (lo, hi)
    bs = {c["batch"] for c in g}
    assert 1 in bs and dg.PIPELINE_MIN - 1 in bs and dg.PIPELINE_MIN in bs
    assert dg.CMAT == 8192 and dg.PB == 32 and dg.KBLK == 64 and dg.Q1R_MAXK == 16 and dg.MAX_LAUNCH_BATCH == 32768


def _check(out_p, out_rankA, out_rankJ2, out_code, out_jpvtJ2, J, rx, A, cx, kind, tag):
    ref = go.gn_subproblem(J, rx, A, cx)
    m, n = J.shape
    assert (out_rankA, out_code) == (ref.rankA, ref.code), tag
    assert np.all(np.isfinite(out_p)), tag
    if kind != "graded":
        assert out_rankJ2 == ref.rankJ2, tag
    tol = 1e-5 if kind == "graded" else (1e-8 if kind != "full" else 1e-10)
    if kind == "rankdefA" and m < n - ref.rankA:
        # a basic solution in a rounding-determined null-space basis: compare the residuals (tests/probes/fuzz_gpu.py)
        rj, ra = np.linalg.norm(J @ out_p + rx), np.linalg.norm(A @ out_p + cx)
        rj0, ra0 = np.linalg.norm(J @ ref.p + rx), np.linalg.norm(A @ ref.p + cx)
        assert abs(rj - rj0) <= 1e-8 * max(1.0, rj0) and abs(ra - ra0) <= 1e-8 * max(1.0, ra0), tag
    else:
        assert rel(out_p, ref.p) <= tol, (tag, rel(out_p, ref.p))
    if kind == "full" and out_jpvtJ2 is not None:
        r = ref.rankJ2
        assert np.array_equal(np.asarray(out_jpvtJ2)[:r], ref.jpvtJ2[:r]), tag


@pytest.mark.gpu
def test_every_grid_case_against_the_oracle_and_its_route():
    import torch
    from enlsip_gn import GNSolver
    from enlsip_gn import workload as wl
    from enlsip_gn.api import route_names
    assert route_names() == dg.header_route_names()
    seen = set()
    cases = dg.grid()
    for ci, c in enumerate(cases):
        batch, m, n, t, kind = c["batch"], c["m"], c["n"], c["t"], c["kind"]
        tag = (ci, batch, m, n, t, kind)
        want = dg.expected_route(batch, m, n, t, kind == "rankdefA")
        s = GNSolver(device=0)                     # fresh handle: no hints, no plan of an earlier shape
        try:
            if batch == 1:
                J, rx, A, cx = GENS[kind](52000 + ci, m, n, t)
                out = s.solve(J, rx, A, cx)
                got = s.route()
                _check(out.p, out.rankA, out.rankJ2, out.code, out.jpvtJ2, J, rx, A, cx, kind, tag)
            elif batch <= 8 or kind != "full":
                probs = [GENS[kind](52000 + 31 * ci + k, m, n, t) for k in range(batch)]
                p, b, d, infos, jA, jL, jJ = s.solve_batched(np.stack([np.ascontiguousarray(P[0].T) for P in probs]),
                                                             np.stack([P[1] for P in probs]),
                                                             np.stack([np.ascontiguousarray(P[2]) for P in probs]) if t else None,
                                                             np.stack([P[3] for P in probs]) if t else None)
                got = s.route()
                for k, (J, rx, A, cx) in enumerate(probs):
                    n2 = n - infos[k][0]
                    _check(p[k], infos[k][0], infos[k][1], infos[k][2], jJ[k][:n2], J, rx, A, cx, kind, tag + (k,))
            else:
                J, rx, At, cx = wl.make_batch(52000 + ci, batch, m, n, t, "cuda:0")
                p = torch.zeros((batch, n), dtype=torch.float64, device="cuda:0")
                jJ = torch.zeros((batch, n), dtype=torch.int64, device="cuda:0")
                info = torch.zeros((batch, 6), dtype=torch.int64, device="cuda:0")
                torch.cuda.synchronize()
                s.solve_batched_dev(batch, m, n, t, J.data_ptr(), m, m * n, rx.data_ptr(), At.data_ptr() if t else 0, max(n, 1), n * t,
                                    cx.data_ptr() if t else 0, dp=p.data_ptr(), dinfo=info.data_ptr(), djJ=jJ.data_ptr())
                got = s.route()
                ph, ih, jh = p.cpu().numpy(), info.cpu().numpy(), jJ.cpu().numpy()
                assert np.all(np.isfinite(ph)), tag
                for k in sorted({0, batch // 2, batch // 2 + 1, batch - 1}):       # both pipeline halves
                    Jh = J[k].cpu().numpy().T
                    Ah = At[k].cpu().numpy().reshape(t, n) if t else np.zeros((0, n))
                    ch = cx[k].cpu().numpy() if t else np.zeros(0)
                    n2 = n - int(ih[k][0])
                    _check(ph[k], int(ih[k][0]), int(ih[k][1]), int(ih[k][2]), jh[k][:n2], Jh, rx[k].cpu().numpy(), Ah, ch, kind, tag + (k,))
                del J, rx, At, cx, p, jJ, info
        finally:
            s.close()
        assert want <= got, (tag, sorted(want - got), sorted(got))
        seen |= got
    missing = set(dg.header_route_names()) - seen - set(dg.COVERED_ELSEWHERE)
    assert not missing, sorted(missing)
