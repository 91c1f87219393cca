"""End-to-end pin of the oracle (and, on a GPU, of the product) on the ONE known answer the
reference publishes: HS65, docs/src/tutorial.md:126-128 and the two statements at :201-211
("solution differs by more than sqrt(eps)", "objective within sqrt(eps)")."""
import math

import numpy as np
import pytest

import hs65

SQRT_EPS = math.sqrt(np.finfo(float).eps)


def _check(res):
    assert res.exit_code > 0                                  # a convergence code (src/cnls_model.jl:166-178)
    assert abs(res.f - hs65.KNOWN_F) < SQRT_EPS                # tutorial.md:207-211: true in the reference
    err = np.abs(res.x - hs65.KNOWN_X).max()
    assert err < 1e-5                                          # "relatively close" ...
    assert not (err < SQRT_EPS)                                # ... tutorial.md:201-205: false in the reference
    # feasibility of the returned point
    assert hs65.c(res.x)[0] > -1e-8 and np.all(res.x <= np.array(hs65.X_UPP) + 1e-12)


def test_hs65_oracle_backend():
    from oracle import enlsip_outer as eo
    res = hs65.run(eo.OracleBackend())
    _check(res)
    assert res.trace[0]["rankA"] == 2 and res.trace[0]["t"] == 3      # rank-deficient start, SURVEY App. C Q10
    assert 1 <= res.nb_subproblem_solves <= 3 * (res.iterations + 1)  # 1-3 solves per update_working_set call


@pytest.mark.gpu
def test_hs65_hip_backend_iteration_for_iteration():
    """Same outer loop, hot path through libenlsip_gn.so: same iterates as with the LAPACK oracle."""
    from enlsip_gn import GNSolver
    from hip_backend import HipBackend
    from oracle import enlsip_outer as eo
    ref = hs65.run(eo.OracleBackend())
    s = GNSolver(device=0)
    res = hs65.run(HipBackend(s))
    s.close()
    _check(res)
    assert res.exit_code == ref.exit_code and res.iterations == ref.iterations
    for a, b in zip(res.trace, ref.trace):
        assert (a["code"], a["t"], a["rankA"], a["rankJ2"]) == (b["code"], b["t"], b["rankA"], b["rankJ2"])
        assert abs(a["alpha"] - b["alpha"]) <= 1e-6 * max(1.0, abs(b["alpha"]))
        assert abs(a["f"] - b["f"]) <= 1e-9 * max(1.0, abs(b["f"]))
    assert np.abs(res.x - ref.x).max() < 1e-9
