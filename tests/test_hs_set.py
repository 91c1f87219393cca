"""Hock-Schittkowski least-squares problems (tests/hs_problems.py; data and solutions from the published collection, the
reference itself ships only HS65) through the restated outer iteration: the oracle backend must reach the published solutions,
and — on a GPU — the HIP library plugged in at the update_working_set seam must follow the oracle run iteration for iteration
(same working sets, ranks, codes, objective values and step lengths)."""
import math

import numpy as np
import pytest

import hs_problems as hp

# hp.NO_KNOWN_ANSWER: runs that end away from the published point (e.g. hs27 in the algorithm's "more than five Newton steps"
# exit -9): kept for the iteration-for-iteration comparison (they walk the Newton and failure branches)
CONVERGING = [k for k in hp.PROBLEMS if k not in hp.NO_KNOWN_ANSWER]
_cache = {}


def _oracle_run(name):
    from oracle import enlsip_outer as eo
    if name not in _cache:
        _cache[name] = hp.run(name, eo.OracleBackend())
    return _cache[name]


@pytest.mark.parametrize("name", CONVERGING)
def test_hs_oracle_backend_reaches_the_published_solution(name):
    P = hp.PROBLEMS[name]()
    res = _oracle_run(name)
    assert math.isfinite(res.f)
    assert abs(res.f - P["f_star"]) <= 1e-7 * max(1.0, abs(P["f_star"]))
    x_tol = hp.X_TOL.get(name, 2e-6)                                       # solutions published to 7 digits
    assert np.abs(res.x - P["x_star"]).max() <= x_tol * max(1.0, np.abs(P["x_star"]).max())
    if "eq" in P:
        assert np.abs(P["eq"](res.x)).max() <= 1e-7


class _SwapCycle(Exception):
    pass


@pytest.mark.parametrize("name", list(hp.CYCLING))
def test_hs20_hs23_spin_in_the_swap_loop_of_evaluate_violated_constraints(name, monkeypatch):
    """HS20 and HS23 are not in the iteration-for-iteration set, and this is why — read from the reference's code, since the
    reference cannot be run here.  `evaluate_violated_constraints` (src/enlsip_functions.jl:608-650) walks the inactive list with
    an index `i`; at capacity (`W.t >= min(l, n)`: both problems have more violated constraints than unknowns) it swaps the
    least violated active inequality out (`remove_constraint!(W, worst_k)`, :636) and then adds `W.inactive[i]` (:642).  But
    `remove_constraint!` (src/structures.jl:234-249) SORTS the inactive list after appending the constraint it removed: when that
    constraint's number is smaller than the candidate's, it lands AT position `i`, `add_constraint!(W, i)` puts it straight back,
    the candidate moves to position `i` again, and the loop meets the very state it started from — `i` never advances, there is no
    iteration bound inside the loop, and `time_limit` is only looked at between outer iterations (:2752).  The Julia reference
    would therefore spin here for ever (no status, no exception); the literal restatement (oracle/enlsip_outer.py) does the same,
    which this test proves by catching the first repeated (active, inactive, i) state inside ONE call.  Whatever backend solves
    the subproblem is irrelevant: the loop only reads constraint values."""
    from oracle import enlsip_outer as eo
    orig = eo.evaluate_violated_constraints
    calls = {"n": 0}

    def watched(cx, W, index_alpha_upp, n, ws_mod):
        seen = set()
        calls["n"] += 1

        class Proxy:
            Constraint = getattr(ws_mod, "Constraint", None)

            @staticmethod
            def remove_constraint(Wx, sidx):
                ws_mod.remove_constraint(Wx, sidx)

            @staticmethod
            def add_constraint(Wx, sidx):
                state = (tuple(int(v) for v in Wx.active), tuple(int(v) for v in Wx.inactive), int(Wx.t), int(sidx))
                if state in seen:
                    raise _SwapCycle(state)
                seen.add(state)
                ws_mod.add_constraint(Wx, sidx)

        return orig(cx, W, index_alpha_upp, n, Proxy)

    monkeypatch.setattr(eo, "evaluate_violated_constraints", watched)
    P = dict(hp.CYCLING[name]())
    kw = dict(P.pop("kw"))
    P.pop("x_star"); P.pop("f_star")
    with pytest.raises(_SwapCycle) as exc:
        eo.solve(P.pop("r"), P.pop("jac_r"), P.pop("n"), P.pop("m"), P.pop("x0"), backend=eo.OracleBackend(), time_limit=30.0, **P, **kw)
    active, inactive, t, i = exc.value.args[0]
    n_unknowns = 2
    # the repeated state sits between the removal and the re-insertion: one below capacity (min(l, n) = n = 2 here)
    assert t == n_unknowns - 1 and calls["n"] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(hp.PROBLEMS))
def test_hs_hip_backend_iteration_for_iteration(name):
    """Every iteration whose direction is above rounding level must agree: working set size, ranks, code, objective, step length.
    Two things are noise in BOTH runs and are not compared: (1) at the converged point the direction is rounding
    (|p| <= 1e-8 max(1, |x|)), so the step length, a degenerate multiplier's sign (working set) and which convergence criteria
    fire (the exit sub-code) are decided by the last bits; (2) after the first Newton iteration (code 2: finite-difference
    Hessians, hs27 only) differences of 1e-12 grow by orders of magnitude per step — there the sequence of codes, working sets and
    ranks is still compared, the values only loosely."""
    from enlsip_gn import GNSolver
    from hip_backend import HipBackend
    ref = _oracle_run(name)
    s = GNSolver(device=0)
    backend = HipBackend(s)
    res = hp.run(name, backend)
    s.close()
    assert res.iterations == ref.iterations
    if any(b["code"] == 2 for b in ref.trace):      # Newton iterations ran: their direction came from enlsip_gn_newton_direction
        assert getattr(backend, "newton_calls", 0) >= 1
    xs = max(1.0, float(np.abs(ref.x).max()))
    newton = False
    last_noise = False
    compared = 0
    for a, b in zip(res.trace, ref.trace):
        last_noise = max(a["pnorm"], b["pnorm"]) <= 1e-8 * xs
        newton = newton or b["code"] == 2
        if not newton:
            assert abs(a["f"] - b["f"]) <= 1e-8 * max(1.0, abs(b["f"]))
        if last_noise:
            continue
        assert (a["code"], a["t"], a["rankA"], a["rankJ2"]) == (b["code"], b["t"], b["rankA"], b["rankJ2"])
        if not newton:
            assert abs(a["alpha"] - b["alpha"]) <= 1e-6 * max(1.0, abs(b["alpha"]))
            assert abs(a["pnorm"] - b["pnorm"]) <= 1e-6 * max(1.0, abs(b["pnorm"]))
            compared += 1
    assert compared >= 1
    if last_noise:
        assert (res.exit_code > 0) == (ref.exit_code > 0) or abs(res.f - ref.f) <= 1e-9 * max(1.0, abs(ref.f))
    else:
        assert res.exit_code == ref.exit_code
    if not newton:
        assert np.abs(res.x - ref.x).max() <= 1e-7 * xs
