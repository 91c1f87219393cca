"""The reference's own test problems beyond HS65 (test/problems/chained_rosenbrock.jl, chained_wood.jl, osborne2.jl) through
the restated outer iteration: with the LAPACK oracle as subproblem backend (CPU), and — on a GPU — with the HIP library
plugged in at the update_working_set seam, iteration for iteration against the oracle run.  What the reference itself asserts
on these problems: a status from its table, a finite objective, `time_limit = -1.0` => time_limit_exceeded (-11)
(chained_rosenbrock.jl:60-73, chained_wood.jl:36-38, osborne2.jl:104-112)."""
import math

import numpy as np
import pytest

import ref_problems as rp

STATUS_CODES = {1, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11}      # keys of dict_status_codes, src/cnls_model.jl:166-186


def _status_key(exit_code):
    return 1 if exit_code > 0 else exit_code


PROBLEMS = {"chained_rosenbrock_1000": lambda: rp.chained_rosenbrock(1000), "chained_wood_20": lambda: rp.chained_wood(20),
            "osborne2": rp.osborne2}
_ref_cache = {}
_env_cache = {}
# problems whose oracle run takes seconds: their step-length tolerance is MEASURED (tests/perturbation_envelope.py); the
# 1000-variable chained Rosenbrock run takes most of a minute per repetition and keeps the fixed bound
ENVELOPE_PROBLEMS = ("chained_wood_20", "osborne2")
ALPHA_FLOOR = 1e-7          # relative step-length agreement asked for where rounding-level perturbations do not move it at all
ENVELOPE_FACTOR = 10.0      # x the spread seen under +-1 ulp perturbations of the subproblem's inputs (six draws)
ALPHA_CAP = 5e-3            # ... but never looser than the bound of the flat end game (rounds 1-3), unless the 1-ulp envelope ITSELF
ENVELOPE_OVER_CAP = 2.0     # is wider than that: then twice the envelope (Osborne 2's last iterations: 1e-2 measured => 2e-2, not 1e-1)


def alpha_tolerance(env_i):
    """Step-length tolerance of iteration i from its measured 1-ulp envelope: ALPHA_FLOOR where rounding does not move the step,
    ENVELOPE_FACTOR x the envelope above that, capped at ALPHA_CAP — a regression of the fast reflector / norm-downdate forms in
    the end game must not hide behind a tenfold envelope."""
    return max(ALPHA_FLOOR, min(ENVELOPE_FACTOR * env_i, max(ALPHA_CAP, ENVELOPE_OVER_CAP * env_i)))


def _alpha_envelope(name):
    """Per-iteration spread of the step length under rounding-level perturbations of the oracle's own inputs."""
    import perturbation_envelope as pe
    if name not in _env_cache:
        ref = _oracle_run(name)
        _env_cache[name] = pe.alpha_envelope(lambda backend: rp.run(PROBLEMS[name](), backend), ref)
    return _env_cache[name]


def _oracle_run(name):
    from oracle import enlsip_outer as eo
    if name not in _ref_cache:
        _ref_cache[name] = rp.run(PROBLEMS[name](), eo.OracleBackend())
    return _ref_cache[name]


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_reference_problem_oracle_backend(name):
    res = _oracle_run(name)
    assert _status_key(res.exit_code) in STATUS_CODES and res.exit_code > 0          # these three converge
    assert math.isfinite(res.f) and res.x.shape == (PROBLEMS[name]()["n"],)
    P = PROBLEMS[name]()
    if "eq" in P:
        assert np.abs(P["eq"](res.x)).max() < 1e-6                                   # equality constraints hold at the solution
    if "x_low" in P:
        assert np.all(res.x >= np.array(P["x_low"]) - 1e-10) and np.all(res.x <= np.array(P["x_upp"]) + 1e-10)


@pytest.mark.parametrize("name", ENVELOPE_PROBLEMS)
def test_step_length_envelope_under_ulp_perturbations(name):
    """The yardstick of the iteration-for-iteration comparison: perturbing J and r of every subproblem by one unit in the last
    place leaves every discrete outcome of the run alone (codes, ranks, working-set sizes, iteration count) and moves the early
    step lengths by rounding noise only — but the line search of Osborne 2's end game amplifies it to 1e-5 .. 1e-2 (which is why a
    fixed 1e-6 bound once vetoed a legitimate change of summation order)."""
    ref = _oracle_run(name)
    env, agree_upto = _alpha_envelope(name)
    assert agree_upto >= len(ref.trace), "a 1-ulp perturbation changed a discrete decision of the run"
    assert env[: min(6, len(env))].max() <= 1e-12
    if name == "osborne2":
        assert env.max() > 1e-6          # the knife edge is real: the old fixed bound sat below the 1-ulp envelope


def test_time_limit_status():
    from oracle import enlsip_outer as eo
    res = rp.run(rp.chained_rosenbrock(200), eo.OracleBackend(), time_limit=-1.0)
    assert res.exit_code == -11                                                      # chained_rosenbrock.jl:71-73


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PROBLEMS))
def test_reference_problem_hip_backend_iteration_for_iteration(name):
    from enlsip_gn import GNSolver
    from hip_backend import HipBackend
    ref = _oracle_run(name)
    s = GNSolver(device=0)
    res = rp.run(PROBLEMS[name](), HipBackend(s))
    s.close()
    # discrete outcomes: exact
    assert res.exit_code == ref.exit_code and res.iterations == ref.iterations
    env = _alpha_envelope(name)[0] if name in ENVELOPE_PROBLEMS else None
    f_prev = None
    for i, (a, b) in enumerate(zip(res.trace, ref.trace)):
        assert (a["code"], a["t"], a["rankA"], a["rankJ2"]) == (b["code"], b["t"], b["rankA"], b["rankJ2"])
        assert abs(a["f"] - b["f"]) <= 1e-8 * max(1.0, abs(b["f"]))
        # step length: within ENVELOPE_FACTOR x what +-1 ulp on the subproblem's inputs does to the ORACLE's own step length at
        # this iteration (measured, tests/perturbation_envelope.py), capped (alpha_tolerance), and within ALPHA_FLOOR where that
        # is nothing.  Osborne 2's end game: 1e-5 .. 1e-2 from iteration 12 on.  Without a measured envelope (chained Rosenbrock): 1e-6, and 5e-3 once the
        # objective has stopped moving (|f_k - f_{k-1}| < 1e-8 f), where the step length is no longer determined.
        if env is not None:
            tol = alpha_tolerance(env[i])
        else:
            flat = f_prev is not None and abs(b["f"] - f_prev) <= 1e-8 * max(1.0, abs(b["f"]))
            tol = 5e-3 if flat else 1e-6
        assert abs(a["alpha"] - b["alpha"]) <= tol * max(1.0, abs(b["alpha"])), (i, a["alpha"], b["alpha"], tol)
        f_prev = b["f"]
    assert np.abs(res.x - ref.x).max() <= 1e-7 * max(1.0, np.abs(ref.x).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hs65", "chained_wood_20", "osborne2"])
def test_fast_reflector_scalars_against_the_ieee_forms(name):
    """The approximation itself, pinned: the product build takes dlarfg's scalars and dlaqp2's norm downdate from
    rcp / rsq with Newton steps (make_reflector, norm_downdate); enlsip.jl_amd/lib/libenlsip_gn_ieee.so is the SAME source built
    with -DENLSIP_FAST_REFLECTOR=0 (IEEE sqrt and divisions).  Whole outer runs through both builds (one process each): identical
    discrete traces — exit code, iteration count, codes, working-set sizes, ranks — objectives to 1e-8 and step lengths within the
    capped 1-ulp envelope of the iteration (HS65, which has no measured envelope: 1e-6), i.e. the bounds of the comparison with the
    oracle run (measured on MI355X: HS65 1.6e-9 in its last iterations, chained Wood 5e-10 relative in f)."""
    import json, os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    ieee = root / "enlsip.jl_amd" / "lib" / "libenlsip_gn_ieee.so"
    if not ieee.exists():
        pytest.skip("the IEEE-scalar partner build (enlsip.jl_amd/lib/libenlsip_gn_ieee.so) is missing: python __graft_entry__.py builds it")
    runs = []
    for lib in (None, ieee):
        env = dict(os.environ)
        env.pop("ENLSIP_GN_LIB", None)
        if lib is not None:
            env["ENLSIP_GN_LIB"] = str(lib)
        out = subprocess.run([sys.executable, str(root / "tests" / "outer_trace_worker.py"), name], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    fast, exact = runs
    assert (fast["exit_code"], fast["iterations"]) == (exact["exit_code"], exact["iterations"])
    env = _alpha_envelope(name)[0] if name in ENVELOPE_PROBLEMS else None
    for i, (a, b) in enumerate(zip(fast["trace"], exact["trace"])):
        assert (a["code"], a["t"], a["rankA"], a["rankJ2"]) == (b["code"], b["t"], b["rankA"], b["rankJ2"]), i
        assert abs(a["f"] - b["f"]) <= 1e-8 * max(1.0, abs(b["f"])), i          # the bound of the comparison with the oracle run
        tol = alpha_tolerance(env[i]) if env is not None else 1e-6
        assert abs(a["alpha"] - b["alpha"]) <= tol * max(1.0, abs(b["alpha"])), (i, a["alpha"], b["alpha"], tol)
