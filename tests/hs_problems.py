"""Least-squares problems of the Hock-Schittkowski collection (W. Hock, K. Schittkowski, "Test Examples for Nonlinear
Programming Codes", LNEMS 187, 1981) as data for oracle/enlsip_outer.py: residuals r with f = ||r||^2 (the objective Enlsip
reports), constraints in Enlsip's sense (equalities = 0, inequalities >= 0, bounds), hand-derived Jacobians, the collection's
starting points and its published solutions.  The reference ships only HS65 (test/problems/HS65.jl, tests/hs65.py here); these
widen the "iteration for iteration" comparison of the HIP backend against the oracle backend.  Test infrastructure."""
import math

import numpy as np

S2 = math.sqrt(2.0)


def _p(**kw):
    kw.setdefault("kw", {})
    return kw


def hs6():
    return _p(n=2, m=1, x0=np.array([-1.2, 1.0]),
              r=lambda x: np.array([1.0 - x[0]]), jac_r=lambda x: np.array([[-1.0, 0.0]]),
              eq=lambda x: np.array([10.0 * (x[1] - x[0] ** 2)]), jac_eq=lambda x: np.array([[-20.0 * x[0], 10.0]]), nb_eq=1,
              x_star=np.array([1.0, 1.0]), f_star=0.0)


def hs27():
    return _p(n=3, m=2, x0=np.array([2.0, 2.0, 2.0]),
              r=lambda x: np.array([0.1 * (x[0] - 1.0), x[1] - x[0] ** 2]),
              jac_r=lambda x: np.array([[0.1, 0.0, 0.0], [-2.0 * x[0], 1.0, 0.0]]),
              eq=lambda x: np.array([x[0] + x[2] ** 2 + 1.0]), jac_eq=lambda x: np.array([[1.0, 0.0, 2.0 * x[2]]]), nb_eq=1,
              x_star=np.array([-1.0, 1.0, 0.0]), f_star=0.04)


def hs28():
    return _p(n=3, m=2, x0=np.array([-4.0, 1.0, 1.0]),
              r=lambda x: np.array([x[0] + x[1], x[1] + x[2]]),
              jac_r=lambda x: np.array([[1.0, 1.0, 0.0], [0.0, 1.0, 1.0]]),
              eq=lambda x: np.array([x[0] + 2 * x[1] + 3 * x[2] - 1.0]), jac_eq=lambda x: np.array([[1.0, 2.0, 3.0]]), nb_eq=1,
              x_star=np.array([0.5, -0.5, 0.5]), f_star=0.0)


def hs32():
    return _p(n=3, m=2, x0=np.array([0.1, 0.7, 0.2]),
              r=lambda x: np.array([x[0] + 3 * x[1] + x[2], 2.0 * (x[0] - x[1])]),
              jac_r=lambda x: np.array([[1.0, 3.0, 1.0], [2.0, -2.0, 0.0]]),
              eq=lambda x: np.array([1.0 - x[0] - x[1] - x[2]]), jac_eq=lambda x: np.array([[-1.0, -1.0, -1.0]]), nb_eq=1,
              ineq=lambda x: np.array([6 * x[1] + 4 * x[2] - x[0] ** 3 - 3.0]),
              jac_ineq=lambda x: np.array([[-3 * x[0] ** 2, 6.0, 4.0]]), nb_ineq=1,
              x_low=[0.0, 0.0, 0.0], x_upp=[np.inf] * 3,
              x_star=np.array([0.0, 0.0, 1.0]), f_star=1.0)


def hs42():
    return _p(n=4, m=4, x0=np.ones(4),
              r=lambda x: x - np.array([1.0, 2.0, 3.0, 4.0]), jac_r=lambda x: np.eye(4),
              eq=lambda x: np.array([x[0] - 2.0, x[2] ** 2 + x[3] ** 2 - 2.0]),
              jac_eq=lambda x: np.array([[1.0, 0, 0, 0], [0, 0, 2 * x[2], 2 * x[3]]]), nb_eq=2,
              x_star=np.array([2.0, 2.0, 0.6 * S2, 0.8 * S2]), f_star=28.0 - 10.0 * S2)


def hs48():
    return _p(n=5, m=3, x0=np.array([3.0, 5.0, -3.0, 2.0, -2.0]),
              r=lambda x: np.array([x[0] - 1.0, x[1] - x[2], x[3] - x[4]]),
              jac_r=lambda x: np.array([[1.0, 0, 0, 0, 0], [0, 1.0, -1.0, 0, 0], [0, 0, 0, 1.0, -1.0]]),
              eq=lambda x: np.array([x.sum() - 5.0, x[2] - 2 * (x[3] + x[4]) + 3.0]),
              jac_eq=lambda x: np.array([[1.0, 1, 1, 1, 1], [0, 0, 1.0, -2.0, -2.0]]), nb_eq=2,
              x_star=np.ones(5), f_star=0.0)


def hs50():
    return _p(n=5, m=4, x0=np.array([35.0, -31.0, 11.0, 5.0, -5.0]),
              r=lambda x: np.array([x[0] - x[1], x[1] - x[2], (x[2] - x[3]) ** 2, x[3] - x[4]]),
              jac_r=lambda x: np.array([[1.0, -1.0, 0, 0, 0], [0, 1.0, -1.0, 0, 0],
                                        [0, 0, 2 * (x[2] - x[3]), -2 * (x[2] - x[3]), 0], [0, 0, 0, 1.0, -1.0]]),
              eq=lambda x: np.array([x[0] + 2 * x[1] + 3 * x[2] - 6.0, x[1] + 2 * x[2] + 3 * x[3] - 6.0, x[2] + 2 * x[3] + 3 * x[4] - 6.0]),
              jac_eq=lambda x: np.array([[1.0, 2, 3, 0, 0], [0, 1.0, 2, 3, 0], [0, 0, 1.0, 2, 3]]), nb_eq=3,
              x_star=np.ones(5), f_star=0.0)


def hs51():
    return _p(n=5, m=4, x0=np.array([2.5, 0.5, 2.0, -1.0, 0.5]),
              r=lambda x: np.array([x[0] - x[1], x[1] + x[2] - 2.0, x[3] - 1.0, x[4] - 1.0]),
              jac_r=lambda x: np.array([[1.0, -1.0, 0, 0, 0], [0, 1.0, 1.0, 0, 0], [0, 0, 0, 1.0, 0], [0, 0, 0, 0, 1.0]]),
              eq=lambda x: np.array([x[0] + 3 * x[1] - 4.0, x[2] + x[3] - 2 * x[4], x[1] - x[4]]),
              jac_eq=lambda x: np.array([[1.0, 3, 0, 0, 0], [0, 0, 1.0, 1.0, -2.0], [0, 1.0, 0, 0, -1.0]]), nb_eq=3,
              x_star=np.ones(5), f_star=0.0)


def _hs52_base():
    return dict(n=5, m=4, x0=np.full(5, 2.0),
                r=lambda x: np.array([4 * x[0] - x[1], x[1] + x[2] - 2.0, x[3] - 1.0, x[4] - 1.0]),
                jac_r=lambda x: np.array([[4.0, -1.0, 0, 0, 0], [0, 1.0, 1.0, 0, 0], [0, 0, 0, 1.0, 0], [0, 0, 0, 0, 1.0]]),
                eq=lambda x: np.array([x[0] + 3 * x[1], x[2] + x[3] - 2 * x[4], x[1] - x[4]]),
                jac_eq=lambda x: np.array([[1.0, 3, 0, 0, 0], [0, 0, 1.0, 1.0, -2.0], [0, 1.0, 0, 0, -1.0]]), nb_eq=3)


def hs52():
    return _p(**_hs52_base(), x_star=np.array([-33.0, 11.0, 180.0, -158.0, 11.0]) / 349.0, f_star=1859.0 / 349.0)


def hs53():
    B = _hs52_base()
    B["r"] = lambda x: np.array([x[0] - x[1], x[1] + x[2] - 2.0, x[3] - 1.0, x[4] - 1.0])
    B["jac_r"] = lambda x: np.array([[1.0, -1.0, 0, 0, 0], [0, 1.0, 1.0, 0, 0], [0, 0, 0, 1.0, 0], [0, 0, 0, 0, 1.0]])
    return _p(**B, x_low=[-10.0] * 5, x_upp=[10.0] * 5,
              x_star=np.array([-33.0, 11.0, 27.0, -5.0, 11.0]) / 43.0, f_star=176.0 / 43.0)


def hs60():
    return _p(n=3, m=3, x0=np.full(3, 2.0),
              r=lambda x: np.array([x[0] - 1.0, x[0] - x[1], (x[1] - x[2]) ** 2]),
              jac_r=lambda x: np.array([[1.0, 0, 0], [1.0, -1.0, 0], [0, 2 * (x[1] - x[2]), -2 * (x[1] - x[2])]]),
              eq=lambda x: np.array([x[0] * (1 + x[1] ** 2) + x[2] ** 4 - 4.0 - 3.0 * S2]),
              jac_eq=lambda x: np.array([[1 + x[1] ** 2, 2 * x[0] * x[1], 4 * x[2] ** 3]]), nb_eq=1,
              x_low=[-10.0] * 3, x_upp=[10.0] * 3,
              x_star=np.array([1.104859024, 1.196674194, 1.535262257]), f_star=0.03256820025)


def hs77():
    return _p(n=5, m=5, x0=np.full(5, 2.0),
              r=lambda x: np.array([x[0] - 1.0, x[0] - x[1], x[2] - 1.0, (x[3] - 1.0) ** 2, (x[4] - 1.0) ** 3]),
              jac_r=lambda x: np.array([[1.0, 0, 0, 0, 0], [1.0, -1.0, 0, 0, 0], [0, 0, 1.0, 0, 0],
                                        [0, 0, 0, 2 * (x[3] - 1.0), 0], [0, 0, 0, 0, 3 * (x[4] - 1.0) ** 2]]),
              eq=lambda x: np.array([x[0] ** 2 * x[3] + math.sin(x[3] - x[4]) - 2.0 * S2,
                                     x[1] + x[2] ** 4 * x[3] ** 2 - 8.0 - S2]),
              jac_eq=lambda x: np.array([[2 * x[0] * x[3], 0, 0, x[0] ** 2 + math.cos(x[3] - x[4]), -math.cos(x[3] - x[4])],
                                         [0, 1.0, 4 * x[2] ** 3 * x[3] ** 2, 2 * x[2] ** 4 * x[3], 0]]), nb_eq=2,
              x_star=np.array([1.166172, 1.182111, 1.380257, 1.506036, 0.6109203]), f_star=0.24150513)


def hs79():
    return _p(n=5, m=5, x0=np.full(5, 2.0),
              r=lambda x: np.array([x[0] - 1.0, x[0] - x[1], x[1] - x[2], (x[2] - x[3]) ** 2, (x[3] - x[4]) ** 2]),
              jac_r=lambda x: np.array([[1.0, 0, 0, 0, 0], [1.0, -1.0, 0, 0, 0], [0, 1.0, -1.0, 0, 0],
                                        [0, 0, 2 * (x[2] - x[3]), -2 * (x[2] - x[3]), 0],
                                        [0, 0, 0, 2 * (x[3] - x[4]), -2 * (x[3] - x[4])]]),
              eq=lambda x: np.array([x[0] + x[1] ** 2 + x[2] ** 3 - 2.0 - 3.0 * S2, x[1] - x[2] ** 2 + x[3] + 2.0 - 2.0 * S2,
                                     x[0] * x[4] - 2.0]),
              jac_eq=lambda x: np.array([[1.0, 2 * x[1], 3 * x[2] ** 2, 0, 0], [0, 1.0, -2 * x[2], 1.0, 0], [x[4], 0, 0, 0, x[0]]]),
              nb_eq=3,
              x_star=np.array([1.191127, 1.362603, 1.472818, 1.635017, 1.679081]), f_star=0.0787768209)


def _rosen_r(x):
    return np.array([10.0 * (x[1] - x[0] ** 2), 1.0 - x[0]])


def _rosen_j(x):
    return np.array([[-20.0 * x[0], 10.0], [-1.0, 0.0]])


def hs1():
    return _p(n=2, m=2, x0=np.array([-2.0, 1.0]), r=_rosen_r, jac_r=_rosen_j, x_low=[-np.inf, -1.5], x_upp=[np.inf, np.inf],
              x_star=np.array([1.0, 1.0]), f_star=0.0)


def hs2():
    a = math.sqrt(598.0 / 1200.0)
    b = 400.0 * a ** 3
    return _p(n=2, m=2, x0=np.array([-2.0, 1.0]), r=_rosen_r, jac_r=_rosen_j, x_low=[-np.inf, 1.5], x_upp=[np.inf, np.inf],
              x_star=np.array([2.0 * a * math.cos(math.acos(1.0 / b) / 3.0), 1.5]), f_star=0.0504261879)


def hs14():
    s7 = math.sqrt(7.0)
    return _p(n=2, m=2, x0=np.array([2.0, 2.0]),
              r=lambda x: np.array([x[0] - 2.0, x[1] - 1.0]), jac_r=lambda x: np.eye(2),
              eq=lambda x: np.array([x[0] - 2 * x[1] + 1.0]), jac_eq=lambda x: np.array([[1.0, -2.0]]), nb_eq=1,
              ineq=lambda x: np.array([-0.25 * x[0] ** 2 - x[1] ** 2 + 1.0]), jac_ineq=lambda x: np.array([[-0.5 * x[0], -2 * x[1]]]),
              nb_ineq=1, x_star=np.array([0.5 * (s7 - 1.0), 0.25 * (s7 + 1.0)]), f_star=9.0 - 2.875 * s7)


def hs15():
    return _p(n=2, m=2, x0=np.array([-2.0, 1.0]), r=_rosen_r, jac_r=_rosen_j,
              ineq=lambda x: np.array([x[0] * x[1] - 1.0, x[0] + x[1] ** 2]),
              jac_ineq=lambda x: np.array([[x[1], x[0]], [1.0, 2 * x[1]]]), nb_ineq=2,
              x_low=[-np.inf, -np.inf], x_upp=[0.5, np.inf], x_star=np.array([0.5, 2.0]), f_star=306.5)


def hs16():
    return _p(n=2, m=2, x0=np.array([-2.0, 1.0]), r=_rosen_r, jac_r=_rosen_j,
              ineq=lambda x: np.array([x[0] + x[1] ** 2, x[0] ** 2 + x[1]]),
              jac_ineq=lambda x: np.array([[1.0, 2 * x[1]], [2 * x[0], 1.0]]), nb_ineq=2,
              x_low=[-2.0, -np.inf], x_upp=[0.5, 1.0], x_star=np.array([0.5, 0.25]), f_star=0.25)


def hs17():
    return _p(n=2, m=2, x0=np.array([-2.0, 1.0]), r=_rosen_r, jac_r=_rosen_j,
              ineq=lambda x: np.array([x[1] ** 2 - x[0], x[0] ** 2 - x[1]]),
              jac_ineq=lambda x: np.array([[-1.0, 2 * x[1]], [2 * x[0], -1.0]]), nb_ineq=2,
              x_low=[-2.0, -np.inf], x_upp=[0.5, 1.0], x_star=np.array([0.0, 0.0]), f_star=1.0)


def hs18():
    return _p(n=2, m=2, x0=np.array([2.0, 2.0]),
              r=lambda x: np.array([0.1 * x[0], x[1]]), jac_r=lambda x: np.array([[0.1, 0.0], [0.0, 1.0]]),
              ineq=lambda x: np.array([x[0] * x[1] - 25.0, x[0] ** 2 + x[1] ** 2 - 25.0]),
              jac_ineq=lambda x: np.array([[x[1], x[0]], [2 * x[0], 2 * x[1]]]), nb_ineq=2,
              x_low=[2.0, 0.0], x_upp=[50.0, 50.0], x_star=np.array([math.sqrt(250.0), math.sqrt(2.5)]), f_star=5.0)


def hs20():
    return _p(n=2, m=2, x0=np.array([-2.0, 1.0]), r=_rosen_r, jac_r=_rosen_j,
              ineq=lambda x: np.array([x[0] + x[1] ** 2, x[0] ** 2 + x[1], x[0] ** 2 + x[1] ** 2 - 1.0]),
              jac_ineq=lambda x: np.array([[1.0, 2 * x[1]], [2 * x[0], 1.0], [2 * x[0], 2 * x[1]]]), nb_ineq=3,
              x_low=[-0.5, -np.inf], x_upp=[0.5, np.inf], x_star=np.array([0.5, math.sqrt(3.0) / 2.0]),
              f_star=81.5 - 25.0 * math.sqrt(3.0))


def hs22():
    return _p(n=2, m=2, x0=np.array([2.0, 2.0]),
              r=lambda x: np.array([x[0] - 2.0, x[1] - 1.0]), jac_r=lambda x: np.eye(2),
              ineq=lambda x: np.array([-x[0] - x[1] + 2.0, -x[0] ** 2 + x[1]]),
              jac_ineq=lambda x: np.array([[-1.0, -1.0], [-2 * x[0], 1.0]]), nb_ineq=2,
              x_star=np.array([1.0, 1.0]), f_star=1.0)


def hs23():
    return _p(n=2, m=2, x0=np.array([3.0, 1.0]),
              r=lambda x: np.array([x[0], x[1]]), jac_r=lambda x: np.eye(2),
              ineq=lambda x: np.array([x[0] + x[1] - 1.0, x[0] ** 2 + x[1] ** 2 - 1.0, 9 * x[0] ** 2 + x[1] ** 2 - 9.0,
                                       x[0] ** 2 - x[1], x[1] ** 2 - x[0]]),
              jac_ineq=lambda x: np.array([[1.0, 1.0], [2 * x[0], 2 * x[1]], [18 * x[0], 2 * x[1]], [2 * x[0], -1.0],
                                           [-1.0, 2 * x[1]]]), nb_ineq=5,
              x_low=[-50.0, -50.0], x_upp=[50.0, 50.0], x_star=np.array([1.0, 1.0]), f_star=2.0)


def hs30():
    return _p(n=3, m=3, x0=np.ones(3), r=lambda x: x.copy(), jac_r=lambda x: np.eye(3),
              ineq=lambda x: np.array([x[0] ** 2 + x[1] ** 2 - 1.0]), jac_ineq=lambda x: np.array([[2 * x[0], 2 * x[1], 0.0]]),
              nb_ineq=1, x_low=[1.0, -10.0, -10.0], x_upp=[10.0, 10.0, 10.0], x_star=np.array([1.0, 0.0, 0.0]), f_star=1.0)


def hs31():
    return _p(n=3, m=3, x0=np.ones(3), r=lambda x: np.array([3 * x[0], x[1], 3 * x[2]]),
              jac_r=lambda x: np.diag([3.0, 1.0, 3.0]),
              ineq=lambda x: np.array([x[0] * x[1] - 1.0]), jac_ineq=lambda x: np.array([[x[1], x[0], 0.0]]), nb_ineq=1,
              x_low=[-10.0, 1.0, -10.0], x_upp=[10.0, 10.0, 1.0],
              x_star=np.array([1.0 / math.sqrt(3.0), math.sqrt(3.0), 0.0]), f_star=6.0)


PROBLEMS = {"hs1": hs1, "hs2": hs2, "hs14": hs14, "hs15": hs15, "hs16": hs16, "hs17": hs17, "hs18": hs18,
            "hs22": hs22, "hs30": hs30, "hs31": hs31, "hs6": hs6, "hs27": hs27, "hs28": hs28, "hs32": hs32, "hs42": hs42, "hs48": hs48, "hs50": hs50, "hs51": hs51,
            "hs52": hs52, "hs53": hs53, "hs60": hs60, "hs77": hs77, "hs79": hs79}


# Not run iteration for iteration: with more violated constraints than unknowns the swap loop of evaluate_violated_constraints
# at capacity (src/enlsip_functions.jl:621-647) exchanges the same two constraints forever — remove_constraint! re-sorts the
# inactive list (src/structures.jl:240), so the constraint just swapped out lands at the loop's index and add_constraint! puts
# it straight back.  The Julia reference would spin there without a status; tests/test_hs_set.py proves the repeated state on
# the literal restatement (test_hs20_hs23_spin_in_the_swap_loop_of_evaluate_violated_constraints).
CYCLING = {"hs20": hs20, "hs23": hs23}

# Runs that end away from the collection's solution with this algorithm (whatever the subproblem backend): HS2 in its second,
# local minimiser (f = 4.9412), HS16 at the iteration limit, HS18 in a failed line search, HS27 in the Newton-step limit.
# They are compared iteration for iteration only.
NO_KNOWN_ANSWER = ("hs2", "hs16", "hs18", "hs27")
X_TOL = {"hs30": 1e-4}       # x2 -> 0 only linearly (the solver stops at its x tolerance)


def run(name, backend, **over):
    from oracle import enlsip_outer as eo
    P = dict(PROBLEMS[name]())
    kw = dict(P.pop("kw"))
    kw.update(over)
    P.pop("x_star"); P.pop("f_star")
    kw.setdefault("time_limit", 60.0)
    return eo.solve(P.pop("r"), P.pop("jac_r"), P.pop("n"), P.pop("m"), P.pop("x0"), backend=backend, **P, **kw)
