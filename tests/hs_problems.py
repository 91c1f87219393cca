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


PROBLEMS = {"hs6": hs6, "hs27": hs27, "hs28": hs28, "hs32": hs32, "hs42": hs42, "hs48": hs48, "hs50": hs50, "hs51": hs51,
            "hs52": hs52, "hs53": hs53, "hs60": hs60, "hs77": hs77, "hs79": hs79}


def run(name, backend, **over):
    from oracle import enlsip_outer as eo
    P = dict(PROBLEMS[name]())
    kw = dict(P.pop("kw"))
    kw.update(over)
    P.pop("x_star"); P.pop("f_star")
    return eo.solve(P.pop("r"), P.pop("jac_r"), P.pop("n"), P.pop("m"), P.pop("x0"), backend=backend, **P, **kw)
