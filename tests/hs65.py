"""HS65 (reference test/problems/HS65.jl:7-17) and the known answer of docs/src/tutorial.md:126-128."""
import numpy as np

KNOWN_X = np.array([3.650461821, 3.65046168, 4.6204170507])
KNOWN_F = 0.9535288567
X0 = np.array([-5.0, 5.0, 0.0])
X_LOW = [-4.5, -4.5, -5.0]
X_UPP = [4.5, 4.5, 5.0]


def r(x):
    return np.array([x[0] - x[1], (x[0] + x[1] - 10.0) / 3.0, x[2] - 5.0])


def jac_r(x):
    return np.array([[1.0, -1.0, 0.0], [1.0 / 3.0, 1.0 / 3.0, 0.0], [0.0, 0.0, 1.0]])


def c(x):
    return np.array([48.0 - x[0] ** 2 - x[1] ** 2 - x[2] ** 2])


def jac_c(x):
    return np.array([[-2 * x[0], -2 * x[1], -2 * x[2]]])


def run(backend, **kw):
    from oracle import enlsip_outer as eo
    return eo.solve(r, jac_r, 3, 3, X0, ineq=c, jac_ineq=jac_c, nb_ineq=1, x_low=X_LOW, x_upp=X_UPP,
                    backend=backend, **kw)
