"""The reference's other test problems (test/problems/chained_rosenbrock.jl, chained_wood.jl, osborne2.jl) as data for
oracle/enlsip_outer.py: residual / constraint functions with hand-derived Jacobians (the reference differentiates the last two
with ForwardDiff), starting points and solver tolerances exactly as in those test files.  Test infrastructure."""
import math

import numpy as np


# ---- Chained Rosenbrock (test/problems/chained_rosenbrock.jl:3-53): n parameters, 2(n-1) residuals, n-2 equalities ----
def chained_rosenbrock(n=1000):
    m = 2 * (n - 1)

    def r(x):
        out = np.empty(m)
        out[: n - 1] = 10.0 * (x[: n - 1] ** 2 - x[1:])
        out[n - 1:] = x[: n - 1] - 1.0
        return out

    def jac_r(x):
        J = np.zeros((m, n))
        i = np.arange(n - 1)
        J[i, i] = 20.0 * x[: n - 1]
        J[i, i + 1] = -10.0
        J[n - 1 + i, i] = 1.0
        return J

    def c(x):
        a, b, d = x[:-2], x[1:-1], x[2:]
        return 3 * b ** 3 + 2 * d - 5 + np.sin(b - d) * np.sin(b + d) + 4 * b - a * np.exp(a - b) - 3

    def jac_c(x):
        A = np.zeros((n - 2, n))
        k = np.arange(n - 2)
        a, b, d = x[:-2], x[1:-1], x[2:]
        A[k, k] = -(a + 1) * np.exp(a - b)
        A[k, k + 1] = 9 * b ** 2 + np.cos(b - d) * np.sin(b + d) + np.sin(b - d) * np.cos(b + d) + 4 + a * np.exp(a - b)
        A[k, k + 2] = 2 - np.cos(b - d) * np.sin(b + d) + np.sin(b - d) * np.cos(b + d)
        return A

    x0 = np.array([-1.2 if (i % 2) == 1 else 1.0 for i in range(1, n + 1)])
    return dict(r=r, jac_r=jac_r, n=n, m=m, x0=x0, eq=c, jac_eq=jac_c, nb_eq=n - 2, kw={})


# ---- Chained Wood (test/problems/chained_wood.jl:3-38): n = 20, m = 6 (n/2 - 1), n - 7 equalities ------------------------
def chained_wood(n=20):
    N = n // 2 - 1
    m = 6 * N
    s = math.sqrt(10.0)

    def r(x):
        i = np.arange(1, N + 1)
        o, e, o2, e2 = x[2 * i - 2], x[2 * i - 1], x[2 * i], x[2 * i + 1]      # x[2i-1], x[2i], x[2i+1], x[2i+2] (1-based)
        return np.concatenate([10 * (o ** 2 - e), o - 1, 3 * s * (o2 ** 2 - e2), o2 - 1, s * (e + e2 - 2), (e - e2) / s])

    def jac_r(x):
        J = np.zeros((m, n))
        for i in range(1, N + 1):
            a, b, c_, d = 2 * i - 2, 2 * i - 1, 2 * i, 2 * i + 1
            J[i - 1, a] = 20 * x[a]; J[i - 1, b] = -10.0
            J[N + i - 1, a] = 1.0
            J[2 * N + i - 1, c_] = 6 * s * x[c_]; J[2 * N + i - 1, d] = -3 * s
            J[3 * N + i - 1, c_] = 1.0
            J[4 * N + i - 1, b] = s; J[4 * N + i - 1, d] = s
            J[5 * N + i - 1, b] = 1 / s; J[5 * N + i - 1, d] = -1 / s
        return J

    def c(x):
        out = np.empty(n - 7)
        for k in range(1, n - 6):
            lo = max(k - 5, 1)
            out[k - 1] = (2 + 5 * x[k + 4] ** 2) * x[k + 4] + 1 + sum(x[i - 1] * (1 + x[i - 1]) for i in range(lo, k + 2))
        return out

    def jac_c(x):
        A = np.zeros((n - 7, n))
        for k in range(1, n - 6):
            A[k - 1, k + 4] += 2 + 15 * x[k + 4] ** 2
            for i in range(max(k - 5, 1), k + 2):
                A[k - 1, i - 1] += 1 + 2 * x[i - 1]
        return A

    x0 = np.array([-2.0 if (i % 2) == 1 else 1.0 for i in range(1, n + 1)])
    return dict(r=r, jac_r=jac_r, n=n, m=m, x0=x0, eq=c, jac_eq=jac_c, nb_eq=n - 7,
                kw=dict(rel_tol=1e-5, x_tol=1e-3, c_tol=1e-6))


# ---- Osborne 2 (test/problems/osborne2.jl:3-110): 65 data points, 11 parameters, box constraints ---------------------------
_OSB_Y = [1.366, 1.191, 1.112, 1.013, 0.991, 0.885, 0.831, 0.847, 0.786, 0.725, 0.746, 0.679, 0.608, 0.655, 0.616, 0.606, 0.602,
          0.626, 0.651, 0.724, 0.649, 0.649, 0.694, 0.644, 0.624, 0.661, 0.612, 0.558, 0.533, 0.495, 0.5, 0.423, 0.395, 0.375,
          0.538, 0.522, 0.506, 0.49, 0.478, 0.467, 0.457, 0.457, 0.457, 0.457, 0.457, 0.457, 0.457, 0.457, 0.457, 0.457, 0.457,
          0.431, 0.431, 0.424, 0.42, 0.414, 0.411, 0.406, 0.406, 0.406, 0.406, 0.406, 0.406, 0.406, 0.406]


def osborne2():
    m, n = 65, 11
    t = 0.1 * np.arange(m)
    y = np.array(_OSB_Y)

    def parts(x):
        e1 = np.exp(-x[4] * t)
        e2 = np.exp(-x[5] * (t - x[8]) ** 2)
        e3 = np.exp(-x[6] * (t - x[9]) ** 2)
        e4 = np.exp(-x[7] * (t - x[10]) ** 2)
        return e1, e2, e3, e4

    def r(x):
        e1, e2, e3, e4 = parts(x)
        return y - (x[0] * e1 + x[1] * e2 + x[2] * e3 + x[3] * e4)

    def jac_r(x):
        e1, e2, e3, e4 = parts(x)
        J = np.zeros((m, n))
        J[:, 0], J[:, 1], J[:, 2], J[:, 3] = -e1, -e2, -e3, -e4
        J[:, 4] = x[0] * t * e1
        J[:, 5] = x[1] * (t - x[8]) ** 2 * e2
        J[:, 6] = x[2] * (t - x[9]) ** 2 * e3
        J[:, 7] = x[3] * (t - x[10]) ** 2 * e4
        J[:, 8] = -x[1] * e2 * 2 * x[5] * (t - x[8])
        J[:, 9] = -x[2] * e3 * 2 * x[6] * (t - x[9])
        J[:, 10] = -x[3] * e4 * 2 * x[7] * (t - x[10])
        return J

    low = [1.31, 0.4314, 0.6336, 0.5, 0.5, 0.6, 1.0, 4.0, 2.0, 4.5689, 5.0]
    upp = [1.4, 0.8, 1.0, 1.0, 1.0, 3.0, 5.0, 7.0, 2.5, 5.0, 6.0]
    x0 = np.array([1.3344098963722457, 0.5572842161127423, 0.6757364753061974, 0.8291980513226953, 0.9233565833014519,
                   0.9588470511477797, 1.9610314699563896, 4.055321823656234, 2.048625993866472, 4.60296578920499,
                   5.95212572157736])
    return dict(r=r, jac_r=jac_r, n=n, m=m, x0=x0, x_low=low, x_upp=upp, kw={})


def run(problem, backend, **over):
    from oracle import enlsip_outer as eo
    P = dict(problem)
    kw = dict(P.pop("kw"))
    kw.update(over)
    return eo.solve(P.pop("r"), P.pop("jac_r"), P.pop("n"), P.pop("m"), P.pop("x0"), backend=backend, **P, **kw)
