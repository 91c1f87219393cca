"""Subproblem backend for oracle/enlsip_outer.py that routes the hot path through the HIP library
(enlsip_gn.working_set.update_working_set + the factor accessors + enlsip_gn_resolve).  Test
infrastructure: it lets the restated outer iteration drive the PRODUCT exactly where the Julia glue
would plug it in."""
import numpy as np

from enlsip_gn import working_set as pws


class _WSAdapter:
    Constraint = pws.Constraint

    @staticmethod
    def WorkingSet(q, t, l, active, inactive):
        return pws.WorkingSet(q, t, l, active, inactive)

    @staticmethod
    def remove_constraint(W, s):
        W.remove_constraint(s)

    @staticmethod
    def add_constraint(W, s):
        W.add_constraint(s)


class HipBackend:
    ws = _WSAdapter

    def __init__(self, solver):
        self.solver = solver
        self._shape = None

    def evaluate_scaling(self, C):          # src/structures.jl:160-178
        t = C.A.shape[0]
        C.diag_scale = np.zeros(t)
        for i in range(t):
            row = float(np.linalg.norm(C.A[i, :]))
            C.diag_scale[i] = row
            if C.scaling:
                if abs(row) < np.finfo(float).eps:
                    row = 1.0
                C.A[i, :] = C.A[i, :] / row
                C.cx[i] = C.cx[i] / row
                C.diag_scale[i] = 1.0 / row

    def update_working_set(self, W, rx, A, C, grad_fx, J, p_gn, it, eps_rank, on_solve):
        rec = pws.IterationRecord()
        rec.delete, rec.index_del = it.delete, it.index_del
        F = pws.update_working_set(self.solver, W, rx, A, C, grad_fx, J, p_gn, rec, eps_rank, on_solve=on_solve)
        it.rankA, it.rankJ2, it.dimA, it.dimJ2 = rec.rankA, rec.rankJ2, rec.dimA, rec.dimJ2
        it.b_gn, it.d_gn, it.grad_res, it.lam = rec.b_gn, rec.d_gn, rec.grad_res, rec.lam
        it.delete, it.index_del = rec.delete, rec.index_del
        self._shape = (J.shape[0], J.shape[1], C.A.shape[0])
        return _Views(F)

    def rmul_Q1(self, J, F_A):
        m, n = J.shape
        return self.solver.JQ1(m, n)

    def newton_direction(self, Gam):
        # newton_search_direction after its Hessian sums (src/enlsip_functions.jl:398-421) on the resident factors
        self.newton_calls = getattr(self, "newton_calls", 0) + 1
        return self.solver.newton_direction(Gam)

    def sub_search_direction(self, J1, rx, cx, F_A, F_L11, F_J2, n, t, rankA, dimA, dimJ2, code):
        m = len(rx)
        return self.solver.resolve(m, n, t, dimA, dimJ2, code)


class _View:
    """FactorView + the `.P` accessor the outer loop uses (F.P[p[i], i] = 1)."""

    def __init__(self, fv):
        self._fv = fv

    def __getattr__(self, name):
        return getattr(self._fv, name)

    @property
    def P(self):
        p = self._fv.p
        n = len(p)
        P = np.zeros((n, n))
        P[p - 1, np.arange(n)] = 1.0
        return P


def _Views(F):
    return tuple(_View(f) for f in F)
