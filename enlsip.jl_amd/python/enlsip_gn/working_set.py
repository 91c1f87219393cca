"""Host mirror of the reference's ``update_working_set`` (src/enlsip_functions.jl:686-795) over the
HIP library: same arguments, same mutations of ``W`` / ``C`` / ``iter_k``, same return triple
``(F_A, F_L11, F_J2)`` — with every ``qr(·, ColumnNorm())`` + ``gn_search_direction`` group replaced
by ONE device solve and the factor objects replaced by device-backed views (``FactorView``).

The multiplier estimates run on the device, on the resident ``F_A``, ``J`` and
``J*F_A.Q`` (``GNSolver.first_lagrange`` / ``second_lagrange``); the host restatements over the accessors
(``.R``, ``.p``, ``Qt_mul``) are kept below as the form the Julia glue would use without those entry points.
The deletion test and the sequencing of 1-3 subproblem solves per call (including quirk Q1: a first-order
deletion is always undone, App. C) are reproduced on the host.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from ._lib import FACTOR_A, FACTOR_L11, FACTOR_J2
from .api import GNSolver

EPS = float(np.finfo(np.float64).eps)
SQRT_EPS = math.sqrt(EPS)


# ---- records mirrored from src/structures.jl -------------------------------------------------------
@dataclass
class WorkingSet:                      # src/structures.jl:209-225
    q: int
    t: int
    l: int
    active: np.ndarray
    inactive: np.ndarray

    @staticmethod
    def create(q: int, l: int) -> "WorkingSet":
        active = np.zeros(l, dtype=np.int64)
        inactive = np.zeros(l - q, dtype=np.int64)
        active[:q] = np.arange(1, q + 1)
        inactive[:] = np.arange(q + 1, l + 1)
        return WorkingSet(q, q, l, active, inactive)

    def remove_constraint(self, s: int) -> None:      # :230-245, s 1-based
        l, t = self.l, self.t
        self.inactive[l - t] = self.active[s - 1]
        self.inactive[: l - t + 1] = np.sort(self.inactive[: l - t + 1])
        for i in range(s, t):
            self.active[i - 1] = self.active[i]
        self.active[t - 1] = 0
        self.t -= 1

    def add_constraint(self, s: int) -> None:         # :250-263, s 1-based index into inactive
        l, t = self.l, self.t
        self.active[t] = self.inactive[s - 1]
        self.active[: t + 1] = np.sort(self.active[: t + 1])
        for i in range(s, l - t):
            self.inactive[i - 1] = self.inactive[i]
        self.inactive[l - t - 1] = 0
        self.t += 1


@dataclass
class Constraint:                      # src/structures.jl:145-150
    cx: np.ndarray
    A: np.ndarray
    scaling: bool
    diag_scale: np.ndarray


@dataclass
class IterationRecord:                 # hot-path fields of src/structures.jl:63-91
    rankA: int = 0
    rankJ2: int = 0
    dimA: int = 0
    dimJ2: int = 0
    b_gn: np.ndarray = field(default_factory=lambda: np.zeros(0))
    d_gn: np.ndarray = field(default_factory=lambda: np.zeros(0))
    grad_res: float = 0.0
    lam: np.ndarray = field(default_factory=lambda: np.zeros(0))
    delete: bool = False
    index_del: int = 0


def pseudo_rank(diag_T, eps_rank: float) -> int:      # src/enlsip_functions.jl:17-31
    l = len(diag_T)
    if l == 0 or abs(diag_T[0]) < eps_rank:
        return 0
    tol = abs(diag_T[0]) * math.sqrt(float(l)) * eps_rank
    r = 1
    while r < l and abs(diag_T[r - 1]) > tol:
        r += 1
    return r - (0 if (r == l and abs(diag_T[r - 1]) > tol) else 1)


def _invperm(p1):
    p0 = np.asarray(p1, dtype=np.int64) - 1
    inv = np.empty_like(p0)
    inv[p0] = np.arange(p0.size)
    return inv


def _tri_solve(T, b, lower):
    """``Triangular(T) \\ b`` on the host for the t x t multiplier systems (t is small)."""
    b = np.asarray(b, dtype=np.float64)
    if b.size == 0:
        return b.copy()
    if np.any(np.diag(T) == 0.0):
        raise np.linalg.LinAlgError("singular triangular system")
    from scipy.linalg import solve_triangular
    return solve_triangular(T, b, lower=lower)


# ---- consumers of the resident factors (src/enlsip_functions.jl:461-537, 574-603) -------------------
def first_lagrange_mult_estimate(A, grad_fx, cx, scaling_done, diag_scale, F_A, it: IterationRecord, eps_rank):
    t, n = A.shape
    R = F_A.R
    p = F_A.p
    inv_p = _invperm(p)
    prankA = pseudo_rank(np.diag(R[: min(R.shape), : min(R.shape)]), eps_rank)
    b = F_A.Qt_mul(grad_fx)
    v = np.zeros(t)
    v[:prankA] = _tri_solve(R[:prankA, :prankA], b[:prankA], lower=False)
    lam_ls = v[inv_p]
    it.grad_res = float(np.linalg.norm(b[prankA:n])) if n > prankA else 0.0
    b2 = -np.asarray(cx, dtype=np.float64)[p - 1]
    y = np.zeros(t)
    y[:prankA] = _tri_solve(R.T[:prankA, :prankA], b2[:prankA], lower=True)
    u = np.zeros(t)
    u[:prankA] = _tri_solve(R[:prankA, :prankA], y[:prankA], lower=False)
    lam = lam_ls + u[inv_p]
    return lam * diag_scale if scaling_done else lam


def second_lagrange_mult_estimate(solver: GNSolver, J, F_A, rx, p_gn, t, scaling, diag_scale, eps_rank=SQRT_EPS):
    m, n = J.shape
    R = F_A.R
    prankA = pseudo_rank(np.diag(R[: min(R.shape), : min(R.shape)]), eps_rank)
    J1 = solver.JQ1(m, n)[:, :t]                       # J * F_A.Q served from the device (:526, quirk Q6)
    b = J1.T @ (rx + J @ p_gn)
    v = np.zeros(t)
    v[:prankA] = _tri_solve(R[:prankA, :prankA], b[:prankA], lower=False)
    lam = v[_invperm(F_A.p)]
    return lam * diag_scale if scaling else lam


def check_constraint_deletion(q, A, lam, scaling, diag_scale, grad_res) -> int:
    t = A.shape[0]
    delta = 10.0
    lam_max = 1.0 if len(lam) == 0 else float(np.max(np.abs(lam)))
    sq_rel = SQRT_EPS * lam_max
    s = 0
    if t > q:
        e = sq_rel
        for i in range(q + 1, t + 1):
            row_i = (1.0 / diag_scale[i - 1]) if scaling else diag_scale[i - 1]
            if row_i * lam[i - 1] <= sq_rel and row_i * lam[i - 1] <= e:
                e = row_i * lam[i - 1]
                s = i
        if grad_res > -e * delta:
            s = 0
    return s


# ---- update_working_set (src/enlsip_functions.jl:686-795) -------------------------------------------
def update_working_set(solver: GNSolver, W: WorkingSet, rx, A, C: Constraint, grad_fx, J, p_gn, it: IterationRecord,
                       eps_rank: float, on_solve=None):
    """Returns (F_A, F_L11, F_J2) as device-backed views, valid until the next solve on ``solver``."""
    m, n = J.shape

    def _direction(factored=False):
        # F_A = qr(C.A'); rankA; F_L11 = qr(F_A.R'); p_gn, F_J2 = gn_search_direction(...)  -> one device solve;
        # factored: the working set is the one just factored at :700, the solve goes on with that factorisation (:768-771)
        out = solver.solve_factored(J, rx, C.A.shape[0], eps_rank) if factored else solver.solve(J, rx, C.A, C.cx, eps_rank)
        p_gn[:] = out.p
        it.rankA, it.rankJ2 = out.rankA, out.rankJ2
        it.dimA, it.dimJ2 = out.rankA, out.rankJ2
        it.b_gn, it.d_gn = out.b, out.d
        if on_solve is not None:
            on_solve()
        return out.rankA

    def _views():
        return solver.factor(FACTOR_A), solver.factor(FACTOR_L11), solver.factor(FACTOR_J2)

    def _second_order(rankA, lam):
        if not (W.t != rankA or it.rankJ2 != min(m, n - rankA)):       # :745 / :773
            # second_lagrange_mult_estimate! on the device: J1 = (J*F_A.Q)[:, 1:t] is resident (:526, quirk Q6)
            lam[:] = solver.second_lagrange(W.t, p_gn, C.diag_scale if C.scaling else None)
            s2 = check_constraint_deletion(W.q, C.A, lam, C.scaling, C.diag_scale, 0.0)
            if s2 != 0:
                index_s2 = int(W.active[s2 - 1])
                lam = np.delete(lam, s2 - 1)
                C.diag_scale = np.delete(C.diag_scale, s2 - 1)
                C.cx = np.delete(C.cx, s2 - 1)
                W.remove_constraint(s2)
                it.delete = True
                it.index_del = index_s2
                C.A = np.delete(C.A, s2 - 1, axis=0)
                rankA = _direction()
        return rankA, lam

    # F_A = qr(C.A', ColumnNorm()) (:700) and first_lagrange_mult_estimate! (:704) on the device, before any solve:
    # the constraint stage alone leaves F_A resident for the estimate
    solver.factor_constraints(J.shape[0], C.A, C.cx, eps_rank)
    lam, it.grad_res = solver.first_lagrange(W.t, grad_fx, C.diag_scale if C.scaling else None, eps_rank)
    s = check_constraint_deletion(W.q, C.A, lam, C.scaling, C.diag_scale, it.grad_res)
    if s != 0:                                                         # :706-765
        cx_s = C.cx[s - 1]
        A_s = C.A[s - 1, :].copy()
        lam_s = lam[s - 1]
        diag_scale_s = C.diag_scale[s - 1]
        index_s = int(W.active[s - 1])
        lam = np.delete(lam, s - 1)
        C.cx = np.delete(C.cx, s - 1)
        C.diag_scale = np.delete(C.diag_scale, s - 1)
        W.remove_constraint(s)
        it.delete = True
        it.index_del = index_s
        C.A = np.delete(C.A, s - 1, axis=0)
        rankA = _direction()
        As_p = 0.0 if rankA <= W.t else float(A_s @ p_gn)              # :728 (quirk Q1)
        feasible = (As_p >= -cx_s) and (As_p > 0)
        if not feasible:
            C.cx = np.insert(C.cx, s - 1, cx_s)
            lam = np.insert(lam, s - 1, lam_s)
            C.diag_scale = np.insert(C.diag_scale, s - 1, diag_scale_s)
            s_inact = int(np.where(W.inactive == index_s)[0][0]) + 1
            W.add_constraint(s_inact)
            it.index_del = 0
            it.delete = False
            rows = A[W.active[: W.t] - 1, :]
            C.A = rows * C.diag_scale[:, None] if C.scaling else rows.copy()
            rankA = _direction()
            rankA, lam = _second_order(rankA, lam)
    else:                                                              # :767-791
        rankA = _direction(factored=True)
        rankA, lam = _second_order(rankA, lam)
    it.lam = lam
    return _views()
