"""CPU oracle for the Gauss-Newton search-direction subproblem of Enlsip.jl.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / reported baseline.  The shipped path is the HIP
library behind ``include/enlsip_gn.h`` and fails loudly if that library is missing.

What this file is
-----------------
A line-for-line *restatement* (not a copy: the reference is Julia, this is NumPy) of the
reference hot path, calling the same LAPACK routines Julia's LinearAlgebra dispatches to
(``dgeqp3`` for ``qr(·, ColumnNorm())``, ``dormqr`` for ``Q'*v`` / ``M*Q``, ``dtrtrs`` for
``Triangular \\ v``) through ``scipy.linalg.lapack``:

* ``pseudo_rank``              <- src/enlsip_functions.jl:17-31
* ``sub_search_direction``     <- src/enlsip_functions.jl:116-153
* ``gn_search_direction``      <- src/enlsip_functions.jl:206-234
* ``first_lagrange_mult_estimate`` / ``second_lagrange_mult_estimate``
                               <- src/enlsip_functions.jl:461-508, 514-537
* ``check_constraint_deletion`` <- src/enlsip_functions.jl:574-603
* ``update_working_set``       <- src/enlsip_functions.jl:686-795
* ``evaluate_scaling``         <- src/structures.jl:160-178
* ``gn_subproblem``            <- the QR / rank / direction lines of update_working_set's
                                  ``s == 0`` branch (src/enlsip_functions.jl:700, 768-771):
                                  one "solve" in the sense of SURVEY.md §8(d).

Pinning status
--------------
The arithmetic lives in a third-party dependency that is absent from /root/reference:
Julia stdlib LinearAlgebra -> libblastrampoline 5.8.0+1 -> OpenBLAS_jll 0.3.23+4
(only pin: docs/Manifest.toml:3, :296-299, :439-442; the package itself only requires
julia ^1.8).  Julia is not installed in the build container, so the reference cannot be
run, and the reference's own tests hold NO numeric assertion on this path.
The only known answer the reference publishes is the HS65 solution / objective in
docs/src/tutorial.md:126-128; ``oracle/enlsip_outer.py`` (host restatement of the outer
loop) reproduces it with this file as its subproblem (tests/test_hs65_known_answer.py:
objective within 1e-10, solution within 5e-7 — and, like the reference, NOT within
sqrt(eps), docs/src/tutorial.md:201-211).  Status:

    pinned end to end on the reference's HS65 known answer;
    no reference vectors exist at the subproblem boundary itself (unpinned there).

Independent of the reference: the same outer loop reaches the published solutions of nineteen further least-squares problems
of the Hock-Schittkowski collection (tests/hs_problems.py, tests/test_hs_set.py) — a check of the restatement against the
literature, not a reference vector.

Conventions: column-major semantics are irrelevant in NumPy, but every permutation
returned is the **1-based LAPACK jpvt** exactly as Julia's ``F.p``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np
from scipy.linalg import lapack

EPS = np.finfo(np.float64).eps
SQRT_EPS = math.sqrt(EPS)  # ε_rank = √eps, src/solver.jl:81


# --------------------------------------------------------------------------------------
# pseudo_rank  — src/enlsip_functions.jl:17-31
# --------------------------------------------------------------------------------------
def pseudo_rank(diag_T: np.ndarray, eps_rank: float) -> int:
    diag_T = np.asarray(diag_T, dtype=np.float64)
    if diag_T.size == 0 or abs(diag_T[0]) < eps_rank:
        return 0
    l_diag = diag_T.size
    tol = abs(diag_T[0]) * math.sqrt(float(l_diag)) * eps_rank
    r = 1  # 1-based like the reference
    while r < l_diag and abs(diag_T[r - 1]) > tol:
        r += 1
    return r - (0 if (r == l_diag and abs(diag_T[r - 1]) > tol) else 1)


# --------------------------------------------------------------------------------------
# QRPivoted  — Julia's LinearAlgebra.QRPivoted as used by the reference
# --------------------------------------------------------------------------------------
@dataclass
class QRPivoted:
    """``qr(A, ColumnNorm())``: compact dgeqp3 factors, tau, 1-based jpvt."""

    factors: np.ndarray  # rows x cols, Fortran order
    tau: np.ndarray      # min(rows, cols)
    jpvt: np.ndarray     # cols, 1-based (int64)

    @property
    def rows(self) -> int:
        return self.factors.shape[0]

    @property
    def cols(self) -> int:
        return self.factors.shape[1]

    @property
    def k(self) -> int:
        return min(self.factors.shape)

    @property
    def R(self) -> np.ndarray:
        """``F.R`` = triu(factors[1:min,:])  (k x cols)."""
        return np.triu(self.factors[: self.k, :])

    @property
    def p(self) -> np.ndarray:
        return self.jpvt

    @property
    def P(self) -> np.ndarray:
        """``F.P`` with ``P[p[i], i] = 1``."""
        n = self.cols
        P = np.zeros((n, n))
        P[self.jpvt - 1, np.arange(n)] = 1.0
        return P

    def diagR(self) -> np.ndarray:
        return np.diag(self.factors[: self.k, : self.k]).copy()

    def _ormqr(self, side: str, trans: str, c: np.ndarray) -> np.ndarray:
        c = np.array(c, dtype=np.float64, order="F", copy=True)
        if self.k == 0 or c.size == 0:
            return c
        a = self.factors[:, : self.k]
        a = np.asfortranarray(a)
        _, work, info = lapack.dormqr(side, trans, a, self.tau, c, -1)
        lwork = int(work[0])
        cq, _, info = lapack.dormqr(side, trans, a, self.tau, c, lwork, overwrite_c=1)
        if info != 0:
            raise RuntimeError(f"dormqr info={info}")
        return cq

    def Qt_mul(self, v: np.ndarray) -> np.ndarray:
        """``F.Q' * v`` (v has ``rows`` entries; Q behaves as rows x rows)."""
        v = np.asarray(v, dtype=np.float64)
        vec = v.ndim == 1
        if v.size == 0:
            return v.copy()
        out = self._ormqr("L", "T", v.reshape(self.rows, -1))
        return out[:, 0].copy() if vec else out

    def Q_mul(self, v: np.ndarray) -> np.ndarray:
        """``F.Q * v``."""
        v = np.asarray(v, dtype=np.float64)
        vec = v.ndim == 1
        if v.size == 0:
            return v.copy()
        out = self._ormqr("L", "N", v.reshape(self.rows, -1))
        return out[:, 0].copy() if vec else out

    def rmul_Q(self, M: np.ndarray) -> np.ndarray:
        """``M * F.Q``  (dormqr('R','N') on copy(M))."""
        return self._ormqr("R", "N", M)


def qr_colnorm(A: np.ndarray) -> QRPivoted:
    """``qr(A, ColumnNorm())`` -> dgeqp3 on a copy, all columns free (jpvt = 0)."""
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    rows, cols = A.shape
    if rows == 0 or cols == 0:
        return QRPivoted(A, np.zeros(0), np.arange(1, cols + 1, dtype=np.int64))
    _, _, _, work, info = lapack.dgeqp3(A, lwork=-1)
    lwork = max(int(work[0]), 3 * (cols + 1))
    qr_, jpvt, tau, _, info = lapack.dgeqp3(A, lwork=lwork, overwrite_a=1)
    if info != 0:
        raise RuntimeError(f"dgeqp3 info={info}")
    return QRPivoted(qr_, tau, jpvt.astype(np.int64))


def _trtrs(T: np.ndarray, b: np.ndarray, lower: bool) -> np.ndarray:
    """``Triangular(T) \\ b`` via dtrtrs; raises on an exactly singular diagonal as Julia does."""
    b = np.asarray(b, dtype=np.float64)
    if b.size == 0:
        return b.copy()
    x, info = lapack.dtrtrs(np.asfortranarray(T), b.reshape(-1, 1).copy(order="F"),
                            lower=1 if lower else 0)
    if info != 0:
        raise np.linalg.LinAlgError(f"singular triangular system (dtrtrs info={info})")
    return x[:, 0].copy()


def invperm(p1: np.ndarray) -> np.ndarray:
    """Julia ``invperm`` on a 1-based permutation, returned 0-based for NumPy indexing:
    ``v[invperm(p)]`` in Julia == ``v[invperm(p1)]`` here."""
    p0 = np.asarray(p1, dtype=np.int64) - 1
    inv = np.empty_like(p0)
    inv[p0] = np.arange(p0.size)
    return inv


# --------------------------------------------------------------------------------------
# Iteration record (hot-path fields only) — src/structures.jl:63-91
# --------------------------------------------------------------------------------------
@dataclass
class IterationRecord:
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


# --------------------------------------------------------------------------------------
# sub_search_direction — src/enlsip_functions.jl:116-153
# --------------------------------------------------------------------------------------
def sub_search_direction(J1, rx, cx, F_A: QRPivoted, F_L11: QRPivoted, F_J2: QRPivoted,
                         n: int, t: int, rankA: int, dimA: int, dimJ2: int, code: int):
    rx = np.asarray(rx, dtype=np.float64)
    cx = np.asarray(cx, dtype=np.float64)
    if code == 1:                                                      # :131-137
        b = -cx[F_A.p - 1]
        p1 = _trtrs(F_A.R.T, b, lower=True)
        d_temp = -J1 @ p1 - rx
        d = F_J2.Qt_mul(d_temp)
        dp2 = _trtrs(F_J2.R[:dimJ2, :dimJ2], d[:dimJ2], lower=False)
        p2 = np.concatenate([dp2, np.zeros(n - t - dimJ2)])[invperm(F_J2.p)]
    elif code == -1:                                                   # :140-148
        b_buff = -cx[F_A.p - 1]
        b = F_L11.Qt_mul(b_buff)
        dp1 = _trtrs(F_L11.R[:dimA, :dimA], b[:dimA], lower=False)
        p1 = np.concatenate([dp1, np.zeros(t - dimA)])[invperm(F_L11.p)][:rankA]
        d_temp = -J1 @ p1 - rx
        d = F_J2.Qt_mul(d_temp)
        dp2 = _trtrs(F_J2.R[:dimJ2, :dimJ2], d[:dimJ2], lower=False)
        p2 = np.concatenate([dp2, np.zeros(n - rankA - dimJ2)])[invperm(F_J2.p)]
    else:
        raise ValueError("code must be 1 or -1")
    p = F_A.Q_mul(np.concatenate([p1, p2]))                            # :151
    return p, b, d


# --------------------------------------------------------------------------------------
# gn_search_direction — src/enlsip_functions.jl:206-234
# --------------------------------------------------------------------------------------
def gn_search_direction(J, rx, cx, F_A: QRPivoted, F_L11: QRPivoted, rankA: int, t: int,
                        eps_rank: float, current_iter: IterationRecord):
    code = 1 if rankA == t else -1                                     # :217
    n = J.shape[1]
    JQ1 = F_A.rmul_Q(J)                                                # :219
    J1, J2 = JQ1[:, :rankA], JQ1[:, rankA:]                            # :220
    F_J2 = qr_colnorm(J2)                                              # :223
    rankJ2 = pseudo_rank(F_J2.diagR(), eps_rank)                       # :224
    p_gn, b_gn, d_gn = sub_search_direction(J1, rx, cx, F_A, F_L11, F_J2, n, t, rankA,
                                            rankA, rankJ2, code)       # :225
    current_iter.rankA = rankA
    current_iter.rankJ2 = rankJ2
    current_iter.dimA = rankA
    current_iter.dimJ2 = rankJ2
    current_iter.b_gn = b_gn
    current_iter.d_gn = d_gn
    return p_gn, F_J2


# --------------------------------------------------------------------------------------
# One "solve" (SURVEY §8d): QR/rank/direction lines of update_working_set, s == 0 branch
# src/enlsip_functions.jl:700, 768-771
# --------------------------------------------------------------------------------------
@dataclass
class GNSolution:
    p: np.ndarray
    b: np.ndarray
    d: np.ndarray
    rankA: int
    rankJ2: int
    code: int
    jpvtA: np.ndarray
    jpvtL: np.ndarray
    jpvtJ2: np.ndarray
    F_A: QRPivoted
    F_L11: QRPivoted
    F_J2: QRPivoted


def gn_subproblem(J: np.ndarray, rx: np.ndarray, A_active: np.ndarray, cx_active: np.ndarray,
                  eps_rank: float = SQRT_EPS) -> GNSolution:
    """A_active is the t x n active-constraint Jacobian (``C.A``); the QR is of its transpose."""
    J = np.asarray(J, dtype=np.float64)
    m, n = J.shape
    A_active = np.asarray(A_active, dtype=np.float64).reshape(-1, n)
    t = A_active.shape[0]
    F_A = qr_colnorm(A_active.T)                                       # :700
    rankA = pseudo_rank(F_A.diagR(), eps_rank)                         # :768
    F_L11 = qr_colnorm(F_A.R.T)                                        # :769
    it = IterationRecord()
    p, F_J2 = gn_search_direction(J, rx, cx_active, F_A, F_L11, rankA, t, eps_rank, it)  # :771
    return GNSolution(p=p, b=it.b_gn, d=it.d_gn, rankA=rankA, rankJ2=it.rankJ2,
                      code=(1 if rankA == t else -1), jpvtA=F_A.jpvt, jpvtL=F_L11.jpvt,
                      jpvtJ2=F_J2.jpvt, F_A=F_A, F_L11=F_L11, F_J2=F_J2)


# --------------------------------------------------------------------------------------
# Constraint record + evaluate_scaling! — src/structures.jl:145-178
# --------------------------------------------------------------------------------------
@dataclass
class Constraint:
    cx: np.ndarray
    A: np.ndarray
    scaling: bool
    diag_scale: np.ndarray


def evaluate_scaling(C: Constraint) -> None:
    t = C.A.shape[0]
    C.diag_scale = np.zeros(t)
    for i in range(t):
        row_i = float(np.linalg.norm(C.A[i, :]))
        C.diag_scale[i] = row_i
        if C.scaling:
            if abs(row_i) < EPS:
                row_i = 1.0
            C.A[i, :] = C.A[i, :] / row_i
            C.cx[i] = C.cx[i] / row_i
            C.diag_scale[i] = 1.0 / row_i


# --------------------------------------------------------------------------------------
# Lagrange multiplier estimates — src/enlsip_functions.jl:461-508, 514-537
# --------------------------------------------------------------------------------------
def first_lagrange_mult_estimate(A, grad_fx, cx, scaling_done: bool, diag_scale, F: QRPivoted,
                                 it: IterationRecord, eps_rank: float) -> np.ndarray:
    t, n = A.shape
    v = np.zeros(t)
    inv_p = invperm(F.p)
    prankA = pseudo_rank(F.diagR(), eps_rank)
    b = F.Qt_mul(grad_fx)
    R = F.R
    v[:prankA] = _trtrs(R[:prankA, :prankA], b[:prankA], lower=False)
    lam_ls = v[inv_p]
    it.grad_res = float(np.linalg.norm(b[prankA:n])) if n > prankA else 0.0
    b2 = -np.asarray(cx, dtype=np.float64)[F.p - 1]
    y = np.zeros(t)
    y[:prankA] = _trtrs(R.T[:prankA, :prankA], b2[:prankA], lower=True)
    u = np.zeros(t)
    u[:prankA] = _trtrs(R[:prankA, :prankA], y[:prankA], lower=False)
    lam = lam_ls + u[inv_p]
    if scaling_done:
        lam = lam * diag_scale
    return lam


def second_lagrange_mult_estimate(J, F_A: QRPivoted, rx, p_gn, t: int, scaling: bool, diag_scale,
                                  eps_rank: float = SQRT_EPS) -> np.ndarray:
    prankA = pseudo_rank(F_A.diagR(), eps_rank)
    J1 = F_A.rmul_Q(J)[:, :t]                                          # :526 (recomputed, quirk Q6)
    b = J1.T @ (rx + J @ p_gn)
    v = np.zeros(t)
    v[:prankA] = _trtrs(F_A.R[:prankA, :prankA], b[:prankA], lower=False)
    lam = v[invperm(F_A.p)]
    if scaling:
        lam = lam * diag_scale
    return lam


# check_constraint_deletion — src/enlsip_functions.jl:574-603 (returns 1-based s, 0 = none)
def check_constraint_deletion(q: int, A, lam, scaling: bool, diag_scale, grad_res: float) -> int:
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


# --------------------------------------------------------------------------------------
# WorkingSet — src/structures.jl:209-267 (1-based indices kept in the arrays)
# --------------------------------------------------------------------------------------
@dataclass
class WorkingSet:
    q: int
    t: int
    l: int
    active: np.ndarray    # length l, first t entries valid (1-based constraint ids)
    inactive: np.ndarray  # length l - q, first l - t entries valid


def new_working_set(q: int, l: int) -> WorkingSet:
    """src/structures.jl:219-225."""
    active = np.zeros(l, dtype=np.int64)
    inactive = np.zeros(l - q, dtype=np.int64)
    active[:q] = np.arange(1, q + 1)
    inactive[:] = np.arange(q + 1, l + 1)
    return WorkingSet(q, q, l, active, inactive)


def remove_constraint(W: WorkingSet, s: int) -> None:
    """src/structures.jl:230-245 remove_constraint! (s is 1-based)."""
    l, t = W.l, W.t
    W.inactive[l - t] = W.active[s - 1]
    W.inactive[: l - t + 1] = np.sort(W.inactive[: l - t + 1])
    for i in range(s, t):            # i = s:t-1 (1-based)
        W.active[i - 1] = W.active[i]
    W.active[t - 1] = 0
    W.t -= 1


def add_constraint(W: WorkingSet, s: int) -> None:
    """src/structures.jl:250-263 add_constraint! (s is a 1-based index into inactive)."""
    l, t = W.l, W.t
    W.active[t] = W.inactive[s - 1]
    W.active[: t + 1] = np.sort(W.active[: t + 1])
    for i in range(s, l - t):        # i = s:l-t-1 (1-based)
        W.inactive[i - 1] = W.inactive[i]
    W.inactive[l - t - 1] = 0
    W.t += 1


# --------------------------------------------------------------------------------------
# update_working_set — src/enlsip_functions.jl:686-795
# --------------------------------------------------------------------------------------
def update_working_set(W: WorkingSet, rx, A, C: Constraint, grad_fx, J, p_gn: np.ndarray,
                       it: IterationRecord, eps_rank: float,
                       solve=None) -> Tuple[QRPivoted, QRPivoted, QRPivoted]:
    """``solve`` is unused by the oracle (hook used by tests to count subproblem solves)."""
    m, n = J.shape

    def _direction(F_A):
        rankA = pseudo_rank(F_A.diagR(), eps_rank)
        F_L11 = qr_colnorm(F_A.R.T)
        p, F_J2 = gn_search_direction(J, rx, C.cx, F_A, F_L11, rankA, W.t, eps_rank, it)
        p_gn[:] = p
        if solve is not None:
            solve()
        return rankA, F_L11, F_J2

    def _second_order(F_A, rankA, F_L11, F_J2, lam):
        if not (W.t != rankA or it.rankJ2 != min(m, n - rankA)):       # :745 / :773
            lam[:] = second_lagrange_mult_estimate(J, F_A, rx, p_gn, W.t, C.scaling, C.diag_scale)
            s2 = check_constraint_deletion(W.q, C.A, lam, C.scaling, C.diag_scale, 0.0)
            if s2 != 0:
                index_s2 = int(W.active[s2 - 1])
                lam = np.delete(lam, s2 - 1)
                C.diag_scale = np.delete(C.diag_scale, s2 - 1)
                C.cx = np.delete(C.cx, s2 - 1)
                remove_constraint(W, s2)
                it.delete = True
                it.index_del = index_s2
                C.A = np.delete(C.A, s2 - 1, axis=0)
                F_A = qr_colnorm(C.A.T)
                rankA, F_L11, F_J2 = _direction(F_A)
        return F_A, rankA, F_L11, F_J2, lam

    F_A = qr_colnorm(C.A.T)                                            # :700
    lam = first_lagrange_mult_estimate(C.A, grad_fx, C.cx, C.scaling, C.diag_scale, F_A, it, eps_rank)
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
        remove_constraint(W, s)
        it.delete = True
        it.index_del = index_s
        C.A = np.delete(C.A, s - 1, axis=0)
        F_A = qr_colnorm(C.A.T)
        rankA, F_L11, F_J2 = _direction(F_A)
        As_p = 0.0 if rankA <= W.t else float(A_s @ p_gn)              # :728 (quirk Q1: always 0.0)
        feasible = (As_p >= -cx_s) and (As_p > 0)
        if not feasible:
            C.cx = np.insert(C.cx, s - 1, cx_s)
            lam = np.insert(lam, s - 1, lam_s)
            C.diag_scale = np.insert(C.diag_scale, s - 1, diag_scale_s)
            s_inact = int(np.where(W.inactive == index_s)[0][0]) + 1
            add_constraint(W, s_inact)
            it.index_del = 0
            it.delete = False
            rows = A[W.active[: W.t] - 1, :]
            C.A = rows * C.diag_scale[:, None] if C.scaling else rows.copy()
            F_A = qr_colnorm(C.A.T)
            rankA, F_L11, F_J2 = _direction(F_A)
            F_A, rankA, F_L11, F_J2, lam = _second_order(F_A, rankA, F_L11, F_J2, lam)
    else:                                                              # :767-791
        rankA, F_L11, F_J2 = _direction(F_A)
        F_A, rankA, F_L11, F_J2, lam = _second_order(F_A, rankA, F_L11, F_J2, lam)
    it.lam = lam
    return F_A, F_L11, F_J2
