"""Host restatement of the ENLSIP outer iteration — everything AROUND the hot
path: working-set initialisation, analysis of the Gauss-Newton direction, subspace minimisation,
Newton direction, penalty weights, merit function, line search, termination tests.

TEST INFRASTRUCTURE (see oracle/gn_oracle.py header).  Purpose: run the reference's HS65 problem
end to end and check it against the ONE known answer the reference publishes
(docs/src/tutorial.md:126-128: solution (3.650461821, 3.65046168, 4.6204170507), objective
0.9535288567; the tutorial itself states the objective matches within sqrt(eps) and the solution
does not, :201-211) — the only end-to-end pin available for the oracle, and, with the HIP backend
plugged in at the ``update_working_set`` seam, for the product.

Statement by statement restatement of src/enlsip_functions.jl (functions cited one by one below)
and src/solver.jl:62-91; array aliasing of the Julia code (``Iteration.rx`` / ``.cx`` alias the live
buffers that ``new_point!`` overwrites, ``copy(iter)`` snapshots them) is preserved because NumPy
arrays have the same reference semantics.

The subproblem backend is injected: ``backend.update_working_set(...)`` returns three factor
objects exposing ``.R`` (k x cols), ``.p`` (1-based), ``.P``, ``Qt_mul(v)``, ``Q_mul(v)`` and the
backend provides ``rmul_Q1(J)`` (= J * F_A.Q) and ``sub_search_direction(...)``.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

EPS = float(np.finfo(np.float64).eps)
SQRT_EPS = math.sqrt(EPS)


# ---------------------------------------------------------------------------------------------
# evaluation wrappers (src/cnls_model.jl:11-62) and data records (src/structures.jl)
# ---------------------------------------------------------------------------------------------
class EvalFunction:
    def __init__(self, f: Callable, jac: Callable):
        self.f, self.jac = f, jac
        self.nb_eval = 0
        self.nb_jac = 0

    def eval_into(self, x, out):
        out[:] = self.f(x)
        self.nb_eval += 1

    def jac_into(self, x, out):
        out[:] = self.jac(x)
        self.nb_jac += 1


@dataclass
class Iteration:                                   # src/structures.jl:63-98
    x: np.ndarray
    p: np.ndarray
    rx: np.ndarray
    cx: np.ndarray
    t: int
    alpha: float
    index_alpha_upp: int
    lam: np.ndarray
    w: np.ndarray
    rankA: int
    rankJ2: int
    dimA: int
    dimJ2: int
    b_gn: np.ndarray
    d_gn: np.ndarray
    predicted_reduction: float
    progress: float
    grad_res: float
    speed: float
    beta: float
    restart: bool
    first: bool
    add: bool
    delete: bool
    index_del: int
    code: int
    nb_newton_steps: int

    def copy(self) -> "Iteration":                 # Base.copy(s::Iteration), :93-98 (deep copies of the arrays)
        return Iteration(self.x.copy(), self.p.copy(), self.rx.copy(), self.cx.copy(), self.t, self.alpha,
                         self.index_alpha_upp, self.lam.copy(), self.w.copy(), self.rankA, self.rankJ2, self.dimA,
                         self.dimJ2, self.b_gn.copy(), self.d_gn.copy(), self.predicted_reduction, self.progress,
                         self.grad_res, self.speed, self.beta, self.restart, self.first, self.add, self.delete,
                         self.index_del, self.code, self.nb_newton_steps)


def new_point(x, r: EvalFunction, c: EvalFunction, rx, cx, J, A):   # src/enlsip_functions.jl:34-52
    r.eval_into(x, rx)
    r.jac_into(x, J)
    c.eval_into(x, cx)
    c.jac_into(x, A)


# ---------------------------------------------------------------------------------------------
# working set (src/enlsip_functions.jl:608-650, 826-859)
# ---------------------------------------------------------------------------------------------
def init_working_set(cx, K, step: Iteration, q: int, l: int, ws_mod):
    delta, eps_, _ = 0.1, 0.01, SQRT_EPS
    for i in range(len(K)):
        K[i] = delta * np.ones(l)
    for i in range(l):
        step.w[i] = min(abs(cx[i]) + eps_, delta)
    active = np.zeros(l, dtype=np.int64)
    inactive = np.zeros(l - q, dtype=np.int64)
    t = q
    lmt = 0
    active[:q] = np.arange(1, q + 1)
    for i in range(q + 1, l + 1):
        if cx[i - 1] <= 0.0:
            t += 1
            active[t - 1] = i
        else:
            lmt += 1
            inactive[lmt - 1] = i
    step.t = t
    return ws_mod.WorkingSet(q, t, l, active, inactive)


def evaluate_violated_constraints(cx, W, index_alpha_upp: int, n: int, ws_mod) -> bool:
    eps_ = SQRT_EPS
    delta = 0.1
    bnd = min(W.l, n)
    added = False
    if W.l > W.t:
        i = 1
        while i <= W.l - W.t:
            k = int(W.inactive[i - 1])
            if cx[k - 1] < eps_ or (k == index_alpha_upp and cx[k - 1] < delta):
                if W.t >= bnd:
                    worst_k, worst_val = 0, -math.inf
                    for j in range(W.q + 1, W.t + 1):
                        jj = int(W.active[j - 1])
                        if cx[jj - 1] > worst_val:
                            worst_val = cx[jj - 1]
                            worst_k = j
                    if worst_k > 0 and worst_val > cx[k - 1]:
                        ws_mod.remove_constraint(W, worst_k)
                    else:
                        i += 1
                        continue
                ws_mod.add_constraint(W, i)
                added = True
            else:
                i += 1
    return added


# ---------------------------------------------------------------------------------------------
# analysis of the GN direction (src/enlsip_functions.jl:864-1291)
# ---------------------------------------------------------------------------------------------
def subspace_min_previous_step(tau, rho, rho_prk, c1, pseudo_rk, previous_dimR, progress,
                               predicted_linear_progress, prelin_previous_dim, previous_alpha):
    stepb, pgb1, pgb2, predb, rlenb, c2 = 2e-1, 3e-1, 1e-1, 7e-1, 2.0, 1e2
    if (previous_alpha < stepb and progress <= pgb1 * predicted_linear_progress ** 2
            and progress <= pgb2 * prelin_previous_dim ** 2):
        dim = max(1, previous_dimR - 1)
        if previous_dimR > 1 and rho[dim - 1] > c1 * rho_prk:
            return dim
    dim = previous_dimR
    if previous_dimR < len(tau) and (((rho[dim - 1] > predb * rho_prk) and (rlenb * tau[dim - 1] < tau[dim]))
                                     or (c2 * tau[dim - 1] < tau[dim])):
        suggested_dim = dim
    else:
        i1 = previous_dimR - 1
        if i1 <= 0:
            suggested_dim = pseudo_rk
        else:
            buff = [i for i in range(i1, previous_dimR + 1) if rho[i - 1] > predb * rho_prk]
            suggested_dim = pseudo_rk if not buff else min(buff)
    return suggested_dim


def gn_previous_step(tau, tau_prk, mindim, rho, rho_prk, pseudo_rank):
    tau_max, rho_min = 2e-1, 5e-1
    pm1 = pseudo_rank - 1
    if mindim > pm1:
        return mindim
    k = pm1
    while (tau[k - 1] >= tau_max * tau_prk or rho[k - 1] <= rho_min * rho_prk) and k > mindim:
        k -= 1
    return k if k > mindim else max(mindim, pm1)


def check_gn_direction(b1nrm, d1nrm, d1nrm_as_km1, dnrm, active_c_sum, iter_number, rankA, n, m, restart,
                       constraint_added, constraint_deleted, W, cx, lam, iter_km1: Iteration, scaling, diag_scale):
    delta = 1e-1
    c1, c2, c3, c4, c5 = 0.5, 0.1, 4.0, 10.0, 0.05
    eps_rel = EPS
    beta_k = math.sqrt(d1nrm ** 2 + b1nrm ** 2)
    method_code = 1
    newton_or_restart = iter_km1.code == 2 or restart
    first_iter = iter_number == 0
    submin_prev_iter = iter_km1.code == -1
    add_or_del = constraint_added or constraint_deleted
    convergence_lower_c1 = beta_k < c1 * iter_km1.beta
    progress_not_close = (iter_km1.progress > c2 * iter_km1.predicted_reduction) and (dnrm <= c3 * beta_k)
    if newton_or_restart or (not first_iter and (submin_prev_iter or not (add_or_del or convergence_lower_c1
                                                                          or progress_not_close))):
        method_code = -1
        non_linearity_k = math.sqrt(d1nrm * d1nrm + active_c_sum)
        non_linearity_km1 = math.sqrt(d1nrm_as_km1 * d1nrm_as_km1 + active_c_sum)
        to_reduce = False
        if W.q < W.t:
            sqr_eps = SQRT_EPS
            rows = np.zeros(W.t - W.q)
            for i in range(W.q + 1, W.t + 1):
                rows[i - W.q - 1] = (1.0 / diag_scale[i - 1]) if scaling else diag_scale[i - 1]
            lw = lam[W.q: W.t]
            lagrange_mult_cond = bool(np.any(lw * rows >= -sqr_eps)) and bool(np.any(lw < 0))
            to_reduce = to_reduce or lagrange_mult_cond
        if W.l - W.t > 0:
            inact_c = [cx[int(W.inactive[j]) - 1] for j in range(W.l - W.t)]
            to_reduce = to_reduce or any(v < delta for v in inact_c)
        newton_previously = iter_km1.code == 2 and not constraint_deleted
        cond4 = active_c_sum > c2
        cond5 = constraint_deleted or constraint_added or to_reduce or (W.t == n and W.t == rankA)
        eps_ = max(1e-2, 10.0 * eps_rel)
        cond6 = (not ((W.l == W.q) or (rankA <= W.t))) and (not ((beta_k < eps_ * dnrm) or (b1nrm < eps_ and m == n - W.t)))
        if newton_previously or not (cond4 or cond5 or cond6):
            cond7 = (iter_km1.alpha < c5 and non_linearity_km1 < c2 * non_linearity_k) or m == n - W.t
            cond8 = not (dnrm <= c4 * beta_k)
            if newton_previously or cond7 or cond8:
                method_code = 2
    return method_code, beta_k


def determine_solving_dim(previous_dimR, rankR, predicted_linear_progress, obj_progress, prelin_previous_dim, R, y,
                          previous_alpha, restart):
    c1 = 0.1
    newdim = rankR
    eta = 1.0
    mindim = 1
    if rankR > 0:
        l_sd, l_rh = np.zeros(rankR), np.zeros(rankR)
        l_sd[0] = abs(y[0])
        l_rh[0] = abs(y[0] / R[0, 0])
        for i in range(2, rankR + 1):
            l_sd[i - 1] = y[i - 1]
            l_rh[i - 1] = y[i - 1] / R[i - 1, i - 1]
            l_rh[i - 1] = np.linalg.norm(l_rh[i - 2: i])
            l_sd[i - 1] = np.linalg.norm(l_sd[i - 2: i])
        nrm_sd = l_sd[rankR - 1]
        nrm_rh = l_rh[rankR - 1]
        dsum = 0.0
        psimax = 0.0
        for i in range(1, rankR + 1):
            dsum += l_sd[i - 1] ** 2
            psi_ = math.sqrt(dsum) * abs(R[i - 1, i - 1])
            if psi_ > psimax:
                psimax = psi_
                mindim = i
        k = mindim
        if not restart:
            suggested_dim = rankR
            if previous_dimR == rankR or previous_dimR <= 0:
                suggested_dim = gn_previous_step(l_sd, nrm_sd, mindim, l_rh, nrm_rh, rankR)
            elif previous_dimR != rankR and previous_dimR > 0:
                suggested_dim = subspace_min_previous_step(l_sd, l_rh, nrm_rh, c1, rankR, previous_dimR, obj_progress,
                                                           predicted_linear_progress, prelin_previous_dim, previous_alpha)
            newdim = max(mindim, suggested_dim)
        else:
            newdim = max(0, min(rankR, previous_dimR))
            if newdim != 0:
                k = max(previous_dimR - 1, 1)
                if l_sd[newdim - 1] != 0:
                    eta = l_sd[k - 1] / l_sd[newdim - 1]
    return newdim, eta


def choose_subspace_dimensions(rx_sum, rx, active_cx_sum, J1, t, rankJ2, rankA, b, F_L11, F_J2,
                               previous_iter: Iteration, restart):
    c1, c2, alpha_low = 0.1, 0.01, 0.2
    previous_alpha = previous_iter.alpha
    if rankA <= 0:
        dimA = 0
        previous_dimA = 0
        d = -rx
    else:
        previous_dimA = abs(previous_iter.dimA) + t - previous_iter.t
        nrm_b_asprev = float(np.linalg.norm(b[:previous_dimA]))
        nrm_b = float(np.linalg.norm(b))
        constraint_progress = float(previous_iter.cx @ previous_iter.cx) - active_cx_sum
        RL = np.triu(F_L11.R)
        dimA, _ = determine_solving_dim(previous_dimA, rankA, nrm_b, constraint_progress, nrm_b_asprev, RL, b,
                                        previous_alpha, restart)
        dp1 = _utri_solve(RL[:dimA, :dimA], b[:dimA])
        PL = F_L11.P
        p1 = PL[:rankA, :rankA] @ np.concatenate([dp1, np.zeros(rankA - dimA)])
        d = -(rx + J1 @ p1)
    if rankJ2 > 0:
        d = F_J2.Qt_mul(d)
    previous_dimJ2 = abs(previous_iter.dimJ2) + previous_iter.t - t
    nrm_d_asprev = float(np.linalg.norm(d[:max(previous_dimJ2, 0)]))
    nrm_d = float(np.linalg.norm(d))
    residual_progress = float(previous_iter.rx @ previous_iter.rx) - rx_sum
    dimJ2, _ = determine_solving_dim(previous_dimJ2, rankJ2, nrm_d, residual_progress, nrm_d_asprev, np.triu(F_J2.R), d,
                                     previous_alpha, restart)
    if (not restart) and previous_alpha >= alpha_low:
        dimA = max(dimA, previous_dimA)
        dimJ2 = max(dimJ2, previous_dimJ2)
    return dimA, dimJ2


def _utri_solve(U, b):
    b = np.asarray(b, dtype=np.float64)
    if b.size == 0:
        return b.copy()
    from scipy.linalg import solve_triangular
    return solve_triangular(U, b, lower=False)


def _ltri_solve(Lo, b):
    b = np.asarray(b, dtype=np.float64)
    if b.size == 0:
        return b.copy()
    from scipy.linalg import solve_triangular
    return solve_triangular(Lo, b, lower=True)


# ---- Newton direction (src/enlsip_functions.jl:243-423) ------------------------------------------------
def hessian_res(r: EvalFunction, x, rx, n, m):
    B = np.zeros((n, n))
    e1 = EPS ** (1.0 / 3.0)
    f1, f2, f3, f4 = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m)
    for k in range(n):
        for j in range(k + 1):
            ek = max(abs(x[k]), 1.0) * e1
            ej = max(abs(x[j]), 1.0) * e1
            xw = x.copy(); xw[j] += ej; xw[k] += ek; r.eval_into(xw, f1)
            xw = x.copy(); xw[j] -= ej; xw[k] += ek; r.eval_into(xw, f2)
            xw = x.copy(); xw[j] += ej; xw[k] -= ek; r.eval_into(xw, f3)
            xw = x.copy(); xw[j] -= ej; xw[k] -= ek; r.eval_into(xw, f4)
            s = 0.0
            for i in range(m):
                s += (f1[i] - f2[i] - f3[i] + f4[i]) * rx[i]
            s /= (4 * ej * ek)
            B[k, j] = s
            B[j, k] = s
    return B


def hessian_cons(c: EvalFunction, x, lam, active, n, l, t):
    B = np.zeros((n, n))
    e1 = EPS ** (1.0 / 3.0)
    f1, f2, f3, f4 = np.zeros(l), np.zeros(l), np.zeros(l), np.zeros(l)
    for k in range(n):
        for j in range(k + 1):
            ek = max(abs(x[k]), 1.0) * e1
            ej = max(abs(x[j]), 1.0) * e1
            xw = x.copy(); xw[j] += ej; xw[k] += ek; c.eval_into(xw, f1)
            xw = x.copy(); xw[j] -= ej; xw[k] += ek; c.eval_into(xw, f2)
            xw = x.copy(); xw[j] += ej; xw[k] -= ek; c.eval_into(xw, f3)
            xw = x.copy(); xw[j] -= ej; xw[k] -= ek; c.eval_into(xw, f4)
            s = 0.0
            for i in range(t):
                ii = int(active[i]) - 1
                s += (f1[ii] - f2[ii] - f3[ii] + f4[ii]) * lam[i]
            s /= (4.0 * ek * ej)
            B[k, j] = s
            B[j, k] = s
    return B


def newton_search_direction(x, c, r, active_cx, W, lam, rx, J, F_A, F_L11, rankA, backend):
    m, n = J.shape
    t, l = W.t, W.l
    if t == rankA:
        b = -active_cx[F_A.p - 1]
        p1 = _ltri_solve(np.triu(F_A.R)[:t, :t].T, b)
    else:
        b = F_L11.Qt_mul(-active_cx[F_A.p - 1])
        dp1 = _utri_solve(np.triu(F_L11.R)[:rankA, :rankA], b[:rankA])
        p1 = F_L11.P[:rankA, :rankA] @ dp1
    if rankA == n:
        return p1, False         # (the reference returns only p1 here; callers unpack two values)
    JQ1 = backend.rmul_Q1(J, F_A)
    J1, J2 = JQ1[:, :rankA], JQ1[:, rankA:]
    r_mat = hessian_res(r, x, rx, n, m)
    c_mat = hessian_cons(c, x, lam, W.active, n, l, t)
    Gam = r_mat - c_mat
    if t == rankA and hasattr(backend, "newton_direction"):
        # the backend under test owns everything after the Hessian sums (the HIP library: enlsip_gn_newton_direction)
        return backend.newton_direction(Gam)
    QtG = np.column_stack([F_A.Qt_mul(Gam[:, j]) for j in range(n)])          # Q' * Gam
    E = np.column_stack([F_A.Qt_mul(QtG[i, :]) for i in range(n)]).T         # (Q' Gam) Q
    if t > rankA:
        vp = F_L11.p - 1
        E = E[np.ix_(vp, vp)]
    E21 = E[rankA:n, :rankA]
    E22 = E[rankA:n, rankA:n]
    W22 = E22 + J2.T @ J2
    W21 = E21 + J2.T @ J1
    d = -W21 @ p1 - J2.T @ rx
    sW22 = (W22 + W22.T) * 0.5
    try:
        Lc = np.linalg.cholesky(sW22)
    except np.linalg.LinAlgError:
        return np.zeros(n), True
    y = _ltri_solve(Lc, d)
    p2 = _utri_solve(Lc.T, y)
    p = F_A.Q_mul(np.concatenate([p1, p2]))
    return p, False


def search_direction_analys(previous_iter: Iteration, current_iter: Iteration, iter_number, x, c, r, rx, cx, active_C,
                            active_cx_sum, p_gn, J, W, F_A, F_L11, F_J2, second_derivatives, backend):
    m, n = J.shape
    rx_sum = float(rx @ rx)
    active_cx = active_C.cx
    scaling = active_C.scaling
    diag_scale = active_C.diag_scale
    lam = current_iter.lam
    b_gn = current_iter.b_gn
    nrm_b1_gn = float(np.linalg.norm(b_gn[:current_iter.dimA]))
    rankA = current_iter.rankA
    d_gn = current_iter.d_gn
    nrm_d_gn = float(np.linalg.norm(d_gn))
    nrm_d1_gn = float(np.linalg.norm(d_gn[:current_iter.dimJ2]))
    rankJ2 = current_iter.rankJ2
    prev_dimJ2m1 = previous_iter.dimJ2 + previous_iter.t - W.t - 1
    nrm_d1_asprev = float(np.linalg.norm(d_gn[:max(prev_dimJ2m1, 0)]))
    restart = current_iter.restart
    error_code = 0
    method_code, beta = check_gn_direction(nrm_b1_gn, nrm_d1_gn, nrm_d1_asprev, nrm_d_gn, active_cx_sum, iter_number,
                                           rankA, n, m, restart, current_iter.add, current_iter.delete, W, cx, lam,
                                           previous_iter, scaling, diag_scale)
    if method_code == 1:
        dimA, dimJ2 = rankA, rankJ2
        p, b, d = p_gn, b_gn, d_gn
    elif method_code == -1:
        JQ1 = backend.rmul_Q1(J, F_A)
        J1 = JQ1[:, :rankA]
        b = F_L11.Qt_mul(-active_cx[F_A.p - 1])
        dimA, dimJ2 = choose_subspace_dimensions(rx_sum, rx, active_cx_sum, J1, W.t, rankJ2, rankA, b, F_L11, F_J2,
                                                 previous_iter, restart)
        p, b, d = backend.sub_search_direction(J1, rx, active_cx, F_A, F_L11, F_J2, n, W.t, rankA, dimA, dimJ2, method_code)
        if dimA == rankA and dimJ2 == rankJ2:
            method_code = 1
    else:
        if second_derivatives:
            p, newton_error = newton_search_direction(x, c, r, active_cx, W, lam, rx, J, F_A, F_L11, rankA, backend)
            b, d = b_gn, d_gn
            dimA = -W.t
            dimJ2 = W.t - n
            current_iter.nb_newton_steps += 1
            if newton_error:
                error_code = -3
        else:
            p, b, d = p_gn, b_gn, d_gn
            dimA, dimJ2 = rankA, rankJ2
            error_code = -4
    current_iter.b_gn = b
    current_iter.d_gn = d
    current_iter.dimA = dimA
    current_iter.dimJ2 = dimJ2
    current_iter.code = method_code
    current_iter.speed = beta / previous_iter.beta if previous_iter.beta != 0 else math.inf
    current_iter.beta = beta
    current_iter.p = p
    return error_code


# ---------------------------------------------------------------------------------------------
# merit function, penalty weights (src/enlsip_functions.jl:1307-1640)
# ---------------------------------------------------------------------------------------------
def psi(x, alpha, p, r, c, w, m, l, t, active, inactive):
    rxb, cxb = np.zeros(m), np.zeros(l)
    x_new = x + alpha * p
    r.eval_into(x_new, rxb)
    c.eval_into(x_new, cxb)
    s = 0.0
    for i in range(t):
        j = int(active[i]) - 1
        s += w[j] * cxb[j] ** 2
    for i in range(l - t):
        j = int(inactive[i]) - 1
        if cxb[j] < 0.0:
            s += w[j] * cxb[j] ** 2
    return 0.5 * (float(rxb @ rxb) + s)


def assort(K, w, t, active):
    for i in range(t):
        for ii in range(4):
            k = int(active[i]) - 1
            if w[k] > K[ii][k]:
                for j in range(3, ii, -1):
                    K[j][k] = K[j - 1][k]
                K[ii][k] = w[k]


def min_norm_w(ctrl, w, w_old, y, tau, pos_index, nb_pos):
    w[:] = w_old
    if nb_pos > 0:
        y = y.copy()
        y_sum = float(y @ y)
        y_norm = float(np.linalg.norm(y))
        if y_norm != 0.0:
            y = y / y_norm
        tau_new = tau
        s = 0.0
        n_runch = nb_pos
        terminated = False
        while not terminated:
            tau_new -= s
            cc = 1.0 if np.max(np.abs(y)) <= EPS else tau_new / y_sum
            y_sum, s = 0.0, 0.0
            i_stop = n_runch
            k = 1
            while k <= n_runch:
                i = int(pos_index[k - 1]) - 1
                buff = cc * y[k - 1] * y_norm
                if buff >= w_old[i]:
                    w[i] = buff
                    y_sum += y[k - 1] ** 2
                    k += 1
                else:
                    s += w_old[i] * y[k - 1] * y_norm
                    n_runch -= 1
                    for j in range(k, n_runch + 1):
                        pos_index[j - 1] = pos_index[j]
                        y[j - 1] = y[j]
            y_sum *= y_norm * y_norm
            terminated = (n_runch <= 0) or (ctrl == 2) or (i_stop == n_runch)


def euclidean_norm_weight_update(vA, cx, active, t, mu, dimA, previous_w, K):
    w = previous_w.copy()
    if t != 0:
        z = vA ** 2
        w_old = K[3]
        act = np.asarray(active[:t], dtype=np.int64) - 1
        ztw = float(z @ w_old[act])
        pos_index = np.zeros(t + 1, dtype=np.int64)
        if ztw >= mu and dimA < t:
            y = np.zeros(t + 1)
            ctrl, nb_pos, gamma = 2, 0, 0.0
            for i in range(t):
                k = int(active[i])
                y_elem = vA[i] * (vA[i] + cx[k - 1])
                if y_elem > 0:
                    nb_pos += 1
                    pos_index[nb_pos - 1] = k
                    y[nb_pos - 1] = y_elem
                else:
                    gamma -= y_elem * w_old[k - 1]
            min_norm_w(ctrl, w, w_old, y[:t], gamma, pos_index, nb_pos)
        elif ztw < mu and dimA < t:
            e = np.zeros(t + 1)
            ctrl, nb_pos, tau = 2, 0, mu
            for i in range(t):
                k = int(active[i])
                e_elem = -vA[i] * cx[k - 1]
                if e_elem > 0:
                    nb_pos += 1
                    pos_index[nb_pos - 1] = k
                    e[nb_pos - 1] = e_elem
                else:
                    tau -= e_elem * w_old[k - 1]
            min_norm_w(ctrl, w, w_old, e[:t], tau, pos_index, nb_pos)
        elif ztw < mu and dimA == t:
            ctrl = 1
            pos_index[:t] = active[:t]
            min_norm_w(ctrl, w, w_old, z.copy(), mu, pos_index, t)
        assort(K, w, t, active)
    return w


def max_norm_weight_update(nrm_Ap, rmy, alpha_w, delta, w, active, t, K):
    mu = 0.0 if abs(alpha_w - 1.0) <= delta else rmy / nrm_Ap
    i1 = int(active[0]) if active[0] != 0 else 1
    previous_w = w[i1 - 1]
    nu = max(mu, K[3][0])
    for i in range(t):
        w[int(active[i]) - 1] = nu
    if mu > previous_w:
        i = 1
        placed = False
        while i <= 4 and not placed:
            if mu > K[i - 1][0]:
                for j in range(4, i, -1):
                    K[j - 1][0] = K[j - 2][0]
                K[i - 1][0] = mu
                placed = True
            i += 1


def penalty_weight_update(w_old, Jp, Ap, K, rx, cx, W, dimA, norm_code):
    delta = 0.25
    active = W.active
    t = W.t
    nrm_Ap = math.sqrt(float(Ap @ Ap))
    idx = np.asarray(active[:max(dimA, 0)], dtype=np.int64) - 1
    nrm_cx = 0.0 if idx.size == 0 else max(0.0, float(np.max(np.abs(cx[idx]))))
    nrm_Jp = math.sqrt(float(Jp @ Jp))
    nrm_rx = math.sqrt(float(rx @ rx))
    if nrm_Jp != 0:
        Jp = Jp / nrm_Jp
    if nrm_Ap != 0:
        Ap = Ap / nrm_Ap
    if nrm_rx != 0:
        rx = rx / nrm_rx
    if nrm_cx != 0:
        cx = cx / nrm_cx
    Jp_rx = float(Jp @ rx) * nrm_Jp * nrm_rx
    AtwA = 0.0
    BtwA = 0.0
    for i in range(dimA):
        k = int(active[i]) - 1
        AtwA += w_old[k] * Ap[i] ** 2
        BtwA += w_old[k] * Ap[i] * cx[k]
    AtwA *= nrm_Ap ** 2
    BtwA *= nrm_Ap * nrm_cx
    alpha_w = 1.0
    if abs(AtwA + nrm_Jp ** 2) > EPS:
        alpha_w = (-BtwA - Jp_rx) / (AtwA + nrm_Jp ** 2)
    rmy = (abs(Jp_rx + nrm_Jp ** 2) / delta) - nrm_Jp ** 2
    if norm_code == 0:
        w = w_old.copy()
        max_norm_weight_update(nrm_Ap, rmy, alpha_w, delta, w, active, t, K)
    else:
        w = euclidean_norm_weight_update(Ap * nrm_Ap, cx * nrm_cx, active, t, rmy, dimA, w_old, K)
    BtwA = 0.0
    AtwA = 0.0
    for i in range(t):
        k = int(active[i]) - 1
        AtwA += w[k] * Ap[i] ** 2
        BtwA += w[k] * Ap[i] * cx[k]
    BtwA *= nrm_Ap * nrm_cx
    AtwA *= nrm_Ap ** 2
    dpsi0 = BtwA + Jp_rx
    return w, dpsi0


# ---------------------------------------------------------------------------------------------
# line search (src/enlsip_functions.jl:1641-2195)
# ---------------------------------------------------------------------------------------------
def concatenate(v, rx, cx, w, m, t, l, active, inactive):
    v[:m] = rx
    for i in range(t):
        k = int(active[i]) - 1
        v[m + k] = math.sqrt(w[k]) * cx[k]
    for j in range(l - t):
        k = int(inactive[j]) - 1
        v[m + k] = 0.0 if cx[k] > 0 else math.sqrt(w[k]) * cx[k]


def coefficients_linesearch(v0, v1, v2, alpha_k, rx, cx, rx_new, cx_new, w, m, t, l, active, inactive):
    concatenate(v0, rx, cx, w, m, t, l, active, inactive)
    vb = np.zeros(m + l)
    concatenate(vb, rx_new, cx_new, w, m, t, l, active, inactive)
    v2[:] = ((vb - v0) / alpha_k - v1) / alpha_k


def minimize_quadratic(x1, y1, x2, y2, x3, y3):
    d1, d2 = y2 - y1, y3 - y1
    s = (x3 - x1) ** 2 * d1 - (x2 - x1) ** 2 * d2
    q = 2 * ((x2 - x1) * d2 - (x3 - x1) * d1)
    return x1 - s / q


def minrn(x1, y1, x2, y2, x3, y3, a_min, a_max, p_max):
    eps_ = SQRT_EPS / p_max
    if abs(x1 - x2) < eps_ or abs(x3 - x1) < eps_ or abs(x3 - x2) < eps_:
        return 0.0, 0.0
    u = minimize_quadratic(x1, y1, x2, y2, x3, y3)
    a = min(max(u, a_min), a_max)
    t1 = (a - x1) * (a - x2) * y3 / ((x3 - x1) * (x3 - x2))
    t2 = (a - x3) * (a - x2) * y1 / ((x1 - x3) * (x1 - x2))
    t3 = (a - x3) * (a - x2) * y2 / ((x2 - x1) * (x2 - x3))
    return a, t1 + t2 + t3


def _polyval(coef, x):      # coefficients in increasing degree (Polynomials.jl convention)
    s = 0.0
    for cfs in reversed(coef):
        s = s * x + cfs
    return s


def _polyder(coef):
    return [i * coef[i] for i in range(1, len(coef))]


def newton_raphson(x_min, Dm, ds, dds):
    a, it = x_min, 0
    eps_, err = 1e-4, 1.0
    while (err > eps_ or it < 3) and it < 50:
        cc = _polyval(dds, a)
        if abs(cc) < EPS:
            break
        h = -_polyval(ds, a) / cc
        a += h
        err = (2 * Dm * h ** 2) / abs(cc)
        it += 1
    return a


def one_root(cc, d, a):
    arg1, arg2 = -cc / 2 + math.sqrt(d), -cc / 2 - math.sqrt(d)
    return float(np.cbrt(arg1) + np.cbrt(arg2)) - a / 3


def two_roots(b, cc, d, a, x_min):
    phi = math.acos(abs(cc / 2) / (-b / 3) ** 1.5)
    tt = 2 * math.sqrt(-b / 3) if cc <= 0 else -2 * math.sqrt(-b / 3)
    b1 = tt * math.cos(phi / 3) - a / 3
    b2 = tt * math.cos((phi + 2 * math.pi) / 3) - a / 3
    b3 = tt * math.cos((phi + 4 * math.pi) / 3) - a / 3
    b1, b2, b3 = sorted([b1, b2, b3])
    return (b1, b3) if x_min <= b2 else (b3, b1)


def parameters_rm(v0, v1, v2, x_min, ds, dds):
    dds_best = _polyval(dds, x_min)
    eta, d = 0.1, 1.0
    normv2 = float(v2 @ v2)
    h0 = abs(_polyval(ds, x_min) / dds_best)
    Dm = abs(6 * float(v1 @ v2) + 12 * x_min * normv2) + 24 * h0 * normv2
    hm = max(h0, 1)
    beta_hat = None
    if dds_best * eta < 2 * Dm * hm:
        cs = [cf / (2 * normv2) for cf in ds]
        # `(a3, a2, a1) = coeffs(ds) / (2 normv2)` destructures the FIRST three coefficients (degree 0, 1, 2)
        a3, a2, a1 = cs[0], cs[1], cs[2]
        b = a2 - (a1 ** 2) / 3
        cc = a3 - a1 * a2 / 3 + 2 * (a1 / 3) ** 3
        d = (cc / 2) ** 2 + (b / 3) ** 3
        if d < 0:
            alpha_hat, beta_hat = two_roots(b, cc, d, a1, x_min)
        else:
            alpha_hat = one_root(cc, d, a1)
    else:
        alpha_hat = newton_raphson(x_min, Dm, ds, dds)
    if d >= 0:
        beta_hat = alpha_hat
    return alpha_hat, beta_hat


def minrm(v0, v1, v2, x_min, a_min, a_max):
    s = [0.5 * float(v0 @ v0), float(v0 @ v1), float(v0 @ v2) + 0.5 * float(v1 @ v1), float(v1 @ v2),
         0.5 * float(v2 @ v2)]
    ds = _polyder(s)
    dds = _polyder(ds)
    a_hat, b_hat = parameters_rm(v0, v1, v2, x_min, ds, dds)
    a_old = a_hat
    a_hat = max(min(a_hat, a_max), a_min)
    sa = _polyval(s, a_hat)
    if a_old == b_hat:
        b_hat, sb = a_hat, _polyval(s, a_hat)
    else:
        b_hat = max(min(b_hat, a_max), a_min)
        sb = _polyval(s, b_hat)
    return a_hat, sa, b_hat, sb


def check_reduction(psi_a, psi_k, approx_k, eta, diff_psi):
    delta = 0.2
    if psi_a - approx_k >= eta * diff_psi:
        return not ((psi_a - psi_k < eta * diff_psi) and (psi_k > delta * psi_a))
    return False


def goldstein_armijo_step(psi0, dpsi0, a_min, tau, p_max, x, a0, p, r, c, w, m, l, t, active, inactive):
    u = a0
    ex = (p_max * u < SQRT_EPS) or (u <= a_min)
    psiu = psi(x, u, p, r, c, w, m, l, t, active, inactive)
    while (not ex) and (psiu > psi0 + tau * u * dpsi0):
        u *= 0.5
        psiu = psi(x, u, p, r, c, w, m, l, t, active, inactive)
        ex = (p_max * u < SQRT_EPS) or (u <= a_min)
    return u, ex


def linesearch_constrained(x, a0, p, r, c, rx, cx, JpAp, w, W, psi0, dpsi0, a_low, a_upp):
    m = len(rx)
    l, t = W.l, W.t
    active, inactive = W.active, W.inactive
    rx_new, cx_new = np.zeros(m), np.zeros(l)
    v0, v2 = np.zeros(m + l), np.zeros(m + l)
    eta, tau, gamma = 0.3, 0.25, 0.4
    a_min, a_max = a_low, a_upp
    a_k = min(a0, a_max)
    a_km1 = 0.0
    psi_km1 = psi0
    p_max = float(np.max(np.abs(p)))
    gac_error = False
    v1 = JpAp
    for i in range(t):
        k = int(active[i]) - 1
        v1[m + k] = math.sqrt(w[k]) * v1[m + k]
    for j in range(l - t):
        k = int(inactive[j]) - 1
        v1[m + k] = 0.0 if cx[k] > 0 else math.sqrt(w[k]) * v1[m + k]
    psi_k = psi(x, a_k, p, r, c, w, m, l, t, active, inactive)
    diff_psi = psi0 - psi_k
    x_new = x + a_k * p
    r.eval_into(x_new, rx_new)
    c.eval_into(x_new, cx_new)
    v0[:] = 0.0
    v2[:] = 0.0
    coefficients_linesearch(v0, v1, v2, a_k, rx, cx, rx_new, cx_new, w, m, t, l, active, inactive)
    x_min = a_k if diff_psi >= 0 else 0.0
    a_kp1, pk, beta, pbeta = minrm(v0, v1, v2, x_min, a_min, a_max)
    if a_kp1 != beta and pbeta < pk and beta <= a_k:
        a_kp1 = beta
        pk = pbeta
    a_km2, psi_km2 = a_km1, psi_km1
    a_km1, psi_km1 = a_k, psi_k
    a_k = a_kp1
    psi_k = psi(x, a_k, p, r, c, w, m, l, t, active, inactive)
    if (-diff_psi <= tau * dpsi0 * a_km1) or (psi_km1 < gamma * psi0):
        diff_psi = psi0 - psi_k
        reduction_likely = check_reduction(psi_km1, psi_k, pk, eta, diff_psi)
        while reduction_likely:
            a_kp1, pk = minrn(a_k, psi_k, a_km1, psi_km1, a_km2, psi_km2, a_min, a_max, p_max)
            a_km2, psi_km2 = a_km1, psi_km1
            a_km1, psi_km1 = a_k, psi_k
            a_k = a_kp1
            psi_k = psi(x, a_k, p, r, c, w, m, l, t, active, inactive)
            diff_psi = psi0 - psi_k
            reduction_likely = check_reduction(psi_km1, psi_k, pk, eta, diff_psi)
        if (psi_km1 - pk >= eta * diff_psi) and (psi_k < psi_km1):
            a_km1, psi_km1 = a_k, psi_k
    else:
        diff_psi = psi0 - psi_k
        if (-diff_psi <= tau * dpsi0 * a_k) or (psi_k < gamma * psi0):
            if psi0 <= psi_km1:
                x_min = a_k
                x_new = x + a_k * p
                r.eval_into(x_new, rx_new)
                c.eval_into(x_new, cx_new)
                v0[:] = 0.0
                v2[:] = 0.0
                coefficients_linesearch(v0, v1, v2, a_k, rx, cx, rx_new, cx_new, w, m, t, l, active, inactive)
                a_kp1, pk, beta, pbeta = minrm(v0, v1, v2, x_min, a_min, a_max)
                if a_kp1 != beta and pbeta < pk and beta <= a_k:
                    a_kp1 = beta
                    pk = pbeta
                a_km1 = 0.0
                psi_km1 = psi0
            else:
                a_kp1, pk = minrn(a_k, psi_k, a_km1, psi_km1, a_km2, psi_km2, a_min, a_max, p_max)
            a_km2, psi_km2 = a_km1, psi_km1
            a_km1, psi_km1 = a_k, psi_k
            a_k = a_kp1
            psi_k = psi(x, a_k, p, r, c, w, m, l, t, active, inactive)
            reduction_likely = check_reduction(psi_km1, psi_k, pk, eta, diff_psi)
            while reduction_likely:
                a_kp1, pk = minrn(a_k, psi_k, a_km1, psi_km1, a_km2, psi_km2, a_min, a_max, p_max)
                a_km2, psi_km2 = a_km1, psi_km1
                a_km1, psi_km1 = a_k, psi_k
                a_k = a_kp1
                psi_k = psi(x, a_k, p, r, c, w, m, l, t, active, inactive)
                reduction_likely = check_reduction(psi_km1, psi_k, pk, eta, diff_psi)
            if (psi_km1 - pk >= eta * diff_psi) and (psi_k < psi_km1):
                a_km1, psi_km1 = a_k, psi_k
        else:
            a_km1, gac_error = goldstein_armijo_step(psi0, dpsi0, a_min, tau, p_max, x, a_k, p, r, c, w, m, l, t,
                                                     active, inactive)
    return a_km1, gac_error


def upper_bound_steplength(A, cx, p, W, index_del):
    inactive = W.inactive
    t, l = W.t, W.l
    a_upper = math.inf
    index = 0
    if len(inactive) and np.max(np.abs(inactive)) > 0:
        for i in range(l - t):
            j = int(inactive[i])
            if j != index_del:
                g = float(A[j - 1, :] @ p)
                if g != 0.0:
                    a_j = -cx[j - 1] / g
                else:
                    a_j = math.inf if cx[j - 1] < 0 else -math.inf
                if cx[j - 1] > 0 and g < 0 and a_j < a_upper:
                    a_upper = a_j
                    index = j
    return min(3.0, a_upper), index


def check_derivatives(dpsi0, psi0, psi_k, x_old, alpha, p, r, c, w, W, m):
    l, t = W.l, W.t
    psi_ma = psi(x_old, -alpha, p, r, c, w, m, l, t, W.active, W.inactive)
    dfw = (psi_k - psi0) / alpha
    dbw = (psi0 - psi_ma) / alpha
    dce = (psi_k - psi_ma) / (2 * alpha)
    max_diff = max(abs(dfw - dce), abs(dfw - dbw), abs(dbw - dce))
    inconsistency = abs(dfw - dpsi0) > max_diff and abs(dce - dpsi0) > max_diff
    return -1 if inconsistency else 0


def compute_steplength(it: Iteration, previous_iter: Iteration, x, r, rx, J, c, cx, A, active_C, W, K, weight_code):
    m = J.shape[0]
    p = it.p
    dimA = it.dimA
    rankJ2 = it.rankJ2
    method_code = it.code
    ind_del = it.index_del
    previous_alpha = previous_iter.alpha
    prev_rankJ2 = previous_iter.rankJ2
    w_old = previous_iter.w
    Jp = J @ p
    Ap = A @ p
    JpAp = np.concatenate([Jp, Ap])
    active_Ap = active_C.A @ p
    active_index = np.asarray(W.active[:W.t], dtype=np.int64) - 1
    if active_C.scaling:
        active_Ap = active_Ap / active_C.diag_scale
    psi_error = 0
    if method_code != 2:
        w, dpsi0 = penalty_weight_update(w_old, Jp, active_Ap, K, rx, cx, W, dimA, weight_code)
        psi0 = 0.5 * (float(rx @ rx) + float(w[active_index] @ (cx[active_index] ** 2)))
        if dpsi0 >= 0:
            alpha = 1.0
            psi_error = -1
            it.index_alpha_upp = 0
        else:
            a_upp, index_a_upp = upper_bound_steplength(A, cx, p, W, ind_del)
            a_low = a_upp / 3000.0
            magfy = 6.0 if rankJ2 < prev_rankJ2 else 3.0
            a0 = min(1.0, magfy * previous_alpha, a_upp)
            alpha, gac_error = linesearch_constrained(x, a0, p, r, c, rx, cx, JpAp, w, W, psi0, dpsi0, a_low, a_upp)
            if gac_error:
                psi_k = psi(x, alpha, p, r, c, w, m, W.l, W.t, W.active, W.inactive)
                psi_error = check_derivatives(dpsi0, psi0, psi_k, x, alpha, p, r, c, w, W, m)
            uppbound = min(1.0, a_upp)
            atwa = float(w[active_index] @ (active_Ap ** 2))
            it.predicted_reduction = uppbound * (-2.0 * float(Jp @ rx) - uppbound * float(Jp @ Jp)
                                                 + (2.0 - uppbound ** 2) * atwa)
            rx_new, cx_new = np.zeros(m), np.zeros(W.l)
            x_new = x + alpha * p
            r.eval_into(x_new, rx_new)
            c.eval_into(x_new, cx_new)
            whsum = float(w[active_index] @ (cx_new[active_index] ** 2))
            it.progress = 2 * psi0 - float(rx_new @ rx_new) - whsum
            it.index_alpha_upp = 0 if (index_a_upp != 0 and abs(alpha - a_upp) > 0.1) else index_a_upp
    else:
        w = w_old
        it.index_alpha_upp = 0
        alpha = 1.0
    return alpha, w, psi_error


# ---------------------------------------------------------------------------------------------
# termination (src/enlsip_functions.jl:540-564, 2399-2517)
# ---------------------------------------------------------------------------------------------
def minmax_lagrangian_mult(lam, W, active_C):
    q, t = W.q, W.t
    sq_rel = SQRT_EPS
    lam_abs_max = 0.0
    sigmin = math.inf
    if t > q:
        lam_abs_max = float(np.max(np.abs(lam)))
        rows = (1.0 / active_C.diag_scale) if active_C.scaling else active_C.diag_scale
        for i in range(q + 1, t + 1):
            li = lam[i - 1]
            if li * rows[i - 1] <= -sq_rel and li < sigmin:
                sigmin = li
    return sigmin, lam_abs_max


def check_termination_criteria(it: Iteration, prev_iter: Iteration, W, active_C, x, cx, rx_sum, grad_fx, max_iter,
                               nb_iter, eps_abs, eps_rel, eps_x, eps_c, error_code, dtime, sigma_min, lam_abs_max,
                               psi_error):
    exit_code = 0
    alfnoi = EPS / (float(np.linalg.norm(it.p)) + EPS)
    preliminary = not (it.restart or (it.code == -1 and alfnoi <= 0.25))
    if preliminary:
        necessary = (not it.delete) and (float(np.linalg.norm(active_C.cx)) < eps_c) and \
                    (it.grad_res < math.sqrt(eps_rel) * (1 + float(np.linalg.norm(grad_fx))))
        if W.l - W.t > 0:
            inact = np.asarray(W.inactive[: W.l - W.t], dtype=np.int64) - 1
            necessary = necessary and bool(np.all(cx[inact] > 0))
        if W.t > W.q:
            factor = (1 + rx_sum) if W.t == 1 else lam_abs_max
            necessary = necessary and (sigma_min >= eps_rel * factor)
        if necessary:
            d1 = it.d_gn[: max(it.dimJ2, 0)]
            x_diff = float(np.linalg.norm(prev_iter.x - x))
            if float(d1 @ d1) <= rx_sum * eps_rel ** 2:
                exit_code += 10000
            if rx_sum <= eps_abs ** 2:
                exit_code += 2000
            if x_diff < eps_x * float(np.linalg.norm(x)):
                exit_code += 300
            if alfnoi > 0.25:
                exit_code += 40
            if exit_code > 0 and W.l - W.t > 0:
                feas = 1
                for ii in range(W.l - W.t):
                    jj = int(W.inactive[ii])
                    if cx[jj - 1] <= 0.0:
                        feas = -1
                        break
                exit_code *= feas
    if exit_code == 0:
        x_diff = float(np.linalg.norm(prev_iter.x - x))
        Atcx_nrm = float(np.linalg.norm(active_C.A.T @ active_C.cx))
        act = np.asarray(W.active[: W.t], dtype=np.int64) - 1
        aps = 0.0 if W.t == 0 else float(it.w[act] @ it.w[act])
        if nb_iter >= max_iter:
            exit_code = -2
        elif -5 <= error_code <= -3:
            exit_code = error_code
        elif it.nb_newton_steps > 5:
            exit_code = -9
        elif psi_error == -1:
            exit_code = -6
        elif x_diff <= 10.0 * eps_x and Atcx_nrm <= 10.0 * eps_c and aps >= 1.0:
            exit_code = -10
        elif dtime > 0:
            exit_code = -11
    return exit_code


# ---------------------------------------------------------------------------------------------
# enlsip (src/enlsip_functions.jl:2638-2880) and solve! defaults (src/solver.jl:62-81)
# ---------------------------------------------------------------------------------------------
@dataclass
class EnlsipResult:
    exit_code: int
    x: np.ndarray
    f: float
    iterations: int
    nb_subproblem_solves: int
    trace: List[dict] = field(default_factory=list)


def enlsip(x0, r: EvalFunction, c: EvalFunction, n, m, q, l, backend, scaling=False, second_derivatives=True,
           weight_code=2, MAX_ITER=100, TIME_LIMIT=1000.0, eps_abs=1e-10, eps_rel=1e-5, eps_x=1e-3, eps_c=1e-4,
           eps_rank=1e-10) -> EnlsipResult:
    ws_mod = backend.ws
    second_derivatives = second_derivatives and (n + m < 1000)
    nb_iteration = 0
    K = [np.zeros(l) for _ in range(4)]
    rx, cx = np.zeros(m), np.zeros(l)
    J, A = np.zeros((m, n)), np.zeros((l, n))
    x0 = np.asarray(x0, dtype=np.float64).copy()
    new_point(x0, r, c, rx, cx, J, A)
    first_iter = Iteration(x0, np.zeros(n), rx, cx, l, 1.0, 0, np.zeros(l), np.zeros(l), 0, 0, 0, 0, np.zeros(n),
                           np.zeros(n), 0.0, 0.0, 0.0, 0.0, 0.0, False, True, False, False, 0, 1, 0)
    start_time = time.time()
    W = init_working_set(cx, K, first_iter, q, l, ws_mod)
    first_iter.t = W.t
    act = np.asarray(W.active[: W.t], dtype=np.int64) - 1
    active_C = ws_mod.Constraint(cx[act].copy(), A[act, :].copy(), scaling, np.zeros(W.t))
    grad_fx = J.T @ rx
    p_gn = np.zeros(n)
    trace = []
    nsolve = [0]

    def uws(it_):
        backend.evaluate_scaling(active_C)
        return backend.update_working_set(W, rx, A, active_C, grad_fx, J, p_gn, it_, eps_rank,
                                          lambda: nsolve.__setitem__(0, nsolve[0] + 1))

    F_A, F_L11, F_J2 = uws(first_iter)
    rx_sum = float(rx @ rx)
    act = np.asarray(W.active[: W.t], dtype=np.int64) - 1
    active_cx_sum = float(cx[act] @ cx[act])
    first_iter.t = W.t
    previous_iter = first_iter.copy()
    error_code = search_direction_analys(previous_iter, first_iter, nb_iteration, x0, c, r, rx, cx, active_C,
                                         active_cx_sum, p_gn, J, W, F_A, F_L11, F_J2, second_derivatives, backend)
    alpha, w, psi_error = compute_steplength(first_iter, previous_iter, x0, r, rx, J, c, cx, A, active_C, W, K, weight_code)
    first_iter.alpha = alpha
    first_iter.w = w
    x = x0 + alpha * first_iter.p
    trace.append(dict(iter=0, code=first_iter.code, t=W.t, rankA=first_iter.rankA, rankJ2=first_iter.rankJ2,
                      alpha=alpha, pnorm=float(np.linalg.norm(first_iter.p)), f=rx_sum))
    new_point(x, r, c, rx, cx, J, A)
    grad_fx = J.T @ rx
    rx_sum = float(rx @ rx)
    first_iter.restart = error_code < 0
    sigma_min, lam_abs_max = minmax_lagrangian_mult(first_iter.lam, W, active_C)
    dtime = (time.time() - start_time) - TIME_LIMIT
    exit_code = check_termination_criteria(first_iter, previous_iter, W, active_C, x, cx, rx_sum, grad_fx, MAX_ITER,
                                           nb_iteration, eps_abs, eps_rel, eps_x, eps_c, error_code, dtime, sigma_min,
                                           lam_abs_max, psi_error)
    first_iter.add = evaluate_violated_constraints(cx, W, first_iter.index_alpha_upp, n, ws_mod)
    act = np.asarray(W.active[: W.t], dtype=np.int64) - 1
    active_C.cx = cx[act].copy()
    active_C.A = A[act, :].copy()
    previous_iter = first_iter.copy()
    first_iter.x = x
    first_iter.rx = rx
    first_iter.cx = cx
    nb_iteration += 1
    it = first_iter.copy()
    it.first = False
    it.add = False
    it.delete = False
    x_opt, f_opt = x, float(rx @ rx)
    while exit_code == 0:
        p_gn[:] = 0.0
        F_A, F_L11, F_J2 = uws(it)
        act = np.asarray(W.active[: W.t], dtype=np.int64) - 1
        active_cx_sum = float(cx[act] @ cx[act])
        it.t = W.t
        error_code = search_direction_analys(previous_iter, it, nb_iteration, x, c, r, rx, cx, active_C,
                                             active_cx_sum, p_gn, J, W, F_A, F_L11, F_J2, second_derivatives, backend)
        alpha, w, psi_error = compute_steplength(it, previous_iter, x, r, rx, J, c, cx, A, active_C, W, K, weight_code)
        it.alpha = alpha
        it.w = w
        trace.append(dict(iter=nb_iteration, code=it.code, t=W.t, rankA=it.rankA, rankJ2=it.rankJ2, alpha=alpha,
                          pnorm=float(np.linalg.norm(it.p)), f=float(rx @ rx)))
        x = x + alpha * it.p
        new_point(x, r, c, rx, cx, J, A)
        rx_sum = float(rx @ rx)
        grad_fx = J.T @ rx
        it.restart = error_code < 0
        sigma_min, lam_abs_max = minmax_lagrangian_mult(it.lam, W, active_C)
        dtime = (time.time() - start_time) - TIME_LIMIT
        exit_code = check_termination_criteria(it, previous_iter, W, active_C, x, cx, rx_sum, grad_fx, MAX_ITER,
                                               nb_iteration, eps_abs, eps_rel, eps_x, eps_c, error_code, dtime,
                                               sigma_min, lam_abs_max, psi_error)
        if exit_code == 0:
            it.add = evaluate_violated_constraints(cx, W, it.index_alpha_upp, n, ws_mod)
            act = np.asarray(W.active[: W.t], dtype=np.int64) - 1
            active_C.cx = cx[act].copy()
            active_C.A = A[act, :].copy()
            nb_iteration += 1
            previous_iter = it.copy()
            it.x = x
            it.rx = rx
            it.cx = cx
            it.delete = False
            it.add = False
        else:
            x_opt = x
            f_opt = float(rx @ rx)
    return EnlsipResult(exit_code, x_opt, f_opt, nb_iteration, nsolve[0], trace)


def solve(residuals, jac_res, n, m, x0, eq=None, jac_eq=None, nb_eq=0, ineq=None, jac_ineq=None, nb_ineq=0,
          x_low=None, x_upp=None, backend=None, max_iter=100, scaling=False, time_limit=1e3,
          abs_tol=EPS, rel_tol=None, c_tol=None, x_tol=None) -> EnlsipResult:
    """solve!(model) of src/solver.jl:62-91 with the constraint ordering of src/cnls_model.jl:410-496:
    [eq ; ineq ; x - x_low (finite) ; x_upp - x (finite)]."""
    rel_tol = math.sqrt(abs_tol) if rel_tol is None else rel_tol
    c_tol = rel_tol if c_tol is None else c_tol
    x_tol = rel_tol if x_tol is None else x_tol
    x_low = np.full(n, -np.inf) if x_low is None else np.asarray(x_low, dtype=np.float64)
    x_upp = np.full(n, np.inf) if x_upp is None else np.asarray(x_upp, dtype=np.float64)
    lo_idx = [i for i in range(n) if np.isfinite(x_low[i])]
    up_idx = [i for i in range(n) if np.isfinite(x_upp[i])]
    I = np.eye(n)

    def cons(x):
        parts = []
        if eq is not None:
            parts.append(np.atleast_1d(eq(x)))
        if ineq is not None:
            parts.append(np.atleast_1d(ineq(x)))
        parts.append((x - x_low)[lo_idx])
        parts.append((x_upp - x)[up_idx])
        return np.concatenate(parts) if parts else np.zeros(0)

    def jac_cons(x):
        parts = []
        if eq is not None:
            parts.append(np.atleast_2d(jac_eq(x)))
        if ineq is not None:
            parts.append(np.atleast_2d(jac_ineq(x)))
        parts.append(I[lo_idx, :])
        parts.append(-I[up_idx, :])
        return np.vstack(parts) if parts else np.zeros((0, n))

    l = nb_eq + nb_ineq + len(lo_idx) + len(up_idx)
    r = EvalFunction(lambda x: np.asarray(residuals(x), dtype=np.float64), lambda x: np.asarray(jac_res(x), dtype=np.float64))
    c = EvalFunction(cons, jac_cons)
    return enlsip(x0, r, c, n, m, nb_eq, l, backend, scaling=scaling, MAX_ITER=max_iter, TIME_LIMIT=time_limit,
                  eps_rel=rel_tol, eps_x=x_tol, eps_c=c_tol, eps_rank=SQRT_EPS)


# ---------------------------------------------------------------------------------------------
# the CPU backend: the oracle's own subproblem
# ---------------------------------------------------------------------------------------------
class OracleBackend:
    """update_working_set / sub_search_direction through oracle/gn_oracle.py (LAPACK)."""

    def __init__(self):
        from oracle import gn_oracle as go

        class _WS:
            WorkingSet = go.WorkingSet
            Constraint = go.Constraint
            remove_constraint = staticmethod(go.remove_constraint)
            add_constraint = staticmethod(go.add_constraint)

        self.go = go
        self.ws = _WS

    def evaluate_scaling(self, C):
        self.go.evaluate_scaling(C)

    def update_working_set(self, W, rx, A, C, grad_fx, J, p_gn, it: Iteration, eps_rank, on_solve):
        rec = self.go.IterationRecord()
        rec.delete, rec.index_del = it.delete, it.index_del     # iter_k keeps these unless a branch overwrites them
        F = self.go.update_working_set(W, rx, A, C, grad_fx, J, p_gn, rec, eps_rank, solve=on_solve)
        it.rankA, it.rankJ2, it.dimA, it.dimJ2 = rec.rankA, rec.rankJ2, rec.dimA, rec.dimJ2
        it.b_gn, it.d_gn, it.grad_res, it.lam = rec.b_gn, rec.d_gn, rec.grad_res, rec.lam
        it.delete, it.index_del = rec.delete, rec.index_del
        return F

    def rmul_Q1(self, J, F_A):
        return F_A.rmul_Q(J)

    def sub_search_direction(self, J1, rx, cx, F_A, F_L11, F_J2, n, t, rankA, dimA, dimJ2, code):
        return self.go.sub_search_direction(J1, rx, cx, F_A, F_L11, F_J2, n, t, rankA, dimA, dimJ2, code)
