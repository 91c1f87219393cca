"""Plain-NumPy restatement of the LAPACK semantics the hot path relies on.

TEST INFRASTRUCTURE (see oracle/gn_oracle.py header).  Julia's ``qr(·, ColumnNorm())`` is
LAPACK ``dgeqp3`` (third-party, OpenBLAS_jll 0.3.23+4 per docs/Manifest.toml:296-299; not
under /root/reference), so its published algorithm is restated here from the LAPACK 3.x
specification: ``dlarfg`` (reflector generation), ``dlaqp2`` (unblocked column-pivoted
step with the Drmac-Bujanovic norm downdate), ``dlarft`` (forward, columnwise T factor).
tests/test_oracle_lapack_semantics.py checks these against scipy's real dgeqp3 so that the
HIP kernels, which follow *this* file step by step, are anchored to LAPACK behaviour.

Also holds ``gn_subproblem_factored``: the arithmetic *plan* of the HIP path (unpivoted
blocked QR of J2 followed by a pivoted QR of its R factor) in NumPy, used by the
CPU tests to show that plan reproduces dgeqp3's pivots, |diag R| and the step p.
"""
from __future__ import annotations

import math

import numpy as np

EPS = np.finfo(np.float64).eps
TOL3Z = math.sqrt(EPS)  # dlaqp2's recompute threshold


def dlarfg(alpha: float, x: np.ndarray):
    """Returns (beta, tau, scale) with v = [1; x*scale], H = I - tau v v', H[alpha;x] = [beta;0].
    Without LAPACK's safmin rescaling loop (inputs here are never that tiny)."""
    xnorm = float(np.linalg.norm(x)) if x.size else 0.0
    if xnorm == 0.0:
        return alpha, 0.0, 0.0
    beta = -math.copysign(math.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    scale = 1.0 / (alpha - beta)
    return beta, tau, scale


def geqp2(A: np.ndarray, nrhs: int = 0):
    """Unblocked column-pivoted Householder QR (dlaqp2 semantics) of A[:, :cols-nrhs]; the last
    ``nrhs`` columns are carried along (reflectors applied, never pivoted).
    Returns (factors, tau, jpvt 1-based)."""
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    rows, ctot = A.shape
    cols = ctot - nrhs
    k = min(rows, cols)
    tau = np.zeros(k)
    jpvt = np.arange(1, cols + 1, dtype=np.int64)
    vn1 = np.linalg.norm(A[:, :cols], axis=0)
    vn2 = vn1.copy()
    for j in range(k):
        pvt = j + int(np.argmax(vn1[j:cols]))          # idamax: FIRST maximal entry
        if pvt != j:
            A[:, [j, pvt]] = A[:, [pvt, j]]
            jpvt[[j, pvt]] = jpvt[[pvt, j]]
            vn1[pvt] = vn1[j]
            vn2[pvt] = vn2[j]
        beta, t_j, scale = dlarfg(A[j, j], A[j + 1:, j])
        tau[j] = t_j
        if t_j != 0.0:
            A[j + 1:, j] *= scale
        v = np.concatenate([[1.0], A[j + 1:, j]])
        if j + 1 < ctot and t_j != 0.0:
            w = v @ A[j:, j + 1:]
            A[j:, j + 1:] -= t_j * np.outer(v, w)
        A[j, j] = beta
        for c in range(j + 1, cols):                    # norm downdate (dlaqp2)
            if vn1[c] != 0.0:
                temp = 1.0 - (abs(A[j, c]) / vn1[c]) ** 2
                temp = max(temp, 0.0)
                temp2 = temp * (vn1[c] / vn2[c]) ** 2
                if temp2 <= TOL3Z:
                    if j + 1 < rows:
                        vn1[c] = float(np.linalg.norm(A[j + 1:, c]))
                        vn2[c] = vn1[c]
                    else:
                        vn1[c] = 0.0
                        vn2[c] = 0.0
                else:
                    vn1[c] *= math.sqrt(temp)
    return A, tau, jpvt


def geqr2(A: np.ndarray):
    """Unpivoted Householder QR (dgeqr2 semantics).  Returns (factors, tau)."""
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    rows, cols = A.shape
    k = min(rows, cols)
    tau = np.zeros(k)
    for j in range(k):
        beta, t_j, scale = dlarfg(A[j, j], A[j + 1:, j])
        tau[j] = t_j
        if t_j != 0.0:
            A[j + 1:, j] *= scale
            v = np.concatenate([[1.0], A[j + 1:, j]])
            if j + 1 < cols:
                w = v @ A[j:, j + 1:]
                A[j:, j + 1:] -= t_j * np.outer(v, w)
        A[j, j] = beta
    return A, tau


def unit_lower(factors: np.ndarray, k: int) -> np.ndarray:
    """V (rows x k) of the compact factors: unit diagonal, strictly-lower part of factors."""
    V = np.tril(factors[:, :k], -1)
    V[np.arange(k), np.arange(k)] = 1.0
    return V


def larft(V: np.ndarray, tau: np.ndarray) -> np.ndarray:
    """dlarft('F','C'): upper-triangular T with H_1...H_k = I - V T V'."""
    k = tau.size
    T = np.zeros((k, k))
    for j in range(k):
        T[j, j] = tau[j]
        if j > 0 and tau[j] != 0.0:
            T[:j, j] = -tau[j] * (T[:j, :j] @ (V[:, :j].T @ V[:, j]))
    return T


def apply_qt(factors: np.ndarray, tau: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Q' c with Q = H_1...H_k stored compactly (dorm2r 'L','T')."""
    c = np.array(c, dtype=np.float64, copy=True)
    for j in range(tau.size):
        if tau[j] != 0.0:
            v = np.concatenate([[1.0], factors[j + 1:, j]])
            c[j:] -= tau[j] * v * (v @ c[j:]) if c.ndim == 1 else tau[j] * np.outer(v, v @ c[j:])
    return c


def apply_q(factors: np.ndarray, tau: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Q c (dorm2r 'L','N')."""
    c = np.array(c, dtype=np.float64, copy=True)
    for j in range(tau.size - 1, -1, -1):
        if tau[j] != 0.0:
            v = np.concatenate([[1.0], factors[j + 1:, j]])
            c[j:] -= tau[j] * v * (v @ c[j:]) if c.ndim == 1 else tau[j] * np.outer(v, v @ c[j:])
    return c


def pseudo_rank(diag_T, eps_rank: float) -> int:
    """Same arithmetic as gn_oracle.pseudo_rank (src/enlsip_functions.jl:17-31); duplicated here
    so this file stands alone as the kernels' step-by-step specification."""
    l = len(diag_T)
    if l == 0 or abs(diag_T[0]) < eps_rank:
        return 0
    tol = abs(diag_T[0]) * math.sqrt(float(l)) * eps_rank
    r = 1
    while r < l and abs(diag_T[r - 1]) > tol:
        r += 1
    return r - (0 if (r == l and abs(diag_T[r - 1]) > tol) else 1)


# --------------------------------------------------------------------------------------
# The arithmetic plan of the HIP path in NumPy
# --------------------------------------------------------------------------------------
def gn_subproblem_factored(J, rx, A_active, cx, eps_rank: float):
    """Same outputs as gn_oracle.gn_subproblem, computed the way the HIP kernels do:

      F_A   = geqp2(A')                 (single-workgroup pivoted QR)
      F_L11 = geqp2(R_A')
      JQ1   = J Q1 via compact WY       (J - (J V) T V')
      J2    = Q0 R0                     (UNPIVOTED blocked QR, [J2 | d_temp] carried together)
      R0 P  = Qt Rt                     (pivoted QR of the kp x n2 factor, z = (Q0'd)[:kp] carried)
      => J2 P = (Q0 diag(Qt, I)) Rt :   same pivots / |diag| as dgeqp3(J2) in exact arithmetic.
    """
    J = np.asarray(J, dtype=np.float64)
    m, n = J.shape
    A_active = np.asarray(A_active, dtype=np.float64).reshape(-1, n)
    t = A_active.shape[0]
    fA, tauA, pA = geqp2(A_active.T)
    kA = min(n, t)
    RA = np.triu(fA[:kA, :])
    rankA = pseudo_rank(np.diag(fA[:kA, :kA]), eps_rank)
    fL, tauL, pL = geqp2(RA.T)
    kL = min(t, kA)
    RL = np.triu(fL[:kL, :])
    code = 1 if rankA == t else -1
    # J Q1 by compact WY
    if kA > 0:
        V = unit_lower(fA, kA)
        T = larft(V, tauA)
        JQ1 = J - (J @ V) @ T @ V.T
    else:
        JQ1 = J.copy()
    J1, J2 = JQ1[:, :rankA], JQ1[:, rankA:]
    n2 = n - rankA
    bA = -np.asarray(cx, dtype=np.float64)[pA - 1]
    if code == 1:
        b = bA
        p1 = np.linalg.solve(RA[:t, :t].T, b) if t else np.zeros(0)
    else:
        b = apply_qt(fL, tauL, bA)
        dimA = rankA
        dp1 = np.linalg.solve(RL[:dimA, :dimA], b[:dimA]) if dimA else np.zeros(0)
        w = np.concatenate([dp1, np.zeros(t - dimA)])
        inv = np.empty(t, dtype=np.int64)
        inv[pL - 1] = np.arange(t)
        p1 = w[inv][:rankA]
    d_temp = -J1 @ p1 - rx
    # unpivoted QR of [J2 | d_temp]
    kp = min(m, n2)
    if n2 > 0:
        f0, tau0 = geqr2(J2)
        d0 = apply_qt(f0, tau0, d_temp)
    else:
        d0 = d_temp.copy()
    R0 = np.triu(f0[:kp, :n2]) if n2 > 0 else np.zeros((0, 0))
    # pivoted QR of [R0 | d0[:kp]]
    if n2 > 0:
        ft, taut, pJ = geqp2(np.column_stack([R0, d0[:kp]]), nrhs=1)
        Rt = np.triu(ft[:kp, :n2])
        d = d0.copy()
        d[:kp] = ft[:kp, n2]
        rankJ2 = pseudo_rank(np.diag(ft[:kp, :kp]), eps_rank)
    else:
        pJ = np.zeros(0, dtype=np.int64)
        Rt = np.zeros((0, 0))
        d = d0
        rankJ2 = 0
    dimJ2 = rankJ2
    dp2 = np.linalg.solve(Rt[:dimJ2, :dimJ2], d[:dimJ2]) if dimJ2 else np.zeros(0)
    w2 = np.concatenate([dp2, np.zeros(n2 - dimJ2)])
    inv2 = np.empty(n2, dtype=np.int64)
    inv2[pJ - 1] = np.arange(n2)
    p2 = w2[inv2]
    p = apply_q(fA, tauA, np.concatenate([p1, p2])) if kA > 0 else np.concatenate([p1, p2])
    return dict(p=p, b=b, d=d, rankA=rankA, rankJ2=rankJ2, code=code, jpvtA=pA, jpvtL=pL, jpvtJ2=pJ,
                diagA=np.diag(fA[:kA, :kA]).copy(), diagL=np.diag(fL[:kL, :kL]).copy(),
                diagJ2=(np.diag(ft[:kp, :kp]).copy() if n2 > 0 else np.zeros(0)))
