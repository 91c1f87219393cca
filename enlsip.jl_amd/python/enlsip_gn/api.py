"""Host-side mirror of the reference's interface for the GN subproblem, over the C ABI.

``GNSolver.solve`` has the argument meaning of the reference's
``gn_search_direction(J, rx, cx, F_A, F_L11, rankA, t, ε_rank, iter)`` preceded by the QR lines
of ``update_working_set`` (src/enlsip_functions.jl:206-234, :700, :768-769) and returns the
quantities those lines write into ``Iteration`` (rankA, rankJ2, dimA, dimJ2, b_gn, d_gn) plus the
three ``QRPivoted``-like factor views whose accessors (``.R``, ``.p``, ``Qt_mul``, ``Q_mul``) are
backed by the device-resident factors.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L

SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)


class GNError(RuntimeError):
    pass


def route_names(lib=None) -> list:
    """Every route bit the library knows, in bit order (enlsip_gn_route_name)."""
    lib = lib or L.load()
    names, bit = [], 0
    while True:
        nm = lib.enlsip_gn_route_name(bit)
        if nm is None:
            return names
        names.append(nm.decode())
        bit += 1


def _fptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class GNResult:
    p: np.ndarray
    b: np.ndarray
    d: np.ndarray
    rankA: int
    rankJ2: int
    code: int
    dimA: int
    dimJ2: int
    status: int
    jpvtA: np.ndarray
    jpvtL: np.ndarray
    jpvtJ2: np.ndarray


class FactorView:
    """QRPivoted-like view of one resident factorisation (valid until the next solve)."""

    def __init__(self, solver: "GNSolver", which: int, prob: int = 0):
        self._s, self._which, self._prob = solver, which, prob

    @property
    def shape(self):
        r, c = C.c_int64(), C.c_int64()
        self._s._chk(self._s._lib.enlsip_gn_factor_shape(self._s._h, self._which, self._prob, C.byref(r), C.byref(c)))
        return int(r.value), int(c.value)

    @property
    def R(self) -> np.ndarray:
        r, c = self.shape
        out = np.zeros((max(r, 1), c), order="F")
        self._s._chk(self._s._lib.enlsip_gn_get_R(self._s._h, self._which, self._prob, _fptr(out), max(r, 1)))
        return out[:r, :]

    def diagR(self) -> np.ndarray:
        r, c = self.shape
        out = np.zeros(min(r, c))
        if out.size:
            self._s._chk(self._s._lib.enlsip_gn_get_diagR(self._s._h, self._which, self._prob, _fptr(out)))
        return out

    @property
    def p(self) -> np.ndarray:
        _, c = self.shape
        out = np.zeros(c, dtype=np.int64)
        if c:
            self._s._chk(self._s._lib.enlsip_gn_get_jpvt(self._s._h, self._which, self._prob, _fptr(out)))
        return out

    def Qt_mul(self, v: np.ndarray) -> np.ndarray:
        out = np.array(v, dtype=np.float64, copy=True)
        self._s._chk(self._s._lib.enlsip_gn_apply_qt(self._s._h, self._which, self._prob, _fptr(out)))
        return out

    def Q_mul(self, v: np.ndarray) -> np.ndarray:
        out = np.array(v, dtype=np.float64, copy=True)
        self._s._chk(self._s._lib.enlsip_gn_apply_q(self._s._h, self._which, self._prob, _fptr(out)))
        return out


class GNSolver:
    """One handle = one HIP stream + device workspace (not thread-safe)."""

    def __init__(self, device: int = -1, flags: int = 0, tile_rows: int = 0, stream: int = 0):
        self._lib = L.load()
        self._h = C.c_void_p()
        opts = L.Opts(device=device, flags=flags, panel_width=0, tile_rows=tile_rows,
                      stream=C.c_void_p(stream) if stream else None)
        rc = self._lib.enlsip_gn_create(C.byref(self._h), C.byref(opts))
        if rc != 0:
            msg = self._lib.enlsip_gn_last_error(None)
            raise GNError(f"enlsip_gn_create failed with code {rc}: {msg.decode() if msg else 'no usable HIP device?'}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.enlsip_gn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            msg = self._lib.enlsip_gn_last_error(self._h)
            raise GNError(f"libenlsip_gn error {rc}: {msg.decode() if msg else ''}")

    # ---- single problem, host buffers -------------------------------------------------------
    def solve(self, J: np.ndarray, rx: np.ndarray, A_active: np.ndarray, cx: np.ndarray,
              eps_rank: float = SQRT_EPS, dimA: int = -1, dimJ2: int = -1) -> GNResult:
        """A_active is the t x n active-constraint Jacobian (the reference's ``C.A``)."""
        J = np.asfortranarray(J, dtype=np.float64)
        m, n = J.shape
        A_active = np.asarray(A_active, dtype=np.float64).reshape(-1, n)
        t = A_active.shape[0]
        At = np.asfortranarray(A_active.T)                  # n x t column-major
        rx = np.ascontiguousarray(rx, dtype=np.float64)
        cx = np.ascontiguousarray(cx, dtype=np.float64)
        kA = min(n, t)
        p, b, d = np.zeros(n), np.zeros(t), np.zeros(m)
        jA, jL, jJ = np.zeros(t, np.int64), np.zeros(kA, np.int64), np.zeros(n, np.int64)
        info = L.Info()
        self._chk(self._lib.enlsip_gn_solve(self._h, m, n, t, _fptr(J), m, _fptr(rx), _fptr(At) if t else None,
                                            max(n, 1), _fptr(cx) if t else None, eps_rank, dimA, dimJ2,
                                            _fptr(p), _fptr(b), _fptr(d), C.byref(info),
                                            _fptr(jA), _fptr(jL), _fptr(jJ)))
        n2 = n - int(info.rankA)
        return GNResult(p, b, d, int(info.rankA), int(info.rankJ2), int(info.code), int(info.dimA),
                        int(info.dimJ2), int(info.status), jA, jL, jJ[:n2].copy())

    # ---- batch, host buffers ----------------------------------------------------------------
    def factor_constraints(self, m: int, A: np.ndarray, cx: np.ndarray, eps_rank: float = SQRT_EPS):
        """Constraint stage alone (src/enlsip_functions.jl:700, :768-769): F_A, rankA, F_L11 left resident for
        first_lagrange (pass grad_fx) and the F_A / F_L11 accessors.  A: (t, n) active constraint Jacobian; m: rows of the
        solve that follows.  Returns (rankA, code, dimA)."""
        A = np.asarray(A, dtype=np.float64)
        t, n = (A.shape if A.size else (0, A.shape[1] if A.ndim == 2 else 0))
        At = np.ascontiguousarray(A) if t else None          # (t, n) C-order == column-major n x t
        cxv = np.ascontiguousarray(cx, dtype=np.float64) if t else None
        info = L.Info()
        self._chk(self._lib.enlsip_gn_factor_constraints(self._h, m, n, t, _fptr(At) if t else None, max(n, 1),
                                                         _fptr(cxv) if t else None, eps_rank, C.byref(info)))
        return int(info.rankA), int(info.code), int(info.dimA)

    def solve_factored(self, J: np.ndarray, rx: np.ndarray, t: int, eps_rank: float = SQRT_EPS, dimJ2: int = -1) -> GNResult:
        """The solve right after factor_constraints with an unchanged working set (src/enlsip_functions.jl:768-771 reuses the
        F_A of :700): only J and rx go in."""
        J = np.asfortranarray(J, dtype=np.float64)
        m, n = J.shape
        rx = np.ascontiguousarray(rx, dtype=np.float64)
        kA = min(n, t)
        p, b, d = np.zeros(n), np.zeros(t), np.zeros(m)
        jA, jL, jJ = np.zeros(t, np.int64), np.zeros(kA, np.int64), np.zeros(n, np.int64)
        info = L.Info()
        self._chk(self._lib.enlsip_gn_solve_factored(self._h, m, n, t, _fptr(J), m, _fptr(rx), eps_rank, dimJ2,
                                                     _fptr(p), _fptr(b), _fptr(d), C.byref(info), _fptr(jA), _fptr(jL), _fptr(jJ)))
        n2 = n - int(info.rankA)
        return GNResult(p, b, d, int(info.rankA), int(info.rankJ2), int(info.code), int(info.dimA),
                        int(info.dimJ2), int(info.status), jA, jL, jJ[:n2].copy())

    def solve_batched(self, J: np.ndarray, rx: np.ndarray, At: np.ndarray, cx: np.ndarray,
                      eps_rank: float = SQRT_EPS):
        """J: (batch, n, m) C-order array holding each m x n problem column-major (i.e. J[k].T is
        the matrix), rx: (batch, m), At: (batch, t, n) C-order = column-major n x t, cx: (batch, t)."""
        batch, n, m = J.shape
        t = At.shape[1] if At is not None and At.size else 0
        kA = min(n, t)
        J = np.ascontiguousarray(J, dtype=np.float64)
        rx = np.ascontiguousarray(rx, dtype=np.float64)
        p, b, d = np.zeros((batch, n)), np.zeros((batch, t)), np.zeros((batch, m))
        jA = np.zeros((batch, t), np.int64)
        jL = np.zeros((batch, kA), np.int64)
        jJ = np.zeros((batch, n), np.int64)
        info = (L.Info * batch)()
        if t:
            At = np.ascontiguousarray(At, dtype=np.float64)
            cx = np.ascontiguousarray(cx, dtype=np.float64)
        self._chk(self._lib.enlsip_gn_solve_batched(
            self._h, batch, m, n, t, _fptr(J), m, m * n, _fptr(rx), _fptr(At) if t else None, max(n, 1), n * t,
            _fptr(cx) if t else None, eps_rank, _fptr(p), _fptr(b), _fptr(d),
            C.cast(info, C.c_void_p), _fptr(jA), _fptr(jL), _fptr(jJ)))
        infos = [(int(i.rankA), int(i.rankJ2), int(i.code), int(i.dimA), int(i.dimJ2), int(i.status)) for i in info]
        return p, b, d, infos, jA, jL, jJ

    # ---- batch, device buffers (raw pointers, e.g. torch tensor .data_ptr()) ------------------
    def solve_batched_dev(self, batch, m, n, t, dJ, ldj, strideJ, drx, dAt, ldat, strideAt, dcx,
                          eps_rank=SQRT_EPS, dp=0, db=0, dd=0, dinfo=0, djA=0, djL=0, djJ=0):
        v = lambda x: C.c_void_p(x) if x else None
        self._chk(self._lib.enlsip_gn_solve_batched_dev(
            self._h, batch, m, n, t, v(dJ), ldj, strideJ, v(drx), v(dAt), ldat, strideAt, v(dcx), eps_rank,
            v(dp), v(db), v(dd), v(dinfo), v(djA), v(djL), v(djJ)))

    # ---- resident factors ---------------------------------------------------------------------
    def factor(self, which: int, prob: int = 0) -> FactorView:
        return FactorView(self, which, prob)

    def JQ1(self, m: int, n: int, prob: int = 0) -> np.ndarray:
        out = np.zeros((m, n), order="F")
        self._chk(self._lib.enlsip_gn_get_JQ1(self._h, prob, _fptr(out), m))
        return out

    def resolve(self, m: int, n: int, t: int, dimA: int, dimJ2: int, code: int = -1, prob: int = 0):
        """sub_search_direction re-entry on the resident factors (src/enlsip_functions.jl:1253)."""
        p, b, d = np.zeros(n), np.zeros(t), np.zeros(m)
        self._chk(self._lib.enlsip_gn_resolve(self._h, prob, dimA, dimJ2, code, _fptr(p), _fptr(b), _fptr(d)))
        return p, b, d

    # ---- multiplier estimates on the resident data (src/enlsip_functions.jl:461-537, :2690) -----------
    def gradient(self, n: int, prob: int = 0) -> np.ndarray:
        g = np.zeros(n)
        self._chk(self._lib.enlsip_gn_gradient(self._h, prob, _fptr(g)))
        return g

    def jacobian_times(self, m: int, t: int, p: np.ndarray, prob: int = 0):
        """(J p, A_active p) of the last solve's J and A (src/enlsip_functions.jl:2226-2229)."""
        pv = np.ascontiguousarray(p, dtype=np.float64)
        Jp, Ap = np.zeros(m), np.zeros(t)
        self._chk(self._lib.enlsip_gn_jacobian_times(self._h, prob, _fptr(pv), _fptr(Jp), _fptr(Ap) if t else None))
        return Jp, Ap

    def full_constraints_times(self, A: np.ndarray, p: np.ndarray) -> np.ndarray:
        """A p with the FULL constraint Jacobian (l x n), src/enlsip_functions.jl:2227."""
        Af = np.asfortranarray(A, dtype=np.float64)
        l, n = Af.shape
        pv = np.ascontiguousarray(p, dtype=np.float64)
        Ap = np.zeros(l)
        if l:
            self._chk(self._lib.enlsip_gn_full_constraints_times(self._h, l, n, _fptr(Af), l, _fptr(pv), _fptr(Ap)))
        return Ap

    def matrix_times_QA(self, M: np.ndarray, prob: int = 0) -> np.ndarray:
        """M * F_A.Q for a host matrix with the row count of the last solve (`J * F_A.Q`, src/enlsip_functions.jl:526, :1249)."""
        Mf = np.asfortranarray(M, dtype=np.float64)
        rows, n = Mf.shape
        out = np.zeros((rows, n), order="F")
        self._chk(self._lib.enlsip_gn_matrix_times_QA(self._h, prob, rows, _fptr(Mf), max(rows, 1), _fptr(out), max(rows, 1)))
        return out

    def first_lagrange(self, t: int, grad_fx: Optional[np.ndarray] = None, diag_scale: Optional[np.ndarray] = None,
                       eps_rank: float = SQRT_EPS, prob: int = 0):
        """first_lagrange_mult_estimate!: returns (lambda, grad_res)."""
        lam = np.zeros(t)
        gres = C.c_double(0.0)
        g = None if grad_fx is None else np.ascontiguousarray(grad_fx, dtype=np.float64)
        ds = None if diag_scale is None else np.ascontiguousarray(diag_scale, dtype=np.float64)
        self._chk(self._lib.enlsip_gn_first_lagrange(self._h, prob, _fptr(g), _fptr(ds), eps_rank, _fptr(lam),
                                                     C.byref(gres)))
        return lam, float(gres.value)

    def second_lagrange(self, t: int, p_gn: np.ndarray, diag_scale: Optional[np.ndarray] = None,
                        eps_rank: float = SQRT_EPS, prob: int = 0) -> np.ndarray:
        """second_lagrange_mult_estimate!: lambda from the resident J1 = (J*F_A.Q)[:, 1:t]."""
        lam = np.zeros(t)
        p = np.ascontiguousarray(p_gn, dtype=np.float64)
        ds = None if diag_scale is None else np.ascontiguousarray(diag_scale, dtype=np.float64)
        self._chk(self._lib.enlsip_gn_second_lagrange(self._h, prob, _fptr(p), _fptr(ds), eps_rank, _fptr(lam)))
        return lam

    def newton_direction(self, Gamma: np.ndarray, prob: int = 0):
        """newton_search_direction (src/enlsip_functions.jl:348-423) after its Hessian sums: Gamma = r_mat - c_mat (n x n).
        Returns (p, error) as the reference does (error = True: W22 not positive definite, p = 0)."""
        G = np.asfortranarray(Gamma, dtype=np.float64)
        n = G.shape[0]
        p = np.zeros(n)
        bad = C.c_int64(0)
        self._chk(self._lib.enlsip_gn_newton_direction(self._h, prob, _fptr(G), n, _fptr(p), C.byref(bad)))
        return p, bool(bad.value)

    # ---- instrumentation ------------------------------------------------------------------------
    def set_profiling(self, on: bool, all_updates: bool = False):
        """HIP-event timing of the stages and the level-0 far updates; all_updates: every trailing-update launch (update_totals)."""
        self._chk(self._lib.enlsip_gn_set_profiling(self._h, (2 if all_updates else 1) if on else 0))

    def stage_ms(self) -> dict:
        arr = (C.c_float * len(L.STAGE_NAMES))()
        self._chk(self._lib.enlsip_gn_get_stage_ms(self._h, arr))
        return dict(zip(L.STAGE_NAMES, [float(x) for x in arr]))

    def update_stats(self):
        ms, cnt, by = C.c_float(), C.c_int64(), C.c_double()
        self._chk(self._lib.enlsip_gn_get_update_stats(self._h, C.byref(ms), C.byref(cnt), C.byref(by)))
        return float(ms.value), int(cnt.value), float(by.value)

    def update_table(self):
        """Per timed level-0 far-update launch of the last profiled solve: [(SURVEY 8d bytes, ms), ...] in sweep order."""
        cap = 256
        by = (C.c_double * cap)()
        ms = (C.c_float * cap)()
        cnt = C.c_int64(0)
        self._chk(self._lib.enlsip_gn_get_update_table(self._h, cap, by, ms, C.byref(cnt)))
        return [(float(by[i]), float(ms[i])) for i in range(min(int(cnt.value), cap))]

    def launch_plan(self):
        """(pipeline_split, panel_pairs, tile_rows) of the last solve: what the library chose on its own."""
        sp, pr, tr = C.c_int64(), C.c_int(), C.c_int64()
        self._chk(self._lib.enlsip_gn_get_launch_plan(self._h, C.byref(sp), C.byref(pr), C.byref(tr)))
        return int(sp.value), bool(pr.value), int(tr.value)

    def route(self) -> set:
        """Names of the kernel-selection branches the last solve took (enlsip_gn_get_route; include/enlsip_gn.h ENLSIP_GN_ROUTE_*)."""
        mask = C.c_uint64(0)
        self._chk(self._lib.enlsip_gn_get_route(self._h, C.byref(mask)))
        return {name for bit, name in enumerate(route_names(self._lib)) if (mask.value >> bit) & 1}

    def pipeline_split(self) -> int:
        return self.launch_plan()[0]

    def plan_uses_pairs(self) -> bool:
        return self.launch_plan()[1]

    def update_totals(self):
        """(far_ms, other_ms, other_launches, all_panels_bytes) of the last profiled solve: HIP-event time of the level-0 far
        passes and of every other trailing-update launch (tree levels, second-panel columns); SURVEY 8d bytes of all panels."""
        far, oth, cnt, by = C.c_float(), C.c_float(), C.c_int64(), C.c_double()
        self._chk(self._lib.enlsip_gn_get_update_totals(self._h, C.byref(far), C.byref(oth), C.byref(cnt), C.byref(by)))
        return float(far.value), float(oth.value), int(cnt.value), float(by.value)

    def measure_stream(self, nbytes: int = 1 << 30, reps: int = 5) -> float:
        """GB/s of an in-place non-temporal read-modify-write stream with the trailing update's access shape on this device."""
        out = C.c_double(0.0)
        self._chk(self._lib.enlsip_gn_measure_stream(self._h, nbytes, reps, C.byref(out)))
        return float(out.value)

    def synchronize(self):
        self._chk(self._lib.enlsip_gn_synchronize(self._h))
