// Constraint stage for SMALL problems (n <= 64, t <= 63): ONE WAVE per problem, no barriers.  Same contract as
// k_constraint (gn_kernels_constraint.hpp):
//   F_A   = qr(C.A', ColumnNorm())                 src/enlsip_functions.jl:700
//   rankA = pseudo_rank(diag(F_A.R))               :768, :17-31
//   F_L11 = qr(F_A.R', ColumnNorm())               :769
//   b, p1 of sub_search_direction                  :131-133 / :141-144
//   block T factor of Q1 (dlarft), built from the Gram entries the factorisation produces on the way
// Both factorisations run with lanes along COLUMNS (gn_wave_qrcp.hpp).  b_buff = -cx[F_A.p] rides through the F_L11
// factorisation as a carried right-hand side, which yields F_L11.Q' b_buff (:141) without a second pass.
#pragma once
#include "gn_kernels_constraint.hpp"
#include "gn_wave_qrcp.hpp"

namespace gn {

// LDS (doubles): tmp[kA * 65] vbuf[64] dg[64] gcol[64] tail[64] Tl[64 * kA]
inline size_t constraint_small_lds_bytes(int kA) { return (size_t)((kA > 0 ? kA : 1) * (65 + 64) + 4 * 64 + 8) * 8; }

template <int NR>
#ifndef ENLSIP_CS_OCC
#define ENLSIP_CS_OCC 3           // NR = 32: 168 registers without spills, 3 waves per SIMD instead of 2 (C5 constraint stage 0.117 -> 0.098 ms)
#endif
#ifndef ENLSIP_CS64_OCC
#define ENLSIP_CS64_OCC 0         // measured: 2 waves per SIMD = 420 spilled registers, C3 constraint stage 0.07 -> 0.13 ms
#endif
__global__ __launch_bounds__(64, (NR == 32 && ENLSIP_CS_OCC) ? ENLSIP_CS_OCC : ((NR == 64 && ENLSIP_CS64_OCC) ? ENLSIP_CS64_OCC : 1)) void k_constraint_small(ConstraintArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = a.n, t = a.t, kA = a.kA;
    const int kd = kA > 0 ? kA : 1;
    double* tmp = smem;
    double* vbuf = tmp + kd * 65;
    double* dg = vbuf + 64;
    double* gcol = dg + 64;
    double* tail = gcol + 64;
    double* Tl = tail + 64;
    const int prob = blockIdx.x + a.prob0;
    const double* At = a.At + prob * a.strideAt;
    const double* cx = a.cx + prob * a.stride_cx;
    double* FA = a.FA + prob * a.sFA;
    double* tauA = a.tauA + prob * a.sTauA;
    long long* jpvtA = a.jpvtA + prob * a.sJA;
    double* FL = a.FL + prob * a.sFL;
    double* tauL = a.tauL + prob * a.sTauL;
    long long* jpvtL = a.jpvtL + prob * a.sJL;
    double* p1 = a.p1 + prob * a.sP1;
    double* bvec = a.bvec + prob * a.sB;
    ProbState* st = a.state + prob;
    const int ln = threadIdx.x;
    int status = 0;

    // ---- F_A: lane c owns column c of A' (n x t) ---------------------------------------------------------------------
    double x[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) x[r] = (r < n && ln < t) ? At[r + (size_t)ln * a.ldat] : 0.0;
    WaveQrcp q{};
    q.rows = n; q.k = kA; q.ncand = t; q.npart = t;
    q.tmp = tmp; q.vbuf = vbuf; q.dg = dg; q.F = FA; q.ldf = n; q.tau = tauA; q.Tl = Tl; q.gcol = gcol;
    int mypos = ln;
    if (kA > 0) wave_qrcp<NR, true>(x, q, ln, mypos);
    if (ln < t) jpvtA[mypos] = ln + 1;
    int* lpos = reinterpret_cast<int*>(vbuf);
    wave_mem_sync();
    if (ln < t) lpos[mypos] = ln;
    wave_mem_sync();
    const int lp = (ln < t) ? lpos[ln] : 0;          // lane i: the lane (= column of A') that sits at position i
    wave_qrcp_store_upper(q, ln, lp, t, FA, n);
    if (kA > 0) {
        // block T factor (one block: kA <= 63), zero outside the upper triangle.  The compact-WY kernels of J*Q1 pad a block to 64
        // reflectors and need the zeros of all 64 columns; with at most 16 reflectors J*Q1 runs reflector by reflector
        // (k_jq1_rows*, k_jq1_factor_small: the diagonal; launch_jq1 of the accessors: the upper triangle of the first kA columns),
        // so only those columns are written — 30 of the 32 KB per problem at C5 (t = 4) were zeros nobody reads: 250 MB per step
        double* T = a.TA + prob * a.sTA;
        const int jcols = jq1_by_reflectors(n, kA) ? kA : KBLK;       // the predicate of launch_jq1_rows
        for (int jc = 0; jc < jcols; ++jc) T[ln + jc * KBLK] = (jc < kA && ln <= jc) ? Tl[ln + 64 * jc] : 0.0;
    }
    const int rankA = wave_pseudo_rank(dg, kA, a.eps_rank, ln, pseudo_rank_abs_threshold(a.eps_rank, a.abs_shift));
    int code = (rankA == t) ? 1 : -1;
    if (a.code_override != 0) code = a.code_override;
    int dimA = (a.dimA_override >= 0) ? a.dimA_override : rankA;
    dimA = dimA < kA ? dimA : kA;         // as k_constraint: never past the factor

    // b_buff = -cx[F_A.p]   (lane i: entry i)
    const double bb = (ln < t) ? -cx[lp] : 0.0;
    if (code == 1) {
        // b = b_buff; p1 = LowerTriangular(R') \ b, with L[i][k] = R[k][i] = tmp[k][lane at position i]     (:132-133)
        if (ln < t) bvec[ln] = bb;
        double y = bb;
        const int dim = t < kA ? t : kA;
        for (int k = 0; k < dim; ++k) {
            const int lk = __builtin_amdgcn_readlane(lp, k);
            const double dkk = tmp[k * 65 + lk];
            if (dkk == 0.0) status |= 1;
            const double yk = wave_bcast(y, k) / dkk;
            if (ln == k) y = yk;
            if (ln > k && ln < t) y -= tmp[k * 65 + lp] * yk;
        }
        if (ln < t) p1[ln] = y;
    }

    // ---- F_L11: L11 = R_A' (t x kA, lower trapezoid); lane j owns column j, lane kA carries b_buff -------------------------
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int li = __builtin_amdgcn_readlane(lp, i);      // lane at position i (column i of R_A)
        const double bi = wave_bcast(bb, i);
        double v = 0.0;
        if (i < t) {
            if (ln < kA) v = (i >= ln) ? tmp[ln * 65 + li] : 0.0;
            else if (ln == kA) v = bi;
        }
        x[i] = v;
    }
    wave_mem_sync();          // every lane has taken its column of L11 out of tmp before the second factorisation reuses it
    WaveQrcp ql{};
    ql.rows = t; ql.k = kA; ql.ncand = kA; ql.npart = kA + 1;
    ql.tmp = tmp; ql.vbuf = vbuf; ql.dg = dg; ql.F = FL; ql.ldf = t; ql.tau = tauL;
    int posL = ln, j0 = 0;
    if (kA > 0) j0 = wave_qrcp<NR, false>(x, ql, ln, posL);
    if (ln < kA) jpvtL[posL] = ln + 1;
    wave_mem_sync();
    if (ln <= kA) lpos[posL] = ln;
    wave_mem_sync();
    const int lpL = (ln <= kA) ? lpos[ln] : 0;
    wave_qrcp_store_upper(ql, ln, lpL, kA, FL, t);
    if (code == -1) {
        // b = F_L11.Q' b_buff: rows < kA are the finished rows of the carried column, rows >= kA are still in its registers
        if (ln == kA) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (j0 + r >= kA && j0 + r < t) tail[j0 + r] = x[r];
        }
        wave_mem_sync();
        if (ln < t) bvec[ln] = (ln < kA) ? tmp[ln * 65 + kA] : tail[ln];
        // dp1 = U(R_L[1:dimA,1:dimA]) \ b[1:dimA]; p1 = ([dp1; 0][invperm(F_L11.p)])[1:rankA]     (:141-144)
        const int ds = dimA < kA ? dimA : kA;
        double zw = (ln < ds) ? tmp[ln * 65 + kA] : 0.0;
        for (int i = ds - 1; i >= 0; --i) {
            const int li = __builtin_amdgcn_readlane(lpL, i);
            const double dkk = tmp[i * 65 + li];
            if (dkk == 0.0) status |= 1;
            const double yi = wave_bcast(zw, i) / dkk;
            if (ln == i) zw = yi;
            if (ln < i) zw -= tmp[ln * 65 + li] * yi;
        }
        if (ln < t) p1[ln] = 0.0;
        if (ln < kA) p1[lpL] = (ln < dimA) ? zw : 0.0;    // distinct targets; lanes >= kA wrote zeros to other entries
    }
    if (ln == 0) {
        st->rankA = rankA;
        st->n2 = n - rankA;
        st->kp = (a.m < n - rankA) ? a.m : (n - rankA);
        st->rankJ2 = 0;
        st->code = code;
        st->dimA = dimA;
        st->dimJ2 = 0;
        st->status = status;
    }
}

// Returns false when the shape is outside the kernel's range (the caller uses k_constraint).
inline bool launch_constraint_small(int batch, hipStream_t s, const ConstraintArgs& a) {
    if (a.n > 64 || a.t > 63 || a.fa_done) return false;
    const size_t lds = constraint_small_lds_bytes(a.kA);
    if (a.n <= 32 && a.t <= 32) { GN_ROUTE(ENLSIP_GN_ROUTE_CONSTRAINT_WAVE32); hipLaunchKernelGGL(k_constraint_small<32>, dim3(batch), dim3(64), lds, s, a); }
    else { GN_ROUTE(ENLSIP_GN_ROUTE_CONSTRAINT_WAVE64); hipLaunchKernelGGL(k_constraint_small<64>, dim3(batch), dim3(64), lds, s, a); }
    return true;
}

}  // namespace gn
