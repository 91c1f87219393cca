// Pivot + solve stage for SMALL factors (kp <= 64 rows, n2 + 1 <= 64 columns): ONE WAVE per problem, no barriers.
// Same contract as k_pivot_solve (gn_kernels_final.hpp, refactor == 1):
//   R0 P = Qt Rt, rankJ2 = pseudo_rank(diag Rt)         src/enlsip_functions.jl:223-224
//   dp2 = U(Rt) \ d[1:dimJ2], p2 = [dp2;0][invperm]     :136-137 / :147-148
//   p = F_A.Q [p1; p2]                                   :151
//
// Layout: lane c owns COLUMN c of [R0 | z] (lane n2 = the carried right-hand side), rows in registers: gn_wave_qrcp.hpp.
#pragma once
#include "gn_kernels_final.hpp"
#include "gn_wave_qrcp.hpp"

namespace gn {

// LDS (doubles): tmp[kpm * 65] vbuf[64] dg[64] pbuf[nv]
inline size_t final_small_lds_bytes(int kpm, int nv) { return (size_t)(kpm * 65 + 64 + 64 + nv + 8) * 8; }

#ifdef GN_PS_STAMPS   // timing-only build: phase stamps (100 MHz wall clock) of problem 0 in the last entries of d_out
#define GN_PS_STAMP(i) do { if (prob == 0 && ln == 0 && a.d_out) stamps[i] = (double)wall_clock64(); } while (0)
#else
#define GN_PS_STAMP(i) do { } while (0)
#endif

template <int NR>
#ifndef ENLSIP_PS_OCC
#define ENLSIP_PS_OCC 0           // measured: 4 waves per SIMD (128 registers, 120 B scratch) more than doubles the kernel's time
#endif
#ifndef ENLSIP_PS64_OCC
#define ENLSIP_PS64_OCC 0         // measured: 2 waves per SIMD = 152 spilled registers, C3 pivot stage 0.19 -> 0.36 ms
#endif
__global__ __launch_bounds__(64, (NR == 32 && ENLSIP_PS_OCC) ? ENLSIP_PS_OCC : ((NR == 64 && ENLSIP_PS64_OCC) ? ENLSIP_PS64_OCC : 1)) void k_pivot_small(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int prob = blockIdx.x + a.prob0;
    ProbState* stp = a.state + prob;
    const int rankA = stp->rankA, n2 = stp->n2, kp = stp->kp;
    // a problem whose J2 is wider than the launch shape (rank-deficient A) is redone by the caller with the true width
    if (kp > NR || n2 + 1 > 64 || kp * 65 > a.matd || (a.n2cap > 0 && n2 > a.n2cap)) return;
    double* tmp = smem;
    double* vbuf = tmp + a.matd;
    double* dg = vbuf + 64;
    double* pbuf = dg + 64;
    const int n = a.n, m = a.m, t = a.t, kA = a.kA, ldr = a.ldr, ldw = a.ldw;
    const double* W = a.W + prob * a.sW;
    double* Rt = a.Rt + prob * a.sRt;
    double* tauJ = a.tauJ + prob * a.sTauJ;
    long long* jpvtJ = a.jpvtJ + prob * a.sJJ;
    const double* FA = a.FA + prob * a.sFA;
    const double* tauA = a.tauA + prob * a.sTauA;
    const double* p1 = a.p1 + prob * a.sP1;
    const double* bvec = a.bvec + prob * a.sB;
    const int ln = threadIdx.x;
    int status = 0;

    GN_PS_STAMP(0);
    // ---- R0 (upper trapezoid) and z: columns through LDS so that the global loads run along rows ---------------
    for (int c0 = 0; c0 <= n2; c0 += 8) {       // eight independent loads in flight, then eight LDS writes
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            v[u] = 0.0;
            if (ln < kp) {
                if (c < n2) v[u] = (ln <= c) ? W[ln + (size_t)(rankA + c) * ldw] : 0.0;
                else if (c == n2) v[u] = W[ln + (size_t)n * ldw];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ln < kp && c0 + u <= n2) tmp[ln * 65 + c0 + u] = v[u];
    }
    if (a.zsave && ln < kp) a.zsave[prob * a.sZ + ln] = W[ln + (size_t)n * ldw];
    wave_mem_sync();
    double x[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) x[r] = (r < kp && ln <= n2) ? tmp[r * 65 + ln] : 0.0;
    WaveQrcp q{};
    q.rows = kp; q.k = kp; q.ncand = n2; q.npart = n2 + 1;
    q.tmp = tmp; q.vbuf = vbuf; q.dg = dg; q.F = Rt; q.ldf = ldr; q.tau = tauJ;
    int mypos = ln;
    GN_PS_STAMP(1);
    if (kp > 0) wave_qrcp<NR, false>(x, q, ln, mypos);
    GN_PS_STAMP(2);

    // ---- permutation, upper parts of the columns, rank ---------------------------------------------------------------
    if (ln < n2) jpvtJ[mypos] = ln + 1;
    // lane-of-position table (positions 0..n2): lp of lane i = lane that sits at position i
    int* lpos = reinterpret_cast<int*>(vbuf);
    wave_mem_sync();
    if (ln <= n2) lpos[mypos] = ln;
    wave_mem_sync();
    const int lp = (ln <= n2) ? lpos[ln] : 0;
    wave_qrcp_store_upper(q, ln, lp, n2 + 1, Rt, ldr);
    const int rankJ2 = wave_pseudo_rank(dg, kp, a.eps_rank, ln);
    int dimJ2 = (a.dimJ2_override >= 0) ? a.dimJ2_override : rankJ2;
    dimJ2 = dimJ2 < kp ? dimJ2 : kp;      // as k_pivot_solve: never past the factor
    GN_PS_STAMP(3);

    // ---- dp2 = U(Rt[1:dimJ2,1:dimJ2]) \ d[1:dimJ2]: lane r carries row r of the right-hand side ------------------------
    double zw = (ln < dimJ2 && ln < kp) ? tmp[ln * 65 + n2] : 0.0;
    for (int i = (dimJ2 < kp ? dimJ2 : kp) - 1; i >= 0; --i) {
        const int li = __builtin_amdgcn_readlane(lp, i);
        const double dkk = tmp[i * 65 + li];
        if (dkk == 0.0) status |= 1;
        const double yi = wave_bcast(zw, i) / dkk;
        if (ln == i) zw = yi;
        if (ln < i) zw -= tmp[ln * 65 + li] * yi;
    }
    GN_PS_STAMP(4);
    // y = [p1 ; p2],  p2[pJ[i]-1] = (i < dimJ2 ? dp2[i] : 0)
    for (int i = ln; i < rankA; i += WAVE) pbuf[i] = p1[i];
    if (ln < n2) pbuf[rankA + lp] = (ln < dimJ2) ? zw : 0.0;
    // p = F_A.Q * y
    wave_mem_sync();
    wave_apply_reflectors<false>(FA, n, tauA, kA, n, pbuf);
    wave_mem_sync();
    GN_PS_STAMP(5);
    if (a.p_out)
        for (int i = ln; i < n; i += WAVE) a.p_out[prob * a.sPo + i] = pbuf[i];
    if (a.b_out)
        for (int i = ln; i < t; i += WAVE) a.b_out[prob * a.sBo + i] = bvec[i];
    if (a.d_out) {
        for (int i = ln; i < m; i += WAVE)
            a.d_out[prob * a.sDo + i] = (i < kp) ? tmp[i * 65 + n2] : W[i + (size_t)n * ldw];
    }
    if (a.jA_out)
        for (int i = ln; i < t; i += WAVE) a.jA_out[prob * a.sJAo + i] = a.jpvtA[prob * a.sJA + i];
    if (a.jL_out)
        for (int i = ln; i < kA; i += WAVE) a.jL_out[prob * a.sJLo + i] = a.jpvtL[prob * a.sJL + i];
    if (a.jJ_out && ln < n2) a.jJ_out[prob * a.sJJo + mypos] = ln + 1;
    if (ln == 0) {
        stp->rankJ2 = rankJ2;
        stp->dimJ2 = dimJ2;
        stp->status |= status;
    }
#ifdef GN_PS_STAMPS
    GN_PS_STAMP(6);
    if (prob == 0 && ln == 0 && a.d_out)
        for (int i = 0; i < 7; ++i) a.d_out[m - 8 + i] = stamps[i];
#endif
}

// Returns false when the launch shape is outside the kernel's range (the caller uses k_pivot_solve).
inline bool launch_pivot_small(int kp_launch, int n2_launch, int batch, hipStream_t s, FinalArgs a) {
    if (kp_launch > 64 || n2_launch + 1 > 64 || a.refactor != 1 || a.dsrc) return false;
    const long long mx = a.n > a.t ? a.n : a.t;
    a.nv = (int)((mx + 7) / 8 * 8);
    a.matd = (kp_launch > 0 ? kp_launch : 1) * 65 + 1;
    a.matd = (a.matd + 1) / 2 * 2;
    const size_t lds = final_small_lds_bytes(0, a.nv) + (size_t)a.matd * 8;
    if (kp_launch <= 32) hipLaunchKernelGGL(k_pivot_small<32>, dim3(batch), dim3(64), lds, s, a);
    else hipLaunchKernelGGL(k_pivot_small<64>, dim3(batch), dim3(64), lds, s, a);
    return true;
}

}  // namespace gn
