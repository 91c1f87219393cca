// Pivot + solve stage, one single-workgroup launch per problem:
//   R0 P = Qt Rt  — column-pivoted QR (dgeqp3 semantics) of the kp x n2 factor R0 left by the
//                   unpivoted CAQR, carrying z = (Q0' d)[1:kp] as a right-hand side.  In exact
//                   arithmetic this yields the pivots, |diag R| and rank of qr(J2, ColumnNorm())
//                   (src/enlsip_functions.jl:223-224).
//   rankJ2 = pseudo_rank(diag(Rt))                               :224
//   dp2 = U(Rt[1:dimJ2,1:dimJ2]) \ d[1:dimJ2]; p2 = [dp2;0][invperm(F_J2.p)]   :136-137 / :147-148
//   p = F_A.Q * [p1; p2]                                          :151
// and the output records b (t), d (m), p (n), jpvtJ2, diag.
#pragma once
#include "gn_wg_linalg.hpp"
#include "gn_kernels_constraint.hpp"

namespace gn {

struct FinalArgs {
    int m, n, t, kA, ldw, ldr;
    double eps_rank;
    int abs_shift;        // 0, or the power of two J and rx were scaled by (pseudo_rank_abs_threshold)
    int dimJ2_override;   // -1 = rankJ2
    int refactor;         // 1: extract R0 and factor it here; 2: Rt already factored (distributed path);
                          // 0: reuse resident Rt (resolve path)
    int prob0;            // problem index offset
    int n2cap;            // > 0: launch shape of the stage; wider problems (state.n2 > n2cap) are skipped (redone by the caller)
    int nv, matd;         // LDS carve: vector length (>= max(n, t)) and matrix/diagonal-block area in doubles
    const double* dsrc;   // refactor == 0: transformed right-hand side Q3'd (length m, one problem)
    const double* W;   long long sW;      // ldw x (n+1): R0 in the upper triangle of the J2 columns, d in column n
    double* Rt;        long long sRt;     // ldr x (n2max + 1) : pivoted factors + carried rhs
    double* tauJ;      long long sTauJ;
    long long* jpvtJ;  long long sJJ;     // n entries reserved
    const double* FA;  long long sFA;
    const double* tauA; long long sTauA;
    const double* p1;  long long sP1;
    const double* bvec; long long sB;
    double* zsave;     long long sZ;      // kp: copy of (Q0' d)[1:kp] before the pivoted stage (resolve path)
    // outputs (device), may be null
    double* p_out;     long long sPo;
    double* b_out;     long long sBo;
    double* d_out;     long long sDo;
    long long* jA_out; long long sJAo;  const long long* jpvtA; long long sJA;
    long long* jL_out; long long sJLo;  const long long* jpvtL; long long sJL;
    long long* jJ_out; long long sJJo;
    ProbState* state;
};

// LDS (doubles): vn1[nv] vn2[nv] ybuf[nv] pbuf[nv] mat[matd] + ints, sized by the host to the problem so that small
// problems run several workgroups per CU.  The 64 x 65 diagonal block of the triangular solve aliases `mat`, which is
// dead once the pivoted factors have been copied back.
inline void final_carve(long long m, long long n, long long t, int& nv, int& matd) {
    const long long mx = n > t ? n : t;
    nv = (int)((mx + 7) / 8 * 8);
    const long long kp = m < n ? m : n;
    long long md = kp * (n + 1);
    if (md > CMAT_DOUBLES) md = CMAT_DOUBLES;
    const long long nb = mx < 64 ? mx : 64;
    if (md < nb * 65) md = nb * 65;
    matd = (int)((md + 1) / 2 * 2);
}
inline size_t final_lds_bytes(int nv, int matd) { return (size_t)(4 * nv + matd + 8) * 8; }

template <int RPL, int G, int NTH>
__global__ __launch_bounds__(NTH) void k_pivot_solve(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* vn1 = smem;
    double* vn2 = vn1 + a.nv;
    double* ybuf = vn2 + a.nv;
    double* pbuf = ybuf + a.nv;
    double* mat = pbuf + a.nv;
    double* blk = mat;
    int* sh_i = reinterpret_cast<int*>(mat + a.matd);

    const int prob = blockIdx.x + a.prob0;
    ProbState* stp = a.state + prob;
    if (a.n2cap > 0 && stp->n2 > a.n2cap) return;     // its pivoted factors were not produced (stale permutation entries)
    const int rankA = stp->rankA, n2 = stp->n2, kp = stp->kp;
    const int n = a.n, m = a.m, t = a.t, kA = a.kA, ldr = a.ldr, ldw = a.ldw;
    const double* W = a.W + prob * a.sW;
    double* Rt = a.Rt + prob * a.sRt;
    double* tauJ = a.tauJ + prob * a.sTauJ;
    long long* jpvtJ = a.jpvtJ + prob * a.sJJ;
    const double* FA = a.FA + prob * a.sFA;
    const double* tauA = a.tauA + prob * a.sTauA;
    const double* p1 = a.p1 + prob * a.sP1;
    const double* bvec = a.bvec + prob * a.sB;
    const int tid = threadIdx.x, nt = blockDim.x;

    if (tid == 0) sh_i[2] = 0;
    int rankJ2 = stp->rankJ2;
    if (a.refactor == 1) {
        const bool lds = (size_t)kp * (n2 + 1) <= (size_t)CMAT_DOUBLES;
        double* WR = lds ? mat : Rt;
        const int ldx = lds ? kp : ldr;
        // extract R0 (upper trapezoid) and z
        for (int e = tid; e < kp * (n2 + 1); e += nt) {
            const int i = e % kp, c = e / kp;
            double v;
            if (c < n2) v = (i <= c) ? W[i + (size_t)(rankA + c) * ldw] : 0.0;
            else v = W[i + (size_t)n * ldw];
            WR[i + (size_t)c * ldx] = v;
        }
        if (a.zsave)
            for (int i = tid; i < kp; i += nt) a.zsave[prob * a.sZ + i] = W[i + (size_t)n * ldw];
        __syncthreads();
        if (kp > 0) wg_geqp2<RPL, G>(WR, ldx, kp, n2, 1, tauJ, jpvtJ, vn1, vn2, sh_i);
        if (lds) {
            for (int e = tid; e < kp * (n2 + 1); e += nt) {
                const int i = e % kp, c = e / kp;
                Rt[i + (size_t)c * ldr] = mat[e];
            }
        }
        __syncthreads();
    }
    if (a.refactor == 1 || a.refactor == 2) {
        // (refactor == 2: the factors were produced by the blocked / distributed pivoted QR.)  The diagonal goes to LDS
        // in parallel first: one thread walking it in global memory pays an L2 round trip per entry (0.3 ms at kp = 448).
        for (int i = tid; i < kp; i += nt) pbuf[i] = Rt[i + (size_t)i * ldr];
        __syncthreads();
        if (tid == 0) {
            sh_i[1] = pseudo_rank_serial(kp, a.eps_rank, [&](int i) { return pbuf[i]; }, pseudo_rank_abs_threshold(a.eps_rank, a.abs_shift));
        }
        __syncthreads();
        rankJ2 = sh_i[1];
    }
    int dimJ2 = (a.dimJ2_override >= 0) ? a.dimJ2_override : rankJ2;
    dimJ2 = dimJ2 < kp ? dimJ2 : kp;      // the host rejects larger overrides; never index past the factor

    // dp2 = U(Rt[1:dimJ2,1:dimJ2]) \ d[1:dimJ2]
    for (int i = tid; i < n2; i += nt)
        ybuf[i] = (i < dimJ2) ? (a.dsrc ? a.dsrc[i] : Rt[i + (size_t)n2 * ldr]) : 0.0;
    __syncthreads();
    wg_trsv<false>(Rt, ldr, dimJ2, ybuf, blk, &sh_i[2]);
    // y = [p1 ; p2],  p2[pJ[i]-1] = (i < dimJ2 ? dp2[i] : 0)
    for (int i = tid; i < rankA; i += nt) pbuf[i] = p1[i];
    for (int i = tid; i < n2; i += nt) pbuf[rankA + (int)jpvtJ[i] - 1] = (i < dimJ2) ? ybuf[i] : 0.0;
    __syncthreads();
    // p = F_A.Q * y
    if (wave_id() == 0) {
        if (n <= 64 * RPL) wave_apply_reflectors_reg<false, RPL>(FA, n, tauA, kA, n, pbuf);
        else wave_apply_reflectors<false>(FA, n, tauA, kA, n, pbuf);
    }
    __syncthreads();
    if (a.p_out)
        for (int i = tid; i < n; i += nt) a.p_out[prob * a.sPo + i] = pbuf[i];
    if (a.b_out)
        for (int i = tid; i < t; i += nt) a.b_out[prob * a.sBo + i] = bvec[i];
    if (a.d_out) {
        for (int i = tid; i < m; i += nt)
            a.d_out[prob * a.sDo + i] =
                a.dsrc ? a.dsrc[i] : ((i < kp) ? Rt[i + (size_t)n2 * ldr] : W[i + (size_t)n * ldw]);
    }
    if (a.jA_out)
        for (int i = tid; i < t; i += nt) a.jA_out[prob * a.sJAo + i] = a.jpvtA[prob * a.sJA + i];
    if (a.jL_out)
        for (int i = tid; i < kA; i += nt) a.jL_out[prob * a.sJLo + i] = a.jpvtL[prob * a.sJL + i];
    if (a.jJ_out)
        for (int i = tid; i < n2; i += nt) a.jJ_out[prob * a.sJJo + i] = jpvtJ[i];
    if (tid == 0) {
        stp->rankJ2 = rankJ2;
        stp->dimJ2 = dimJ2;
        stp->status |= sh_i[2];
    }
}

}  // namespace gn
