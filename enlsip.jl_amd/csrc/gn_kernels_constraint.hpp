// Constraint stage: everything of the subproblem that only involves the active-constraint
// Jacobian, in ONE single-workgroup launch per problem:
//   F_A   = qr(C.A', ColumnNorm())                 src/enlsip_functions.jl:700
//   rankA = pseudo_rank(diag(F_A.R))               src/enlsip_functions.jl:768, :17-31
//   F_L11 = qr(F_A.R', ColumnNorm())               src/enlsip_functions.jl:769
//   b, p1 of sub_search_direction                  src/enlsip_functions.jl:131-133 / :141-144
//   block T factors of Q1 (dlarft) for the compact-WY product J*Q1 (:219)
// Small matrices are factored out of LDS (160 KB/CU), larger ones in place in L2-resident
// global memory through the same code (generic pointers).
#pragma once
#include "gn_wg_linalg.hpp"

namespace gn {

constexpr int KBLK = 64;           // reflectors per compact-WY block of Q1
constexpr int CMAT_DOUBLES = 8192; // LDS matrix area (64 KB)

struct ConstraintArgs {
    int n, t, kA, m;
    double eps_rank;
    int abs_shift;       // 0, or the power of two At and cx were scaled by (pseudo_rank_abs_threshold)
    int dimA_override;   // -1 = rankA
    int code_override;   // 0 = derive from rankA; +1 / -1 force (resolve path)
    int prob0;           // problem index offset
    int fa_done;         // 1: F_A, tau_A, jpvt_A and the block T factor were produced by k_geqp3_reg
    int nv, blkd, gld, matd;   // LDS carve in doubles (constraint_carve)
    // many constraints (run_constraint_dist): both factorisations were produced by the distributed pivoted QR
    int fl_done;         // 1: F_L11, tau_L, jpvt_L are final; Lmat = the unfactored L11 = R_A' (ld ldL), qb = F_L11.Q' b_buff (t)
    int need_T;          // 1: build the block T factors of Q1 here although F_A was produced elsewhere
    const double* Lmat;  long long ldL, sL;
    const double* qb;    long long sQb;
    // inputs
    const double* At;    long long ldat, strideAt;   // n x t
    const double* cx;    long long stride_cx;        // t
    // per-problem workspaces
    double* FA;     long long sFA;      // n x t, ld n
    double* tauA;   long long sTauA;    // kA
    long long* jpvtA; long long sJA;    // t
    double* FL;     long long sFL;      // t x kA, ld t
    double* tauL;   long long sTauL;    // kA
    long long* jpvtL; long long sJL;    // kA
    double* TA;     long long sTA;      // nblk x KBLK x KBLK
    double* p1;     long long sP1;      // t   (first rankA entries used)
    double* bvec;   long long sB;       // t
    ProbState* state;
};

// LDS carve (doubles): vn1[nv] vn2[nv] ybuf[nv] blk[blkd] gl[gld] mat[matd] + ints, sized by the host to the problem
// (small problems then run several workgroups per CU).  blk: 64 x 65 diagonal block of the triangular solves, the
// 32 x 65 staging chunk of V and the 64-strided T image; gl: 64-strided Gram matrix of one reflector block.
inline void constraint_carve(long long n, long long t, int fa_done, int& nv, int& blkd, int& gld, int& matd, int need_T = 0,
                             int fl_done = 0) {
    const long long mx = n > t ? n : t;
    const long long kA = n < t ? n : t;
    nv = (int)((mx + 7) / 8 * 8);
    blkd = mx <= 32 ? 32 * 65 : 64 * 65;
    gld = ((fa_done && !need_T) || kA == 0) ? 0 : (int)(64 * (kA < 64 ? kA : 64));
    long long md = (!fa_done && n * t > t * kA) ? n * t : t * kA;     // F_A is not staged when k_geqp3_reg produced it
    if (fl_done) md = 2;                                               // nothing is staged at all
    if (md > CMAT_DOUBLES) md = CMAT_DOUBLES;
    matd = (int)((md + 1) / 2 * 2);
}
inline size_t constraint_lds_bytes(int nv, int blkd, int gld, int matd) { return (size_t)(3 * nv + blkd + gld + matd + 8) * 8; }

template <int RPL, int G, int NTH>
__global__ __launch_bounds__(NTH) void k_constraint(ConstraintArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* vn1 = smem;
    double* vn2 = vn1 + a.nv;
    double* ybuf = vn2 + a.nv;
    double* blk = ybuf + a.nv;
    double* gl = blk + a.blkd;
    double* mat = gl + a.gld;
    int* sh_i = reinterpret_cast<int*>(mat + a.matd);

    const int prob = blockIdx.x + a.prob0;
    const int n = a.n, t = a.t, kA = a.kA;
    const double* At = a.At + prob * a.strideAt;
    const double* cx = a.cx + prob * a.stride_cx;
    double* FA = a.FA + prob * a.sFA;
    double* tauA = a.tauA + prob * a.sTauA;
    long long* jpvtA = a.jpvtA + prob * a.sJA;
    double* FL = a.FL + prob * a.sFL;
    double* tauL = a.tauL + prob * a.sTauL;
    long long* jpvtL = a.jpvtL + prob * a.sJL;
    double* TA = a.TA + prob * a.sTA;
    double* p1 = a.p1 + prob * a.sP1;
    double* bvec = a.bvec + prob * a.sB;
    ProbState* st = a.state + prob;
    const int tid = threadIdx.x, nt = blockDim.x;

    if (tid == 0) sh_i[2] = 0;  // status accumulator
    // ---- F_A ------------------------------------------------------------------------------
    if (!a.fa_done) {
        const bool fa_lds = (size_t)n * t <= (size_t)CMAT_DOUBLES;
        double* WA = fa_lds ? mat : FA;
        for (int e = tid; e < n * t; e += nt) {
            const int r = e % n, c = e / n;
            WA[r + (size_t)c * n] = At[r + (size_t)c * a.ldat];
        }
        __syncthreads();
        if (t > 0) wg_geqp2<RPL, G>(WA, n, n, t, 0, tauA, jpvtA, vn1, vn2, sh_i);
        if (fa_lds) {
            for (int e = tid; e < n * t; e += nt) FA[e] = mat[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < kA; i += nt) ybuf[i] = FA[i + (size_t)i * n];      // diagonal -> LDS in parallel (see k_pivot_solve)
    __syncthreads();
    if (tid == 0) {
        int rk = pseudo_rank_serial(kA, a.eps_rank, [&](int i) { return ybuf[i]; }, pseudo_rank_abs_threshold(a.eps_rank, a.abs_shift));
        sh_i[1] = rk;
    }
    __syncthreads();
    const int rankA = sh_i[1];
    int code = (rankA == t) ? 1 : -1;
    if (a.code_override != 0) code = a.code_override;
    int dimA = (a.dimA_override >= 0) ? a.dimA_override : rankA;
    dimA = dimA < kA ? dimA : kA;         // the host rejects larger overrides; never index past the factor

    // ---- L11 = R_A'  (t x kA, lower trapezoid) ---------------------------------------------
    const bool fl_lds = !a.fl_done && (size_t)t * kA <= (size_t)CMAT_DOUBLES;
    double* WL = fl_lds ? mat : FL;
    if (!a.fl_done) {
        for (int e = tid; e < t * kA; e += nt) {
            const int i = e % t, j = e / t;  // L[i][j] = R[j][i], j <= i
            WL[i + (size_t)j * t] = (j <= i) ? FA[j + (size_t)i * n] : 0.0;
        }
    }
    // b_buff = -cx[F_A.p]
    for (int i = tid; i < t; i += nt) ybuf[i] = -cx[jpvtA[i] - 1];
    __syncthreads();
    if (code == 1) {
        // b = -cx[p]; p1 = LowerTriangular(R') \ b        (:132-133)
        for (int i = tid; i < t; i += nt) bvec[i] = ybuf[i];
        __syncthreads();
        if (a.fl_done) wg_trsv<true>(a.Lmat + prob * a.sL, (int)a.ldL, t, ybuf, blk, &sh_i[2]);
        else wg_trsv<true>(WL, t, t, ybuf, blk, &sh_i[2]);
        for (int i = tid; i < t; i += nt) p1[i] = ybuf[i];
        __syncthreads();
    }
    // ---- F_L11 -----------------------------------------------------------------------------
    if (!a.fl_done) {
        if (t > 0 && kA > 0) wg_geqp2<RPL, G>(WL, t, t, kA, 0, tauL, jpvtL, vn1, vn2, sh_i);
        if (fl_lds) {
            for (int e = tid; e < t * kA; e += nt) FL[e] = mat[e];
        }
    }
    __syncthreads();
    if (code == -1) {
        // b = F_L11.Q' * b_buff ; dp1 = U(R_L[1:dimA,1:dimA]) \ b[1:dimA] ;
        // p1 = ([dp1; 0][invperm(F_L11.p)])[1:rankA]       (:141-144)
        if (a.fl_done) {
            for (int i = tid; i < t; i += nt) ybuf[i] = a.qb[prob * a.sQb + i];   // carried through the factorisation
        } else if (wave_id() == 0) wave_apply_reflectors<true>(WL, t, tauL, kA, t, ybuf);
        __syncthreads();
        for (int i = tid; i < t; i += nt) bvec[i] = ybuf[i];
        __syncthreads();
        wg_trsv<false>(WL, t, dimA, ybuf, blk, &sh_i[2]);
        // scatter: out[pL[i]-1] = (i < dimA ? dp1[i] : 0), keep first rankA entries.
        // jpvtL has kA entries; positions kA..t-1 of the padded vector are zero (only reachable
        // when t > n, where Julia's invperm would reject the shorter permutation anyway).
        for (int i = tid; i < t; i += nt) p1[i] = 0.0;
        __syncthreads();
        for (int i = tid; i < kA; i += nt) {
            const int dst = (int)jpvtL[i] - 1;
            p1[dst] = (i < dimA) ? ybuf[i] : 0.0;
        }
        __syncthreads();
    }

    // ---- block T factors of Q1 (dlarft, forward columnwise), KBLK reflectors per block --------
    // Gram G = V'V accumulated from 32-row chunks of V staged in LDS; T recurrence in LDS.
    const int nblk = (a.fa_done && !a.need_T) ? 0 : (kA + KBLK - 1) / KBLK;
    for (int blkid = 0; blkid < nblk; ++blkid) {
        const int c0 = blkid * KBLK;
        const int kb = (kA - c0) < KBLK ? (kA - c0) : KBLK;
        constexpr int NWV = NTH / 64, NQ = 64 / NWV;
        const int c1 = tid & 63, c2q = tid >> 6;  // c2 = c2q + NWV q
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
        for (int row0 = c0; row0 < n; row0 += 32) {
            for (int e = tid; e < 32 * KBLK; e += nt) {
                const int r = e & 31, c = e >> 5;
                const int gr = row0 + r, gc = c0 + c;
                double v = 0.0;
                if (c < kb && gr < n) v = (gr > gc) ? FA[gr + (size_t)gc * n] : (gr == gc ? 1.0 : 0.0);
                blk[r * 65 + c] = v;
            }
            __syncthreads();
            for (int r = 0; r < 32; ++r) {
                const double x1 = blk[r * 65 + c1];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] += x1 * blk[r * 65 + c2q + NWV * q];
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (c2q + NWV * q < kb) gl[c1 + 64 * (c2q + NWV * q)] = acc[q];
        __syncthreads();
        // T in LDS (blk reused, columns 0..kb-1 of a 64-strided image)
        for (int e = tid; e < KBLK * kb; e += nt) blk[e] = 0.0;
        __syncthreads();
        for (int j = 0; j < kb; ++j) {
            const double tj = tauA[c0 + j];
            // T[0:j, j] = -tau_j * T[0:j, 0:j] * G[0:j, j]
            if (tid < j) {
                double s = 0.0;
                for (int l = tid; l < j; ++l) s += blk[tid + l * KBLK] * gl[l + 64 * j];
                blk[tid + j * KBLK] = -tj * s;
            }
            if (tid == j) blk[j + j * KBLK] = tj;
            __syncthreads();
        }
        double* T = TA + (size_t)blkid * KBLK * KBLK;
        for (int e = tid; e < KBLK * KBLK; e += nt) T[e] = (e < KBLK * kb) ? blk[e] : 0.0;
        __syncthreads();
    }

    if (tid == 0) {
        st->rankA = rankA;
        st->n2 = n - rankA;
        st->kp = (a.m < n - rankA) ? a.m : (n - rankA);
        st->rankJ2 = 0;
        st->code = code;
        st->dimA = dimA;
        st->dimJ2 = 0;
        st->status = sh_i[2];
    }
}

}  // namespace gn
