// Small one-tile problems (m <= 256, n <= 32, at most 16 reflectors in Q1, J2 narrower than one panel: the batched configuration
// C5 = 65536 x (256, 32, 4)):  J * F_A.Q, d_temp = -J1 p1 - rx (src/enlsip_functions.jl:219, :134 / :145) AND the unpivoted
// Householder factorisation of [J2 | d] (the CAQR of one 256-row tile with d as passenger column, :223 first half) in ONE launch.
//
// The two kernels this replaces (k_jq1_rows, k_caqr_factor<4, 4>) both hold the problem in registers — the first with a lane per
// ROW (the product with Q1 is row-local: no reduction over lanes), the second with a wave per group of COLUMNS (lanes along
// rows, reflector broadcast) — and handed it over through HBM: 2 x 8 (n2 + 1) m bytes per problem written and read back, the
// larger part of the pipeline's traffic at this size, plus a launch.  Here the hand-over is a transposition through 16 KB of
// LDS in slabs of 8 columns: thread = row writes its 8 entries of the slab, wave w picks up the panel columns w, w + 4, ... that
// fall into it.  J1 (the first rankA columns of J Q1, needed by the multiplier estimates and the re-solve) still goes to W.
#pragma once
#include "gn_kernels_caqr.hpp"
#include "gn_kernels_q1_rows.hpp"

namespace gn {

constexpr int SF_SLAB = 8;       // columns per transposition slab

template <int NMAX>
__global__ __launch_bounds__(256, 4) void k_jq1_factor_small(JQ1Args q, CaqrArgs a) {
    constexpr int RPL = 4, NW = 4, NC = PB / NW;
    __shared__ double vsh[2][64 * RPL];
    __shared__ double taush[PB];
    __shared__ double gsh[PB][PB + 1];
    __shared__ __attribute__((aligned(16))) double Vs[Q1R_MAXK * NMAX];   // Vs[k][c] = v_k[c] (unit diagonal, zeros above)
    __shared__ double taus[Q1R_MAXK];
    __shared__ double p1s[NMAX];
    __shared__ double slab[SF_SLAB][64 * RPL];

    const int n = q.n, m = q.m, kA = q.kA, ldw = q.ldw;
    const int prob = blockIdx.y + a.prob0;
    const ProbState st = a.state[prob];
    const double* Jin = q.J + prob * q.strideJ;
    const double* rx = q.rx + prob * q.stride_rx;
    const double* FA = q.FA + prob * q.sFA;
    const double* TA = q.TA + prob * q.sTA;
    const double* p1 = q.p1 + prob * q.sP1;
    double* W = q.W + prob * q.sW;
    const int rankA = st.rankA;
    const int tid = threadIdx.x, ln = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());

    // ---- J Q1 and d_temp, a lane per row (k_jq1_rows) --------------------------------------------------------------------
    // Order of the requests: the SMALL operands (reflectors of Q1, their tau, p1: a few values per thread) first, then the row of J
    // and its rx entry, then the small operands go to LDS.  Loads retire in order under vmcnt, so a wait for a value requested
    // AFTER the row also waits for the whole row; requested before it, the staging (a global round trip and a barrier) runs
    // beside the row's 32 loads.  Every load is unconditional (row / column clamped into the matrix) and masked afterwards:
    // guarded loads (`live && c < n ? J[..] : 0`) were partly compiled into branches with the wait right behind the load, and
    // the row arrived in five memory round trips.
    const int row = tid;                                   // m <= 256: one workgroup holds every row
    const bool live = row < m;
    static_assert(Q1R_MAXK * NMAX <= 2 * 256, "two staging entries per thread");
    double fv[2];
    bool fon[2], fone[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = tid + 256 * h;
        const int k = e / NMAX, c = e % NMAX;
        fon[h] = (e < kA * NMAX) && (c < n) && (c > k);
        fone[h] = (e < kA * NMAX) && (c < n) && (c == k);
        fv[h] = FA[fon[h] ? c + (size_t)k * n : 0];
    }
    const double tauv = TA[(tid < kA ? tid : 0) * (KBLK + 1)];        // dlarft: diag(T) = tau
    const double p1v = p1[tid < NMAX ? tid : 0];
    __builtin_amdgcn_sched_barrier(0);                     // ... in THIS order
    double xr[NMAX];
    const int rowc = live ? row : 0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) xr[c] = __builtin_nontemporal_load(&Jin[rowc + (size_t)(c < n ? c : 0) * q.ldj]);
    double rxv = rx[rowc];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = tid + 256 * h;
        if (e < kA * NMAX) Vs[e] = fon[h] ? fv[h] : (fone[h] ? 1.0 : 0.0);
    }
    if (tid < kA) taus[tid] = tauv;
    if (tid < NMAX) p1s[tid] = (tid < rankA) ? p1v : 0.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
        const unsigned long long mk = (live && c < n) ? ~0ull : 0ull;
        xr[c] = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(xr[c]) & mk));
    }
    rxv = live ? rxv : 0.0;
    __syncthreads();
    // A reflector is read from LDS in HALVES, once for the dot product and once for the update (broadcast reads: every lane the same
    // address).  Read whole and kept, its 32 values (64 registers) beside the 64 of the row pushed two quads of the row into spill
    // slots inside this loop — two scratch round trips per reflector, each behind a vmcnt(0) (tests/probes/isa_scratch_in_loops.py).
    constexpr int NH = NMAX / 2;
    for (int k = 0; k < kA; ++k) {
        const double* vk = Vs + k * NMAX;
        double dot = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double vv[NH];
#pragma unroll
            for (int c = 0; c < NH; ++c) vv[c] = vk[NH * h + c];
#pragma unroll
            for (int c = 0; c < NH; ++c) dot += xr[NH * h + c] * vv[c];
            __builtin_amdgcn_sched_barrier(0);
        }
        const double s = taus[k] * dot;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double vv[NH];
#pragma unroll
            for (int c = 0; c < NH; ++c) vv[c] = vk[NH * h + c];
#pragma unroll
            for (int c = 0; c < NH; ++c) xr[NH * h + c] -= s * vv[c];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double ds = 0.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) ds += xr[c] * p1s[c];
    const double dtemp = live ? (-ds - rxv) : 0.0;
    // J1 stays in W for the consumers of J Q1 (multiplier estimates, re-solve); rankA <= kA <= 16
    if (row < ldw) {
#pragma unroll
        for (int c = 0; c < Q1R_MAXK; ++c)
            if (c < rankA && c < NMAX) W[row + (size_t)c * ldw] = xr[c];
    }
    const int kp = st.kp;
    if (kp <= 0) {          // J2 is empty (rankA = n): nothing to factor, d = d_temp
        if (row < ldw) W[row + (size_t)n * ldw] = dtemp;
        return;
    }
    // ---- row owners -> column owners through LDS: panel column c (J2-local) = column rankA + c of J Q1, passenger = d ---------
    const int bw = kp < PB ? kp : PB;
    const int bwp = (a.npass && bw < PB) ? bw + 1 : bw;
    double x[NC][RPL];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int i = 0; i < RPL; ++i) x[cc][i] = 0.0;
#pragma unroll
    for (int p = 0; p < (NMAX + 1 + SF_SLAB - 1) / SF_SLAB; ++p) {
#pragma unroll
        for (int cj = 0; cj < SF_SLAB; ++cj) {
            const int col = SF_SLAB * p + cj;              // column of [J Q1 | d], compile-time
            double v = 0.0;
            if (col < NMAX) v = (col < n) ? xr[col] : 0.0;
            v = (col == n) ? dtemp : v;
            slab[cj][row] = v;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int c = w + NW * cc;                     // panel column of this wave (uniform)
            const int col = rankA + c - SF_SLAB * p;       // its place in the slab (uniform)
            if (c < bwp && col >= 0 && col < SF_SLAB) {
#pragma unroll
                for (int i = 0; i < RPL; ++i) x[cc][i] = slab[col][ln + 64 * i];
            }
        }
        __syncthreads();
    }
    // ---- the panel factorisation with the tile already in registers; V, R and T go where the other kernels expect them -------
    // (always a passenger launch on a panel narrower than 32: the T factor is never read beyond its diagonal)
    caqr_factor_core<RPL, NW, true, false>(a, prob, st, x, vsh, taush, gsh);
}

// true when the fused kernel serves the launch shape (one 256-row tile, one panel narrower than 32 with d as passenger)
inline bool small_fused_applies(long long m, long long n, int kA, int n2_launch) {
    return m <= 256 && n <= 32 && kA <= Q1R_MAXK && n2_launch >= 1 && n2_launch < PB && m >= n2_launch;
}
inline void launch_jq1_factor_small(const JQ1Args& q, const CaqrArgs& a, int batch, hipStream_t s) {
    GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_FUSED_SMALL);
    hipLaunchKernelGGL(k_jq1_factor_small<32>, dim3(1, batch), dim3(256), 0, s, q, a);
}

}  // namespace gn
