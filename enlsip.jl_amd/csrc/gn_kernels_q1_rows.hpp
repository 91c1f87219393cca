// J * F_A.Q for SMALL problems (src/enlsip_functions.jl:219), fused with d_temp = -J1 p1 - rx (:134 / :145):
// n <= 64 parameters and at most 16 reflectors (the batched configurations C3: n = 64, t = 8 and C5: n = 32, t = 4).
// One lane owns one ROW of J in registers and multiplies it by H_0 H_1 ... H_{kA-1} reflector by reflector; the
// reflectors sit in LDS and are read by broadcast.  Loads and stores run with lanes along rows (coalesced for
// column-major J); there is no reduction across lanes and one barrier.  The compact-WY kernels (gn_kernels_q1_v2.hpp,
// gn_kernels_q1_mfma.hpp) pad every block to 64 reflectors, which costs 8-16x the arithmetic at these sizes.
#pragma once
#include "gn_kernels_q1.hpp"

namespace gn {

constexpr int Q1R_MAXK = 16;
// ONE predicate for "J*Q1 of this shape runs reflector by reflector (k_jq1_rows*, k_jq1_factor_small)": the dispatch below
// and the writer of the T block of Q1 (k_constraint_small, which then writes only the kA columns those kernels and the accessors
// read instead of the 64 the compact-WY kernels need) must agree on it.
__host__ __device__ inline bool jq1_by_reflectors(int n, int kA) { return n <= 64 && kA <= Q1R_MAXK; }

template <int NMAX>
#ifndef ENLSIP_JR_OCC
#define ENLSIP_JR_OCC 0           // measured: no gain at 4 waves per SIMD, 5x slower at 5 (380 B scratch)
#endif
#ifndef ENLSIP_JR64_OCC
#define ENLSIP_JR64_OCC 0         // measured: 2 waves per SIMD = 88 spilled registers, C3 J*Q1 stage 0.19 -> 0.29 ms
#endif
__global__ __launch_bounds__(256, (NMAX == 32 && ENLSIP_JR_OCC) ? ENLSIP_JR_OCC : ((NMAX == 64 && ENLSIP_JR64_OCC) ? ENLSIP_JR64_OCC : 1)) void k_jq1_rows(JQ1Args a) {
    __shared__ __attribute__((aligned(16))) double Vs[Q1R_MAXK * NMAX];   // Vs[k][c] = v_k[c] (unit diagonal, zeros above)
    __shared__ double taus[Q1R_MAXK];
    __shared__ double p1s[NMAX];
    const int n = a.n, m = a.m, kA = a.kA, ldw = a.ldw;
    const int prob = blockIdx.y + a.prob0;
    const double* Jin = a.J + prob * a.strideJ;
    const double* rx = a.rx + prob * a.stride_rx;
    const double* FA = a.FA + prob * a.sFA;
    const double* TA = a.TA + prob * a.sTA;
    const double* p1 = a.p1 + prob * a.sP1;
    double* W = a.W + prob * a.sW;
    const int rankA = a.state[prob].rankA;
    const int tid = threadIdx.x;

    // the row and its rx entry are requested first: the staging of the reflectors runs beside them, not in front of them
    const int row = blockIdx.x * 256 + tid;
    const bool live = row < m;
    double x[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) x[c] = (live && c < n) ? __builtin_nontemporal_load(&Jin[row + (size_t)c * a.ldj]) : 0.0;
    const double rxv = live ? rx[row] : 0.0;
    for (int e = tid; e < kA * NMAX; e += 256) {
        const int k = e / NMAX, c = e % NMAX;
        double v = 0.0;
        if (c < n) v = (c > k) ? FA[c + (size_t)k * n] : (c == k ? 1.0 : 0.0);
        Vs[e] = v;
    }
    if (tid < kA) taus[tid] = TA[tid + tid * KBLK];       // dlarft: diag(T) = tau
    if (tid < NMAX) p1s[tid] = (tid < rankA) ? p1[tid] : 0.0;
    __syncthreads();
    if (row >= ldw) return;
    for (int k = 0; k < kA; ++k) {
        const double* vk = Vs + k * NMAX;
        double dot = 0.0;
#pragma unroll
        for (int c = 0; c < NMAX; ++c) dot += x[c] * vk[c];
        const double s = taus[k] * dot;
#pragma unroll
        for (int c = 0; c < NMAX; ++c) x[c] -= s * vk[c];
    }
    double ds = 0.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) ds += x[c] * p1s[c];
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
        if (c < n) W[row + (size_t)c * ldw] = x[c];
    W[row + (size_t)n * ldw] = live ? (-ds - rxv) : 0.0;
}

// Two lanes per row for 32 < n <= 64 (C3): lane l < 32 of a wave owns columns 0..31 of row l, lane l + 32 columns 32..63 of the
// SAME row; a dot product is two half sums joined by one cross-half add.  Half the registers per lane (the one-lane form
// needs 255 and runs at two workgroups per CU: the kernel moves 0.54 GB at 2.8 TB/s because too few loads are in flight), so
// four workgroups per CU; the row data is requested before the reflectors are staged, rx with it.
template <int dummy = 0>
__global__ __launch_bounds__(256, 3) void k_jq1_rows2(JQ1Args a) {
    constexpr int NH = 32;                                                // columns per lane
    __shared__ __attribute__((aligned(16))) double Vs[Q1R_MAXK * 64];    // Vs[k][c] = v_k[c] (unit diagonal, zeros above)
    __shared__ double taus[Q1R_MAXK];
    __shared__ double p1s[64];
    const int n = a.n, m = a.m, kA = a.kA, ldw = a.ldw;
    const int prob = blockIdx.y + a.prob0;
    const double* Jin = a.J + prob * a.strideJ;
    const double* rx = a.rx + prob * a.stride_rx;
    const double* FA = a.FA + prob * a.sFA;
    const double* TA = a.TA + prob * a.sTA;
    const double* p1 = a.p1 + prob * a.sP1;
    double* W = a.W + prob * a.sW;
    const int rankA = a.state[prob].rankA;
    const int tid = threadIdx.x, ln = tid & 63, w = tid >> 6;
    const int half = ln >> 5;                                            // which 32 columns
    const int row = blockIdx.x * 128 + 32 * w + (ln & 31);
    const bool inrange = row < ldw, live = row < m;
    const int c0 = NH * half;

    // addresses = uniform base of column c + ONE 32-bit lane offset (row and column half): no 64-bit address per column
    const unsigned joff = (unsigned)(((size_t)row + (size_t)c0 * a.ldj) * 8), woff = (unsigned)(((size_t)row + (size_t)c0 * ldw) * 8);
    double x[NH];
#pragma unroll
    for (int c = 0; c < NH; ++c)
        x[c] = (live && c0 + c < n) ? __builtin_nontemporal_load((const double*)((const char*)(Jin + (size_t)c * a.ldj) + joff)) : 0.0;
    const double rxv = (live && half == 0) ? rx[row] : 0.0;
    for (int e = tid; e < kA * 64; e += 256) {
        const int k = e >> 6, c = e & 63;
        double v = 0.0;
        if (c < n) v = (c > k) ? FA[c + (size_t)k * n] : (c == k ? 1.0 : 0.0);
        Vs[e] = v;
    }
    if (tid < kA) taus[tid] = TA[tid + tid * KBLK];       // dlarft: diag(T) = tau
    if (tid < 64) p1s[tid] = (tid < rankA) ? p1[tid] : 0.0;
    __syncthreads();
    if (!inrange) return;
    for (int k = 0; k < kA; ++k) {
        const double* vk = Vs + k * 64 + c0;
        double dot = 0.0;
#pragma unroll
        for (int c = 0; c < NH; ++c) dot += x[c] * vk[c];
        const double s = taus[k] * xor32_sum(dot);
#pragma unroll
        for (int c = 0; c < NH; ++c) x[c] -= s * vk[c];
    }
    double ds = 0.0;
#pragma unroll
    for (int c = 0; c < NH; ++c) ds += x[c] * p1s[c0 + c];
    ds = xor32_sum(ds);
#pragma unroll
    for (int c = 0; c < NH; ++c)
        if (c0 + c < n) *(double*)((char*)(W + (size_t)c * ldw) + woff) = x[c];
    if (half == 0) W[row + (size_t)n * ldw] = live ? (-ds - rxv) : 0.0;
}

// Returns false when the shape is outside this kernel's range (the caller falls through to the compact-WY kernels).
inline bool launch_jq1_rows(const JQ1Args& a, int batch, hipStream_t s) {
    if (!jq1_by_reflectors(a.n, a.kA)) return false;
    const dim3 grid((a.ldw + 255) / 256, batch);
#ifdef ENLSIP_GN_LAB
    static const bool two = !(getenv("ENLSIP_GN_JQ1_ROWS2") && getenv("ENLSIP_GN_JQ1_ROWS2")[0] == '0');      // A/B switch
#else
    constexpr bool two = true;
#endif
    if (a.n <= 32) { GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_ROWS32); hipLaunchKernelGGL(k_jq1_rows<32>, grid, dim3(256), 0, s, a); }
    else if (two && a.ldj < (1 << 23) && a.ldw < (1 << 23)) { GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_ROWS2); hipLaunchKernelGGL(k_jq1_rows2<0>, dim3((a.ldw + 127) / 128, batch), dim3(256), 0, s, a); }
    else { GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_ROWS64); hipLaunchKernelGGL(k_jq1_rows<64>, grid, dim3(256), 0, s, a); }
    return true;
}

}  // namespace gn
