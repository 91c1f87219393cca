// Null-space projection: JQ1 = J * F_A.Q (src/enlsip_functions.jl:219) by compact WY,
//   J Q1 = J (I - V1 T1 V1') (I - V2 T2 V2') ...      (KBLK reflectors per block),
// fused with the right-hand side of the second system
//   d_temp = -J1 * p1 - rx                             (src/enlsip_functions.jl:134 / :145)
// which is written as the augmented column n of the workspace, so Q3'd costs no extra pass
// (SURVEY §2.1 K8/K9).  The product is row-local: one workgroup owns RB rows of J, keeps them in
// LDS across all reflector blocks, and writes each element of the workspace exactly once.
// Rows m..ldw-1 of the workspace (CAQR tile padding) are written as zeros here.
#pragma once
#include "gn_kernels_constraint.hpp"

namespace gn {

constexpr int Q1_KC = 32;   // V rows staged per chunk

struct JQ1Args {
    int m, n, kA, ldw;
    const double* J;   long long ldj, strideJ;
    const double* rx;  long long stride_rx;
    const double* FA;  long long sFA;     // n x t (ld n) compact factors
    const double* TA;  long long sTA;     // blocks of KBLK x KBLK
    const double* p1;  long long sP1;
    double* W;         long long sW;      // ldw x (n + 1)
    const ProbState* state;
    int prob0;
    double* VT;        long long sVT;     // k_jq1_v2 only: V T' of the (single) reflector block, n x 64 (ld n), built by k_vt
};

// LDS (doubles): Jl[RB * n] Vl[KC * 65] Wv[RB * 64] Wt[RB * 64] red[256]
__host__ __device__ inline size_t jq1_lds_doubles(int n, int RB) {
    return (size_t)RB * n + Q1_KC * 65 + 2 * RB * KBLK + 256;
}

// RB rows per workgroup (16, or 8 when 16 rows of J would not fit LDS); 256 threads = RB rows x NG groups
template <int Q1_RB>
__global__ __launch_bounds__(256) void k_jq1(JQ1Args a) {
    constexpr int NG = 256 / Q1_RB;        // thread groups along columns
    constexpr int CI = KBLK / NG;          // W columns per thread in phase 1/2
    constexpr int HK = Q1_KC / NG;         // k rows per thread per chunk in phase 3
    constexpr int RSH = (Q1_RB == 16) ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = a.n, m = a.m, kA = a.kA;
    double* Jl = smem;                       // [r + RB * k]
    double* Vl = Jl + (size_t)Q1_RB * n;     // [kk * 65 + c]
    double* Wv = Vl + Q1_KC * 65;            // [r + RB * c]
    double* Wt = Wv + Q1_RB * KBLK;
    double* red = Wt + Q1_RB * KBLK;

    const int prob = blockIdx.y + a.prob0;
    const int row0 = blockIdx.x * Q1_RB;
    const double* J = a.J + prob * a.strideJ;
    const double* rx = a.rx + prob * a.stride_rx;
    const double* FA = a.FA + prob * a.sFA;
    const double* TA = a.TA + prob * a.sTA;
    const double* p1 = a.p1 + prob * a.sP1;
    double* W = a.W + prob * a.sW;
    const int rankA = a.state[prob].rankA;
    const int tid = threadIdx.x;
    const int r = tid & (Q1_RB - 1), g = tid >> RSH;

    if (row0 >= m) {  // pure padding rows: zeros
        for (int e = tid; e < Q1_RB * (n + 1); e += 256) {
            const int rr = e & (Q1_RB - 1), c = e >> RSH;
            if (row0 + rr < a.ldw) W[(row0 + rr) + (size_t)c * a.ldw] = 0.0;
        }
        return;
    }
    // load the row block (rows >= m read as zero)
    for (int e = tid; e < Q1_RB * n; e += 256) {
        const int rr = e & (Q1_RB - 1), c = e >> RSH;
        Jl[e] = (row0 + rr < m) ? J[(row0 + rr) + (size_t)c * a.ldj] : 0.0;
    }
    __syncthreads();

    const int nblk = (kA + KBLK - 1) / KBLK;
    for (int b = 0; b < nblk; ++b) {
        const int c0 = b * KBLK;
        const int kb = (kA - c0) < KBLK ? (kA - c0) : KBLK;
        // phase 1: Wv = Jblk * V_b
        double acc[CI];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) acc[ci] = 0.0;
        for (int k0 = c0; k0 < n; k0 += Q1_KC) {
            for (int e = tid; e < Q1_KC * KBLK; e += 256) {
                const int kk = e & (Q1_KC - 1), c = e >> 5;
                const int gr = k0 + kk, gc = c0 + c;
                double v = 0.0;
                if (c < kb && gr < n) v = (gr > gc) ? FA[gr + (size_t)gc * n] : (gr == gc ? 1.0 : 0.0);
                Vl[kk * 65 + c] = v;
            }
            __syncthreads();
            const int kend = (n - k0) < Q1_KC ? (n - k0) : Q1_KC;
            for (int kk = 0; kk < kend; ++kk) {
                const double x = Jl[r + Q1_RB * (k0 + kk)];
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) acc[ci] += x * Vl[kk * 65 + g + NG * ci];
            }
            __syncthreads();
        }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) Wv[r + Q1_RB * (g + NG * ci)] = acc[ci];
        __syncthreads();
        // phase 2: Wt = Wv * T_b   (T upper triangular)
        const double* T = TA + (size_t)b * KBLK * KBLK;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const int c = g + NG * ci;
            double s = 0.0;
            if (c < kb)
                for (int cp = 0; cp <= c; ++cp) s += Wv[r + Q1_RB * cp] * T[cp + c * KBLK];
            Wt[r + Q1_RB * c] = s;
        }
        __syncthreads();
        // phase 3: Jblk -= Wt * V_b'
        for (int k0 = c0; k0 < n; k0 += Q1_KC) {
            for (int e = tid; e < Q1_KC * KBLK; e += 256) {
                const int kk = e & (Q1_KC - 1), c = e >> 5;
                const int gr = k0 + kk, gc = c0 + c;
                double v = 0.0;
                if (c < kb && gr < n) v = (gr > gc) ? FA[gr + (size_t)gc * n] : (gr == gc ? 1.0 : 0.0);
                Vl[kk * 65 + c] = v;
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < HK; ++h) {
                const int kk = g + NG * h;
                if (k0 + kk < n) {
                    double s = 0.0;
                    for (int c = 0; c < kb; ++c) s += Wt[r + Q1_RB * c] * Vl[kk * 65 + c];
                    Jl[r + Q1_RB * (k0 + kk)] -= s;
                }
            }
            __syncthreads();
        }
    }

    // d_temp = -J1 p1 - rx  (J1 = first rankA columns of the projected block)
    double part = 0.0;
    for (int c = g; c < rankA; c += NG) part += Jl[r + Q1_RB * c] * p1[c];
    red[tid] = part;
    __syncthreads();
    if (tid < Q1_RB) {
        double s = 0.0;
        for (int gg = 0; gg < NG; ++gg) s += red[tid + Q1_RB * gg];
        const int row = row0 + tid;
        if (row < a.ldw) W[row + (size_t)n * a.ldw] = (row < m) ? (-s - rx[row]) : 0.0;
    }
    for (int e = tid; e < Q1_RB * n; e += 256) {
        const int rr = e & (Q1_RB - 1), c = e >> RSH;
        if (row0 + rr < a.ldw) W[(row0 + rr) + (size_t)c * a.ldw] = Jl[e];
    }
}

inline void launch_jq1(const JQ1Args& a, int batch, hipStream_t s) {
    GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_PLAIN);
    const size_t lds16 = jq1_lds_doubles(a.n, 16) * 8;
    if (lds16 <= 160 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_jq1<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipLaunchKernelGGL(k_jq1<16>, dim3((a.ldw + 15) / 16, batch), dim3(256), lds16, s, a);
    } else {
        const size_t lds8 = jq1_lds_doubles(a.n, 8) * 8;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_jq1<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
        hipLaunchKernelGGL(k_jq1<8>, dim3((a.ldw + 7) / 8, batch), dim3(256), lds8, s, a);
    }
}

}  // namespace gn
