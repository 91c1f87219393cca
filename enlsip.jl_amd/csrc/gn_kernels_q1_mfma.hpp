// J * F_A.Q on v_mfma_f64_16x16x4_f64 (src/enlsip_functions.jl:219), fused with
// d_temp = -J1 p1 - rx (:134 / :145).  Same mathematics as gn_kernels_q1.hpp (kept as the
// plain-FMA A/B partner): per block of <= 64 reflectors
//     W1 = Jrb V        (64 x n)(n x 64)      phase 1, K streamed in 32-column chunks
//     W2 = -W1 T        (64 x 64)(64 x 64)    phase 2
//     Jrb += W2 V'      (64 x 64)(64 x n)     phase 3, output streamed in 32-column chunks
// One workgroup owns 64 rows; J is read twice (HBM, then L2) and the workspace written once.
// Both streaming loops are software-pipelined: the global loads of chunk c+1 are issued before
// the MFMAs of chunk c and only waited for at the next LDS write.
//
// Fragment maps (cdna_hip_programming.md §3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D[i = (l>>4) + 4 r][j = l&15].  LDS leading dimensions are chosen so the two 16-lane groups of
// a half-wave (k and k+1) fall on disjoint banks: 80 for images read with lanes along rows,
// 34 / 66 for images read with lanes along columns.
#pragma once
#include "gn_kernels_q1.hpp"
#include "gn_kernels_update_v4.hpp"

namespace gn {

constexpr int QM_RB = 64;    // rows per workgroup
constexpr int QM_KC = 32;    // chunk of columns of J (= rows of V)
constexpr int QM_LDJ = 80;   // Jl / Wl leading dimension (rows contiguous)
constexpr int QM_LDV = 34;   // Vl leading dimension (k contiguous)
constexpr int QM_LDT = 66;

__global__ __launch_bounds__(256, 2) void k_jq1_mfma(JQ1Args a) {
    __shared__ __attribute__((aligned(16))) double lds_a[QM_KC * QM_LDJ + KBLK * QM_LDV];  // Jl | Vl, aliased by Tl
    __shared__ __attribute__((aligned(16))) double Wl[KBLK * QM_LDJ];                      // W1 then W2 (negated)
    __shared__ double dred[QM_RB];
    static_assert(QM_KC * QM_LDJ + KBLK * QM_LDV >= KBLK * QM_LDT, "T image must fit the aliased region");
    double* Jl = lds_a;                       // [col * 80 + row]   (32 cols x 64 rows)
    double* Vl = lds_a + QM_KC * QM_LDJ;      // [c * 34 + kk]      (64 reflectors x 32 rows of V)
    double* Tl = lds_a;                       // [j * 66 + k]

    const int n = a.n, m = a.m, kA = a.kA, ldw = a.ldw;
    const int prob = blockIdx.y + a.prob0;
    const int row0 = blockIdx.x * QM_RB;
    const double* Jin = a.J + prob * a.strideJ;
    const double* rx = a.rx + prob * a.stride_rx;
    const double* FA = a.FA + prob * a.sFA;
    const double* TA = a.TA + prob * a.sTA;
    const double* p1 = a.p1 + prob * a.sP1;
    double* W = a.W + prob * a.sW;
    const int rankA = a.state[prob].rankA;
    const int tid = threadIdx.x, ln = lane_id(), w = wave_id();
    const int lr = ln & 15, lq = ln >> 4;
    const int srow = tid & 63, scg = tid >> 6;   // J staging: 64 rows x 4 column groups
    const int vkk = tid & 31, vcg = tid >> 5;    // V staging: 32 rows of V x 8 reflector groups

    if (row0 >= m) {  // pure padding rows
        for (int e = tid; e < QM_RB * (n + 1); e += 256) {
            const int rr = e & 63, c = e >> 6;
            if (row0 + rr < ldw) W[(row0 + rr) + (size_t)c * ldw] = 0.0;
        }
        return;
    }
    if (tid < QM_RB) dred[tid] = 0.0;
    const int nblk = (kA + KBLK - 1) / KBLK;
    if (nblk == 0) {  // no constraints: Q1 = I
        for (int e = tid; e < QM_RB * n; e += 256) {
            const int rr = e & 63, c = e >> 6;
            const int row = row0 + rr;
            if (row < ldw) W[row + (size_t)c * ldw] = (row < m) ? Jin[row + (size_t)c * a.ldj] : 0.0;
        }
        if (tid < QM_RB && row0 + tid < ldw) W[(row0 + tid) + (size_t)n * ldw] = (row0 + tid < m) ? -rx[row0 + tid] : 0.0;
        return;
    }

    for (int b = 0; b < nblk; ++b) {
        const int c0 = b * KBLK;
        const int kb = (kA - c0) < KBLK ? (kA - c0) : KBLK;
        const double* src = (b == 0) ? Jin : W;          // later blocks re-read what phase 3 wrote
        const long long lds_src = (b == 0) ? a.ldj : ldw;
        const int rows_src = (b == 0) ? m : ldw;
        const bool last = (b == nblk - 1);
        const int kfirst = (c0 / QM_KC) * QM_KC;          // V_b is zero above row c0
        auto load_j = [&](int k0, double* jr) {
            const int row = row0 + srow;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int col = k0 + scg + 4 * u;
                jr[u] = (col < n && row < rows_src) ? src[row + (size_t)col * lds_src] : 0.0;
            }
        };
        auto load_v = [&](int k0, double* vr) {
            const int gr = k0 + vkk;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = vcg + 8 * u;
                const int gc = c0 + c;
                double v = 0.0;
                if (c < kb && gr < n) {
                    const double x = FA[gr + (size_t)gc * n];
                    v = (gr > gc) ? x : (gr == gc ? 1.0 : 0.0);
                }
                vr[u] = v;
            }
        };
        // ---- phase 1: W1 = Jrb * V_b  (wave w: row tile w, 4 column tiles) ---------------------
        mfma_d4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        double jreg[8], vreg[8];
        load_j(kfirst, jreg);
        load_v(kfirst, vreg);
        for (int k0 = kfirst; k0 < n; k0 += QM_KC) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) Jl[(scg + 4 * u) * QM_LDJ + srow] = jreg[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) Vl[(vcg + 8 * u) * QM_LDV + vkk] = vreg[u];
            if (k0 + QM_KC < n) {
                load_j(k0 + QM_KC, jreg);
                load_v(k0 + QM_KC, vreg);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < QM_KC / 4; ++ks) {
                const int kk = 4 * ks + lq;
                const double av = Jl[kk * QM_LDJ + 16 * w + lr];          // A[row][k]
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const double bv = Vl[(16 * ct + lr) * QM_LDV + kk];   // B[k][reflector]
                    acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[ct], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        // W1 -> LDS (Wl[c * 80 + row]); T_b -> LDS (aliases Jl/Vl)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Wl[(16 * ct + lr) * QM_LDJ + 16 * w + lq + 4 * r] = acc[ct][r];
        {
            const double* T = TA + (size_t)b * KBLK * KBLK;
            for (int e = tid; e < KBLK * KBLK; e += 256) {
                const int k = e & 63, j = e >> 6;
                Tl[j * QM_LDT + k] = T[k + j * KBLK];
            }
        }
        __syncthreads();
        // ---- phase 2: W2 = -W1 * T_b ---------------------------------------------------------------
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KBLK / 4; ++ks) {
            const int kk = 4 * ks + lq;
            const double av = Wl[kk * QM_LDJ + 16 * w + lr];              // A[row][k] = W1[row][k]
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const double bv = Tl[(16 * ct + lr) * QM_LDT + kk];       // B[k][j] = T[k][j]
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[ct], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Wl[(16 * ct + lr) * QM_LDJ + 16 * w + lq + 4 * r] = -acc[ct][r];
        // ---- phase 3: out chunk = J chunk + W2n * V_b'   (D[i = col][j = row]) ------------------------
        double dpart = 0.0;
        const int row = row0 + 16 * w + lr;
        auto load_o = [&](int k0, mfma_d4* o) {
            // C fragments straight from global: o[it][r] = J[row][k0 + 16 it + lq + 4 r]
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = k0 + 16 * it + lq + 4 * r;
                    o[it][r] = (col < n && row < rows_src) ? src[row + (size_t)col * lds_src] : 0.0;
                }
        };
        mfma_d4 o[2], onext[2];
        load_v(kfirst, vreg);
        load_o(kfirst, o);
        for (int k0 = kfirst; k0 < n; k0 += QM_KC) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) Vl[(vcg + 8 * u) * QM_LDV + vkk] = vreg[u];
            if (k0 + QM_KC < n) {
                load_v(k0 + QM_KC, vreg);
                load_o(k0 + QM_KC, onext);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < KBLK / 4; ++ks) {
                const int kr = 4 * ks + lq;                                  // reflector index
                const double bv = Wl[kr * QM_LDJ + 16 * w + lr];             // B[k][row] = W2n[row][k]
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const double av = Vl[kr * QM_LDV + 16 * it + lr];        // A[col][k] = V[col][k]
                    o[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, o[it], 0, 0, 0);
                }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = k0 + 16 * it + lq + 4 * r;
                    if (col < n && row < ldw) {
                        W[row + (size_t)col * ldw] = o[it][r];
                        if (last && col < rankA) dpart += o[it][r] * p1[col];
                    }
                }
            o[0] = onext[0];
            o[1] = onext[1];
        }
        if (last) {
            // columns left of the last block's first chunk still contribute to d (blocks > 0 only)
            for (int col = lq; col < kfirst && col < rankA; col += 4)
                if (row < ldw) dpart += W[row + (size_t)col * ldw] * p1[col];
            dpart += __shfl_xor(dpart, 16, WAVE);
            dpart += __shfl_xor(dpart, 32, WAVE);
            if (lq == 0) dred[16 * w + lr] = dpart;
        }
        __syncthreads();
    }
    if (tid < QM_RB) {
        const int row = row0 + tid;
        if (row < ldw) W[row + (size_t)n * ldw] = (row < m) ? (-dred[tid] - rx[row]) : 0.0;
    }
}

inline void launch_jq1_mfma(const JQ1Args& a, int batch, hipStream_t s) {
    GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_MFMA);
    hipLaunchKernelGGL(k_jq1_mfma, dim3((a.ldw + QM_RB - 1) / QM_RB, batch), dim3(256), 0, s, a);
}

}  // namespace gn
