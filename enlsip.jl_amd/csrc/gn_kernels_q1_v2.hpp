// J * F_A.Q on v_mfma_f64_16x16x4_f64 (src/enlsip_functions.jl:219), fused with d_temp = -J1 p1 - rx
// (:134 / :145), second form (n <= 512).  Same mathematics as gn_kernels_q1_mfma.hpp — per block of <= 64
// reflectors  W1 = Jrb V,  W2 = -W1 T,  Jrb += W2 V'  — laid out like the trailing update
// (gn_kernels_update_v4.hpp):
//
//  * a workgroup (4 waves) owns 32 rows of J and ALL n columns; the n/16 column tiles are dealt round-robin
//    to the waves and live in registers from the first load to the final store (J crosses HBM once, W once):
//        jt[T][r][p] = J[row0 + 2 lr + p][16 tile(T) + lq + 4 r]        (lane lr = l & 15, lq = l >> 4)
//    i.e. 16-byte row pairs, 4 columns x 256 contiguous bytes per wave-instruction (full-rate load path);
//  * the SAME registers are the A operand of product 1 (k-step r contracts over the columns lq + 4 r of the
//    tile; the V operand uses the same bijection) and the C/D fragment of product 3 in its transposed form
//    D^T[i = column lq + 4 r][j = row pair lr], once for the even rows (p = 0) and once for the odd rows;
//  * V (n x 64 per block, L2-resident, shared by every workgroup of the problem) is staged tile by tile
//    — 16 rows x 64 reflectors, loaded as row pairs one tile ahead of the MFMAs — into a wave-private LDS
//    image [reflector][16] (ld 18), read as B operand of product 1 and as A operand of product 3: no workgroup
//    barrier in either product;
//  * the only workgroup-wide step is the reduction of W1 over the waves (through 32 KB of LDS that alias
//    the V images).  T is applied to V once per problem instead (k_vt: Vt = V T', 4 MFLOP), so that
//    J -= (J V) Vt' needs no second small product between the two big ones: that step used to cost 11 of the
//    53 us of a workgroup (T from L2, 2 x 16 dependent MFMAs behind three barriers).
// Work per workgroup: 2 x 1024 + 128 MFMAs for 32 rows, i.e. the kernel is bound by the FP64 matrix pipe
// (4 m n t flop: 1.96 ms for the C2 batch of 256 at the 70 TF/s the pipe sustains; measured 3.9 ms).
// The body must stay free of runtime loops around the tile code: loop-invariant code motion otherwise hoists every
// address and mask of the unrolled tiles and spills hundreds of bytes per lane.
#pragma once
#include "gn_kernels_q1.hpp"
#include "gn_kernels_update_v4.hpp"

#ifndef ENLSIP_JQ1_PREFETCH_VT
#define ENLSIP_JQ1_PREFETCH_VT 1
#endif
#ifndef ENLSIP_JQ1_ABLATE
#define ENLSIP_JQ1_ABLATE 0     // timing-only ablations for tests/microbench/jq1_bench.hip: 1 = no MFMAs, 2 = no J loads / W stores,
#endif                          // 3 = no V staging (wrong results in every non-zero mode)

#ifdef ENLSIP_JQ1_STAMPS        // harness only: phase stamps (100 MHz) of a few sample workgroups
__device__ long long g_jq1_stamps[8 * 8];
#define JQ1_STAMP(i) do { if (blockIdx.x == 37 && (blockIdx.y & 31) == 3 && blockIdx.y < 256 && threadIdx.x == 0) g_jq1_stamps[(blockIdx.y >> 5) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define JQ1_STAMP(i) do { } while (0)
#endif

namespace gn {

constexpr int Q2_LDV = 18;                   // V image: [reflector][16 tile columns], conflict-free as B and as A operand
constexpr int Q2_VS = KBLK * Q2_LDV;         // doubles per wave image

// Vt = V T' for one reflector block (kA = 64): Vt[c][k] = sum_{j = k}^{min(c, 63)} V[c][j] T[k][j], V the unit lower
// trapezoid held in FA, T upper triangular.  One thread per (row c, reflector k); T through LDS.
__global__ __launch_bounds__(256) void k_vt(JQ1Args a) {
    __shared__ __attribute__((aligned(16))) double Ts[KBLK * KBLK];      // Ts[k][j] = T[k][j] (zero below the diagonal)
    const int n = a.n;
    const int prob = blockIdx.y + a.prob0;
    const double* FA = a.FA + prob * a.sFA;
    const double* TA = a.TA + prob * a.sTA;
    double* VT = a.VT + prob * a.sVT;
    for (int e = threadIdx.x; e < KBLK * KBLK; e += 256) {
        const int k = e & 63, j = e >> 6;                                  // TA[k + j * 64]
        Ts[k * KBLK + j] = (k <= j) ? TA[e] : 0.0;
    }
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cc = c < n ? c : n - 1;
    // the thread's row of V in registers (lanes along rows: coalesced), then 16 reflectors k = g, g + 4, ...
    double v[KBLK];
#pragma unroll
    for (int j = 0; j < KBLK; ++j) {
        const double x = FA[cc + (size_t)j * n];
        v[j] = (cc > j) ? x : (cc == j ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int k = threadIdx.x >> 6; k < KBLK; k += 4) {
        const double* tk = Ts + k * KBLK;          // broadcast reads; entries left of the diagonal are zero
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j < KBLK; j += 2) {
            s0 += v[j] * tk[j];
            s1 += v[j + 1] * tk[j + 1];
        }
        if (c < n) VT[c + (size_t)k * n] = s0 + s1;
    }
}

template <int NTW, int Q2_NW, int NP>     // NTW tiles per wave, Q2_NW waves: n = 16 * NTW * Q2_NW; NP row fragments: 16 NP rows
__global__ __launch_bounds__(64 * Q2_NW, 2) void k_jq1_v2(JQ1Args a) {
    constexpr int Q2_RB = 16 * NP;
    constexpr int Q2_PS = Q2_RB * KBLK;          // doubles of one W1 partial (D layout, lane-linear)
    constexpr int Q2_LDW = (NP == 2) ? 48 : 16;  // W2 image: [reflector][row slots], lq step = 32 banks
    constexpr int NS = Q2_NW / 2;          // partial-sum slots of the reduction
    // Vs[w] (phases 1 / 3) and the partial-sum slots share one region
    __shared__ __attribute__((aligned(16))) double ush[(Q2_NW * Q2_VS > NS * Q2_PS) ? Q2_NW * Q2_VS : NS * Q2_PS];
    __shared__ __attribute__((aligned(16))) double W2s[KBLK * Q2_LDW];
    __shared__ double dred[Q2_NW][Q2_RB];

    const int n = a.n, ldw = a.ldw;
    const int prob = blockIdx.y + a.prob0;
    const int row0 = blockIdx.x * Q2_RB;
    const double* Jin = a.J + prob * a.strideJ;
    const double* rx = a.rx + prob * a.stride_rx;
    const double* FA = a.FA + prob * a.sFA;
    const double* VT = a.VT + prob * a.sVT;
    const double* p1 = a.p1 + prob * a.sP1;
    double* W = a.W + prob * a.sW;
    const int rankA = a.state[prob].rankA;
    const int tid = threadIdx.x, ln = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int lr = ln & 15, lq = ln >> 4;
    double* Vs = ush + w * Q2_VS;

    JQ1_STAMP(0);
    // ---- J tiles -> registers (shape restrictions of launch_jq1_v2: every tile and every row pair exists) -------
    // addresses = wave-uniform base + one 32-bit lane offset (saddr + voffset form, no 64-bit address registers)
    const unsigned jlane = (unsigned)(((size_t)lq * a.ldj + NP * lr) * 8);     // row (pair) NP lr, column lq of a 4-column step
    const unsigned wlane = (unsigned)(((size_t)lq * ldw + NP * lr) * 8);
    // p1 entries of the wave's first tile (columns < kA = 64 live in tiles 0..3 = T 0 of every wave) and the rx entry of the
    // final row: the epilogue must have no load queued behind its own stores (loads and stores retire in order under vmcnt: a
    // load after 32 tile stores waits for their HBM acknowledgements).  Until round 5 they were fetched here, at kernel entry,
    // and their ten registers were live through both products: the <8, 4, 2> instance (C2) spilled four J values (76 B of scratch,
    // reloaded in the middle of product 1).  They are now requested at the start of the LAST tile of product 3, when the
    // accumulators of product 1 are long dead: the epilogue still finds them arrived, and nothing spills.
    static_assert(16 * Q2_NW >= KBLK, "p1 columns must lie in the first tile of every wave");
    double pcv[4];
    double rxv = 0.0;
    double jt[NTW][4][NP];
#pragma unroll
    for (int T = 0; T < NTW; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const char* src = (const char*)(Jin + (size_t)(16 * (w + Q2_NW * T) + 4 * r) * a.ldj + row0) + jlane;
            if (ENLSIP_JQ1_ABLATE == 2) {
                jt[T][r][0] = (double)tid; jt[T][r][NP - 1] = 1.0;
            } else if (NP == 2) {
                const v4_d2 x = __builtin_nontemporal_load((const v4_d2*)src);     // J is streamed once
                jt[T][r][0] = x[0];
                jt[T][r][NP - 1] = x[1];
            } else {
                jt[T][r][0] = *(const double*)src;
            }
        }

    {   // one block of 64 reflectors (launch_jq1_v2 restricts the fast path to kA == 64)
        constexpr int b = 0, c0 = 0;
        // V image of one tile: rows 16 tile .. +15 of V_b (64 reflectors), unit lower trapezoid masked in.
        // lane: row pair (ln & 7), reflectors (ln >> 3) + 8 q.  fetch_v issues the loads, put_v finishes them.
        const int rp = ln & 7, jc = ln >> 3;
        const unsigned vlane = (unsigned)(((size_t)jc * n + 2 * rp) * 8);
        auto fetch_v = [&](int tile, v4_d2 (&x)[8]) {
            if (ENLSIP_JQ1_ABLATE == 3) return;
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = *(const v4_d2*)((const char*)(FA + (size_t)(c0 + 8 * q) * n + 16 * tile) + vlane);
        };
        auto put_v = [&](int tile, const v4_d2 (&x)[8]) {
            if (ENLSIP_JQ1_ABLATE == 3) return;
            const int gr = 16 * tile + 2 * rp;
            if (16 * tile >= c0 + KBLK) {          // tile below the trapezoid's triangle (wave-uniform): V is dense there
#pragma unroll
                for (int q = 0; q < 8; ++q) *(v4_d2*)&Vs[(jc + 8 * q) * Q2_LDV + 2 * rp] = x[q];
                return;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int gc = c0 + jc + 8 * q;
                v4_d2 y;
                y[0] = (gr > gc) ? x[q][0] : ((gr == gc) ? 1.0 : 0.0);
                y[1] = (gr + 1 > gc) ? x[q][1] : ((gr + 1 == gc) ? 1.0 : 0.0);
                *(v4_d2*)&Vs[(jc + 8 * q) * Q2_LDV + 2 * rp] = y;
            }
        };
        auto fetch_vt = [&](int tile, v4_d2 (&x)[8]) {     // the same tile of Vt = V T' (dense: no structure to mask in)
            if (ENLSIP_JQ1_ABLATE == 3) return;
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = *(const v4_d2*)((const char*)(VT + (size_t)(8 * q) * n + 16 * tile) + vlane);
        };
        auto put_vt = [&](int tile, const v4_d2 (&x)[8]) {
            if (ENLSIP_JQ1_ABLATE == 3) return;
#pragma unroll
            for (int q = 0; q < 8; ++q) *(v4_d2*)&Vs[(jc + 8 * q) * Q2_LDV + 2 * rp] = x[q];
        };
        auto tile_on = [&](int T) -> bool {               // wave-uniform: V_b is zero above row c0
            const int tile = w + Q2_NW * T;
            return T < NTW && 16 * tile + 15 >= c0;
        };
        // ---- phase 1: per-wave partial W1 = J V_b over the wave's tiles --------------------------------
        JQ1_STAMP(1);
        mfma_d4 acc[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) acc[p][jg] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
        {
            v4_d2 vx[8];
            if (tile_on(0)) fetch_v(w, vx);
#pragma unroll
            for (int T = 0; T < NTW; ++T) {
                const int tile = w + Q2_NW * T;
                if (tile_on(T)) {
                    put_v(tile, vx);
                    if (tile_on(T + 1)) fetch_v(tile + Q2_NW, vx);        // next tile's V travels during the MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jg = 0; jg < 4; ++jg) {
                        double bv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) bv[r] = Vs[(16 * jg + lr) * Q2_LDV + lq + 4 * r];   // B[k = col][j = reflector]
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (ENLSIP_JQ1_ABLATE == 1) acc[p][jg][r] += jt[T][r][p] + bv[r];
                                else acc[p][jg] = __builtin_amdgcn_mfma_f64_16x16x4f64(jt[T][r][p], bv[r], acc[p][jg], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);     // keep the chain of 4 on one accumulator together
                        }
                    }
                } else if (tile_on(T + 1)) {
                    fetch_v(tile + Q2_NW, vx);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- phase 2: W1 = sum over waves (slots of waves 4..7, then += waves 0..3), W2 = -W1 T_b ------------
        // partial layout: lane-linear, index ((p * 4 + jg) * 4 + r) * 64 + ln
        JQ1_STAMP(2);
        // the first Vt tile of phase 3 travels during the reduction (it used to be requested after it: one exposed L2 round trip
        // per workgroup)
        v4_d2 vx3[8];
        if (ENLSIP_JQ1_PREFETCH_VT && tile_on(0)) fetch_vt(w, vx3);
        __syncthreads();                                   // every wave is done with its V image
        JQ1_STAMP(3);
        {
            double* ps = ush + (w % NS) * Q2_PS;
            if (w >= NS) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int jg = 0; jg < 4; ++jg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ps[((p * 4 + jg) * 4 + r) * 64 + ln] = acc[p][jg][r];
            }
            __syncthreads();
            if (w < NS) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int jg = 0; jg < 4; ++jg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ps[((p * 4 + jg) * 4 + r) * 64 + ln] += acc[p][jg][r];
            }
            __syncthreads();
        }
        // W2s[reflector][row] = -(sum of the partial slots); wave w takes the fragments jg = w, w + Q2_NW, ...
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            if ((jg % Q2_NW) != w) continue;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = ((p * 4 + jg) * 4 + r) * 64 + ln;
                    double sacc = ush[e];
#pragma unroll
                    for (int q = 1; q < NS; ++q) sacc += ush[q * Q2_PS + e];
                    W2s[(16 * jg + lr) * Q2_LDW + 16 * p + lq + 4 * r] = -sacc;
                }
        }
        __syncthreads();
        JQ1_STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 3: tile^T += V_b W2^T  (D^T[i = column lq + 4 r][j = row pair lr]) -------------------------
        {
            v4_d2 vx[8];
            if (tile_on(0)) {
                if (ENLSIP_JQ1_PREFETCH_VT) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) vx[q] = vx3[q];
                } else fetch_vt(w, vx);
            }
#pragma unroll
            for (int T = 0; T < NTW; ++T) {
                const int tile = w + Q2_NW * T;
                if (T == NTW - 1) {                        // the epilogue's small operands (see the top of the kernel)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = 16 * w + lq + 4 * r;
                        pcv[r] = (col < rankA) ? p1[col < rankA ? col : 0] : 0.0;
                    }
                    rxv = (tid < Q2_RB) ? rx[row0 + tid] : 0.0;
                }
                if (tile_on(T)) {
                    put_vt(tile, vx);
                    if (tile_on(T + 1)) fetch_vt(tile + Q2_NW, vx);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_d4 fr[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) fr[p] = (mfma_d4){jt[T][0][p], jt[T][1][p], jt[T][2][p], jt[T][3][p]};
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) {
                        double av[4], bw[NP][4];
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const int k = 16 * kq + 4 * k4 + lq;                      // reflector
                            av[k4] = Vs[k * Q2_LDV + lr];                             // A[i = column][k]
#pragma unroll
                            for (int p = 0; p < NP; ++p) bw[p][k4] = W2s[k * Q2_LDW + 16 * p + lr];   // B[k][j = row]
                        }
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
#pragma unroll
                            for (int k4 = 0; k4 < 4; ++k4)
                                if (ENLSIP_JQ1_ABLATE == 1) fr[p][k4] += av[k4] + bw[p][k4];
                                else fr[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k4], bw[p][k4], fr[p], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int p = 0; p < NP; ++p) jt[T][r][p] = fr[p][r];
                } else if (tile_on(T + 1)) {
                    fetch_vt(tile + Q2_NW, vx);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        JQ1_STAMP(5);
        __syncthreads();      // W2s and the V images are rewritten by the next block
    }

    // ---- store the tiles, accumulate d = -J1 p1 - rx ---------------------------------------------------------
    double dp[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        dp[p] = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[p] += jt[0][r][p] * pcv[r];
    }
#pragma unroll
    for (int T = 0; T < NTW; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            char* dst = (char*)(W + (size_t)(16 * (w + Q2_NW * T) + 4 * r) * ldw + row0) + wlane;   // W 256-byte aligned
            if (ENLSIP_JQ1_ABLATE == 2) { if (jt[T][r][0] == 1.2345) *(double*)dst = 0.0; }
            else if (NP == 2) __builtin_nontemporal_store((v4_d2){jt[T][r][0], jt[T][r][NP - 1]}, (v4_d2*)dst);
            else *(double*)dst = jt[T][r][0];
        }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        dp[p] += __shfl_xor(dp[p], 16, WAVE);
        dp[p] += __shfl_xor(dp[p], 32, WAVE);
        if (lq == 0) dred[w][NP * lr + p] = dp[p];
    }
    __syncthreads();
    if (tid < Q2_RB) {
        const int row = row0 + tid;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < Q2_NW; ++q) s += dred[q][tid];
        W[row + (size_t)n * ldw] = -s - rxv;
    }
    JQ1_STAMP(6);
}

// The second form is the straight-line fast path for regular shapes: n a multiple of 128 (<= 512), m a multiple of
// 32, kA = 64 (one reflector block), 16-byte aligned even-ld J.  Returns false otherwise (the caller uses launch_jq1_mfma).
inline bool launch_jq1_v2(const JQ1Args& a, int batch, hipStream_t s) {
    if (a.n > 512 || a.n % 128 != 0 || a.m % 32 != 0 || a.kA != KBLK) return false;
    if ((a.ldj & 1) || (a.strideJ & 1) || ((size_t)a.J & 15)) return false;
    // NP = 2 (32 rows per workgroup) measured 3.9 ms on the C2 batch of 256, NP = 1 4.3 ms, the first form 4.2 ms
    if (!a.VT) return false;
    hipLaunchKernelGGL(k_vt, dim3((a.n + 63) / 64, batch), dim3(256), 0, s, a);
    dim3 grid(a.m / 32, batch);
    dim3 blk(256);
    switch (a.n / 128) {
        case 1: GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_V2_N128); hipLaunchKernelGGL((k_jq1_v2<2, 4, 2>), grid, blk, 0, s, a); break;
        case 2: GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_V2_N256); hipLaunchKernelGGL((k_jq1_v2<4, 4, 2>), grid, blk, 0, s, a); break;
        case 3: GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_V2_N384); hipLaunchKernelGGL((k_jq1_v2<6, 4, 2>), grid, blk, 0, s, a); break;
        default: GN_ROUTE(ENLSIP_GN_ROUTE_JQ1_V2_N512); hipLaunchKernelGGL((k_jq1_v2<8, 4, 2>), grid, blk, 0, s, a); break;
    }
    return true;
}

}  // namespace gn
