// EXPERIMENT (round 2), NOT part of the library: the level-0 trailing update as ONE STREAM over the units of a tile,
//     for every 32-row unit:  load C, C += V_k W2_k (W2_k given), store C, then W1_{k+1} += V_{k+1}' C_new
// i.e. the structure the update would have with CAQR tiles at FIXED row boundaries and a one-panel look-ahead (W2 of panel k comes
// from the previous pass, the reflectors of panel k + 1 exist before the bulk update of panel k runs): no reduction step in the
// middle of the kernel, the block of C is never held beyond one unit.  Same MFMA work, same HBM bytes (+ the 8 KB W1 / W2 records)
// as the fourth form.  The harness (update_bench_stream.hip) feeds synthetic W2 and uses the panel's own V in the role of V_{k+1};
// it checks C against C - V W2 and W1 against V' C_new in host loops.
#pragma once
#include "gn_kernels_update_v4.hpp"

namespace gn {

struct StreamArgs {
    CaqrArgs a;
    const double* W2in;   // per (problem, tile, column block): 32 x 32, [k][j] row-major as W2l of the fourth form (unswizzled)
    double* W1out;        // same indexing: V_next' C_new of the tile
    long long sW2;        // doubles per problem
    int ncb;              // column blocks per tile
};

#ifndef ENLSIP_STREAM_OCC
#define ENLSIP_STREAM_OCC 2
#endif

template <int RPL>
__global__ __launch_bounds__(256, ENLSIP_STREAM_OCC) void k_update_stream(StreamArgs sa) {
    const CaqrArgs& a = sa.a;
    __shared__ __attribute__((aligned(16))) double stage[4][V4_STAGE];   // per-wave transpose images / W1 partials
    __shared__ __attribute__((aligned(16))) double W2l[PB * PB];         // W2 of panel k   [k][j], swizzled as in the fourth form
    constexpr int NGW = RPL / 2;
    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0 = a.panel * PB;
    const int g = blockIdx.x;
    const int col0 = st.rankA + r0;
    const int first = r0 + PB;
    const int cb0 = blockIdx.y * 32;
    const double* Wm = a.W + prob * a.sW;
    double* C = a.W + prob * a.sW + (size_t)(st.rankA + first) * a.ldw;
    const long long tile_row0 = (long long)r0 + (long long)g * a.F * 32;
    const int ldw = a.ldw;
    const int ln = lane_id(), lr = ln & 15, lq = ln >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const size_t rec = ((size_t)g * sa.ncb + blockIdx.y) * (PB * PB);
    const double* W2g = sa.W2in + prob * sa.sW2 + rec;
    // W2 of this (tile, column block) into LDS
    for (int e = threadIdx.x; e < PB * PB; e += 256) {
        const int k = e >> 5, j = e & 31;
        W2l[k * PB + (j ^ (16 * ((k >> 0) & 1)))] = W2g[e];
    }
    double* Cs = stage[w];
    double* Vs = stage[w] + V4_IMG;
    const unsigned lane_byte = (unsigned)(((size_t)lq * ldw + 2 * lr) * sizeof(double));
    auto rowu = [&](int gg) -> long long { return tile_row0 + 32 * (w + 4 * gg); };
    auto cptr = [&](int gg, int ct, int r) -> double* {
        double* ub = C + (size_t)(cb0 + 16 * ct + 4 * r) * ldw + rowu(gg);
        return (double*)((char*)ub + lane_byte);
    };
    auto vptr = [&](int gg, int ks) -> const double* {
        const double* ub = Wm + (size_t)(col0 + 4 * ks) * ldw + rowu(gg);
        return (const double*)((const char*)ub + lane_byte);
    };
    auto finish_v = [&](int gg, int h, v4_d2 (&vh)[4]) {
        const int s0 = 32 * (w + 4 * gg) + 2 * lr;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int j = 16 * h + 4 * k4 + lq;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int s = s0 + p;
                double x = vh[k4][p];
                if (gg == 0) x = (s > j) ? x : ((s == j) ? 1.0 : 0.0);
                vh[k4][p] = x;
            }
        }
    };
    __syncthreads();

    v4_d4 acc[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[it][ct] = (v4_d4){0.0, 0.0, 0.0, 0.0};

    v4_d2 cp[2][2][4];        // C of the current and the next unit
    v4_d2 vb[2][2][4];        // V of the current and the next unit, halves
    auto issue = [&](int gg, int slot) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[slot][ct][r] = __builtin_nontemporal_load((const v4_d2*)cptr(gg, ct, r));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) vb[slot][h][k4] = *(const v4_d2*)vptr(gg, 4 * h + k4);
    };
    issue(0, 0);
#pragma unroll
    for (int gg = 0; gg < NGW; ++gg) {
        const int cur = gg & 1;
        if (gg + 1 < NGW) issue(gg + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        finish_v(gg, 0, vb[cur][0]);
        finish_v(gg, 1, vb[cur][1]);
        // ---- product 2 with the given W2: D^T[col][row pair] += W2^T V^T ------------------------------------------------------
        v4_d4 fr[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) fr[p][ct] = (v4_d4){cp[cur][ct][0][p], cp[cur][ct][1][p], cp[cur][ct][2][p], cp[cur][ct][3][p]};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                double a2[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) a2[k4] = W2l[(16 * h + 4 * k4 + lq) * PB + ((16 * ct + lr) ^ (16 * (lq & 1)))];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        fr[p][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[k4], vb[cur][h][k4][p], fr[p][ct], 0, 0, 0);
            }
        // ---- store the updated unit -------------------------------------------------------------------------------------------
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_nontemporal_store((v4_d2){fr[0][ct][r], fr[1][ct][r]}, (v4_d2*)cptr(gg, ct, r));
        // ---- product 1 of the NEXT panel on the updated unit: W1 += V_next' C_new (the harness uses this panel's V) ----------------
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) *(v4_d2*)&Vs[(4 * ks + lq) * V4_LD + 2 * lr] = vb[cur][ks >> 2][ks & 3];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) *(v4_d2*)&Cs[(16 * ct + lq + 4 * r) * V4_LD + 2 * lr] = (v4_d2){fr[0][ct][r], fr[1][ct][r]};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            double av[2][4], bv[2][4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int kr = 16 * kh + 4 * k4 + lq;
#pragma unroll
                for (int it = 0; it < 2; ++it) av[it][k4] = Vs[(16 * it + lr) * V4_LD + kr];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bv[ct][k4] = Cs[(16 * ct + lr) * V4_LD + kr];
            }
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        acc[it][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[it][k4], bv[ct][k4], acc[it][ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- W1 of the tile: reduce the waves' partials, hand the record to the next pass ------------------------------------------
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[w][(16 * it + lq + 4 * r) * PB + 16 * ct + lr] = acc[it][ct][r];
    __syncthreads();
    double* W1g = sa.W1out + prob * sa.sW2 + rec;
    for (int e = threadIdx.x; e < PB * PB; e += 256) W1g[e] = (stage[0][e] + stage[1][e]) + (stage[2][e] + stage[3][e]);
}

inline void launch_update_stream(const StreamArgs& sa, int groups, int ncols, int batch, hipStream_t s) {
    dim3 grid(groups, ncols / 32, batch);
    hipLaunchKernelGGL((k_update_stream<8>), grid, dim3(256), 0, s, sa);
}

}  // namespace gn
