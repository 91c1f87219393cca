// EXPERIMENT (round 2), NOT part of the library: measured slower than the fourth form on MI355X in every variant
// (tests/microbench/update_bench5.hip, profiles/r2_notes.md: panel 0, batch 384: v4 2.27 ms; 16 columns + MFMA transposes at
// 3 workgroups per CU 2.62 ms, at 2 per CU 2.82, at 4 per CU (148 spilled registers) 3.78; 16 columns + LDS images at 3 per CU
// 2.44; 32 columns + MFMA transposes 2.57).  Kept as the record of what was tried and as a harness for the next idea.
//
// Level-0 trailing update of the CAQR, fifth form: C <- C - V (T' (V' C)) for one 64*RPL-row tile x one block of 16 NCT
// trailing columns per workgroup.  What changed against the fourth form (gn_kernels_update_v4.hpp, still used for the tree
// levels and the gathered block updates of the pivoted QR), and why:
//
//   * The fourth form sits at 0.55 of the HBM spec AND 0.55 of the matrix pipe: neither saturated.  A workgroup lives 28 us
//     against 23 us of data movement, and only ONE other workgroup per CU covers its reduction step and first-byte latencies
//     (profiles/r1_notes.md): 256 registers per lane (the 512 x 32 block of C stays in registers from load to store) and 78 KB of
//     LDS (per-wave transpose images of C and V) pin it at two workgroups per CU.
//   * Here the V operand of product 1 (W1 = V' C) comes from a second copy of the tile's reflectors that the panel factorisation
//     leaves in OPERAND ORDER (`Vop`: per 32-row unit, lane l and register q hold V[row slot(q, l >> 4)][column 16 h + (l & 15)],
//     unit diagonal and zeros written out) — one 16-byte full-line load per two MFMAs, no transposition, no structure selects.
//   * C is transposed on the matrix pipe instead of through LDS: with the loaded row pairs as the A operand and a 0/1 selector
//     as B, four chained v_mfma_f64_16x16x4 turn C[row pair lr][column lq + 4 r] into D[row (l >> 4) + 4 reg][column l & 15],
//     which IS the B operand layout of product 1.  Exact (x * 1 + 0), +25 % MFMA work, no LDS traffic, no LDS latency.
//   * With a column block of 16 (NCT = 1) the block of C is 64 registers and LDS holds only the 32 x 16 W1 partials and W2:
//     three to four workgroups per CU instead of two.  V is then read twice as often per byte of C — from the XCD's L2, where the
//     tile's reflectors stay (grid x = tile index = XCD at C2's eight tiles).
//
// Everything else is the fourth form's: row-pair (16-byte, full-line) accesses of C and V, product 2 on the loaded registers in
// the transposed form, units of 32 rows dealt round-robin to the 4 waves, counted vmcnt waits, non-temporal C stream.
#pragma once
#include "gn_kernels_update_v4.hpp"

namespace gn {

// ---- Vop: the reflectors of a level-0 tile in the operand order of W1 = V' C (built on the host by the harness) ----------------
#ifndef ENLSIP_V5_MT
#define ENLSIP_V5_MT 1           // 1: C transposed by selector MFMAs; 0: through per-wave LDS images (A/B partner)
#endif
constexpr bool V5_MT = ENLSIP_V5_MT != 0;
constexpr int V5_UNIT = PB * PB;                   // doubles of one 32-row unit of Vop
// row of a unit (0..31) that register q (0..7) contracts over in lane group k = l >> 4 (0..3) of product 1.  MFMA transposes
// deliver q = 4 p + reg <-> row 2 (k + 4 reg) + p; the LDS form reads its images at rows 4 q + k (conflict-free at ld 34).
__host__ __device__ constexpr int v5_slot(int q, int k) { return V5_MT ? 2 * (k + 4 * (q & 3)) + (q >> 2) : 4 * q + k; }
// position of (unit-local row s, panel column j) inside a unit of Vop
__host__ __device__ inline int v5_vop_index(int s, int j) {
    int q, k;
    if (V5_MT) { const int p = s & 1, i = s >> 1; k = i & 3; q = 4 * p + (i >> 2); }
    else { q = s >> 2; k = s & 3; }
    const int h = j >> 4, l = (k << 4) | (j & 15);
    return (((h * 4 + (q >> 1)) * 64 + l) << 1) | (q & 1);
}
struct V5Args { CaqrArgs a; const double* Vop; long long sVop; };

#ifndef ENLSIP_V5_NCT
#define ENLSIP_V5_NCT 1          // 16-column tiles per workgroup (1: 16 columns, 2: 32 columns)
#endif
#ifndef ENLSIP_V5_OCC
#define ENLSIP_V5_OCC (ENLSIP_V5_NCT == 1 ? 3 : 2)     // workgroups per CU the register budget is sized for
#endif

constexpr int V5_NCT = ENLSIP_V5_NCT;
constexpr int V5_CW = 16 * V5_NCT;                 // columns per workgroup

struct V5Ctx {
    V4Ctx c;
    const double* Vop;       // this tile's units, operand order
};

constexpr int V5_IMG = V5_MT ? 0 : V5_CW * V4_LD;                              // LDS form: C image per wave
constexpr int V5_PART = PB * V5_CW;                                            // W1 partial per wave
constexpr int V5_STAGE = V5_IMG > V5_PART ? V5_IMG : V5_PART;

template <int RPL, bool CFULL, int NGW>
__device__ __forceinline__ void v5_body(const V5Ctx& x, const int w, double (*stage)[V5_STAGE], double* W2l, const unsigned vmask) {
    constexpr int NCT = V5_NCT;
    const V4Ctx& c = x.c;
    const int ln = lane_id();
    const int lr = ln & 15, lq = ln >> 4;
    const int it2 = NCT == 2 ? (w >> 1) : w, ct2 = NCT == 2 ? (w & 1) : 0;    // W2 tile produced by this wave (waves < 2 NCT)
    const bool red = w < 2 * NCT;
    const unsigned lane_byte = (unsigned)(((size_t)lq * c.ldw + 2 * lr) * sizeof(double));
    auto slot0 = [&](int g) -> int { return 32 * (w + 4 * g); };
    auto rowu = [&](int g) -> long long { return c.tile_row0 + slot0(g); };
    auto cptr = [&](int g, int ct, int r) -> double* {
        double* ub = c.C + (size_t)(c.cb0 + 16 * ct + 4 * r) * c.ldw + rowu(g);      // uniform
        return (double*)((char*)ub + lane_byte);
    };
    auto vptr = [&](int g, int ks) -> const double* {
        const double* ub = c.Wm + (size_t)(c.col0 + 4 * ks) * c.ldw + rowu(g);       // uniform
        return (const double*)((const char*)ub + lane_byte);
    };
    auto voptr = [&](int g, int h, int q2) -> const v4_d2* {
        return (const v4_d2*)(x.Vop + (size_t)(w + 4 * g) * V5_UNIT + ((h * 4 + q2) * 64 + ln) * 2);
    };

    v4_d2 cp[NGW > 0 ? NGW : 1][NCT][4];
    auto issue_c = [&](int g) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[g][ct][r] = __builtin_nontemporal_load((const v4_d2*)cptr(g, ct, r));
    };
    auto finish_c = [&](int g) {
        if (CFULL) return;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (vmask >> (4 * ct + r)) & 1u;
                cp[g][ct][r][0] = ok ? cp[g][ct][r][0] : 0.0;
                cp[g][ct][r][1] = ok ? cp[g][ct][r][1] : 0.0;
            }
    };
    auto issue_vo = [&](int g, int h, v4_d2 (&vo)[4]) {
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) vo[q2] = *voptr(g, h, q2);
    };
    auto issue_v = [&](int g, int h, v4_d2 (&vh)[4]) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) vh[k4] = *(const v4_d2*)vptr(g, 4 * h + k4);
    };
    auto finish_v = [&](int g, int h, v4_d2 (&vh)[4]) {   // structure of V: unit lower trapezoid (only unit 0 of the tile meets it)
        const int s0 = slot0(g) + 2 * lr;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int j = 16 * h + 4 * k4 + lq;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int s = s0 + p;
                double v = vh[k4][p];
                if (g == 0) v = (s > j) ? v : ((s == j) ? 1.0 : 0.0);
                if (!CFULL) v = (j < c.bw) ? v : 0.0;
                vh[k4][p] = v;
            }
        }
    };

    // ---- product 1: per-wave partial W1 = V' C ------------------------------------------------------------------------
    v4_d4 acc[2][NCT];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[it][ct] = (v4_d4){0.0, 0.0, 0.0, 0.0};
    double sel[4];                                     // selector B operand of the transposes: S_r[k = lq][j = lr] = (j == k + 4 r)
#pragma unroll
    for (int r = 0; r < 4; ++r) sel[r] = (lr == lq + 4 * r) ? 1.0 : 0.0;
    double tA[8];
    if (NGW == 0 && red) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) tA[ks] = c.T[4 * ks + lq + (16 * it2 + lr) * PB];
    }
    if (NGW > 0) {
        v4_d2 vo[2][4];
        issue_vo(0, 0, vo[0]);
        issue_vo(0, 1, vo[1]);
        issue_c(0);
        if (NGW > 1) issue_c(1);
#pragma unroll
        for (int g = 0; g < NGW; ++g) {
            finish_c(g);
            v4_d4 Ct[NCT][2];
            if (V5_MT) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        v4_d4 t = (v4_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; ++r) t = __builtin_amdgcn_mfma_f64_16x16x4f64(cp[g][ct][r][p], sel[r], t, 0, 0, 0);
                        Ct[ct][p] = t;
                    }
            } else {
                double* Cs = stage[w];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) *(v4_d2*)&Cs[(16 * ct + lq + 4 * r) * V4_LD + 2 * lr] = cp[g][ct][r];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int q = 0; q < 8; ++q) Ct[ct][q >> 2][q & 3] = Cs[(16 * ct + lr) * V4_LD + 4 * q + lq];
            }
            if (g + 2 < NGW) issue_c(g + 2);
            if (g == NGW - 1 && red) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) tA[ks] = c.T[4 * ks + lq + (16 * it2 + lr) * PB];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        acc[h][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(vo[h][q >> 1][q & 1], Ct[ct][q >> 2][q & 3], acc[h][ct], 0, 0, 0);
                if (g + 1 < NGW) issue_vo(g + 1, h, vo[h]);      // same registers, one unit ahead
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    v4_d2 vb[2][4];                                      // ring over (unit, half)
    if (NGW > 0) issue_v(0, 0, vb[0]);                   // travels during the reduction step

    // ---- reduction over waves fused with W2 = -T' W1 ---------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[w][(16 * it + lq + 4 * r) * V5_CW + 16 * ct + lr] = acc[it][ct][r];
    __syncthreads();
    if (red) {
        v4_d4 t = (v4_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int l = 4 * ks + lq, k = 16 * it2 + lr;
            const int o = l * V5_CW + 16 * ct2 + lr;
            const double b = (stage[0][o] + stage[1][o]) + (stage[2][o] + stage[3][o]);
            const double ta = (l <= k && k < c.bw) ? tA[ks] : 0.0;
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, b, t, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * ct2 + lr;
            W2l[(16 * it2 + lq + 4 * r) * V5_CW + (NCT == 2 ? (col ^ (16 * (lq & 1))) : col)] = -t[r];
        }
    }
    __syncthreads();

    // ---- product 2: D^T[col][row pair] += W2^T V^T, even and odd rows of each unit, stored at once --------------------------
#pragma unroll
    for (int g = 0; g < NGW; ++g) {
        v4_d4 fr[2][NCT];                                // [p][ct]
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) fr[p][ct] = (v4_d4){cp[g][ct][0][p], cp[g][ct][1][p], cp[g][ct][2][p], cp[g][ct][3][p]};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int u = 2 * g + h;                     // piece index; piece u + 1 is fetched while u is used
            if (u + 1 < 2 * NGW) issue_v((u + 1) >> 1, (u + 1) & 1, vb[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            finish_v(g, h, vb[u & 1]);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                double a2[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int col = 16 * ct + lr;
                    a2[k4] = W2l[(16 * h + 4 * k4 + lq) * V5_CW + (NCT == 2 ? (col ^ (16 * (lq & 1))) : col)];
                }
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        fr[p][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[k4], vb[u & 1][k4][p], fr[p][ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (CFULL || ((vmask >> (4 * ct + r)) & 1u))
                    __builtin_nontemporal_store((v4_d2){fr[0][ct][r], fr[1][ct][r]}, (v4_d2*)cptr(g, ct, r));
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int RPL, bool CFULL>
__device__ __forceinline__ void v5_dispatch(const V5Ctx& x, int w, int ngw, double (*stage)[V5_STAGE], double* W2l, const unsigned vmask) {
    constexpr int NG = RPL / 2;
    // every variant executes exactly two workgroup barriers, so waves of one workgroup may take different ones
    if (ngw >= NG) v5_body<RPL, CFULL, NG>(x, w, stage, W2l, vmask);
    else if (NG > 3 && ngw == 3) v5_body<RPL, CFULL, (NG > 3 ? 3 : 0)>(x, w, stage, W2l, vmask);
    else if (NG > 2 && ngw == 2) v5_body<RPL, CFULL, (NG > 2 ? 2 : 0)>(x, w, stage, W2l, vmask);
    else if (NG > 1 && ngw == 1) v5_body<RPL, CFULL, (NG > 1 ? 1 : 0)>(x, w, stage, W2l, vmask);
    else v5_body<RPL, CFULL, 0>(x, w, stage, W2l, vmask);
}

template <int RPL>
__global__ __launch_bounds__(256, ENLSIP_V5_OCC) void k_caqr_update_v5(V5Args va) {
    const CaqrArgs& a = va.a;
    __shared__ __attribute__((aligned(16))) double stage[4][V5_STAGE];   // per-wave W1 partials (LDS form: also the C image)
    __shared__ __attribute__((aligned(16))) double W2l[PB * V5_CW];      // W2 = -T' W1   [k][j]

    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0 = a.panel * PB;
    if (r0 >= st.kp) return;
    V5Ctx x;
    V4Ctx& c = x.c;
    c.bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
    c.col0 = st.rankA + r0;
    const int g = blockIdx.x;
    const int first = r0 + c.bw;
    const int ncols = st.n2 + 1 - first;
    c.cb0 = blockIdx.y * V5_CW;
    if (c.cb0 >= ncols) return;
    c.rows_valid = 0;
    c.dshift = 0;
    c.Wm = a.W + prob * a.sW;
    c.C = a.W + prob * a.sW + (size_t)(st.rankA + first) * a.ldw;
    c.T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
    c.gblk0 = (long long)g * a.F;
    c.r0 = r0;
    c.S = a.S;
    c.ldw = a.ldw;
    c.tile_row0 = (long long)r0 + c.gblk0 * 32;
    x.Vop = va.Vop + prob * va.sVop + (size_t)g * a.F * V5_UNIT;
    const long long blocks_here = (long long)a.nblocks - (long long)g * a.F;
    const int nvu = (int)(blocks_here < a.F ? blocks_here : a.F);        // valid 32-row units of this group
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int ngw = nvu > w ? (nvu - w + 3) / 4 : 0;                     // units w, w + 4, ... < nvu
    const bool cfull = (ncols - c.cb0 >= V5_CW) && (c.bw == PB);
    unsigned cmask = 0u;       // valid (= stored) columns of this lane
    {
        const int lq = lane_id() >> 4;
#pragma unroll
        for (int b = 0; b < 4 * V5_NCT; ++b) cmask |= (c.cb0 + 16 * (b >> 2) + 4 * (b & 3) + lq < ncols) ? (1u << b) : 0u;
    }
    if (cfull) v5_dispatch<RPL, true>(x, w, ngw, stage, W2l, ~0u);
    else v5_dispatch<RPL, false>(x, w, ngw, stage, W2l, cmask);
}

inline void launch_update_v5(int RPL, const V5Args& a, int groups, int ncols, int batch, hipStream_t s) {
    dim3 grid(groups, (ncols + V5_CW - 1) / V5_CW, batch);
    if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_v5<8>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_caqr_update_v5<4>), grid, dim3(256), 0, s, a);
}

}  // namespace gn
