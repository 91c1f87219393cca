// Trailing update of the CAQR, fourth form: C <- C - V (T' (V' C)) for one group (a tile of 64*RPL rows
// at level 0, F stacked 32-row blocks at a tree level) x one block of 32 trailing columns per workgroup,
// both GEMM-shaped products on v_mfma_f64_16x16x4_f64.
//
// What bounds this kernel on MI355X is the per-CU load path (TA), not HBM and not the matrix pipe:
// a fragment-shaped global access (16 columns x 64 B per wave-instruction, i.e. the 4 lanes of a quad
// in 4 different cache lines) is serviced at ~18 B/clk/CU, a full-line access (quads contiguous) at
// ~60 B/clk/CU (tests/microbench/l2_hit_bw.hip).  The third form (gn_kernels_update_v3.hpp) fetched C
// and V fragment-shaped and spent 25k clk per workgroup in the TA, exactly its run time.  Here every
// global access is full-line:
//
//   the group's rows are dealt to the 4 waves in units of 32 (unit G = w + 4 g belongs to wave w), so a
//   partial group keeps all waves busy; lane (lr = l & 15, lq = l >> 4) owns, in its wave's unit g, the
//   row pair (2 lr, 2 lr + 1) — one 16-byte access — of column (lq + 4 r) of a 16-column tile:
//       cp[g][ct][r] = C[rows of unit + 2 lr + {0,1}][col 16 ct + lq + 4 r]
//       v[ks]        = V[same rows                  ][vcol 4 ks + lq]
//   i.e. a wave-instruction touches 4 columns x 256 contiguous bytes.  Addresses are a wave-uniform base
//   (scalar registers) plus ONE 32-bit per-lane byte offset (saddr + voffset form).
//
//   product 2 (C += V W2) runs directly on these registers in the transposed form
//       D^T[i = column lq + 4 r][j = row pair lr] += W2^T[i][k] V^T[k][j],
//   once for the even rows (p = 0) and once for the odd rows (p = 1) of the unit: v[ks][p] is the B
//   operand, W2^T comes from LDS, {cp[g][ct][0..3][p]} is the C/D fragment.
//   product 1 (W1 = V' C, contraction over rows) needs both operands with lr along COLUMNS: each wave
//   transposes its own 32 x 32 pieces of C and V through a private LDS image [col][row] (ld 34:
//   ds_write_b128 and ds_read_b64 both conflict-free), no workgroup barrier involved.
//   The only workgroup-wide step is the 32 x 32 reduction of W1 over the 4 waves, fused with
//   W2 = -T' W1 on the matrix pipe (two barriers).
//
// Loads are issued one unit ahead (V; C two units ahead) and only their finishing selects sit in the
// consuming iteration, so the compiler's waits are counted vmcnt(N) — loads and stores retire in issue
// order on this target and a wait placed right behind a load would also wait for every older store.
// Each accumulator receives its MFMAs back to back (dependent chains issue every ~30 ns, rotations only
// every 36-44 ns, tests/microbench/mfma_f64_rate.hip).
//
// MFMA f64 16x16x4 fragment maps (MI355X_MICROARCH.md): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D[i = (l>>4) + 4 r][j = l&15].
#pragma once
#include "gn_kernels_caqr.hpp"

namespace gn {

typedef double v4_d4 __attribute__((ext_vector_type(4)));
typedef v4_d4 mfma_d4;   // name used by the other MFMA kernels
typedef double v4_d2 __attribute__((ext_vector_type(2)));

#ifndef ENLSIP_V4_ABLATE
#define ENLSIP_V4_ABLATE 0
#endif
// timing experiments only (results wrong when != 0): 2 no V loads, 3 no C traffic, 4 no MFMA, 5 MFMA only, 6 no C stores, 7 no C loads
constexpr int V4_ABLATE = ENLSIP_V4_ABLATE;
#ifdef ENLSIP_V4_STAMPS         // harness only (tests/microbench/update_bench.hip): phase stamps (100 MHz) of sample workgroups
__device__ long long g_v4_stamps[8 * 8];
__device__ int g_v4_stamp_x = 3;      // group index (blockIdx.x) of the sampled workgroups (tree levels with one group: 0)
#define V4_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == g_v4_stamp_x && blockIdx.y == 5 && (blockIdx.z & 31) == 7 && blockIdx.z < 256 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_v4_stamps[(blockIdx.z >> 5) * 8 + (i)] = wall_clock64(); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define V4_STAMP(i) do { } while (0)
#endif

#ifndef ENLSIP_V4_PREFETCH_NEXT
#define ENLSIP_V4_PREFETCH_NEXT 1
#endif
// Pair kernel, last pass: the stores of unit g are issued at the START of unit g + 1, right behind the request for that unit's
// second piece of V, instead of at the end of unit g.  vmcnt counts loads and stores in ONE in-order queue on this target: a
// wait for a load that was issued after a store also waits for that store's write acknowledgement from HBM.  With the stores
// at the end of a unit the very next piece of V (requested behind them) paid that acknowledgement half a unit later.
#ifndef ENLSIP_V4_LATE_STORE
#define ENLSIP_V4_LATE_STORE 0
#endif
#ifndef ENLSIP_V4_NT_STORE
#define ENLSIP_V4_NT_STORE 1
#endif
// depth of the V ring of product 2 in pieces of half a unit: 2 = piece u + 1 is requested when piece u is consumed (half a
// unit = 16 MFMAs ahead), 3 = piece u + 2 (a whole unit ahead; 16 more registers)
#ifndef ENLSIP_V4_VB_DEPTH
#define ENLSIP_V4_VB_DEPTH 2
#endif
#ifndef ENLSIP_V4_C_AHEAD
#define ENLSIP_V4_C_AHEAD 2
#endif
#ifndef ENLSIP_V4_IMM_UNIT
#define ENLSIP_V4_IMM_UNIT 1
#endif
#ifndef ENLSIP_V4_TRI_UNIT_VOFF
#define ENLSIP_V4_TRI_UNIT_VOFF 1
#endif
#ifndef ENLSIP_V4_START_FENCE
#define ENLSIP_V4_START_FENCE 1
#endif
constexpr int V4_LD = 34;                    // leading dimension of the per-wave transpose images
constexpr int V4_IMG = PB * V4_LD;           // doubles per image (C or V), 32 columns
constexpr int V4_STAGE = 2 * V4_IMG;         // per wave: C image + V image (also hosts the wave's W1 partial)

struct V4Ctx {
    const double* Wm;     // problem base (V lives in columns col0 .. col0 + bw - 1)
    double* C;            // first trailing column
    const double* T;
    long long tile_row0, r0, gblk0, S;
    int ldw, col0, bw, cb0;
    int rows_valid;       // RMASK only: slots >= rows_valid do not exist (element granularity)
    int dshift;           // level 0: the diagonal of reflector j sits in slot j + dshift (0 for the CAQR; 32 for the second panel
                          // of a pair; the blocked pivoted QR starts its units one row early when its first row is odd, for
                          // 16-byte alignment)
    int mode;             // TRI: row geometry of the level (CaqrArgs::mode; r0 = base)
    // NAP = 2: the second block reflector applied to the same block of C (level 0 of the pair's second panel)
    const double* T2;
    int col02, bw2, dshift2;
};

struct V4NoPost {         // default post-update hook: nothing
    template <class F> __device__ __forceinline__ void operator()(int, const F&) const {}
};

// TRI   = false: level 0 (contiguous rows, V = unit lower trapezoid);  true: tree node (unit G is block
//         gblk0 + G at row stride S; V = identity in unit 0, upper triangles in the others).
// CFULL = true : 32 valid columns and a full panel (bw = 32): no validity selects.
// NGW   = number of 32-row units this wave owns (0..RPL/2), a compile-time constant so that every path
//         is straight-line code.
// NCT   = 16-column tiles in use: 1 for a last block of <= 16 columns (the carried right-hand side makes
//         the column count 32 k + 1, so every panel ends with such a block).
// RMASK = true : the last unit may be partial at element granularity (c.rows_valid); used by the blocked
//         pivoted QR, whose row range starts at an arbitrary pivot step.
// vmask / smask (per lane, bit 4 ct + r <-> column 16 ct + lq + 4 r of the block; ignored when CFULL):
//         column holds valid data / column is written back.
// post(g, fr): called with the updated fragments of unit g (fr[p][ct][r] = row pair element p of column
//         16 ct + lq + 4 r) before they are stored.
// GATHER = true: the block's 32 columns are arbitrary columns of C: coff[4 ct + r] is this lane's byte offset of
//         column (16 ct + lq + 4 r) from c.C (the blocked pivoted QR updates only its still-active columns).
// NAP   = block reflectors applied one after the other to the block of C while it stays in registers (1, or 2 for a panel
//         pair: c.T2 / col02 / bw2 / dshift2 describe the second one; level 0 only).
template <int RPL, bool TRI, bool CFULL, int NGW, int NCT, bool RMASK = false, class Post = V4NoPost, bool GATHER = false, int NAP = 1>
__device__ __forceinline__ void v4_body(const V4Ctx& c, const int w, double (*stage)[V4_STAGE], double* W2l,
                                        const unsigned vmask = ~0u, const unsigned smask = ~0u, const Post& post = Post(),
                                        const unsigned* coff = nullptr) {
    static_assert(NAP == 1 || !TRI, "pairs are applied at level 0 only");
    const int ln = lane_id();
    const int lr = ln & 15, lq = ln >> 4;
    const int it2 = w >> 1, ct2 = w & 1;             // W2 tile produced by this wave in the reduction step
    double* Cs = stage[w];
    double* Vs = stage[w] + V4_IMG;
    const unsigned lane_byte = (unsigned)(((size_t)lq * c.ldw + 2 * lr) * sizeof(double));   // < 2^32: m <= 2^27 (check_limits)
    auto slot0 = [&](int g) -> int { return 32 * (w + 4 * g); };                              // first slot of the wave's unit g
    auto rowu = [&](int g) -> long long {            // uniform: global row of that slot
        if (!TRI) return c.tile_row0 + slot0(g);
        const long long q = c.gblk0 + (w + 4 * g);
        if (c.mode == 2) return c.r0 + ((q + 1) >> 1) * c.S + ((q + 1) & 1) * 32;
        return c.r0 + q * c.S;
    };
    auto ap_T = [&](int ai) -> const double* { return ai ? c.T2 : c.T; };
    auto ap_col0 = [&](int ai) -> int { return ai ? c.col02 : c.col0; };
    auto ap_bw = [&](int ai) -> int { return ai ? c.bw2 : c.bw; };
    auto ap_dshift = [&](int ai) -> int { return ai ? c.dshift2 : c.dshift; };
    const bool dense_unit = TRI && c.mode == 2 && (w & 1);   // mode 2: odd blocks are dense (gblk0 is even)
    // Level 0: unit g of this wave starts 128 g doubles behind its unit 0, so ONE uniform base per column (independent of g)
    // serves every unit and 1024 g bytes ride in the immediate offset of the access: 8 + 8 per block reflector scalar bases
    // instead of one per (unit, column) — those did not fit the scalar file (1200 scalar spills into vector lanes, the reloads
    // in the middle of the MFMA chains).
    // Tree levels: the units of a wave are S rows apart (not a compile-time distance), and one uniform base per (unit, column) —
    // 64 of them for C and V — overflowed the scalar file: the tree kernel carried 158 scalar spills into vector lanes and ~1000
    // v_readlane reloads, 46 k instructions against 33 k for the pair kernel that does twice the work.  The unit's distance from the
    // wave's unit 0 now rides in the per-lane offset (ONE more 32-bit register per unit): 8 + 8 uniform bases as at level 0.
    // (m <= 2^27 - 4096, check_limits: lane offset + unit distance < 2^32 bytes.)
    const long long row00 = rowu(0);                                                           // uniform
    unsigned uoff[NGW > 0 ? NGW : 1];
    if (TRI && !GATHER && ENLSIP_V4_TRI_UNIT_VOFF) {
#pragma unroll
        for (int g = 0; g < NGW; ++g) uoff[g] = lane_byte + (unsigned)((rowu(g) - row00) * (long long)sizeof(double));
    }
    auto cptr = [&](int g, int ct, int r) -> double* {
        if (GATHER) return (double*)((char*)(c.C + rowu(g)) + (coff[4 * ct + r] + 16u * (unsigned)lr));
        if (!TRI && ENLSIP_V4_IMM_UNIT) {
            double* ub = c.C + (size_t)(c.cb0 + 16 * ct + 4 * r) * c.ldw + (c.tile_row0 + 32 * w);      // uniform
            return (double*)((char*)ub + lane_byte) + 128 * g;
        }
        if (TRI && ENLSIP_V4_TRI_UNIT_VOFF) {
            double* ub = c.C + (size_t)(c.cb0 + 16 * ct + 4 * r) * c.ldw + row00;     // uniform, the same for every unit
            return (double*)((char*)ub + uoff[g]);
        }
        double* ub = c.C + (size_t)(c.cb0 + 16 * ct + 4 * r) * c.ldw + rowu(g);      // uniform
        return (double*)((char*)ub + lane_byte);
    };
    auto vptr = [&](int ai, int g, int ks) -> const double* {
        if (!TRI && ENLSIP_V4_IMM_UNIT) {
            const double* ub = c.Wm + (size_t)(ap_col0(ai) + 4 * ks) * c.ldw + (c.tile_row0 + 32 * w);      // uniform
            return (const double*)((const char*)ub + lane_byte) + 128 * g;
        }
        if (TRI && !GATHER && ENLSIP_V4_TRI_UNIT_VOFF) {
            const double* ub = c.Wm + (size_t)(ap_col0(ai) + 4 * ks) * c.ldw + row00;      // uniform, the same for every unit
            return (const double*)((const char*)ub + uoff[g]);
        }
        const double* ub = c.Wm + (size_t)(ap_col0(ai) + 4 * ks) * c.ldw + rowu(g);       // uniform
        return (const double*)((const char*)ub + lane_byte);
    };

    // ---- loaders (issue) and their finishing selects (run in the consuming iteration) -------------------
    // Columns past the last valid one are still read (the workspace carries 32 spare columns per problem,
    // make_plan) and discarded by finish_*.
    v4_d2 cp[NGW > 0 ? NGW : 1][2][4];
    auto issue_c = [&](int g) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (V4_ABLATE == 3 || V4_ABLATE == 5 || V4_ABLATE == 7) cp[g][ct][r] = (v4_d2){(double)(g + r), (double)ct};
                else cp[g][ct][r] = __builtin_nontemporal_load((const v4_d2*)cptr(g, ct, r));   // streamed once: keep it out of L2's way
            }
    };
    auto finish_c = [&](int g) {
        if (CFULL) return;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (vmask >> (4 * ct + r)) & 1u;
                const int s0 = slot0(g) + 2 * lr;
                cp[g][ct][r][0] = (ok && (!RMASK || s0 < c.rows_valid)) ? cp[g][ct][r][0] : 0.0;
                cp[g][ct][r][1] = (ok && (!RMASK || s0 + 1 < c.rows_valid)) ? cp[g][ct][r][1] : 0.0;
            }
    };
    // V operands are fetched in halves of 4 contraction steps: vh[k4] = V[row pair][vcol 16 h + 4 k4 + lq]
    auto issue_v = [&](int ai, int g, int h, v4_d2 (&vh)[4]) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            if (V4_ABLATE == 2 || V4_ABLATE == 5) vh[k4] = (v4_d2){(double)(g + k4), 1.0};
            else vh[k4] = *(const v4_d2*)vptr(ai, g, 4 * h + k4);
        }
    };
    // lrx / lqx: the calling phase's own (opaque) copy of the lane coordinates, so that the masks are recomputed per phase
    // instead of being kept — i.e. spilled — from product 1 to product 2
    auto finish_v = [&](int ai, int g, int h, v4_d2 (&vh)[4], const int lrx, const int lqx) {   // structure of V: unit trapezoid / identity + triangles
        const int s0 = slot0(g) + 2 * lrx;
        const int dsh = ap_dshift(ai);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int j = 16 * h + 4 * k4 + lqx;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int s = s0 + p;
                double x = vh[k4][p];
                if (!TRI) {
                    if (g == 0) x = (s > j + dsh) ? x : ((s == j + dsh) ? 1.0 : 0.0);   // only units 0..3 can meet the diagonal (dshift <= 32)
                } else {
                    const double tri = (dense_unit || (s & 31) <= j) ? x : 0.0;
                    x = (g == 0 && s < PB) ? ((s == j) ? 1.0 : 0.0) : tri;
                }
                if (!CFULL) x = (j < ap_bw(ai) && (!RMASK || s < c.rows_valid)) ? x : 0.0;
                vh[k4][p] = x;
            }
        }
    };

    // One pass per block reflector (NAP of them); the block of C stays in registers from the loads of the first pass to the
    // stores of the last one: as loaded (row-pair quads cp) through product 1 of the first pass, in the accumulator layout of
    // product 2 (cf) from then on.  Later passes build their transpose images from cf with 64-bit LDS stores and the rows of
    // a unit permuted (row 2 lr + p -> slot lr + 16 p, in BOTH images: the contraction index of product 1 may be any
    // bijection), so that no pass needs the block in two register layouts at once.
    v4_d4 cf[NAP > 1 ? (NGW > 0 ? NGW : 1) : 1][2][2];
    v4_d2 vnext[2][4];                       // NAP > 1: V of the NEXT pass's first unit, requested during the last unit of product 2
#pragma unroll
    for (int ai = 0; ai < NAP; ++ai) {
        const bool first_app = (ai == 0), last_app = (ai == NAP - 1);
        // LDS addresses are formed from per-phase opaque copies of the lane index: the compiler otherwise computes the
        // address registers of EVERY phase of BOTH passes at kernel entry (common subexpressions) and spills them across the
        // phases in between — and a scratch reload is a vector-memory operation: its wait is vmcnt(0), which also drains
        // every prefetch in flight (19 such drains per workgroup in the hot variant).
        int lnp = ln;
        asm volatile("" : "+v"(lnp));
        const int lrp = lnp & 15, lqp = lnp >> 4;
        // ---- product 1: per-wave partial W1 = V' C ------------------------------------------------------------
        if (first_app) V4_STAMP(0);
        v4_d4 acc[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[it][ct] = (v4_d4){0.0, 0.0, 0.0, 0.0};
        // T' operand of the fused reduction step: A[i = k][kk = l] = T[l][k] (upper triangular)
        double tA[8];
        if (NGW == 0) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) tA[ks] = ap_T(ai)[4 * ks + lq + (16 * it2 + lr) * PB];
        }
        if (NGW > 0) {
            v4_d2 vp[2][4];
            if (first_app || !ENLSIP_V4_PREFETCH_NEXT) {
                issue_v(ai, 0, 0, vp[0]);
                issue_v(ai, 0, 1, vp[1]);
            } else {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) { vp[0][k4] = vnext[0][k4]; vp[1][k4] = vnext[1][k4]; }
            }
            if (first_app) {
                issue_c(0);
                if (NGW > 1) issue_c(1);
                if (ENLSIP_V4_C_AHEAD > 2) {                 // every unit's block of C is requested at once (its registers exist anyway)
#pragma unroll
                    for (int g2 = 2; g2 < NGW; ++g2) issue_c(g2);
                }
                // every request of the first two units is on its way before the first select on a loaded value (the scheduler
                // used to hoist V's structure selects above the C loads: two memory latencies in a row at kernel start)
                if (ENLSIP_V4_START_FENCE) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < NGW; ++g) {
                // transpose images of unit g (the previous unit's reads were issued before: LDS keeps a wave's order)
                finish_v(ai, g, 0, vp[0], lrp, lqp);
                finish_v(ai, g, 1, vp[1], lrp, lqp);
                if (first_app) {
                    finish_c(g);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) *(v4_d2*)&Vs[(4 * ks + lqp) * V4_LD + 2 * lrp] = vp[ks >> 2][ks & 3];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) *(v4_d2*)&Cs[(16 * ct + lqp + 4 * r) * V4_LD + 2 * lrp] = cp[g][ct][r];
                } else {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                        for (int p = 0; p < 2; ++p) Vs[(4 * ks + lqp) * V4_LD + 16 * p + lrp] = vp[ks >> 2][ks & 3][p];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int p = 0; p < 2; ++p) Cs[(16 * ct + lqp + 4 * r) * V4_LD + 16 * p + lrp] = cf[g][p][ct][r];
                }
                // next operands: V one unit ahead (same registers), C two units ahead (its own registers)
                if (g + 1 < NGW) {
                    issue_v(ai, g + 1, 0, vp[0]);
                    issue_v(ai, g + 1, 1, vp[1]);
                }
                if (first_app && ENLSIP_V4_C_AHEAD <= 2 && g + 2 < NGW) issue_c(g + 2);
                if (g == NGW - 1) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) tA[ks] = ap_T(ai)[4 * ks + lq + (16 * it2 + lr) * PB];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    double av[2][4], bv[2][4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const int kr = 16 * kh + 4 * k4 + lqp;           // row of the unit = contraction index
#pragma unroll
                        for (int it = 0; it < 2; ++it) av[it][k4] = Vs[(16 * it + lrp) * V4_LD + kr];
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) bv[ct][k4] = Cs[(16 * ct + lrp) * V4_LD + kr];
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                            for (int k4 = 0; k4 < 4; ++k4) {
                                if (V4_ABLATE == 4) acc[it][ct][k4] += av[it][k4] + bv[ct][k4];
                                else acc[it][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[it][k4], bv[ct][k4], acc[it][ct], 0, 0, 0);
                            }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (first_app) V4_STAMP(1);
        if (!first_app) V4_STAMP(3);          // pair: product 1 of the second pass done
        constexpr int VBD = ENLSIP_V4_VB_DEPTH;
        v4_d2 vb[VBD][4];                                    // ring over (unit, half)
        if (NGW > 0) issue_v(ai, 0, 0, vb[0]);               // travels during the reduction step
        if (NGW > 0 && VBD > 2) issue_v(ai, 0, 1, vb[1]);

        // ---- reduction over waves fused with W2 = -T' W1 (the partial goes to the wave's own stage) ----------
        int lnr = ln;
        asm volatile("" : "+v"(lnr));
        const int lrr = lnr & 15, lqr = lnr >> 4;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[w][(16 * it + lqr + 4 * r) * PB + 16 * ct + lrr] = acc[it][ct][r];
        if (first_app && NAP == 1) V4_STAMP(2);
        __syncthreads();
        if (first_app && NAP == 1) V4_STAMP(3);
        {
            v4_d4 t = (v4_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int l = 4 * ks + lqr, k = 16 * it2 + lrr;
                const int o = l * PB + 16 * ct2 + lrr;
                const double b = (stage[0][o] + stage[1][o]) + (stage[2][o] + stage[3][o]);
                const double ta = (l <= k && k < ap_bw(ai)) ? tA[ks] : 0.0;
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, b, t, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) W2l[(16 * it2 + lqr + 4 * r) * PB + ((16 * ct2 + lrr) ^ (16 * (lqr & 1)))] = -t[r];   // swizzled, see the read
        }
        __syncthreads();
        if (first_app) V4_STAMP(4);
        if (!first_app) V4_STAMP(7);          // pair: second reduction done

        // ---- product 2: D^T[col][row pair] += W2^T V^T, even and odd rows of each unit; stored by the last pass,
        //      back into the registers otherwise ---------------------------------------------------------------------
        int lnq = ln;
        asm volatile("" : "+v"(lnq));
        const int lrq = lnq & 15, lqq = lnq >> 4;
        constexpr bool LATE = (NAP > 1) && ENLSIP_V4_LATE_STORE && !RMASK;
        auto store_unit = [&](int g, const v4_d4 (&fr)[2][2]) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (V4_ABLATE == 3 || V4_ABLATE == 5 || V4_ABLATE == 6) {
                        if (fr[0][ct][r] == 1.2345e301) c.C[0] = fr[1][ct][r];
                    } else if (CFULL || ((smask >> (4 * ct + r)) & 1u)) {
                        if (!RMASK) {
                            if (ENLSIP_V4_NT_STORE) __builtin_nontemporal_store((v4_d2){fr[0][ct][r], fr[1][ct][r]}, (v4_d2*)cptr(g, ct, r));
                            else *(v4_d2*)cptr(g, ct, r) = (v4_d2){fr[0][ct][r], fr[1][ct][r]};
                        } else {
                            const int s0 = slot0(g) + 2 * lr;
                            if (s0 + 1 < c.rows_valid) *(v4_d2*)cptr(g, ct, r) = (v4_d2){fr[0][ct][r], fr[1][ct][r]};
                            else if (s0 < c.rows_valid) *cptr(g, ct, r) = fr[0][ct][r];
                        }
                    }
                }
        };
#pragma unroll
        for (int g = 0; g < NGW; ++g) {
            v4_d4 fr[2][2];                                  // [p][ct]
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int ct = NCT; ct < 2; ++ct) fr[p][ct] = (v4_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    if (first_app) fr[p][ct] = (v4_d4){cp[g][ct][0][p], cp[g][ct][1][p], cp[g][ct][2][p], cp[g][ct][3][p]};
                    else fr[p][ct] = cf[g][p][ct];
                }
            if (ENLSIP_V4_PREFETCH_NEXT && !last_app && g == NGW - 1) {       // the next pass's first unit of V travels during this unit
                issue_v(ai + 1, 0, 0, vnext[0]);
                issue_v(ai + 1, 0, 1, vnext[1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int u = 2 * g + h;                     // piece index; piece u + 1 is fetched while u is used
                constexpr int AH = VBD - 1;                  // pieces ahead
                if (u + AH < 2 * NGW) issue_v(ai, (u + AH) >> 1, (u + AH) & 1, vb[(u + AH) % VBD]);
                if (LATE && last_app && h == 0 && g > 0) {    // the previous unit's results, parked in cf: both pieces of THIS unit are in flight
                    __builtin_amdgcn_sched_barrier(0);
                    store_unit(g - 1, cf[g - 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                finish_v(ai, g, h, vb[u % VBD], lrq, lqq);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    double a2[4];
#pragma unroll
                    // A[i = col][k]; rows k and k + 1 (the two 16-lane groups of a half-wave) are 256 B apart = same banks, so odd
                    // rows are stored with their two column halves exchanged: conflict-free ds_read_b64
                    for (int k4 = 0; k4 < 4; ++k4) a2[k4] = W2l[(16 * h + 4 * k4 + lqq) * PB + ((16 * ct + lrq) ^ (16 * (lqq & 1)))];
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            if (V4_ABLATE == 4) fr[p][ct][k4] += a2[k4] + vb[u % VBD][k4][p];
                            else fr[p][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[k4], vb[u % VBD][k4][p], fr[p][ct], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!last_app) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) cf[g][p][ct] = fr[p][ct];
                if (g == NGW - 1) V4_STAMP(2);   // pair: product 2 of the first pass done
                continue;
            }
            post(g, fr);
            if (LATE && g + 1 < NGW) {                       // stored at the start of the next unit (see ENLSIP_V4_LATE_STORE)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) cf[g][p][ct] = fr[p][ct];
            } else store_unit(g, fr);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    V4_STAMP(5);
}

template <int RPL, bool TRI, bool CFULL, int NCT, bool RMASK = false, class Post = V4NoPost, bool GATHER = false, int NAP = 1>
__device__ __forceinline__ void v4_dispatch(const V4Ctx& c, int w, int ngw, double (*stage)[V4_STAGE], double* W2l,
                                            const unsigned vmask = ~0u, const unsigned smask = ~0u, const Post& post = Post(),
                                            const unsigned* coff = nullptr) {
    constexpr int NG = RPL / 2;
    // every variant executes exactly two workgroup barriers per block reflector, so waves of one workgroup may take different ones
    if (ngw >= NG) v4_body<RPL, TRI, CFULL, NG, NCT, RMASK, Post, GATHER, NAP>(c, w, stage, W2l, vmask, smask, post, coff);
    else if (NG > 3 && ngw == 3) v4_body<RPL, TRI, CFULL, (NG > 3 ? 3 : 0), NCT, RMASK, Post, GATHER, NAP>(c, w, stage, W2l, vmask, smask, post, coff);
    else if (NG > 2 && ngw == 2) v4_body<RPL, TRI, CFULL, (NG > 2 ? 2 : 0), NCT, RMASK, Post, GATHER, NAP>(c, w, stage, W2l, vmask, smask, post, coff);
    else if (NG > 1 && ngw == 1) v4_body<RPL, TRI, CFULL, (NG > 1 ? 1 : 0), NCT, RMASK, Post, GATHER, NAP>(c, w, stage, W2l, vmask, smask, post, coff);
    else v4_body<RPL, TRI, CFULL, 0, NCT, RMASK, Post, GATHER, NAP>(c, w, stage, W2l, vmask, smask, post, coff);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The carried right-hand side d = Q'(-J1 p1 - rx) is the LAST trailing column of every panel, so the trailing column count is
// always 32 j + 1: in the block kernel above that one column cost a whole 16-column block per tile and panel — 13 % of the
// level-0 workgroups of a C2 solve, each reading 15 spare columns from HBM for nothing.  This routine applies the tile's block
// reflector(s) to that single column instead: d -= V (T' (V' d)) with the tile's V read ONCE (row pairs, 16-byte full-line loads,
// kept in registers), the 32 dot products reduced with the batched wave reductions, T' applied by one wave; for a panel pair the
// second panel's reflector follows while d stays in registers.
// Runs as the LAST block index (blockIdx.y) of the block kernel's own grid, so that the tile's V is read while the column blocks of
// the same tile keep it hot in the XCD's L2 (as a launch of its own, after the block kernel, it re-read 400 MB of reflectors per
// panel from HBM and gave the gain back).  256 threads: thread = row pair of a 512-row tile (RPL = 8); RPL = 4 uses 128 of them.
// ---------------------------------------------------------------------------------------------------------------------------------
// The 32 x 32 product w2 = -T' w1 used to read T from global memory inside a lane-bounded loop (one exposed load latency per
// iteration: ~20 us per block reflector, which made this workgroup as expensive as a whole 32-column block of the MFMA body:
// ~0.3 ms of every far-update launch of a C2 step).  T is now requested first of all (4 doubles per thread, coalesced), parked
// in LDS with leading dimension 33 (lane k reads T[l + 33 k]: conflict-free) and the product runs fully unrolled on registers.
template <int RPL, bool PAIR>
__device__ __forceinline__ void v4_rhs_body(const CaqrArgs& a, double (*part)[PB], double* w2s, double* Tl, const int g) {
    constexpr int TLD = PB + 1;
    constexpr int NWV = 4;                                               // waves of the block kernel's workgroup
    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0a = a.panel * PB;
    if (r0a >= st.kp) return;
    const double* Wm = a.W + prob * a.sW;
    double* dcol = a.W + prob * a.sW + (size_t)a.n * a.ldw;
    const long long tile_row0 = a.base + (long long)g * a.F * 32;
    const long long blocks_here = (long long)a.nblocks - (long long)g * a.F;
    const int nvu = (int)(blocks_here < a.F ? blocks_here : a.F);        // valid 32-row units of this tile
    const int tid = threadIdx.x, ln = lane_id(), w = wave_id();
    const int s0 = 2 * tid;                                              // first slot of the thread's row pair
    const bool live = (s0 < 64 * RPL) && (s0 >> 5) < nvu;
    v4_d2 dd = (v4_d2){0.0, 0.0};
    if (live) dd = *(const v4_d2*)(dcol + tile_row0 + s0);
    const bool has_next = caqr_next_bw(st.kp, r0a) > 0;                  // uniform over the workgroup
    if (PAIR && !has_next) return;                                       // served by the plain kernel (a.pair == 2)
    if (!PAIR && a.pair == 2 && has_next) return;
    constexpr int napp = PAIR ? 2 : 1;
    // the T blocks of every application: 1024 doubles each = 4 per thread (256 threads), in flight before anything else
    v4_d2 tl[napp][2];
#pragma unroll
    for (int ai = 0; ai < napp; ++ai) {
        const double* T = a.Tbuf + prob * a.sT + ((ai ? a.tOff2 : a.tOff) + g) * (long long)(PB * PB);
        tl[ai][0] = *(const v4_d2*)(T + 2 * tid);
        tl[ai][1] = *(const v4_d2*)(T + 512 + 2 * tid);
    }
#pragma unroll
    for (int ai = 0; ai < napp; ++ai) {
        const int r0 = r0a + PB * ai;
        const int bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
        const int col0 = st.rankA + r0;
        const int dsh = 32 * (a.skip + ai);                              // slot of reflector j's diagonal = j + dsh
        // V row pairs (structure of the tile's first rows: zero above the diagonal, one on it; columns >= bw do not exist)
        v4_d2 v[PB];
        const double* vrow = Wm + (size_t)col0 * a.ldw + tile_row0 + s0;
        if (live) {
#pragma unroll
            for (int j = 0; j < PB; ++j) v[j] = *(const v4_d2*)(vrow + (size_t)j * a.ldw);
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int s = s0 + p;
                double x = live ? v[j][p] : 0.0;
                if (s0 < PB + dsh) x = (s > j + dsh) ? x : ((s == j + dsh) ? 1.0 : 0.0);
                v[j][p] = (j < bw && live) ? x : 0.0;
            }
        }
        // w1 = V' d: per-thread partial products, batched wave reductions, partials of the waves through LDS
#pragma unroll
        for (int j8 = 0; j8 < PB; j8 += 8) {
            double pr[8], rs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pr[u] = v[j8 + u][0] * dd[0] + v[j8 + u][1] * dd[1];
            wave_allsum8(pr, rs);
            if (ln == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) part[w][j8 + u] = rs[u];
            }
        }
        // T (column-major, element e = l + 32 k) to LDS as Tl[l + 33 k]; the previous application's readers passed the barrier
        // that closed it
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int e = 512 * q + 2 * tid + e2;
                Tl[(e & 31) + TLD * (e >> 5)] = tl[ai][q][e2];
            }
        __syncthreads();
        // w2 = -T' w1 (T upper triangular, column-major): lane k of wave 0; w1[l] travels by readlane
        if (w == 0) {
            double w1 = 0.0;
            if (ln < PB) {
#pragma unroll
                for (int ww = 0; ww < NWV; ++ww) w1 += part[ww][ln];
            }
            double acc = 0.0;
            const int lk = ln & 31;
#pragma unroll
            for (int l = 0; l < PB; ++l) {
                const double w1l = readlane_f64(w1, l);
                const double tv = Tl[l + TLD * lk];
                acc += (l <= lk) ? tv * w1l : 0.0;
            }
            if (ln < PB) w2s[ln] = (ln < bw) ? -acc : 0.0;
        }
        __syncthreads();
        if (live) {
            double d0 = dd[0], d1 = dd[1];
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                const double wk = w2s[k];
                d0 += v[k][0] * wk;
                d1 += v[k][1] * wk;
            }
            dd = (v4_d2){d0, d1};
        }
    }
    if (live) *(v4_d2*)(dcol + tile_row0 + s0) = dd;
}

#ifndef ENLSIP_V4_CW
#define ENLSIP_V4_CW 32          // columns per workgroup (experiment: 16 with 3 workgroups per CU)
#endif

// Context of one workgroup of the block kernels below.  Returns false when the workgroup has nothing to do.
// PAIR: the far update of a panel pair (both level-0 reflectors in one pass): only problems that HAVE the pair's second panel;
// the plain kernel with a.pair == 2 serves the others (mixed-rank batches in the second attempt of a solve).
template <bool TRI, bool PAIR>
__device__ __forceinline__ bool v4_setup(const CaqrArgs& a, V4Ctx& c, int& nvu, int& ncols, const int g, const int yb) {
    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0 = a.panel * PB;
    if (r0 >= st.kp) return false;
    c.bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
    c.col0 = st.rankA + r0;
    // column window (pairs): the next panel's columns only / everything beyond them
    const int bwn = (a.win || a.pair) ? caqr_next_bw(st.kp, r0) : 0;
    if (PAIR && bwn == 0) return false;
    if (!PAIR && a.pair == 2 && bwn > 0) return false;
    int first = r0 + c.bw;
    ncols = st.n2 + 1 - first - ((!TRI && a.skip_rhs) ? 1 : 0);
    if (a.win == 1) ncols = bwn;
    else if (a.win == 2) { first += bwn; ncols -= bwn; }
    first += a.sub0; ncols -= a.sub0;                                    // sub-range of the window (look-ahead sweep)
    if (a.subn > 0 && ncols > a.subn) ncols = a.subn;
    c.cb0 = yb * ENLSIP_V4_CW;
    if (c.cb0 >= ncols) return false;
    c.rows_valid = 0;
    c.dshift = 32 * a.skip;
    c.mode = a.mode;
    c.Wm = a.W + prob * a.sW;
    c.C = a.W + prob * a.sW + (size_t)(st.rankA + first) * a.ldw;
    c.T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
    c.gblk0 = (long long)g * a.F;
    c.r0 = a.base;
    c.S = a.S;
    c.ldw = a.ldw;
    c.tile_row0 = a.base + c.gblk0 * 32;
    c.T2 = a.Tbuf + prob * a.sT + (a.tOff2 + g) * (long long)(PB * PB);
    c.col02 = c.col0 + PB;
    c.bw2 = bwn;
    c.dshift2 = c.dshift + 32;
    const long long blocks_here = (long long)a.nblocks - (long long)g * a.F;
    nvu = (int)(blocks_here < a.F ? blocks_here : a.F);                  // valid 32-row units of this group
    return true;
}
__device__ __forceinline__ unsigned v4_colmask(int cb0, int ncols) {     // valid (= stored) columns of this lane
    unsigned cmask = 0u;
    const int lq = lane_id() >> 4;
#pragma unroll
    for (int b = 0; b < 8; ++b) cmask |= (cb0 + 16 * (b >> 2) + 4 * (b & 3) + lq < ncols) ? (1u << b) : 0u;
    return cmask;
}

template <int RPL, bool TRI>
__global__ __launch_bounds__(256, ENLSIP_V4_CW == 16 ? 3 : 2) void k_caqr_update_v4(CaqrArgs a) {
    __shared__ __attribute__((aligned(16))) double stage[4][V4_STAGE];   // per-wave transpose images / W1 partials
    __shared__ __attribute__((aligned(16))) double W2l[PB * PB];         // W2 = -T' W1   [k][j]

    V4_STAMP(6);
    if (!TRI && a.skip_rhs && blockIdx.y == gridDim.y - 1) {             // the carried right-hand side of this tile
        v4_rhs_body<RPL, false>(a, reinterpret_cast<double (*)[PB]>(&stage[0][0]), W2l, &stage[1][0], blockIdx.x);
        return;
    }
    V4Ctx c;
    int nvu, ncols;
    if (!v4_setup<TRI, false>(a, c, nvu, ncols, blockIdx.x, blockIdx.y)) return;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int ngw = nvu > w ? (nvu - w + 3) / 4 : 0;                     // units w, w + 4, ... < nvu
    const bool cfull = (ncols - c.cb0 >= 32) && (c.bw == PB);
    const unsigned cmask = v4_colmask(c.cb0, ncols);
    if (ENLSIP_V4_CW == 16) {
        if ((ncols - c.cb0 >= 16) && (c.bw == PB)) v4_dispatch<RPL, TRI, true, 1>(c, w, ngw, stage, W2l);
        else v4_dispatch<RPL, TRI, false, 1>(c, w, ngw, stage, W2l, cmask, cmask);
    } else if (cfull) v4_dispatch<RPL, TRI, true, 2>(c, w, ngw, stage, W2l);
    else if (ncols - c.cb0 <= 16) v4_dispatch<RPL, TRI, false, 1>(c, w, ngw, stage, W2l, cmask, cmask);
    else v4_dispatch<RPL, TRI, false, 2>(c, w, ngw, stage, W2l, cmask, cmask);
}

// Far update of a panel pair: level-0 reflectors of panel a.panel AND of a.panel + 1 applied to every column beyond the pair
// in ONE pass (the block of C stays in registers between the two): half the HBM traffic of two passes.
template <int RPL>
__global__ __launch_bounds__(256, 2) void k_caqr_update_v4_pair(CaqrArgs a) {
    __shared__ __attribute__((aligned(16))) double stage[4][V4_STAGE];
    __shared__ __attribute__((aligned(16))) double W2l[PB * PB];

    V4_STAMP(6);
    // Grid.  Batches: x = tile, y = column block, z = problem — the 8 tiles x ~13 column blocks of a problem are consecutive
    // workgroups and share the tile's V out of the XCD's L2.  ONE problem with hundreds of tiles (a.xmap: C4, a 32768-row shard):
    // in that order the 13 uses of a tile's V lie 511 workgroups = 131 MB of other tiles' V apart, every one of them a miss of the
    // 4 MB L2.  Consecutive workgroups go to the 8 XCDs round-robin, so the launch is made 8 * ceil(tiles / 8) wide and the
    // linear index L is read as (XCD = L mod 8, s = L / 8): the XCD runs through ITS tiles (those = XCD mod 8) with the column
    // blocks of a tile back to back: s = tile_local * ny + column block.  A tile's V is then fetched once per pass.
    int g = blockIdx.x, yb = blockIdx.y, ny = gridDim.y;
    if (a.xmap) {
        const int L = blockIdx.x + gridDim.x * blockIdx.y;
        const int sidx = L >> 3;
        const int tl = sidx / ny;
        yb = sidx - tl * ny;
        g = 8 * tl + (L & 7);
        if (g >= a.xmap) return;                                         // a.xmap = number of tiles; the padding of the last octet
    }
    if (a.skip_rhs && yb == ny - 1) {
        v4_rhs_body<RPL, true>(a, reinterpret_cast<double (*)[PB]>(&stage[0][0]), W2l, &stage[1][0], g);
        return;
    }
    V4Ctx c;
    int nvu, ncols;
    if (!v4_setup<false, true>(a, c, nvu, ncols, g, yb)) return;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int ngw = nvu > w ? (nvu - w + 3) / 4 : 0;
    const bool cfull = (ncols - c.cb0 >= 32) && (c.bw == PB) && (c.bw2 == PB);
    const unsigned cmask = v4_colmask(c.cb0, ncols);
    if (cfull) v4_dispatch<RPL, false, true, 2, false, V4NoPost, false, 2>(c, w, ngw, stage, W2l);
    else if (ncols - c.cb0 <= 16) v4_dispatch<RPL, false, false, 1, false, V4NoPost, false, 2>(c, w, ngw, stage, W2l, cmask, cmask);
    else v4_dispatch<RPL, false, false, 2, false, V4NoPost, false, 2>(c, w, ngw, stage, W2l, cmask, cmask);
}

inline void launch_update_v4(int RPL, const CaqrArgs& a, int groups, int ncols, int batch, hipStream_t s) {
    // level 0 with skip_rhs: ncols counts the J2 columns only; one more block index carries the right-hand side
    dim3 grid(groups, (ncols + ENLSIP_V4_CW - 1) / ENLSIP_V4_CW + ((a.level == 0 && a.skip_rhs) ? 1 : 0), batch);
    if (a.level == 0 && a.pair == 1) {
        if (a.xmap) grid.x = 8 * ((groups + 7) / 8);                     // whole octets of tiles (see the kernel)
        if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_v4_pair<8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_caqr_update_v4_pair<4>), grid, dim3(256), 0, s, a);
    } else if (a.level == 0) {
        if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_v4<8, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_caqr_update_v4<4, false>), grid, dim3(256), 0, s, a);
    } else {
        if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_v4<8, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_caqr_update_v4<4, true>), grid, dim3(256), 0, s, a);
    }
}

}  // namespace gn
