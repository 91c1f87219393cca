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
// timing experiments only (results wrong when != 0): 2 no V loads, 3 no C traffic, 4 no MFMA, 5 MFMA only
constexpr int V4_ABLATE = ENLSIP_V4_ABLATE;
#ifdef ENLSIP_V4_STAMPS         // harness only (tests/microbench/update_bench.hip): phase stamps (100 MHz) of sample workgroups
__device__ long long g_v4_stamps[8 * 8];
#define V4_STAMP(i) do { if (blockIdx.x == 3 && blockIdx.y == 5 && (blockIdx.z & 31) == 7 && blockIdx.z < 256 && threadIdx.x == 0) g_v4_stamps[(blockIdx.z >> 5) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define V4_STAMP(i) do { } while (0)
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
    int dshift;           // level 0: the diagonal of reflector j sits in slot j + dshift (0 for the CAQR; the blocked
                          // pivoted QR starts its units one row early when its first row is odd, for 16-byte alignment)
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
template <int RPL, bool TRI, bool CFULL, int NGW, int NCT, bool RMASK = false, class Post = V4NoPost, bool GATHER = false>
__device__ __forceinline__ void v4_body(const V4Ctx& c, const int w, double (*stage)[V4_STAGE], double* W2l,
                                        const unsigned vmask = ~0u, const unsigned smask = ~0u, const Post& post = Post(),
                                        const unsigned* coff = nullptr) {
    const int ln = lane_id();
    const int lr = ln & 15, lq = ln >> 4;
    const int it2 = w >> 1, ct2 = w & 1;             // W2 tile produced by this wave in the reduction step
    double* Cs = stage[w];
    double* Vs = stage[w] + V4_IMG;
    const unsigned lane_byte = (unsigned)(((size_t)lq * c.ldw + 2 * lr) * sizeof(double));   // < 2^32: m <= 2^27 (check_limits)
    auto slot0 = [&](int g) -> int { return 32 * (w + 4 * g); };                              // first slot of the wave's unit g
    auto rowu = [&](int g) -> long long {            // uniform: global row of that slot
        if (!TRI) return c.tile_row0 + slot0(g);
        return c.r0 + (c.gblk0 + (w + 4 * g)) * c.S;
    };
    auto cptr = [&](int g, int ct, int r) -> double* {
        if (GATHER) return (double*)((char*)(c.C + rowu(g)) + (coff[4 * ct + r] + 16u * (unsigned)lr));
        double* ub = c.C + (size_t)(c.cb0 + 16 * ct + 4 * r) * c.ldw + rowu(g);      // uniform
        return (double*)((char*)ub + lane_byte);
    };
    auto vptr = [&](int g, int ks) -> const double* {
        const double* ub = c.Wm + (size_t)(c.col0 + 4 * ks) * c.ldw + rowu(g);       // uniform
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
                if (V4_ABLATE == 3 || V4_ABLATE == 5) cp[g][ct][r] = (v4_d2){(double)(g + r), (double)ct};
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
    auto issue_v = [&](int g, int h, v4_d2 (&vh)[4]) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            if (V4_ABLATE == 2 || V4_ABLATE == 5) vh[k4] = (v4_d2){(double)(g + k4), 1.0};
            else vh[k4] = *(const v4_d2*)vptr(g, 4 * h + k4);
        }
    };
    auto finish_v = [&](int g, int h, v4_d2 (&vh)[4]) {   // structure of V: unit trapezoid / identity + triangles
        const int s0 = slot0(g) + 2 * lr;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int j = 16 * h + 4 * k4 + lq;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int s = s0 + p;
                double x = vh[k4][p];
                if (!TRI) {
                    if (g == 0) x = (s > j + c.dshift) ? x : ((s == j + c.dshift) ? 1.0 : 0.0);   // only units 0..3 can meet the diagonal
                } else {
                    const double tri = ((s & 31) <= j) ? x : 0.0;
                    x = (g == 0 && s < PB) ? ((s == j) ? 1.0 : 0.0) : tri;
                }
                if (!CFULL) x = (j < c.bw && (!RMASK || s < c.rows_valid)) ? x : 0.0;
                vh[k4][p] = x;
            }
        }
    };

    // ---- product 1: per-wave partial W1 = V' C ------------------------------------------------------------
    V4_STAMP(0);
    v4_d4 acc[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[it][ct] = (v4_d4){0.0, 0.0, 0.0, 0.0};
    // T' operand of the fused reduction step: A[i = k][kk = l] = T[l][k] (upper triangular)
    double tA[8];
    if (NGW == 0) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) tA[ks] = c.T[4 * ks + lq + (16 * it2 + lr) * PB];
    }
    if (NGW > 0) {
        v4_d2 vp[2][4];
        issue_v(0, 0, vp[0]);
        issue_v(0, 1, vp[1]);
        issue_c(0);
        if (NGW > 1) issue_c(1);
#pragma unroll
        for (int g = 0; g < NGW; ++g) {
            // transpose images of unit g (the previous unit's reads were issued before: LDS keeps a wave's order)
            finish_v(g, 0, vp[0]);
            finish_v(g, 1, vp[1]);
            finish_c(g);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) *(v4_d2*)&Vs[(4 * ks + lq) * V4_LD + 2 * lr] = vp[ks >> 2][ks & 3];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) *(v4_d2*)&Cs[(16 * ct + lq + 4 * r) * V4_LD + 2 * lr] = cp[g][ct][r];
            // next operands: V one unit ahead (same registers), C two units ahead (its own registers)
            if (g + 1 < NGW) {
                issue_v(g + 1, 0, vp[0]);
                issue_v(g + 1, 1, vp[1]);
            }
            if (g + 2 < NGW) issue_c(g + 2);
            if (g == NGW - 1) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) tA[ks] = c.T[4 * ks + lq + (16 * it2 + lr) * PB];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                double av[2][4], bv[2][4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int kr = 16 * kh + 4 * k4 + lq;            // row of the unit = contraction index
#pragma unroll
                    for (int it = 0; it < 2; ++it) av[it][k4] = Vs[(16 * it + lr) * V4_LD + kr];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) bv[ct][k4] = Cs[(16 * ct + lr) * V4_LD + kr];
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
    V4_STAMP(1);
    v4_d2 vb[2][4];                                      // ring over (unit, half)
    if (NGW > 0) issue_v(0, 0, vb[0]);                   // travels during the reduction step

    // ---- reduction over waves fused with W2 = -T' W1 (the partial goes to the wave's own stage) ----------
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[w][(16 * it + lq + 4 * r) * PB + 16 * ct + lr] = acc[it][ct][r];
    V4_STAMP(2);
    __syncthreads();
    V4_STAMP(3);
    {
        v4_d4 t = (v4_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int l = 4 * ks + lq, k = 16 * it2 + lr;
            const int o = l * PB + 16 * ct2 + lr;
            const double b = (stage[0][o] + stage[1][o]) + (stage[2][o] + stage[3][o]);
            const double ta = (l <= k && k < c.bw) ? tA[ks] : 0.0;
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, b, t, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) W2l[(16 * it2 + lq + 4 * r) * PB + ((16 * ct2 + lr) ^ (16 * (lq & 1)))] = -t[r];   // swizzled, see the read
    }
    __syncthreads();
    V4_STAMP(4);

    // ---- product 2: D^T[col][row pair] += W2^T V^T, even and odd rows of each unit, stored at once ----------
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
                // A[i = col][k]; rows k and k + 1 (the two 16-lane groups of a half-wave) are 256 B apart = same banks, so odd
                // rows are stored with their two column halves exchanged: conflict-free ds_read_b64
                for (int k4 = 0; k4 < 4; ++k4) a2[k4] = W2l[(16 * h + 4 * k4 + lq) * PB + ((16 * ct + lr) ^ (16 * (lq & 1)))];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        if (V4_ABLATE == 4) fr[p][ct][k4] += a2[k4] + vb[u & 1][k4][p];
                        else fr[p][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[k4], vb[u & 1][k4][p], fr[p][ct], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        post(g, fr);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (V4_ABLATE == 3 || V4_ABLATE == 5) {
                    if (fr[0][ct][r] == 1.2345e301) c.C[0] = fr[1][ct][r];
                } else if (CFULL || ((smask >> (4 * ct + r)) & 1u)) {
                    if (!RMASK) {
                        __builtin_nontemporal_store((v4_d2){fr[0][ct][r], fr[1][ct][r]}, (v4_d2*)cptr(g, ct, r));
                    } else {
                        const int s0 = slot0(g) + 2 * lr;
                        if (s0 + 1 < c.rows_valid) *(v4_d2*)cptr(g, ct, r) = (v4_d2){fr[0][ct][r], fr[1][ct][r]};
                        else if (s0 < c.rows_valid) *cptr(g, ct, r) = fr[0][ct][r];
                    }
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    V4_STAMP(5);
}

template <int RPL, bool TRI, bool CFULL, int NCT, bool RMASK = false, class Post = V4NoPost, bool GATHER = false>
__device__ __forceinline__ void v4_dispatch(const V4Ctx& c, int w, int ngw, double (*stage)[V4_STAGE], double* W2l,
                                            const unsigned vmask = ~0u, const unsigned smask = ~0u, const Post& post = Post(),
                                            const unsigned* coff = nullptr) {
    constexpr int NG = RPL / 2;
    // every variant executes exactly two workgroup barriers, so waves of one workgroup may take different ones
    if (ngw >= NG) v4_body<RPL, TRI, CFULL, NG, NCT, RMASK, Post, GATHER>(c, w, stage, W2l, vmask, smask, post, coff);
    else if (NG > 3 && ngw == 3) v4_body<RPL, TRI, CFULL, (NG > 3 ? 3 : 0), NCT, RMASK, Post, GATHER>(c, w, stage, W2l, vmask, smask, post, coff);
    else if (NG > 2 && ngw == 2) v4_body<RPL, TRI, CFULL, (NG > 2 ? 2 : 0), NCT, RMASK, Post, GATHER>(c, w, stage, W2l, vmask, smask, post, coff);
    else if (NG > 1 && ngw == 1) v4_body<RPL, TRI, CFULL, (NG > 1 ? 1 : 0), NCT, RMASK, Post, GATHER>(c, w, stage, W2l, vmask, smask, post, coff);
    else v4_body<RPL, TRI, CFULL, 0, NCT, RMASK, Post, GATHER>(c, w, stage, W2l, vmask, smask, post, coff);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The carried right-hand side d = Q'(-J1 p1 - rx) is the LAST trailing column of every panel, so the trailing column count is
// always 32 j + 1: in the block kernel above that one column cost a whole 16-column block per tile and panel — 13 % of the
// level-0 workgroups of a C2 solve, each reading 15 spare columns from HBM for nothing.  This kernel applies the tile's block
// reflector to that single column instead: d -= V (T' (V' d)) with the tile's V read ONCE (row pairs, 16-byte full-line loads,
// kept in registers), the 32 dot products reduced with the batched wave reductions, T' applied by one wave.
// Runs as the LAST block index (blockIdx.y) of the block kernel's own grid, so that the tile's V is read while the column blocks of
// the same tile keep it hot in the XCD's L2 (as a launch of its own, after the block kernel, it re-read 400 MB of reflectors per
// panel from HBM and gave the gain back).  256 threads: thread = row pair of a 512-row tile (RPL = 8); RPL = 4 uses 128 of them.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int RPL>
__device__ __forceinline__ void v4_rhs_body(const CaqrArgs& a, double (*part)[PB], double* w2s) {
    constexpr int NWV = 4;                                               // waves of the block kernel's workgroup
    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0 = a.panel * PB;
    if (r0 >= st.kp) return;
    const int bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
    if (st.n2 + 1 - (r0 + bw) < 1) return;
    const int col0 = st.rankA + r0;
    const int g = blockIdx.x;
    const double* Wm = a.W + prob * a.sW;
    double* dcol = a.W + prob * a.sW + (size_t)a.n * a.ldw;
    const double* T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
    const long long tile_row0 = (long long)r0 + (long long)g * a.F * 32;
    const long long blocks_here = (long long)a.nblocks - (long long)g * a.F;
    const int nvu = (int)(blocks_here < a.F ? blocks_here : a.F);        // valid 32-row units of this tile
    const int tid = threadIdx.x, ln = lane_id(), w = wave_id();
    const int s0 = 2 * tid;                                              // first slot of the thread's row pair
    const bool live = (s0 < 64 * RPL) && (s0 >> 5) < nvu;
    // V row pairs (structure of the tile's first 32 rows: unit lower trapezoid; columns >= bw do not exist), d row pair
    v4_d2 v[PB];
    v4_d2 dd = (v4_d2){0.0, 0.0};
    const double* vrow = Wm + (size_t)col0 * a.ldw + tile_row0 + s0;
    if (live) {
#pragma unroll
        for (int j = 0; j < PB; ++j) v[j] = *(const v4_d2*)(vrow + (size_t)j * a.ldw);
        dd = *(const v4_d2*)(dcol + tile_row0 + s0);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int s = s0 + p;
            double x = live ? v[j][p] : 0.0;
            if (s0 < PB) x = (s > j) ? x : ((s == j) ? 1.0 : 0.0);
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
    __syncthreads();
    // w2 = -T' w1 (T upper triangular, column-major): lane k of wave 0
    if (w == 0 && ln < PB) {
        double acc = 0.0;
        for (int l = 0; l <= ln; ++l) {
            double w1 = 0.0;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) w1 += part[ww][l];
            acc += T[l + ln * PB] * w1;
        }
        w2s[ln] = (ln < bw) ? -acc : 0.0;
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
        *(v4_d2*)(dcol + tile_row0 + s0) = (v4_d2){d0, d1};
    }
}

#ifndef ENLSIP_V4_CW
#define ENLSIP_V4_CW 32          // columns per workgroup (experiment: 16 with 3 workgroups per CU)
#endif
template <int RPL, bool TRI>
__global__ __launch_bounds__(256, ENLSIP_V4_CW == 16 ? 3 : 2) void k_caqr_update_v4(CaqrArgs a) {
    __shared__ __attribute__((aligned(16))) double stage[4][V4_STAGE];   // per-wave transpose images / W1 partials
    __shared__ __attribute__((aligned(16))) double W2l[PB * PB];         // W2 = -T' W1   [k][j]

    V4_STAMP(6);
    if (!TRI && a.skip_rhs && blockIdx.y == gridDim.y - 1) {             // the carried right-hand side of this tile
        v4_rhs_body<RPL>(a, reinterpret_cast<double (*)[PB]>(&stage[0][0]), W2l);
        return;
    }
    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0 = a.panel * PB;
    if (r0 >= st.kp) return;
    V4Ctx c;
    c.bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
    c.col0 = st.rankA + r0;
    const int g = blockIdx.x;
    const int first = r0 + c.bw;
    const int ncols = st.n2 + 1 - first - ((!TRI && a.skip_rhs) ? 1 : 0);
    c.cb0 = blockIdx.y * ENLSIP_V4_CW;
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
    const long long blocks_here = (long long)a.nblocks - (long long)g * a.F;
    const int nvu = (int)(blocks_here < a.F ? blocks_here : a.F);        // valid 32-row units of this group
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int ngw = nvu > w ? (nvu - w + 3) / 4 : 0;                     // units w, w + 4, ... < nvu
    const bool cfull = (ncols - c.cb0 >= 32) && (c.bw == PB);
    unsigned cmask = 0u;       // valid (= stored) columns of this lane
    {
        const int lq = lane_id() >> 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) cmask |= (c.cb0 + 16 * (b >> 2) + 4 * (b & 3) + lq < ncols) ? (1u << b) : 0u;
    }
    if (ENLSIP_V4_CW == 16) {
        if ((ncols - c.cb0 >= 16) && (c.bw == PB)) v4_dispatch<RPL, TRI, true, 1>(c, w, ngw, stage, W2l);
        else v4_dispatch<RPL, TRI, false, 1>(c, w, ngw, stage, W2l, cmask, cmask);
    } else if (cfull) v4_dispatch<RPL, TRI, true, 2>(c, w, ngw, stage, W2l);
    else if (ncols - c.cb0 <= 16) v4_dispatch<RPL, TRI, false, 1>(c, w, ngw, stage, W2l, cmask, cmask);
    else v4_dispatch<RPL, TRI, false, 2>(c, w, ngw, stage, W2l, cmask, cmask);
}

inline void launch_update_v4(int RPL, const CaqrArgs& a, int groups, int ncols, int batch, hipStream_t s) {
    // level 0 with skip_rhs: ncols counts the J2 columns only; one more block index carries the right-hand side
    dim3 grid(groups, (ncols + ENLSIP_V4_CW - 1) / ENLSIP_V4_CW + ((a.level == 0 && a.skip_rhs) ? 1 : 0), batch);
    if (a.level == 0) {
        if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_v4<8, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_caqr_update_v4<4, false>), grid, dim3(256), 0, s, a);
    } else {
        if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_v4<8, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_caqr_update_v4<4, true>), grid, dim3(256), 0, s, a);
    }
}

}  // namespace gn
