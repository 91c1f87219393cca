// Unpivoted blocked Householder QR of [J2 | d_temp] as communication-avoiding QR (CAQR):
// every panel of PB = 32 columns is factored tile-locally (tiles of TR = 64*RPL rows, one
// workgroup each, tile in registers, reflector broadcast through LDS), the per-tile R factors are
// combined by a small tree of the same kernel, and the trailing matrix is updated tile-locally,
// so every element of the trailing matrix crosses HBM once per direction per panel
// (B_trail = 8 (2 m_k n_k + m_k b + b^2)).
//
// Geometry.  Rows are handled in blocks of 32.  For panel k (r0 = 32 k) and tree level l, block
// `bidx` starts at row r0 + bidx * S_l with S_l = 32 * F^l, F = TR / 32 blocks per group; group
// g of level l consists of blocks g*F .. g*F+F-1 (those < nblocks_l) and leaves its R factor in
// its first block, which is block g of level l+1.  Level 0 groups are plain tiles of TR
// contiguous rows.  Lane `ln` of every wave owns row slots s = ln + 64 i (block q = s >> 5,
// row-in-block rb = s & 31); wave w owns panel columns c = w + 4 cc.
//
// Storage (in place, LAPACK-like): level 0 leaves V strictly below the diagonal of the tile and
// R on/above it; a level >= 1 group reads only the upper triangles of its blocks, leaves R in
// block 0 and the reflector entries of blocks q >= 1 in their upper triangles (TSQR storage; the
// block-0 part of those reflectors is the identity).  T factors (dlarft) go to a side buffer.
#pragma once
#include "gn_device_utils.hpp"

namespace gn {

constexpr int PB = 32;  // panel width

struct CaqrArgs {
    int m, n, ldw;
    int panel;          // k
    int level;          // tree level l
    int F;              // blocks per group (= TR / 32)
    int nblocks;        // number of 32-row blocks at this level for this panel
    long long S;        // block stride in rows at this level (32 * F^level)
    long long tOff;     // index of the first T block of (panel, level) in Tbuf
    double* W;          long long sW;     // ldw x (n + 1) per problem
    double* Tbuf;       long long sT;     // T blocks, 32 x 32 each
    const ProbState* state;
    // update kernel only:
    int ext_cols;       // > 0: apply to the external matrix C (ldw x ext_cols) instead of the trailing columns
    double* C;          long long sC;
    int reverse;        // 1: apply reflectors in reverse order (Q instead of Q')
    int prob0;          // problem index offset (accessors address one problem of a batch)
    int skip_rhs;       // level-0 MFMA update: 1 = the carried right-hand side (the last trailing column) is left to k_caqr_update_rhs
    int npass;          // factor kernel: 1 = the column right after a panel narrower than 32 (the carried right-hand side
                        // d, when it is the ONLY trailing column) rides through the factorisation as a passenger: it receives
                        // every reflector but is never factored, and no trailing-update launch is needed for the panel
    // Row geometry of the level (see "Panel pairs" below): block q of the level starts at row
    //   mode 0 / 1:  base + q * S                         (level 0: S = 32, base = first row of tile 0)
    //   mode 2    :  base + ((q + 1) >> 1) * S + ((q + 1) & 1) * 32      (first tree level of a pair's second panel)
    int mode;
    long long base;
    int skip;           // level 0: the first `skip` 32-row units of every tile do not belong to this panel (second panel of a pair: 1)
    // trailing update only:
    int win;            // 0: every trailing column; 1: only the columns of the NEXT panel (narrow update inside a pair);
                        // 2: every trailing column BEYOND the next panel's columns (far update of a pair's first panel)
    int sub0, subn;     // a sub-range of the column window: skip sub0 columns, then at most subn (0: to the end).  The look-ahead
                        // sweep splits a pair's far update into the next pair's columns (first) and the rest (on a second stream)
    int pair;           // level-0 far update: 1 = the workgroup also applies the level-0 reflectors of panel + 1 (T blocks from
                        // tOff2) to its block of C while it sits in registers: one pass over the trailing matrix for two panels
    long long tOff2;
    int xmap;           // level-0 far update of a pair, one problem with many tiles: > 0 = number of tiles, the grid is read
                        // XCD-locally (gn_kernels_update_v4.hpp: k_caqr_update_v4_pair)
};

// ---------------------------------------------------------------------------------------------
// Panel pairs (two panels per pass over the trailing matrix).
// With b = 32 the trailing update moves 16 bytes per 8 flops: HBM bound.  Applying TWO panels' reflectors in one pass halves
// that traffic, but in the plain CAQR sweep panel k + 1's tile reflectors do not commute with panel k's tree reflectors
// (the tree mixes the first 32 rows of every tile, which panel k + 1's tiles would contain).  So the second panel of a
// pair (a, b) keeps the tiles of a and simply leaves out their first 32 rows:
//     level 0 of b : tile g = rows [anchor + TR g + 32, anchor + TR (g + 1)), reflector j's diagonal in tile slot 32 + j;
//     level 1 of b : per tile the R factor just produced (rows 32..63 of the tile, upper triangular) AND, for tiles g >= 1,
//                    the rows 0..31 left over from a (dense in b's columns once a's tree has been applied to them);
//                    block order: (tile 0, rows 32..63), then (tile g, rows 0..31), (tile g, rows 32..63) for g = 1, 2, ...
//                    = mode 2 above, dense blocks at odd q; higher levels combine the group leaders as usual.
// Now level 0 of b (rows 32.. of every tile) and the tree of a (rows 0..31 of every tile) touch disjoint rows and commute:
//     Q' = tree(b)' tree(a)' [L0(b)' L0(a)'],
// and the bracket is tile-local: one workgroup applies both to its block of C between one load and one store.
// ---------------------------------------------------------------------------------------------
// width of the panel after the one that starts at J2 column r0 (0: there is none)
__device__ __forceinline__ int caqr_next_bw(int kp, int r0) {
    const int left = kp - (r0 + PB);
    return left <= 0 ? 0 : (left < PB ? left : PB);
}
__device__ __forceinline__ long long caqr_block_row(const CaqrArgs& a, long long q) {
    if (a.mode == 2) return a.base + ((q + 1) >> 1) * a.S + ((q + 1) & 1) * 32;
    return a.base + q * a.S;
}

// ---------------------------------------------------------------------------------------------
// panel factorisation of one group (tile or tree node)
// ---------------------------------------------------------------------------------------------
// NW waves per workgroup; wave w owns the NC = 32 / NW panel columns w + NW cc.  NW = 8 halves the per-step
// dependency chain of a wave (4 dot products + one 4-way transposed reduction + 4 column updates) and doubles
// the waves per SIMD that hide it.
#ifndef ENLSIP_FACTOR_OCC8
#define ENLSIP_FACTOR_OCC8 4
#endif
#ifndef ENLSIP_FACTOR_OCC4
#define ENLSIP_FACTOR_OCC4 4      // one-tile problems of <= 256 rows (RPL 4, 4 waves): 128 registers, 4 workgroups per CU: C5 +4 % (2: 150 registers)
#endif
#ifndef ENLSIP_FACTOR_OCC16
#define ENLSIP_FACTOR_OCC16 8
#endif
// timing experiments only (tests/microbench/factor_bench.hip; results wrong when != 0): 1 no reflector arithmetic (norm reduction,
// dlarfg), 2 no reduction of the dot products, 3 no T factor, 4 no barrier inside the step loop, 5 no step loop, 6 no dot products /
// column updates
#ifndef ENLSIP_FACTOR_ABLATE
#define ENLSIP_FACTOR_ABLATE 0
#endif
constexpr int FACTOR_ABLATE = ENLSIP_FACTOR_ABLATE;
#ifdef ENLSIP_FACTOR_STAMPS     // harness only: phase stamps (100 MHz) of one workgroup
__device__ long long g_factor_stamps[16];
#define FACTOR_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 3 && blockIdx.y == 7 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_factor_stamps[i] = wall_clock64(); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FACTOR_STAMP(i) do { } while (0)
#endif
// The factorisation of one group with the tile in registers.  PRE = true: x already holds the tile (the fused small-problem
// kernel gn_kernels_small_fused.hpp hands it over without the round trip through HBM).
// All-reduce of the LAST `live` entries of dot (the columns of a wave that are still being updated), smallest transposed butterfly
// that holds them; the other entries of out are not written.
template <int NC>
__device__ __forceinline__ void wave_allsum_last(const double (&dot)[NC], double (&out)[NC], const int live) {
    if (NC > 4 && live > 4) {
        wave_allsumN(dot, out);
    } else if (NC >= 4 && live > 2) {
        double d4[4], o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d4[i] = dot[NC - 4 + i];
        wave_allsum4(d4, o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) out[NC - 4 + i] = o4[i];
    } else if (NC >= 2 && live > 1) {
        double d2[2], o2[2];
        d2[0] = dot[NC - 2]; d2[1] = dot[NC - 1];
        wave_allsumN(d2, o2);
        out[NC - 2] = o2[0]; out[NC - 1] = o2[1];
    } else {
        out[NC - 1] = wave_allsum(dot[NC - 1]);
    }
}

// GRAM = false: the group's T factor is never read beyond its diagonal (a last, narrow panel whose only trailing column rides
// through the factorisation as a passenger: no block update follows, and the reflector-by-reflector consumers read tau from
// diag(T)).  The Gram entries v_c' v_j — the dot products of a step with the FINISHED columns — are then not needed: a step
// forms dot products only for the columns it still updates (cc >= jj: columns w + NW cc with cc < jj are finished in every
// wave), reduces them with the smallest butterfly that holds them, and the T epilogue writes diag(T) = tau.
template <int RPL, int NW, bool PRE, bool GRAM = true>
__device__ __forceinline__ void caqr_factor_core(const CaqrArgs& a, const int prob, const ProbState& st, double (&x)[PB / NW][RPL],
                                                 double (*vsh)[64 * RPL], double* taush, double (*gsh)[PB + 1]) {   // 2nd bound = waves per SIMD
    constexpr int NC = PB / NW;
    constexpr int NT = 64 * NW;
    const int r0 = a.panel * PB;
    if (r0 >= st.kp) return;
    const int bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
    const int bwp = (a.npass && bw < PB) ? bw + 1 : bw;      // panel columns + passenger
    const int col0 = st.rankA + r0;
    const int g = blockIdx.x;
    const int ln = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    double* W = a.W + prob * a.sW;
    const bool tri = a.level > 0;

    FACTOR_STAMP(0);
    if (GRAM)
        for (int e = threadIdx.x; e < PB * (PB + 1); e += NT) (&gsh[0][0])[e] = 0.0;
    if (threadIdx.x < PB) taush[threadIdx.x] = 0.0;

    // slot geometry of this lane: slot ln + 64 i = block q = (ln >> 5) + 2 i of the group, row rb of that block.
    // Address = wave-uniform part (column, block pair i) + ONE per-lane offset: no per-slot address registers.
    const int rb = ln & 31;
    const int qh = ln >> 5;
    // row geometry (CaqrArgs): blocks 2 i and 2 i + 1 of the group sit S rows apart (modes 0, 1); in mode 2 the pair is
    // (rows 32..63 of one tile, rows 0..31 of the NEXT tile) and the odd block is dense
    const bool split = a.mode == 2;
    const bool dns = split && qh == 1;
    const long long lane_off = split ? (qh ? a.S : 32) + rb : (long long)qh * a.S + rb;
    const long long u0 = a.base + (split ? ((long long)g * a.F / 2) * a.S : ((long long)g * a.F) * a.S);
    const long long ustep = split ? a.S : 2 * a.S;
    const int dsh = 32 * a.skip;                      // slot of reflector j's diagonal = j + dsh (level 0 of a pair's second panel: 32)
    auto ubase = [&](int c, int i) -> size_t {        // uniform
        return (size_t)(col0 + c) * a.ldw + (size_t)(u0 + i * ustep);
    };
    auto bval = [&](int i) -> bool { return (long long)g * a.F + qh + 2 * i < a.nblocks && qh + 2 * i >= a.skip; };
    // load the tile: x[cc][i] = element (slot ln + 64 i, column w + NW cc)   (PRE: the caller has filled x already)
    if (!PRE) {
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int c = w + NW * cc;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                // tree levels hold upper triangles; the passenger has entries in every row a reflector touches
                const bool ok = (c < bwp) && bval(i) && (!tri || dns || (c < bw ? rb <= c : rb < bw));
                x[cc][i] = ok ? W[ubase(c, i) + lane_off] : 0.0;
            }
        }
    }
    __syncthreads();
    FACTOR_STAMP(1);

#pragma unroll
    for (int jj = 0; jj < NC; ++jj) {
        for (int jw = 0; jw < (FACTOR_ABLATE == 5 ? 0 : NW); ++jw) {
            int lnl = ln;                      // opaque per-iteration copy: keeps LICM from hoisting (and spilling) the row masks
            asm volatile("" : "+v"(lnl));
            const int j = NW * jj + jw;  // wave-uniform
            if (j < bw) {
                const int buf = j & 1;
                if (w == jw) {
                    // reflector of column j: pivot slot dj = j + dsh lives in lane dj, register 0
                    // dj = j + 32 skip <= 63: only register 0 (slots 0..63) can hold the diagonal or rows above it, the slots
                    // of registers i >= 1 are all below it — no row masks there (the owner's part is the step's critical path)
                    const int dj = j + dsh;
                    double xn2 = (lnl > dj) ? x[jj][0] * x[jj][0] : 0.0;
#pragma unroll
                    for (int i = 1; i < RPL; ++i) xn2 += x[jj][i] * x[jj][i];
                    if (FACTOR_ABLATE != 1) xn2 = wave_allsum(xn2);
                    const double alpha = (FACTOR_ABLATE == 1) ? x[jj][0] : wave_bcast(x[jj][0], dj);
                    const Reflector h = (FACTOR_ABLATE == 1) ? Reflector{0.5 * alpha, 1.5, 0.25 * xn2} : make_reflector(alpha, xn2);
                    {
                        const double v = (lnl > dj) ? x[jj][0] * h.scale : (lnl == dj ? 1.0 : 0.0);
                        vsh[buf][lnl] = v;
                        x[jj][0] = (lnl > dj) ? v : (lnl == dj ? h.beta : x[jj][0]);
                    }
#pragma unroll
                    for (int i = 1; i < RPL; ++i) {
                        const double v = x[jj][i] * h.scale;
                        vsh[buf][lnl + 64 * i] = v;
                        x[jj][i] = v;
                    }
                    if (lnl == 0) taush[j] = h.tau;
                }
                if (FACTOR_ABLATE != 4) __syncthreads();
                double v[RPL];
#pragma unroll
                for (int i = 0; i < RPL; ++i) v[i] = vsh[buf][lnl + 64 * i];
                const double tj = taush[j];
                const int cc_lo = GRAM ? 0 : jj;          // first column slot that can still be live (a constant once jj is unrolled)
                double dot[NC];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    dot[cc] = 0.0;
                    if (cc >= cc_lo) {
#pragma unroll
                        for (int i = 0; i < (FACTOR_ABLATE == 6 ? 1 : RPL); ++i) dot[cc] += x[cc][i] * v[i];
                    }
                }
                double ds[NC];
                if (FACTOR_ABLATE == 2) {
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) ds[cc] = dot[cc];
                } else if (GRAM) wave_allsumN(dot, ds);     // NC reductions sharing one transposed butterfly
                else wave_allsum_last<NC>(dot, ds, NC - jj);
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    if (cc < cc_lo) continue;
                    const int c = w + NW * cc;
                    if (c > j) {
                        const double wd = tj * ds[cc];
#pragma unroll
                        for (int i = 0; i < (FACTOR_ABLATE == 6 ? 1 : RPL); ++i) x[cc][i] -= wd * v[i];
                    } else if (GRAM && c < j) {
                        // Gram entry v_c' v_j for the T factor: rows above slot j are masked by
                        // v (zero there), so the R entries held in x[cc] do not contribute
                        if (lnl == 0) gsh[c][j] = ds[cc];
                    }
                }
            }
        }
    }
    __syncthreads();
    FACTOR_STAMP(2);
    // store the tile back (V below / R on and above the diagonal; tree levels: upper triangles).  The lane's slot geometry is formed
    // AGAIN here, from an opaque copy of the lane index: kept from the load phase, the 64-bit lane offset and the row masks stayed
    // live across the step loop — 27 of the 126 registers of the 8-wave form (now 99; the 4-wave forms 207 -> 171 and 124 -> 101).
    {
        int lns = ln;
        asm volatile("" : "+v"(lns));
        const int rbs = lns & 31, qhs = lns >> 5;
        const bool dnss = split && qhs == 1;
        const long long lane_off_s = split ? (qhs ? a.S : 32) + rbs : (long long)qhs * a.S + rbs;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int c = w + NW * cc;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const bool in = (long long)g * a.F + qhs + 2 * i < a.nblocks && qhs + 2 * i >= a.skip;
                const bool ok = (c < bwp) && in && (!tri || dnss || (c < bw ? rbs <= c : rbs < bw));
                if (ok) W[ubase(c, i) + lane_off_s] = x[cc][i];
            }
        }
    }
    FACTOR_STAMP(3);
    // T factor (dlarft forward/columnwise) by wave 0: lanes r and r + 32 build row r, each holding every second entry of it
    if (!GRAM) {
        if (w == 0) {                // diag(T) = tau, zeros elsewhere (lane r + 32 hh writes every second entry of row r)
            double* T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
            // from an opaque copy of the lane index: `ln & 31` is also what the fused small kernel's staging computes at its very
            // start, and as one common value it lived across the step loop in a spill slot, reloaded in every step
            int lnt = ln;
            asm volatile("" : "+v"(lnt));
            const int r = lnt & 31, hh = lnt >> 5;
            const double tr = taush[r];
#pragma unroll
            for (int j2 = 0; j2 < PB / 2; ++j2) {
                const int j = 2 * j2 + hh;
                T[r + j * PB] = (r == j) ? tr : 0.0;
            }
        }
    } else if (w == 0 && FACTOR_ABLATE != 3) {
        double* T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
        // Row r of T in registers, loops fully unrolled and branch-free: T is upper triangular, so trow[l] = 0 for l < r and the
        // plain sum over l < j is the dlarft sum over r <= l < j; gsh / taush are zero where no reflector exists.  (A lane-bounded
        // loop over LDS rows cost ~20 us.)  The row is split over the two half-waves — half hh holds trh[i] = T[r][2 i + hh] and
        // sums its own l, one cross-half add per column: with the whole row in one lane (64 registers) and the Gram column reads
        // hoisted above it the epilogue spilled, and a scratch reload waits (vmcnt(0)) for the acknowledgement of the tile's stores
        // issued just before — seven HBM write round trips inside a 7 us epilogue that the other seven waves of the workgroup sit
        // out.
        const int r = ln & 31, hh = ln >> 5;
        double trh[PB / 2];
#pragma unroll
        for (int i = 0; i < PB / 2; ++i) trh[i] = 0.0;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const double tj = taush[j];
            double s = 0.0;
#pragma unroll
            for (int i = 0; 2 * i < j; ++i) s += trh[i] * gsh[2 * i + hh][j];   // l = 2 i + hh <= j; the entry l = j is still zero
            s = xor32_sum(s);
            const double tv = (r == j) ? tj : ((r < j) ? -tj * s : 0.0);
            trh[j >> 1] = ((j & 1) == hh) ? tv : trh[j >> 1];
            if ((j & 1) == hh) T[r + j * PB] = tv;
        }
    }
    FACTOR_STAMP(4);
}

template <int RPL, int NW>
__global__ __launch_bounds__(64 * NW, NW == 16 ? ENLSIP_FACTOR_OCC16 : (NW == 8 ? ENLSIP_FACTOR_OCC8 : (RPL == 4 ? ENLSIP_FACTOR_OCC4 : 2))) void k_caqr_factor(CaqrArgs a) {   // 2nd bound = waves per SIMD
    __shared__ double vsh[2][64 * RPL];
    __shared__ double taush[PB];
    __shared__ double gsh[PB][PB + 1];
    const int prob = blockIdx.y + a.prob0;
    const ProbState st = a.state[prob];
    double x[PB / NW][RPL];
    // a passenger launch factors last panels narrower than 32 whose only trailing column rides along: no T beyond its diagonal
    const int left = st.kp - a.panel * PB;
    if (a.npass && left < PB) caqr_factor_core<RPL, NW, false, false>(a, prob, st, x, vsh, taush, gsh);
    else caqr_factor_core<RPL, NW, false, true>(a, prob, st, x, vsh, taush, gsh);
}

// ---------------------------------------------------------------------------------------------
// trailing update, reference form: the group's reflectors are applied one after the other
// (dorm2r), columns in registers.  Used for tree levels, for applying Q'/Q to vectors, and as the
// A/B partner of the MFMA kernel (gn_kernels_update_v4.hpp).
// ---------------------------------------------------------------------------------------------
template <int RPL>
__global__ __launch_bounds__(256) void k_caqr_update_refl(CaqrArgs a) {
    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0 = a.panel * PB;
    if (r0 >= st.kp) return;
    const int bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
    const int col0 = st.rankA + r0;
    const int g = blockIdx.x;
    const int ln = lane_id(), w = wave_id();
    const double* W = a.W + prob * a.sW;
    const double* T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
    const bool tri = a.level > 0;

    // target columns
    double* C;
    int ncols;
    if (a.ext_cols > 0) {
        C = a.C + prob * a.sC;
        ncols = a.ext_cols;
    } else {
        int first = r0 + bw;                   // J2-local index of the first trailing column
        ncols = st.n2 + 1 - first;             // includes the augmented column (local index n2)
        if (a.win) {                           // pair: only / all but the next panel's columns
            const int bwn = caqr_next_bw(st.kp, r0);
            if (a.win == 1) ncols = bwn;
            else { first += bwn; ncols -= bwn; }
        }
        first += a.sub0; ncols -= a.sub0;
        if (a.subn > 0 && ncols > a.subn) ncols = a.subn;
        C = a.W + prob * a.sW + (size_t)(st.rankA + first) * a.ldw;
    }
    const int cbase = blockIdx.y * 32;
    if (cbase >= ncols) return;

    const int rb = ln & 31;
    long long rowoff[RPL];
    bool bval[RPL], dns[RPL];
    int qq[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int q = (ln >> 5) + 2 * i;
        const long long bidx = (long long)g * a.F + q;
        qq[i] = q;
        bval[i] = bidx < a.nblocks && q >= a.skip;
        dns[i] = a.mode == 2 && ((bidx + 1) & 1) == 0;
        rowoff[i] = caqr_block_row(a, bidx) + rb;
    }
    const int dsh = 32 * a.skip;
    double x[8][RPL];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const int c = cbase + w + 4 * cc;
#pragma unroll
        for (int i = 0; i < RPL; ++i)
            x[cc][i] = (c < ncols && bval[i]) ? C[rowoff[i] + (size_t)c * a.ldw] : 0.0;
    }
    auto load_v = [&](int j, double* v) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int s = ln + 64 * i;
            double val = 0.0;
            if (bval[i]) {
                if (!tri) {
                    if (s > j + dsh) val = W[rowoff[i] + (size_t)(col0 + j) * a.ldw];
                    else if (s == j + dsh) val = 1.0;
                } else {
                    if (qq[i] == 0) val = (rb == j) ? 1.0 : 0.0;
                    else if (dns[i] || rb <= j) val = W[rowoff[i] + (size_t)(col0 + j) * a.ldw];
                }
            }
            v[i] = val;
        }
    };
    double v[RPL], vn[RPL];
    const int jfirst = a.reverse ? bw - 1 : 0;
    const int jstep = a.reverse ? -1 : 1;
    load_v(jfirst, v);
    for (int it = 0; it < bw; ++it) {
        const int j = jfirst + it * jstep;
        if (it + 1 < bw) load_v(j + jstep, vn);   // prefetch the next reflector
        const double tj = T[j + j * PB];
        if (tj != 0.0) {
            double dot[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                dot[cc] = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) dot[cc] += x[cc][i] * v[i];
            }
            double ds[8];
            wave_allsum8(dot, ds);
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const double wd = tj * ds[cc];
#pragma unroll
                for (int i = 0; i < RPL; ++i) x[cc][i] -= wd * v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = vn[i];
    }
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const int c = cbase + w + 4 * cc;
#pragma unroll
        for (int i = 0; i < RPL; ++i)
            if (c < ncols && bval[i]) C[rowoff[i] + (size_t)c * a.ldw] = x[cc][i];
    }
}

}  // namespace gn
