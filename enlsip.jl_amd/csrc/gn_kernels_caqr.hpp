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
};

// ---------------------------------------------------------------------------------------------
// panel factorisation of one group (tile or tree node)
// ---------------------------------------------------------------------------------------------
// NW waves per workgroup; wave w owns the NC = 32 / NW panel columns w + NW cc.  NW = 8 halves the per-step
// dependency chain of a wave (4 dot products + one 4-way transposed reduction + 4 column updates) and doubles
// the waves per SIMD that hide it.
template <int RPL, int NW>
#ifndef ENLSIP_FACTOR_OCC8
#define ENLSIP_FACTOR_OCC8 4
#endif
#ifndef ENLSIP_FACTOR_OCC4
#define ENLSIP_FACTOR_OCC4 4      // one-tile problems of <= 256 rows (RPL 4, 4 waves): 128 registers, 4 workgroups per CU: C5 +4 % (2: 150 registers)
#endif
#ifndef ENLSIP_FACTOR_OCC16
#define ENLSIP_FACTOR_OCC16 8
#endif
__global__ __launch_bounds__(64 * NW, NW == 16 ? ENLSIP_FACTOR_OCC16 : (NW == 8 ? ENLSIP_FACTOR_OCC8 : (RPL == 4 ? ENLSIP_FACTOR_OCC4 : 2))) void k_caqr_factor(CaqrArgs a) {   // 2nd bound = waves per SIMD
    constexpr int NC = PB / NW;
    constexpr int NT = 64 * NW;
    __shared__ double vsh[2][64 * RPL];
    __shared__ double taush[PB];
    __shared__ double gsh[PB][PB + 1];

    const int prob = blockIdx.y + a.prob0;
    const ProbState st = a.state[prob];
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

    for (int e = threadIdx.x; e < PB * (PB + 1); e += NT) (&gsh[0][0])[e] = 0.0;
    if (threadIdx.x < PB) taush[threadIdx.x] = 0.0;

    // slot geometry of this lane: slot ln + 64 i = block q = (ln >> 5) + 2 i of the group, row rb of that block.
    // Address = wave-uniform part (column, block pair i) + ONE per-lane offset: no per-slot address registers.
    const int rb = ln & 31;
    const int qh = ln >> 5;
    const long long lane_off = (long long)qh * a.S + rb;
    auto ubase = [&](int c, int i) -> size_t {        // uniform
        return (size_t)(col0 + c) * a.ldw + (size_t)(r0 + ((long long)g * a.F + 2 * i) * a.S);
    };
    auto bval = [&](int i) -> bool { return (long long)g * a.F + qh + 2 * i < a.nblocks; };
    // load the tile: a[cc][i] = element (slot ln + 64 i, column w + NW cc)
    double x[NC][RPL];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const int c = w + NW * cc;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            // tree levels hold upper triangles; the passenger has entries in every row a reflector touches
            const bool ok = (c < bwp) && bval(i) && (!tri || (c < bw ? rb <= c : rb < bw));
            x[cc][i] = ok ? W[ubase(c, i) + lane_off] : 0.0;
        }
    }
    __syncthreads();

#pragma unroll
    for (int jj = 0; jj < NC; ++jj) {
        for (int jw = 0; jw < NW; ++jw) {
            int lnl = ln;                      // opaque per-iteration copy: keeps LICM from hoisting (and spilling) the row masks
            asm volatile("" : "+v"(lnl));
            const int j = NW * jj + jw;  // wave-uniform
            if (j < bw) {
                const int buf = j & 1;
                if (w == jw) {
                    // reflector of column j: pivot slot j lives in lane j, register 0
                    double xn2 = 0.0;
#pragma unroll
                    for (int i = 0; i < RPL; ++i)
                        if (lnl + 64 * i > j) xn2 += x[jj][i] * x[jj][i];
                    xn2 = wave_allsum(xn2);
                    const double alpha = wave_bcast(x[jj][0], j);
                    const Reflector h = make_reflector(alpha, xn2);
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        const int s = lnl + 64 * i;
                        const double v = (s > j) ? x[jj][i] * h.scale : (s == j ? 1.0 : 0.0);
                        vsh[buf][s] = v;
                        if (s > j) x[jj][i] = v;
                        if (s == j) x[jj][i] = h.beta;
                    }
                    if (lnl == 0) taush[j] = h.tau;
                }
                __syncthreads();
                double v[RPL];
#pragma unroll
                for (int i = 0; i < RPL; ++i) v[i] = vsh[buf][lnl + 64 * i];
                const double tj = taush[j];
                double dot[NC];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    dot[cc] = 0.0;
#pragma unroll
                    for (int i = 0; i < RPL; ++i) dot[cc] += x[cc][i] * v[i];
                }
                double ds[NC];
                wave_allsumN(dot, ds);     // NC reductions sharing one transposed butterfly
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    const int c = w + NW * cc;
                    if (c > j) {
                        const double wd = tj * ds[cc];
#pragma unroll
                        for (int i = 0; i < RPL; ++i) x[cc][i] -= wd * v[i];
                    } else if (c < j) {
                        // Gram entry v_c' v_j for the T factor: rows above slot j are masked by
                        // v (zero there), so the R entries held in x[cc] do not contribute
                        if (lnl == 0) gsh[c][j] = ds[cc];
                    }
                }
            }
        }
    }
    __syncthreads();
    // store the tile back (V below / R on and above the diagonal; tree levels: upper triangles)
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const int c = w + NW * cc;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const bool ok = (c < bwp) && bval(i) && (!tri || (c < bw ? rb <= c : rb < bw));
            if (ok) W[ubase(c, i) + lane_off] = x[cc][i];
        }
    }
    // T factor (dlarft forward/columnwise): lane r of wave 0 builds row r
    if (w == 0) {
        double* T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
        if (ln < PB) {
            // row ln of T in registers, loops fully unrolled and branch-free: T is upper triangular, so trow[l] = 0
            // for l < ln and the plain sum over l < j is the dlarft sum over ln <= l < j; gsh / taush are zero where no
            // reflector exists.  (The tile registers are dead here.  A lane-bounded loop over LDS rows cost ~20 us.)
            double trow[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const double tj = taush[j];
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < j; ++l) s += trow[l] * gsh[l][j];
                const double tv = (ln == j) ? tj : ((ln < j) ? -tj * s : 0.0);
                trow[j] = tv;
                T[ln + j * PB] = tv;
            }
        }
    }
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
        const int first = r0 + bw;             // J2-local index of the first trailing column
        ncols = st.n2 + 1 - first;             // includes the augmented column (local index n2)
        C = a.W + prob * a.sW + (size_t)(st.rankA + first) * a.ldw;
    }
    const int cbase = blockIdx.y * 32;
    if (cbase >= ncols) return;

    const int rb = ln & 31;
    long long rowoff[RPL];
    bool bval[RPL];
    int qq[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int q = (ln >> 5) + 2 * i;
        const long long bidx = (long long)g * a.F + q;
        qq[i] = q;
        bval[i] = bidx < a.nblocks;
        rowoff[i] = r0 + bidx * a.S + rb;
    }
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
                    if (s > j) val = W[rowoff[i] + (size_t)(col0 + j) * a.ldw];
                    else if (s == j) val = 1.0;
                } else {
                    if (qq[i] == 0) val = (rb == j) ? 1.0 : 0.0;
                    else if (rb <= j) val = W[rowoff[i] + (size_t)(col0 + j) * a.ldw];
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
