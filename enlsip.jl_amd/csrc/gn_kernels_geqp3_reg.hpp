// Column-pivoted Householder QR (dgeqp3 pivot rule, dlaqp2 arithmetic) of ONE small matrix per workgroup with the whole
// matrix in registers: rows <= 64 * RPL (lane = row), cols <= 64 (8 waves x 8 columns).  Used for F_A = qr(C.A',
// ColumnNorm()) (src/enlsip_functions.jl:700) when A' does not fit the LDS area of k_constraint (C2: 512 x 64 = 256 KB),
// where the generic workgroup routine works out of L2 and pays three dependent memory round trips per pivot step.
// Same step structure as k_sb_factor_reg (gn_kernels_qrcp_block_reg.hpp); columns never move, positions are logical, and
// the LAPACK layout (columns in pivot order, V below / R on and above the diagonal, tau, 1-based jpvt) is produced by the
// write-back.  The dot products with already retired columns are the Gram entries of the dlarft T factor of the (single)
// compact-WY block of Q1, which is written too (kA <= 64).
#pragma once
#include "gn_kernels_constraint.hpp"

namespace gn {

struct Geqp3RegArgs {
    int rows, cols;            // A is rows x cols (n x t)
    const double* A;  long long lda, strideA;
    double* F;        long long sF;      // rows x cols, ld = rows (compact factors)
    double* tau;      long long sTau;
    long long* jpvt;  long long sJ;
    double* T;        long long sT;      // 64 x 64 block T factor (column-major, zero padded); may be null
    int prob0;
};

template <int RPL>
__global__ __launch_bounds__(512) void k_geqp3_reg(Geqp3RegArgs a) {
    constexpr int NWV = 8, NCW = 8;
    __shared__ double vsh[2][64 * RPL];
    __shared__ double cvn1[2][64], cvn2[64], taul[64];
    __shared__ int cpos[2][64], colat[64], tslot[64];
    __shared__ double gram[64 * 65];
    __shared__ double tau_s[2], beta_s[2];
    const int prob = blockIdx.x + a.prob0;
    const int rows = a.rows, cols = a.cols;
    const int tid = threadIdx.x, ln = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const double* A = a.A + prob * a.strideA;
    double* F = a.F + prob * a.sF;
    const double tol3z = 1.4901161193847656e-08;
    const int kmax = rows < cols ? rows : cols;

    // columns -> registers: column c = w + 8 cc, row ln + 64 i
    double x[NCW][RPL];
    double nrm[NCW];
#pragma unroll
    for (int cc = 0; cc < NCW; ++cc) {
        const int c = w + NWV * cc;
        nrm[cc] = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            x[cc][i] = (c < cols && r < rows) ? A[r + (size_t)c * a.lda] : 0.0;
            nrm[cc] += x[cc][i] * x[cc][i];
        }
    }
    {
        double ns[NCW];
        wave_allsum8(nrm, ns);
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) {
            const int c = w + NWV * cc;
            if (ln == 0 && c < 64) {
                const double v0 = (c < cols) ? sqrt(ns[cc]) : 0.0;
                cvn1[0][c] = v0;
                cvn2[c] = v0;
                cpos[0][c] = (c < cols) ? c : -1000;      // -1000: no such column
                colat[c] = c;
            }
        }
    }
    for (int e = tid; e < 64 * 65; e += 512) gram[e] = 0.0;
    if (tid < 64) taul[tid] = 0.0;
    __syncthreads();

    int s = 0;
    for (; s < kmax; ++s) {
        int lnl = ln, wl = w;                              // opaque copies: keep LICM from hoisting the per-column masks
        asm volatile("" : "+v"(lnl));
        asm volatile("" : "+s"(wl));
        const int rd = s & 1, wr = rd ^ 1;
        // pivot: largest partial norm among the active columns, ties -> lowest current position (idamax)
        const int cp = cpos[rd][lnl];                      // both words unconditionally: one LDS round trip
        const double cv = cvn1[rd][lnl];
        const bool in = cp >= 0;
        const ArgMax am = wave_argmax(in ? pivot_key(cv) : -1.0, in ? cp : 0x7fffffff, in ? lnl : -1);
        // s < kmax <= cols: an active column exists, am.idx >= 0.  The clamp keeps every index below valid without a second
        // loop exit (a `break` here makes the compiler carry two copies of the column block through the loop, see
        // k_sb_factor_reg)
        const int ci = am.idx < 0 ? 0 : am.idx, q = am.pos;
        // the owner only READS its registers here (the retired column takes its reflector / beta in the column loop below);
        // s < 64: the rows of registers i >= 1 are all below the pivot row, no masks there
        if (wl == ci % NWV) {
#pragma unroll
            for (int cc = 0; cc < NCW; ++cc) {
                if (cc == ci / NWV) {
                    double xn2 = (lnl > s) ? x[cc][0] * x[cc][0] : 0.0;
#pragma unroll
                    for (int i = 1; i < RPL; ++i) xn2 += x[cc][i] * x[cc][i];
                    xn2 = wave_allsum(xn2);
                    const double alpha = wave_bcast(x[cc][0], s);      // s < 64: the pivot row sits in register 0
                    const Reflector h = make_reflector(alpha, xn2);
                    vsh[rd][lnl] = (lnl > s) ? x[cc][0] * h.scale : (lnl == s ? 1.0 : 0.0);
#pragma unroll
                    for (int i = 1; i < RPL; ++i) vsh[rd][lnl + 64 * i] = x[cc][i] * h.scale;
                    if (lnl == 0) {
                        taul[s] = h.tau;
                        tau_s[rd] = h.tau;
                        beta_s[rd] = h.beta;
                        tslot[s] = ci;
                    }
                }
            }
        }
        __syncthreads();
        double v[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = vsh[rd][lnl + 64 * i];
        // the other LDS words of the step in the same round trip: positions of the wave's columns, the owner's beta, the
        // norms of the downdate (lane u < 8 <-> column u of this wave)
        int pkv[NCW];
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) pkv[cc] = cpos[rd][wl + NWV * cc];
        const double beta_o = beta_s[rd];
        const int u = lnl & (NCW - 1);
        const int cu = wl + NWV * u;
        const double o1_in = cvn1[rd][cu], o2_in = cvn2[cu];
        const int pk_u = cpos[rd][cu];
        const double tj = uniform_f64(tau_s[rd]);
        double dot[NCW], ds[NCW];
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) {
            dot[cc] = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) dot[cc] += x[cc][i] * v[i];
        }
        wave_allsum8(dot, ds);
        // every column without a branch: factor 0 for the pivot itself, retired columns and slots without a column (as in
        // k_sb_factor_reg: 64 independent multiply-adds instead of eight compare / branch / update / broadcast sequences)
        double wdv[NCW];
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) {
            const int c = wl + NWV * cc;
            const bool live = (c != ci) && (pkv[cc] >= 0) && (tj != 0.0);
            wdv[cc] = live ? tj * ds[cc] : 0.0;
        }
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc)
#pragma unroll
            for (int i = 0; i < RPL; ++i) x[cc][i] -= wdv[cc] * v[i];
        if (wl == ci % NWV) {       // the retired column keeps its reflector below the diagonal, beta on it
#pragma unroll
            for (int cc = 0; cc < NCW; ++cc) {
                if (cc == ci / NWV) {
                    x[cc][0] = (lnl > s) ? v[0] : (lnl == s ? beta_o : x[cc][0]);
#pragma unroll
                    for (int i = 1; i < RPL; ++i) x[cc][i] = v[i];
                }
            }
        }
        // lane u < 8 <-> column u of this wave: its entry in the pivot row and its dot product (Gram entry of a retired column)
        double ajc = 0.0, dsu = 0.0;
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) {
            const double rv = wave_bcast(x[cc][0], s);
            ajc = (u == cc) ? rv : ajc;
            dsu = (u == cc) ? ds[cc] : dsu;
        }
        const bool mine_col = (lnl < NCW) && (cu != ci);
        if (mine_col && pk_u < 0 && pk_u > -1000) gram[(-1 - pk_u) * 65 + s] = dsu;      // v_a' v_s, a = step that retired this column
        {   // dlaqp2 norm downdate, lane u < 8 <-> column u of this wave
            const bool mine = mine_col && pk_u >= 0;
            double o1 = mine ? o1_in : 0.0;
            const double o2 = mine ? o2_in : 1.0;
            bool need = false;
            if (mine && o1 != 0.0) {
                const NormDown nd = norm_downdate(o1, o2, ajc, tol3z);
                if (nd.need) need = true;
                else o1 = nd.vn1;
            }
            unsigned nm = (unsigned)(__ballot(need) & 0xffull);
            while (nm) {
                const int uu = __ffs((int)nm) - 1;
                nm &= nm - 1;
                double sq = 0.0;
#pragma unroll
                for (int cc = 0; cc < NCW; ++cc)
                    if (cc == uu) {
#pragma unroll
                        for (int i = 0; i < RPL; ++i)
                            if (lnl + 64 * i > s) sq += x[cc][i] * x[cc][i];
                    }
                sq = wave_allsum(sq);
                if (lnl == uu) {
                    o1 = (s + 1 < rows) ? sqrt(sq) : 0.0;
                    cvn2[cu] = o1;
                }
            }
            if (mine) cvn1[wr][cu] = o1;
        }
        // positions: the pivot (at position q) trades places with the column that sat at position s
        if (wl == 0) {
            const int cj = colat[s];
            const int old = cpos[rd][lnl];
            cpos[wr][lnl] = (lnl == ci) ? (-1 - s) : ((lnl == cj && cj != ci) ? q : old);
            if (lnl == 0) {
                colat[s] = ci;
                colat[q] = cj;
            }
        }
        __syncthreads();
    }
    const int fin = s & 1;
    // ---- write-back in LAPACK layout: the column at final position p goes to F[:, p] -----------------------------
#pragma unroll
    for (int cc = 0; cc < NCW; ++cc) {
        const int c = w + NWV * cc;
        if (c >= cols) continue;
        const int pk = cpos[fin][c];
        const int p = (pk < 0) ? (-1 - pk) : pk;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            if (r < rows) F[r + (size_t)p * rows] = x[cc][i];
        }
        if (ln == 0) a.jpvt[prob * a.sJ + p] = c + 1;
    }
    if (tid < kmax) a.tau[prob * a.sTau + tid] = taul[tid];
    // ---- T factor of the compact-WY block (dlarft forward / columnwise), unrolled in registers -------------------
    if (a.T != nullptr && w == 0) {
        double* T = a.T + prob * a.sT;
        double trow[64];
#pragma unroll
        for (int b = 0; b < 64; ++b) {
            const double tb = (b < s) ? taul[b] : 0.0;
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < b; ++l) acc += trow[l] * gram[l * 65 + b];
            const double tv = (ln == b) ? tb : ((ln < b) ? -tb * acc : 0.0);
            trow[b] = tv;
            T[ln + b * KBLK] = tv;
        }
    }
}

}  // namespace gn
