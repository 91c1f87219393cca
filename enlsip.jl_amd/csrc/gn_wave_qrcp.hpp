// Column-pivoted Householder QR (LAPACK dgeqp3 / dlaqp2 semantics) of a SMALL matrix by ONE WAVE:
// lane c owns COLUMN c, its rows live in registers (at most 64 rows, at most 64 columns including carried right-hand
// sides).  A pivot step is lane-local except for the arg-max over the partial norms (one DPP butterfly), the broadcast
// of the Householder vector (the pivot lane writes it to LDS, every lane reads it back by broadcast) and three scalars
// (v_readlane): column norms, dot products, updates and norm downdates need no reduction, where the workgroup form
// (wg_geqp2, lanes along rows) pays a wave reduction per column per step and four barriers per step.  Columns are never
// swapped: every lane tracks its LAPACK position, which also reproduces jpvt of the columns that never become pivots.
//
// Register rows are addressed statically: the step loop runs in blocks of eight unrolled sub-steps, after which the
// rows shift up by eight.  Finished rows (R entries) are parked in a wave-private LDS image tmp[row * 65 + lane]; the
// Householder vectors go straight to the factor matrix in memory (LAPACK layout, by position).
//
// GRAM: also build the dlarft T factor (forward, columnwise; 64-strided LDS image Tl) of the reflectors: the lanes of
// already processed columns still hold their own v below the diagonal, so the Gram entry v_l' v_j is the same dot
// product the unprocessed lanes compute for the update.
#pragma once
#include "gn_device_utils.hpp"

namespace gn {

struct WaveQrcp {
    int rows, k;          // matrix rows, number of steps (min(rows, pivot candidates))
    int ncand, npart;     // lanes [0, ncand) are pivot candidates; lanes [ncand, npart) carried right-hand sides
    double* tmp;          // LDS: k x 65 finished rows
    double* vbuf;         // LDS: 64 (current Householder vector, block-relative rows)
    double* dg;           // LDS: 64 (diagonal of R)
    double* F;            // factor matrix (LAPACK layout), leading dimension ldf: receives v below the diagonal
    long long ldf;
    double* tau;          // k
    double* Tl;           // GRAM: LDS 64 x k image of T (ld 64);  gcol: LDS 64
    double* gcol;
};

// One pivot step; S = block-relative row of the pivot (the rows shift up by WQ_UNROLL after every block of steps).
// The pivot lane publishes its RAW column below the diagonal; every lane forms the raw dot product with it — which, in
// the pivot lane itself, is the squared norm dlarfg needs — and the scale 1 / (alpha - beta) of the Householder vector is
// folded into the scalars: v = scale * raw, so  v' x = x[S] + scale * (raw' x)  and  x -= (tau v' x) scale * raw.
// Registers therefore keep raw columns; a processed lane remembers its own scale for the Gram entries.
constexpr int WQ_UNROLL = 4;

template <int NR, int S, bool GRAM>
__device__ __forceinline__ void wave_qrcp_substep(double (&x)[NR], const WaveQrcp& q, const int j, const int rem, const int ln,
                                                  int& mypos, double& vn1, double& vn2, double& myscale) {
    const double tol3z = 1.4901161193847656e-08;  // sqrt(eps), dlaqp2
    // (a) pivot = first position of the largest partial norm among the unprocessed columns
    const bool cand = (ln < q.ncand) && (mypos >= j);
    const ArgMax am = wave_argmax(cand ? pivot_key(vn1) : -1.0, mypos, ln);
    const int pl = am.idx, pp = am.pos;
    // (b) LAPACK's swap of positions j <-> pp
    if (mypos == j) mypos = pp;
    if (ln == pl) mypos = j;
    // (c) the pivot lane publishes its column below the diagonal (rows beyond the live range are zero)
    if (ln == pl) {
#pragma unroll
        for (int r = S + 1; r < NR; ++r) q.vbuf[r] = x[r];
    }
    wave_mem_sync();
    double v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = (r > S) ? q.vbuf[r] : 0.0;
    // (d) raw dot products, one short chain per 8 rows
    double dotraw = 0.0;
#pragma unroll
    for (int ch = 0; ch < NR / 8; ++ch) {
        if (8 * ch + 7 > S) {
            double pc = 0.0;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = 8 * ch + rr;
                if (r > S) pc += x[r] * v[r];
            }
            dotraw += pc;
        }
    }
    const Reflector h = make_reflector(wave_bcast(x[S], pl), wave_bcast(dotraw, pl));    // uniform
    const double tau = h.tau, scale = h.scale;
    const bool upd = (mypos > j) && (ln < q.npart);
    const double dot = x[S] + scale * dotraw;          // v' x
    if (GRAM && (mypos < j) && (ln < q.npart)) q.gcol[mypos] = myscale * dot;   // G[l][j] = v_l' v_j, l = this lane's position
    if (ln == pl) {
        x[S] = h.beta;
        myscale = scale;
    }
    if (upd && tau != 0.0) {
        const double wd = tau * dot;
        const double ws = wd * scale;
        x[S] -= wd;
#pragma unroll
        for (int r = S + 1; r < NR; ++r) x[r] -= ws * v[r];
    }
    if constexpr (GRAM) {
        wave_mem_sync();      // gcol is complete
        // T[0:j, j] = -tau_j T[0:j, 0:j] G[0:j, j]; T[j][j] = tau_j   (lane l builds row l)
        if (ln < j) {
            double sacc = 0.0;
            for (int i = ln; i < j; ++i) sacc += q.Tl[ln + 64 * i] * q.gcol[i];
            q.Tl[ln + 64 * j] = -tau * sacc;
        }
        if (ln == j) q.Tl[j + 64 * j] = tau;
    }
    // (e) row j is final in every column at a position >= j; the Householder vector goes to column j of F
    if (mypos >= j && ln < q.npart) q.tmp[j * 65 + ln] = x[S];
    if (ln == 0) {
        q.tau[j] = tau;
        q.dg[j] = h.beta;
    }
    if (j + 1 + ln < q.rows) q.F[(size_t)j * q.ldf + j + 1 + ln] = q.vbuf[S + 1 + ln] * scale;   // row j+1+ln <- block row S+1+ln
    wave_mem_sync();          // every reader of vbuf is done before the next publish
    // (f) dlaqp2 norm downdate, lane-local
    if (upd && ln < q.ncand && vn1 != 0.0) {
        const double ajc = x[S];
        const NormDown nd = norm_downdate(vn1, vn2, ajc, tol3z);
        if (nd.need) {
            double s2 = 0.0;
#pragma unroll
            for (int r = S + 1; r < NR; ++r) s2 += x[r] * x[r];
            const double nv = (j + 1 < q.rows) ? sqrt(s2) : 0.0;
            vn1 = nv;
            vn2 = nv;
        } else {
            vn1 = nd.vn1;
        }
    }
}

// The step loop on NRX register rows from row offset j0 on: blocks of WQ_UNROLL steps, the rows shifted up after each, until
// no step is left or — rem_stop > 0 — the live rows (q.rows - j0) fit rem_stop registers.  Returns true when steps remain
// (the rows have been shifted; x[r] is row j0 + r).
template <int NRX, bool GRAM>
__device__ __forceinline__ bool wave_qrcp_blocks(double (&x)[NRX], const WaveQrcp& q, const int ln, int& mypos, double& vn1, double& vn2,
                                                 double& myscale, int& j0, const int rem_stop) {
    // single exit at the bottom (the shift of the last block is skipped, not jumped over): with a `break` in the middle the
    // compiler carries the exit copy and the next-iteration copy of the register rows through the loop (two register sets and
    // a full copy per block; see k_sb_factor_reg)
    bool more, go;
    do {
        const int rem = q.rows - j0;
#define GN_WQ_STEP(S) \
    if (j0 + S < q.k) wave_qrcp_substep<NRX, S, GRAM>(x, q, j0 + S, rem, ln, mypos, vn1, vn2, myscale);
        GN_WQ_STEP(0) GN_WQ_STEP(1) GN_WQ_STEP(2) GN_WQ_STEP(3)
#undef GN_WQ_STEP
        more = j0 + WQ_UNROLL < q.k;
        if (more) {
#pragma unroll
            for (int r = 0; r < NRX; ++r) x[r] = (r + WQ_UNROLL < NRX) ? x[(r + WQ_UNROLL < NRX) ? r + WQ_UNROLL : r] : 0.0;
            j0 += WQ_UNROLL;
        }
        go = more && (q.rows - j0 > rem_stop);
    } while (go);
    return more;
}
// register rows of the next, smaller instance of the step loop (0: none)
constexpr int wq_next_rows(int nr) { return nr > 32 ? nr - 16 : (nr > 8 ? nr - 8 : 0); }
// ... and the rest of the factorisation on FEWER register rows once the live rows fit them (a step costs its instructions —
// one publish, one read-back, two FMAs per register row whether the row is live or not — and the later steps of a
// factorisation have few live rows).  Rows beyond the live range are zero, so the arithmetic is the same to the last bit.
template <int NRX, bool GRAM>
__device__ __forceinline__ void wave_qrcp_run(double (&x)[NRX], const WaveQrcp& q, const int ln, int& mypos, double& vn1, double& vn2,
                                              double& myscale, int& j0) {
    constexpr int NXT = wq_next_rows(NRX);
    // (a matrix with few rows — F_L11 is t x rankA — starts on the smallest instance that holds it, not with a block on this one)
    bool more = j0 < q.k;
    if (NXT == 0 || q.rows - j0 > NXT) more = wave_qrcp_blocks<NRX, GRAM>(x, q, ln, mypos, vn1, vn2, myscale, j0, NXT);
    if constexpr (NXT > 0) {
        if (more) {
            double y[NXT];
#pragma unroll
            for (int r = 0; r < NXT; ++r) y[r] = x[r];
            wave_qrcp_run<NXT, GRAM>(y, q, ln, mypos, vn1, vn2, myscale, j0);
#pragma unroll
            for (int r = 0; r < NRX; ++r) x[r] = (r < NXT) ? y[(r < NXT) ? r : 0] : 0.0;
        }
    }
}

// Factorisation driver.  x: the lane's column (rows 0..NR-1, zero beyond q.rows and in idle lanes).  On return mypos is
// the LAPACK position of the lane's column and the register rows are relative to the returned row offset j0 (the last
// block is not shifted out, so rows >= k of a carried right-hand side are still in x[row - j0]).
template <int NR, bool GRAM>
__device__ __forceinline__ int wave_qrcp(double (&x)[NR], const WaveQrcp& q, const int ln, int& mypos) {
    double vn1;
    {
        double s = 0.0;
#pragma unroll
        for (int ch = 0; ch < NR / 8; ++ch) {
            double pc = 0.0;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) pc += x[8 * ch + rr] * x[8 * ch + rr];
            s += pc;
        }
        vn1 = sqrt(s);
    }
    double vn2 = vn1, myscale = 0.0;
    mypos = ln;
    int j0 = 0;
    wave_qrcp_run<NR, GRAM>(x, q, ln, mypos, vn1, vn2, myscale, j0);
    return j0;
}

// Upper parts of the factor columns from the finished-row image: column at position P, rows <= min(P, k - 1).
// lp: lane i holds the lane that sits at position i.  ncols: positions written.
__device__ __forceinline__ void wave_qrcp_store_upper(const WaveQrcp& q, const int ln, const int lp, const int ncols,
                                                      double* F, long long ldf) {
    wave_mem_sync();
    for (int P = 0; P < ncols; ++P) {
        const int src = __builtin_amdgcn_readlane(lp, P);
        if (ln < q.k && ln <= P) F[ln + (size_t)P * ldf] = q.tmp[ln * 65 + src];
    }
}

// pseudo_rank (src/enlsip_functions.jl:17-31) of the diagonal in LDS (len <= 64), evaluated by the whole wave.
__device__ __forceinline__ int wave_pseudo_rank(const double* dg, const int len, const double eps_rank, const int ln, const double eps_abs) {
    if (len <= 0) return 0;
    const double d0 = fabs(dg[0]);
    if (d0 < eps_abs) return 0;
    const double tol = d0 * sqrt((double)len) * eps_rank;
    const bool fail = (ln < len) && !(fabs(dg[ln < len ? ln : 0]) > tol);
    const unsigned long long mk = __ballot(fail);
    return mk ? (int)__builtin_ctzll(mk) : len;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// TWO problems per wave: lanes 0..31 carry the columns of one problem, lanes 32..63 those of another (at most 32 columns
// including carried right-hand sides, at most NR <= 32 rows each).  With n2 + 1 = 29 columns (C5) a one-problem wave leaves
// 35 of its 64 lanes idle; the pivot stage is latency bound, so halving the number of waves nearly halves its time.
// Everything wave-uniform in the one-problem form (pivot lane, reflector scalars, row counts) becomes uniform per HALF.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ ArgMax half_argmax_exact(double val, int pos, int idx, int ln);
// as wave_argmax: a plain max tournament per half and a ballot when each half has ONE largest value, the exact tournament otherwise
__device__ __forceinline__ ArgMax half_argmax(double val, int pos, int idx, int ln) {
    double x = val;
    x = fmax(x, dpp_f64<0xB1>(x));
    x = fmax(x, dpp_f64<0x4E>(x));
    x = fmax(x, dpp_f64<0x141>(x));
    x = fmax(x, dpp_f64<0x140>(x));
    const double mA = fmax(readlane_f64(x, 0), readlane_f64(x, 16)), mB = fmax(readlane_f64(x, 32), readlane_f64(x, 48));
    const bool up = (ln & 32) != 0;
    const double m = up ? mB : mA;
    const unsigned long long mk = __ballot(val == m);
    const unsigned lo = (unsigned)mk, hi = (unsigned)(mk >> 32);
    if (__popc(lo) == 1 && __popc(hi) == 1) {
        const int lA = __ffs((int)lo) - 1, lB = 32 + __ffs((int)hi) - 1;
        ArgMax r;
        r.val = m;
        const int pA = __builtin_amdgcn_readlane(pos, lA), pB = __builtin_amdgcn_readlane(pos, lB);
        const int iA = __builtin_amdgcn_readlane(idx, lA), iB = __builtin_amdgcn_readlane(idx, lB);
        r.pos = up ? pB : pA;
        r.idx = up ? iB : iA;
        return r;
    }
    return half_argmax_exact(val, pos, idx, ln);
}
__device__ __forceinline__ ArgMax half_argmax_exact(double val, int pos, int idx, int ln) {
    ArgMax a = {val, pos, idx};
    am_step<0xB1>(a);
    am_step<0x4E>(a);
    am_step<0x141>(a);
    am_step<0x140>(a);
    // field by field, in named scalars: a per-lane choice between two STRUCTS is compiled as an indexed private array, i.e.
    // scratch stores + a scratch load per step, and a scratch load's wait (vmcnt(0)) also waits for the acknowledgement of the
    // step's global stores (tau, the Householder vector): ~1 us of HBM write latency on the critical path of every pivot step
    double rv[2];
    int rp[2], ri[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        rv[h] = readlane_f64(a.val, 32 * h);
        rp[h] = __builtin_amdgcn_readlane(a.pos, 32 * h);
        ri[h] = __builtin_amdgcn_readlane(a.idx, 32 * h);
        const double ov = readlane_f64(a.val, 32 * h + 16);
        const int op = __builtin_amdgcn_readlane(a.pos, 32 * h + 16);
        const int oi = __builtin_amdgcn_readlane(a.idx, 32 * h + 16);
        if (am_better(ov, op, rv[h], rp[h])) { rv[h] = ov; rp[h] = op; ri[h] = oi; }
    }
    const bool up = (ln & 32) != 0;
    ArgMax r;
    r.val = up ? rv[1] : rv[0];
    r.pos = up ? rp[1] : rp[0];
    r.idx = up ? ri[1] : ri[0];
    return r;
}
// x of lane srcA (lower half) / srcB (upper half), both wave-uniform, in every lane of the respective half
__device__ __forceinline__ double half_bcast(double x, int srcA, int srcB, int ln) {
    const double a = readlane_f64(x, srcA), b = readlane_f64(x, srcB);
    return (ln & 32) ? b : a;
}

struct WaveQrcp2 {        // per-lane values are those of the lane's half
    int rows, k;          // rows / steps of the lane's problem
    int ncand, npart;     // columns (pivot candidates) / columns + carried right-hand sides, counted inside the half
    double* tmp;          // LDS: finished rows of both problems, tmp[row * 65 + lane]
    double* vbuf;         // LDS: 2 x 64
    double* dg;           // LDS: 2 x 64
    double* F;            // the lane's problem: factor matrix (receives v below the diagonal), leading dimension ldf
    long long ldf;
    double* tau;          // the lane's problem
};

template <int NR, int S>
__device__ __forceinline__ void wave_qrcp2_substep(double (&x)[NR], const WaveQrcp2& q, const int j, const int ln, int& mypos,
                                                   double& vn1, double& vn2) {
    const double tol3z = 1.4901161193847656e-08;  // sqrt(eps), dlaqp2
    const int lh = ln & 31, hb = ln & 32;
    const bool act = j < q.k;
    const bool cand = act && (lh < q.ncand) && (mypos >= j);
    const ArgMax am = half_argmax(cand ? pivot_key(vn1) : -1.0, mypos, ln, ln);
    const int pl = am.idx, pp = am.pos;
    const int plA = __builtin_amdgcn_readlane(pl, 0), plB = __builtin_amdgcn_readlane(pl, 32);
    if (act) {
        if (mypos == j) mypos = pp;
        if (ln == pl) mypos = j;
    }
    double* vb = q.vbuf + 2 * hb;                  // [half][64]
    if (act && ln == pl) {
#pragma unroll
        for (int r = S + 1; r < NR; ++r) vb[r] = x[r];
    }
    wave_mem_sync();
    double v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = (r > S) ? vb[r] : 0.0;
    double dotraw = 0.0;
#pragma unroll
    for (int ch = 0; ch < NR / 8; ++ch) {
        if (8 * ch + 7 > S) {
            double pc = 0.0;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = 8 * ch + rr;
                if (r > S) pc += x[r] * v[r];
            }
            dotraw += pc;
        }
    }
    const Reflector h = make_reflector(half_bcast(x[S], plA, plB, ln), half_bcast(dotraw, plA, plB, ln));    // uniform per half
    const double tau = h.tau, scale = h.scale;
    const bool upd = act && (mypos > j) && (lh < q.npart);
    const double dot = x[S] + scale * dotraw;          // v' x
    if (act && ln == pl) x[S] = h.beta;
    if (upd && tau != 0.0) {
        const double wd = tau * dot;
        const double ws = wd * scale;
        x[S] -= wd;
#pragma unroll
        for (int r = S + 1; r < NR; ++r) x[r] -= ws * v[r];
    }
    // row j is final in every column at a position >= j; the Householder vector goes to column j of the half's F
    if (act && mypos >= j && lh < q.npart) q.tmp[j * 65 + ln] = x[S];
    if (act && lh == 0) {
        q.tau[j] = tau;
        q.dg[2 * hb + j] = h.beta;
    }
    if (act && j + 1 + lh < q.rows && S + 1 + lh < NR) q.F[(size_t)j * q.ldf + j + 1 + lh] = vb[S + 1 + lh] * scale;
    wave_mem_sync();          // every reader of vb is done before the next publish
    if (upd && lh < q.ncand && vn1 != 0.0) {
        const double ajc = x[S];
        const NormDown nd = norm_downdate(vn1, vn2, ajc, tol3z);
        if (nd.need) {
            double s2 = 0.0;
#pragma unroll
            for (int r = S + 1; r < NR; ++r) s2 += x[r] * x[r];
            const double nv = (j + 1 < q.rows) ? sqrt(s2) : 0.0;
            vn1 = nv;
            vn2 = nv;
        } else {
            vn1 = nd.vn1;
        }
    }
}

// Step loop of the two-problem form on NRX register rows (see wave_qrcp_blocks); rows_max = the larger row count of the two problems.
template <int NRX>
__device__ __forceinline__ bool wave_qrcp2_blocks(double (&x)[NRX], const WaveQrcp2& q, const int kmax, const int rows_max, const int ln,
                                                  int& mypos, double& vn1, double& vn2, int& j0, const int rem_stop) {
    bool more, go;                // single exit at the bottom, as in wave_qrcp_blocks
    do {
#define GN_WQ2_STEP(S) \
    if (j0 + S < kmax) wave_qrcp2_substep<NRX, S>(x, q, j0 + S, ln, mypos, vn1, vn2);
        GN_WQ2_STEP(0) GN_WQ2_STEP(1) GN_WQ2_STEP(2) GN_WQ2_STEP(3)
#undef GN_WQ2_STEP
        more = j0 + WQ_UNROLL < kmax;
        if (more) {
#pragma unroll
            for (int r = 0; r < NRX; ++r) x[r] = (r + WQ_UNROLL < NRX) ? x[(r + WQ_UNROLL < NRX) ? r + WQ_UNROLL : r] : 0.0;
            j0 += WQ_UNROLL;
        }
        go = more && (rows_max - j0 > rem_stop);
    } while (go);
    return more;
}
template <int NRX>
__device__ __forceinline__ void wave_qrcp2_run(double (&x)[NRX], const WaveQrcp2& q, const int kmax, const int rows_max, const int ln,
                                               int& mypos, double& vn1, double& vn2, int& j0) {
    constexpr int NXT = wq_next_rows(NRX);
    bool more = j0 < kmax;
    if (NXT == 0 || rows_max - j0 > NXT) more = wave_qrcp2_blocks<NRX>(x, q, kmax, rows_max, ln, mypos, vn1, vn2, j0, NXT);
    if constexpr (NXT > 0) {
        if (more) {
            double y[NXT];
#pragma unroll
            for (int r = 0; r < NXT; ++r) y[r] = x[r];
            wave_qrcp2_run<NXT>(y, q, kmax, rows_max, ln, mypos, vn1, vn2, j0);
#pragma unroll
            for (int r = 0; r < NRX; ++r) x[r] = (r < NXT) ? y[(r < NXT) ? r : 0] : 0.0;
        }
    }
}

// Driver of the two-problem form; kmax = the larger step count of the two problems (wave-uniform).  Returns the row offset of the
// registers (as wave_qrcp).
template <int NR>
__device__ __forceinline__ int wave_qrcp2(double (&x)[NR], const WaveQrcp2& q, const int kmax, const int ln, int& mypos) {
    double vn1;
    {
        double s = 0.0;
#pragma unroll
        for (int ch = 0; ch < NR / 8; ++ch) {
            double pc = 0.0;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) pc += x[8 * ch + rr] * x[8 * ch + rr];
            s += pc;
        }
        vn1 = sqrt(s);
    }
    double vn2 = vn1;
    mypos = ln & 31;
    int j0 = 0;
    const int ra = __builtin_amdgcn_readlane(q.rows, 0), rb = __builtin_amdgcn_readlane(q.rows, 32);
    wave_qrcp2_run<NR>(x, q, kmax, ra > rb ? ra : rb, ln, mypos, vn1, vn2, j0);
    return j0;
}

}  // namespace gn
