// EXPERIMENT (round 2), NOT part of the library.  A literal dgeqp3 of [J2 | d] for small tall problems (C5: 256 x 32, C3: 512 x 64)
// in the registers of one workgroup, lanes along COLUMNS, rows split over the waves, solves included — meant to replace
// k_caqr_factor + k_pivot_small (two chains of n2 steps) by one chain.  It was integrated, passed the whole parity suite (82 GPU
// tests incl. accessors on the plain-QRPivoted storage) and was then measured SLOWER than the two-kernel path:
//   C5 8192 problems: 1.50-1.62 ms for this kernel against 0.86 ms for panel + pivot (4.3-4.5 M vs 6.9 M solves/s);
//   C3 1024 problems: 1.35-1.67 ms against 0.45 ms.
// Why (ISA + phase stamps, profiles/r2_notes.md): a pivot step costs ~450-500 instructions PER WAVE (64 + 64 f64 FMAs, as many
// LDS reads of the published pivot column, 32 LDS writes to publish it, and ~250 instructions of arg-max / dlarfg / norm
// downdate that every wave repeats redundantly), i.e. ~900 wave-instructions per problem and step — no fewer than the
// ~820 of the two-kernel path; on top of that the 64-row column slice (128 registers) plus the step's temporaries do not fit 256
// registers in any of the four code shapes tried (static row shifting, run-time row switch, top rows in LDS): 48-600 spilled
// registers, whose reloads sit inside the step.  Kept as the record of the design; see the notes before trying again.
//
// F_J2 = qr(J2, ColumnNorm()) and the solves of sub_search_direction for SMALL TALL problems in ONE launch
// (src/enlsip_functions.jl:223-224, :134-137 / :145-148, :151): the batched configurations C3 (512 x 64, t = 8) and
// C5 (256 x 32, t = 4), and every other shape with n2 + 1 <= 64 columns, at most 64 reflectors and m <= 512 rows.
//
// Bigger problems factor J2 without pivoting (CAQR) and pivot the small R afterwards, because a literal dgeqp3 walks the
// whole trailing matrix once per pivot step.  Here the whole matrix [J2 | d] fits the REGISTERS of one workgroup, so the
// literal algorithm (dlaqp2: pivot = first largest partial norm, dlarfg, Drmac-Bujanovic norm downdates) is also the
// fastest one: one chain of n2 pivot steps instead of two (unpivoted panel, then pivoted R), no reductions over lanes:
//
//   lane c owns COLUMN c (lane n2 = the carried right-hand side d), wave w owns ROWS 64 w .. 64 w + 63 in registers
//   (gn_wave_qrcp.hpp's layout, rows split over the waves).  A pivot step: every wave finds the pivot among its own copy of
//   the partial norms (replicated, bit-identical), the pivot lane publishes its 64 raw entries to the wave's LDS slot, every
//   lane forms its partial raw dot product with them, ONE workgroup barrier exchanges the partials (and the pivot row, which
//   lives in wave 0), and all waves finish the step redundantly: reflector scalars, update of their own rows, norm downdate.
//   The squared norm dlarfg needs IS the pivot lane's own raw dot product; v = scale * raw is never formed.
//   With n2 + 1 <= 32 a wave carries TWO problems (lanes 0..31 / 32..63): C5 keeps 58 of 64 lanes busy.
//
// The Householder vectors go to W (LAPACK layout: column rankA + position, below the diagonal), R to Rt, tau to tauJ — a
// plain QRPivoted, which the accessors read through the `direct` flag of the handle (gn_accessors.inc).  Rows of R live in a
// workgroup LDS image tmp[row * 65 + lane] until the solve.  Afterwards wave h finishes problem h of the workgroup exactly
// as k_pivot_small does: jpvt, rank, back substitution, p = F_A.Q [p1; p2], output records.
#pragma once
#include "gn_kernels_final.hpp"
#include "gn_wave_qrcp.hpp"

namespace gn {

#ifdef GN_TALL_STAMPS      // timing-only build: phase stamps (100 MHz wall clock) of workgroup 0, wave 0 and the last wave
__device__ long long g_tall_stamps[2][16];
#define GN_TALL_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x == 0 || threadIdx.x == blockDim.x - 64)) g_tall_stamps[threadIdx.x ? 1 : 0][i] = wall_clock64(); } while (0)
#else
#define GN_TALL_STAMP(i) do { } while (0)
#endif

constexpr int TALL_STATUS_FALLBACK = 8;    // state.status bit: the problem does not fit this kernel (J2 wider than the lanes of its half)

// ---- half-wave helpers (PPW problems per wave; half = 64 / PPW lanes) ----------------------------------------------------
template <int PPW>
__device__ __forceinline__ ArgMax half_argmax(double val, int pos, int idx, int ln) {
    if (PPW == 1) return wave_argmax(val, pos, idx);
    ArgMax a = {val, pos, idx};
    am_step<0xB1>(a);
    am_step<0x4E>(a);
    am_step<0x141>(a);
    am_step<0x140>(a);
    ArgMax r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        r[h] = {readlane_f64(a.val, 32 * h), __builtin_amdgcn_readlane(a.pos, 32 * h), __builtin_amdgcn_readlane(a.idx, 32 * h)};
        const double ov = readlane_f64(a.val, 32 * h + 16);
        const int op = __builtin_amdgcn_readlane(a.pos, 32 * h + 16);
        const int oi = __builtin_amdgcn_readlane(a.idx, 32 * h + 16);
        if (am_better(ov, op, r[h].val, r[h].pos)) r[h] = {ov, op, oi};
    }
    return (ln & 32) ? r[1] : r[0];
}
// x of lane src[h] (wave-uniform per half) in every lane of half h
template <int PPW>
__device__ __forceinline__ double half_bcast(double x, int srcA, int srcB, int ln) {
    const double a = readlane_f64(x, srcA);
    if (PPW == 1) return a;
    const double b = readlane_f64(x, srcB);
    return (ln & 32) ? b : a;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which on this target also
// waits for every outstanding GLOBAL store (s_waitcnt vmcnt(0)): with the Householder vector of every step streamed to W that
// is one HBM write acknowledgement (~1-2 us) per pivot step.  Nothing a step stores to global memory is read back in the kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct TallCtx {
    double* tmp;      // LDS: finished rows, tmp[row * 65 + lane]
    double* ex;       // LDS: [2][NW + 1][64] partial raw dot products per wave + the pivot row (double buffered by step parity)
    double* exr;      // LDS: [NW][64] partial sums of squares of a norm recomputation
    double* vb;       // LDS: this wave's [PPW][64] published pivot column entries
    double* dg;       // LDS: [PPW][64] diagonal of R
    int rows;         // m
    int kpm;          // rows of the top block (launch-wide bound on the reflector count): they live in tmp, not in registers
    int ncand, npart; // per lane (its half's problem): n2, n2 + 1
    int kp;           // per lane: reflectors of its problem
    double* taup;     // per lane: its problem's tauJ
    int ldw;
    // wave-uniform, per half: the problem's W, first J2 column (rankA) and reflector count
    double* Wh[2];
    int cbh[2], kph[2];
};

// One pivot step, the same code in every wave.
// Rows 0 .. kpm-1 (kpm = launch-wide bound on the reflector count, <= 64) are the only rows that ever become pivot rows: they
// live in the LDS image tmp[row * 65 + lane] from the start — no register row is ever addressed by a run-time index — and are
// dealt to the waves round-robin (row r belongs to wave r mod NW, which is the only wave that ever touches it).  Every wave
// keeps 64 further rows of its lanes' columns in registers; all of them take part in every step without a mask.
template <int NW, int PPW>
__device__ __forceinline__ void tall_step(double (&x)[64], const TallCtx& q, const int j, const int w, const int ln,
                                          int& mypos, double& vn1, double& vn2, double& myscale) {
    constexpr int HALF = 64 / PPW;
    const double tol3z = 1.4901161193847656e-08;    // sqrt(eps), dlaqp2
    const int lh = ln & (HALF - 1);
    const bool act = j < q.kp;                      // this lane's problem still has steps to do
    // (a) pivot = first position of the largest partial norm among the unprocessed columns of the half
    const bool cand = act && (lh < q.ncand) && (mypos >= j);
    const ArgMax am = half_argmax<PPW>(cand ? pivot_key(vn1) : -1.0, mypos, ln, ln);
    const int pl = am.idx, pp = am.pos;
    const int plA = __builtin_amdgcn_readlane(pl, 0), plB = PPW == 2 ? __builtin_amdgcn_readlane(pl, 32) : 0;
    // (b) LAPACK's swap of positions j <-> pp
    if (act) {
        if (mypos == j) mypos = pp;
        if (ln == pl) mypos = j;
    }
    // (c) the pivot lane publishes its raw column entries of this wave's register rows
    double* vb = q.vb + (PPW == 2 ? (ln & 32) * 2 : 0);          // [half][64]
    if (act && ln == pl) {
#pragma unroll
        for (int r = 0; r < 64; ++r) vb[r] = x[r];
    }
    wave_mem_sync();
    // (d) partial raw dot products: this wave's rows of the top block below the pivot row ...
    double dotraw = 0.0;
    const int rfirst = j + 1 + ((w - (j + 1)) & (NW - 1));       // first row > j that belongs to this wave (NW is a power of two)
    for (int r = rfirst; r < q.kpm; r += NW) dotraw += q.tmp[r * 65 + ln] * q.tmp[r * 65 + pl];
    // ... and its register rows: chunks of 8, the next chunk of the published column in flight while the current one is
    // used.  The scheduling fences keep the compiler from fetching all 64 entries first (128 more registers on top of the
    // 128 of the column: the step loop then spills, and every reload is a trip to memory).
    {
        double va[8], vn[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) va[rr] = vb[rr];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            if (ch + 1 < 8) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) vn[rr] = vb[8 * (ch + 1) + rr];
            }
            double pc = 0.0;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) pc += x[8 * ch + rr] * va[rr];
            dotraw += pc;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) va[rr] = vn[rr];
        }
    }
    double* ex = q.ex + (j & 1) * (NW + 1) * 64;
    const bool owner = (j & (NW - 1)) == w;          // this wave owns the pivot row
    ex[w * 64 + ln] = act ? dotraw : 0.0;
    if (owner) ex[NW * 64 + ln] = q.tmp[j * 65 + ln];
    lds_barrier();                                   // the one workgroup barrier of the step
    double dr = 0.0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) dr += ex[ww * 64 + ln];      // same order in every wave: bit-identical
    const double xs = ex[NW * 64 + ln];                          // pivot row entry of this lane's column
    const Reflector h = make_reflector(half_bcast<PPW>(xs, plA, plB, ln), half_bcast<PPW>(dr, plA, plB, ln));   // uniform per half
    const double tau = h.tau, scale = h.scale;
    const bool upd = act && (mypos > j) && (lh < q.npart);
    const double dot = xs + scale * dr;              // v' x
    const double wd = tau * dot, ws = wd * scale;
    const bool doupd = upd && tau != 0.0;
    const double newx = (ln == pl) ? h.beta : (doupd ? xs - wd : xs);     // row j of this lane's column after the step
    if (act && ln == pl) myscale = scale;
    const double wsm = doupd ? ws : 0.0;            // every lane walks the rows (uniform control flow)
    for (int r = rfirst; r < q.kpm; r += NW) q.tmp[r * 65 + ln] -= wsm * q.tmp[r * 65 + pl];
    {
        double va[8], vn[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) va[rr] = vb[rr];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            if (ch + 1 < 8) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) vn[rr] = vb[8 * (ch + 1) + rr];
            }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) x[8 * ch + rr] -= wsm * va[rr];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) va[rr] = vn[rr];
        }
    }
    if (owner) {
        // row j is final: R in the columns at a position >= j; left of the diagonal stay the (raw) Householder entries of the
        // columns processed earlier
        if (act && lh < q.npart && mypos >= j) q.tmp[j * 65 + ln] = newx;
        if (act && lh == 0) {
            q.taup[j] = tau;
            q.dg[(PPW == 2 ? (ln & 32) * 2 : 0) + j] = h.beta;
        }
    }
    wave_mem_sync();          // every reader of vb (and of this wave's tmp rows) is done before the next publish / update
    // (f) dlaqp2 norm downdate, replicated in every wave
    bool redo = false;
    double temp = 0.0;
    if (upd && lh < q.ncand && vn1 != 0.0) {
        temp = 1.0 - (fabs(newx) / vn1) * (fabs(newx) / vn1);
        temp = temp > 0.0 ? temp : 0.0;
        const double qq = vn1 / vn2;
        redo = temp * qq * qq <= tol3z;
        if (!redo) vn1 *= sqrt(temp);
    }
    if (__ballot(redo) != 0ull) {      // rare; the same lanes in every wave, so every wave meets the barrier
        double s2 = 0.0;
        for (int r = rfirst; r < q.kpm; r += NW) s2 += q.tmp[r * 65 + ln] * q.tmp[r * 65 + ln];
#pragma unroll
        for (int r = 0; r < 64; ++r) s2 += x[r] * x[r];
        q.exr[w * 64 + ln] = s2;
        lds_barrier();
        double tot = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tot += q.exr[ww * 64 + ln];
        if (redo) {
            const double nv = (j + 1 < q.rows) ? sqrt(tot) : 0.0;
            vn1 = nv;
            vn2 = nv;
        }
    }
}

// LDS (doubles): tmp[kpm * 65] ex[2 (NW + 1) 64] exr[NW 64] vb[NW PPW 64] dg[PPW 64] lpos[PPW 64] pbuf[PPW nv]
inline size_t tall_lds_bytes(int NW, int PPW, int kpm, int nv) {
    return (size_t)(kpm * 65 + 2 * (NW + 1) * 64 + NW * 64 + NW * PPW * 64 + 2 * PPW * 64 + PPW * nv + 8) * 8;
}

template <int NW, int PPW>
__global__ __launch_bounds__(64 * NW, 2) void k_qrcp_tall(FinalArgs a, int nprob) {
    constexpr int HALF = 64 / PPW;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int kpm = a.matd;                          // rows of the finished-row image (launch-wide bound on kp)
    double* tmp = smem;
    double* ex = tmp + kpm * 65;
    double* exr = ex + 2 * (NW + 1) * 64;
    double* vball = exr + NW * 64;
    double* dg = vball + NW * PPW * 64;
    int* lposall = reinterpret_cast<int*>(dg + PPW * 64);
    double* pball = dg + 2 * PPW * 64;

    GN_TALL_STAMP(0);
    const int ln = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int half = PPW == 2 ? (ln >> 5) : 0;
    const int lh = ln & (HALF - 1);
    const int prob_l = a.prob0 + blockIdx.x * PPW + half;            // this lane's problem
    const bool have = blockIdx.x * PPW + half < nprob;
    const int m = a.m, n = a.n, ldw = a.ldw;
    int rankA = 0, n2 = 0, kp = 0;
    bool fits = false;
    if (have) {
        const ProbState st = a.state[prob_l];
        rankA = st.rankA; n2 = st.n2; kp = st.kp;
        fits = (n2 + 1 <= HALF) && (kp <= kpm) && (kp <= 64);
        if (!fits && w == 0 && lh == 0) a.state[prob_l].status = st.status | TALL_STATUS_FALLBACK;      // the host redoes the batch on the general path
        if (!fits) kp = 0;
    }
    double* Wp = const_cast<double*>(a.W) + (size_t)(have ? prob_l : a.prob0) * a.sW;
    TallCtx q;
    q.tmp = tmp; q.ex = ex; q.exr = exr; q.vb = vball + w * PPW * 64; q.dg = dg;
    q.rows = m; q.kpm = kpm; q.ncand = fits ? n2 : 0; q.npart = fits ? n2 + 1 : 0; q.kp = kp;
    q.taup = a.tauJ + (size_t)(have ? prob_l : a.prob0) * a.sTauJ; q.ldw = ldw;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int src = (PPW == 2) ? 32 * hh : 0;
        const int ph = a.prob0 + blockIdx.x * PPW + (hh < PPW ? hh : 0);
        q.Wh[hh] = const_cast<double*>(a.W) + (size_t)(blockIdx.x * PPW + hh < nprob && hh < PPW ? ph : a.prob0) * a.sW;
        q.cbh[hh] = __builtin_amdgcn_readlane(rankA, src);
        q.kph[hh] = hh < PPW ? __builtin_amdgcn_readlane(kp, src) : 0;
    }

    // ---- this lane's column: the top block (rows < kpm) goes to the LDS image, row r by wave r mod NW; 64 further rows to the
    //      registers of every wave (row kpm + 64 w + r).  Every load is unconditional and in range (a guarded load per row
    //      would serialise 64 round trips) ----------------------------------------------------------------------------------
    double x[64];
    double s0 = 0.0;
    {
        const bool mine = have && fits && lh <= n2;
        const double* col = Wp + (size_t)(mine ? (lh < n2 ? rankA + lh : n) : 0) * ldw;
        for (int r = w; r < kpm; r += NW) {
            const double v = (mine && r < m) ? col[r < m ? r : m - 1] : 0.0;
            tmp[r * 65 + ln] = v;
            s0 += v * v;
        }
        const int rbase = kpm + 64 * w;
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int row = rbase + r;
            x[r] = col[row < m ? row : m - 1];
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) x[r] = (mine && rbase + r < m) ? x[r] : 0.0;
    }
    GN_TALL_STAMP(1);
    // partial norms: replicated in every wave from the exchanged sums of squares
    {
        double s = s0;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            double pc = 0.0;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) pc += x[8 * ch + rr] * x[8 * ch + rr];
            s += pc;
        }
        exr[w * 64 + ln] = s;
    }
    __syncthreads();
    double vn1 = 0.0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) vn1 += exr[ww * 64 + ln];
    vn1 = sqrt(vn1);
    double vn2 = vn1, myscale = 0.0;
    int mypos = lh;
    int kmax = kp;                                    // steps of the workgroup: the larger kp of its problems
    if (PPW == 2) {
        const int k0 = __builtin_amdgcn_readlane(kp, 0), k1 = __builtin_amdgcn_readlane(kp, 32);
        kmax = k0 > k1 ? k0 : k1;
    } else {
        kmax = __builtin_amdgcn_readfirstlane(kp);
    }
    __syncthreads();                                  // exr is free again
    GN_TALL_STAMP(2);
    for (int j = 0; j < kmax; ++j) tall_step<NW, PPW>(x, q, j, w, ln, mypos, vn1, vn2, myscale);
    GN_TALL_STAMP(3);
    __syncthreads();                                  // tmp, dg, tau complete

    // ---- Householder vectors: the lane that became pivot p still holds its raw column below the diagonal (processed columns
    //      are never touched again): v = scale * raw goes to column rankA + p of W.  Nothing is stored inside the step loop: a
    //      global store there drags an s_waitcnt vmcnt(0) — one HBM write acknowledgement — into every step.
    {
        const bool piv = have && fits && lh < n2 && mypos < kp;
        double* Vc = Wp + (size_t)(rankA + (piv ? mypos : 0)) * ldw;
        // rows of the top block from the image (row r by wave r mod NW), the others from the registers
        for (int row = w; row < kpm; row += NW)
            if (piv && row > mypos && row < m) Vc[row] = myscale * tmp[row * 65 + ln];
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int row = kpm + 64 * w + r;
            if (piv && row < m) Vc[row] = myscale * x[r];
        }
    }
    GN_TALL_STAMP(4);
    // ---- d beyond the reflectors: rows >= kp of the carried column, lanes along rows ------------------------------------------
    if (a.d_out) {
        double* vb = q.vb;
#pragma unroll
        for (int hh = 0; hh < PPW; ++hh) {
            const int src = 32 * hh * (PPW - 1);
            const int okh = __builtin_amdgcn_readlane((int)(have && fits), src);
            const int n2h = __builtin_amdgcn_readlane(n2, src), kph = __builtin_amdgcn_readlane(kp, src);
            const int ph = a.prob0 + blockIdx.x * PPW + hh;
            if (!okh) continue;
            if (ln == src + n2h) {
#pragma unroll
                for (int r = 0; r < 64; ++r) vb[hh * 64 + r] = x[r];
            }
            wave_mem_sync();
            const int rowg = kpm + 64 * w + ln;
            if (rowg < m) a.d_out[(size_t)ph * a.sDo + rowg] = vb[hh * 64 + ln];
            wave_mem_sync();
        }
    }
    GN_TALL_STAMP(5);
    // ---- wave h finishes problem h: permutation, R, rank, back substitution, p = F_A.Q [p1; p2], records -----------------------
    // (a one-wave workgroup carrying two problems finishes both, one after the other)
    for (int hsel = w; hsel < PPW; hsel += NW) {
    const int hb = hsel * HALF;                       // first lane of the problem's half
    const int prob = a.prob0 + blockIdx.x * PPW + hsel;
    if (blockIdx.x * PPW + hsel >= nprob) continue;
    // bring the half's lane state to lanes 0 .. HALF-1 of this wave
    const int myposh = __shfl(mypos, hb + lh);
    const int n2u = __builtin_amdgcn_readlane(n2, hb), kpu = __builtin_amdgcn_readlane(kp, hb), rAu = __builtin_amdgcn_readlane(rankA, hb);
    const int fitu = __builtin_amdgcn_readlane((int)fits, hb);
    if (!fitu) continue;
    ProbState* stp = a.state + prob;
    int* lpos = lposall + hsel * 64;
    double* pbuf = pball + hsel * a.nv;
    double* dgh = dg + hsel * 64;
    double* Rt = a.Rt + (size_t)prob * a.sRt;
    long long* jpvtJ = a.jpvtJ + (size_t)prob * a.sJJ;
    const double* FA = a.FA + (size_t)prob * a.sFA;
    const double* tauA = a.tauA + (size_t)prob * a.sTauA;
    const double* p1 = a.p1 + (size_t)prob * a.sP1;
    const double* bvec = a.bvec + (size_t)prob * a.sB;
    const int t = a.t, kA = a.kA, ldr = a.ldr;
    const bool inh = ln < HALF;                       // worker lanes: lane i <-> column i / row i of the problem
    int status = 0;
    if (inh && ln < n2u) jpvtJ[myposh] = ln + 1;
    if (inh && ln <= n2u) lpos[myposh] = ln;
    wave_mem_sync();
    const int lp = (inh && ln <= n2u) ? lpos[ln] : 0;       // lane i: the column that sits at position i
    for (int P = 0; P <= n2u; ++P) {                          // upper parts of the factor columns (and the carried one)
        const int src = lpos[P];
        if (ln < kpu && ln <= P) Rt[ln + (size_t)P * ldr] = tmp[ln * 65 + hb + src];
    }
    int rankJ2 = 0;
    if (kpu > 0) {
        const double d0 = fabs(dgh[0]);
        if (!(d0 < a.eps_rank)) {
            const double tol = d0 * sqrt((double)kpu) * a.eps_rank;
            const bool fail = (ln < kpu) && !(fabs(dgh[ln < kpu ? ln : 0]) > tol);
            const unsigned long long mk = __ballot(fail);
            rankJ2 = mk ? (int)__builtin_ctzll(mk) : kpu;
        }
    }
    int dimJ2 = (a.dimJ2_override >= 0) ? a.dimJ2_override : rankJ2;
    dimJ2 = dimJ2 < kpu ? dimJ2 : kpu;
    // dp2 = U(Rt[1:dimJ2,1:dimJ2]) \ d[1:dimJ2]: lane r carries row r of the right-hand side
    double zw = (ln < dimJ2) ? tmp[ln * 65 + hb + n2u] : 0.0;
    for (int i = dimJ2 - 1; i >= 0; --i) {
        const int li = lpos[i];
        const double dkk = tmp[i * 65 + hb + li];
        if (dkk == 0.0) status |= 1;
        const double yi = wave_bcast(zw, i) / dkk;
        if (ln == i) zw = yi;
        if (ln < i) zw -= tmp[ln * 65 + hb + li] * yi;
    }
    // y = [p1 ; p2],  p2[pJ[i]-1] = (i < dimJ2 ? dp2[i] : 0)
    for (int i = ln; i < rAu; i += WAVE) pbuf[i] = p1[i];
    if (inh && ln < n2u) pbuf[rAu + lp] = (ln < dimJ2) ? zw : 0.0;
    wave_mem_sync();
    wave_apply_reflectors<false>(FA, n, tauA, kA, n, pbuf);      // p = F_A.Q * y
    wave_mem_sync();
    if (a.p_out)
        for (int i = ln; i < n; i += WAVE) a.p_out[(size_t)prob * a.sPo + i] = pbuf[i];
    if (a.b_out)
        for (int i = ln; i < t; i += WAVE) a.b_out[(size_t)prob * a.sBo + i] = bvec[i];
    if (a.d_out && ln < kpm && ln < m) a.d_out[(size_t)prob * a.sDo + ln] = tmp[ln * 65 + hb + n2u];      // the whole top block: rows < kp are (Q3' d)[1:kp]
    if (a.jA_out)
        for (int i = ln; i < t; i += WAVE) a.jA_out[(size_t)prob * a.sJAo + i] = a.jpvtA[(size_t)prob * a.sJA + i];
    if (a.jL_out)
        for (int i = ln; i < kA; i += WAVE) a.jL_out[(size_t)prob * a.sJLo + i] = a.jpvtL[(size_t)prob * a.sJL + i];
    if (a.jJ_out && inh && ln < n2u) a.jJ_out[(size_t)prob * a.sJJo + myposh] = ln + 1;
    if (ln == 0) {
        stp->rankJ2 = rankJ2;
        stp->dimJ2 = dimJ2;
        stp->status |= status;
    }
    }
    GN_TALL_STAMP(6);
}

// Returns false when the launch shape is outside the kernel's range (the caller takes the CAQR path).
inline bool tall_shape_ok(long long m, long long n, long long kA) { return n - kA + 1 <= 64 && m <= 512 && m >= 1; }

// the general path takes over a batch in which some problem did not fit: drop the marker from the state records
__global__ void k_tall_clear(ProbState* st, int nprob) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nprob) st[i].status &= ~TALL_STATUS_FALLBACK;
}

inline bool launch_qrcp_tall(int batch, hipStream_t s, FinalArgs a) {
    if (!tall_shape_ok(a.m, a.n, a.kA)) return false;
    const int kA = a.kA;
    const int n2min = a.n - kA;                       // full-rank A: the narrowest J2 a problem of this batch can have
    const long long mx = a.n > a.t ? a.n : a.t;
    a.nv = (int)((mx + 7) / 8 * 8);
    int kpm = (int)(a.m < a.n ? a.m : a.n);
    if (kpm > 64) kpm = 64;
    a.matd = kpm > 0 ? kpm : 1;
    const int NW = a.m <= 64 ? 1 : (a.m <= 128 ? 2 : (a.m <= 256 ? 4 : 8));
    const int PPW = (n2min + 1 <= 32) ? 2 : 1;        // problems of lower rankA (wider J2) report TALL_STATUS_FALLBACK
    const size_t lds = tall_lds_bytes(NW, PPW, a.matd, a.nv);
    const dim3 grid((batch + PPW - 1) / PPW);
#define GN_TALL(NWv, PPWv) do { big_lds_tall(k_qrcp_tall<NWv, PPWv>, lds); hipLaunchKernelGGL((k_qrcp_tall<NWv, PPWv>), grid, dim3(64 * NWv), lds, s, a, batch); } while (0)
    auto big_lds_tall = [](auto k, size_t bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); };
    if (PPW == 2) {
        if (NW == 1) GN_TALL(1, 2); else if (NW == 2) GN_TALL(2, 2); else if (NW == 4) GN_TALL(4, 2); else GN_TALL(8, 2);
    } else {
        if (NW == 1) GN_TALL(1, 1); else if (NW == 2) GN_TALL(2, 1); else if (NW == 4) GN_TALL(4, 1); else GN_TALL(8, 1);
    }
#undef GN_TALL
    return true;
}

}  // namespace gn
