// Column-pivoted QR (dgeqp3 pivot rule, dlaqp2 arithmetic) of the kp x n2 factor R0 in BLOCKS
// of pivot steps with exact pivots ("speculate on the candidates, verify against a bound").
//
// A pivot step needs the partial norms of ALL trailing columns, which is what makes dgeqp3 half
// BLAS-2.  But partial norms only decrease.  So at the start of a block (all norms exact):
//   * the K columns with the largest norms (ties -> lowest LAPACK position) are pulled into the
//     REGISTERS of ONE workgroup (gn_kernels_qrcp_block_reg.hpp; a first form kept them in an LDS slab: 14 us per
//     step against 4.3, retired), and B := the best norm among the columns left outside is recorded;
//   * the workgroup runs ordinary pivot steps on its K columns only — reflector, update, exact
//     norm downdate (Drmac-Bujanovic rule) — as long as the best remaining inside norm is
//     STRICTLY larger than B: then no outside column (whose current norm is <= its norm at block
//     start <= B) can be the pivot, and the choice is exactly LAPACK's.  The first step of a
//     block always passes (the globally best column is inside), so every block makes progress;
//   * a second kernel applies the block's s reflectors to all outside columns (and the carried
//     right-hand side) with the same step-by-step norm downdates, after which all norms are exact
//     again and the next block starts.
// On the synthetic C2 problems a block of K = 32 candidates advances 10-24 pivot steps (mean 14),
// so the 448-step factorisation takes ~32 block launches instead of 448 step launches, and the
// trailing matrix crosses the memory system once per block instead of once per step.
// Positions are logical (pos[] / colat[]), columns never move physically; k_qd_assemble gathers
// the LAPACK-style compact factors at the end (shared with gn_kernels_qrcp_dist.hpp).
#pragma once
#include "gn_kernels_qrcp_dist.hpp"
#include "gn_kernels_update_v4.hpp"

namespace gn {

constexpr int SB_KMAX = 64;                  // candidate columns per block (LDS permitting)
constexpr int SB_NMAX = 1024;

#ifdef ENLSIP_SB_STEP_STAMPS     // diagnostic build only (tests/probes/sb_step_probe.py)
__device__ long long g_sb_phase[16];
__device__ long long g_sb_blk[8];      // block-level phases of k_sb_factor_reg, [7] = blocks
#endif

struct SbInfo {       // per problem, device
    int j0;           // first pivot step of the next block
    int s;            // steps done by the last select/factor launch
    int blk;          // id of the last block in which `s` steps were done
    int pad;          // first pivot step of that block
    int nact;         // columns the block update has to touch (list in `act`): still trailing, not a candidate, + the rhs
    int pad2[3];
};

struct SbArgs {
    QdArgs q;         // shared buffers (M, Vb, diag, tau, vn1, vn2, chosen, pos, colat, jpvt, state)
    SbInfo* info;     // per problem
    int* inblk;       long long sIn;    // per physical column: id of the last block it was a candidate in
    int blkid;
    double* Tsb;      long long sTsb;   // per problem: dlarft T (32 x 32, column-major) of the last block's reflectors
    int* act;         long long sAct;   // per problem: compact list of the columns the block update touches
    long long* dbg;   // optional (diagnostic builds): 8 realtime stamps per block of problem prob0
    // per handle: [blkid] = max over the problems of the row count kp - j0 a block started with, [SB_STAT_BLKS + blkid] =
    // SB_STAT_OFF - min of it.  The host reads them back with the end-of-stage check and launches, in the NEXT solve of the same
    // shape, only the forms of the select / factor kernel whose row ranges they touch (run_qrcp_block).
    int* rows_stat;
};
constexpr int SB_STAT_BLKS = 1024;     // block ids (a block makes at least one step: kp <= 512 here)
constexpr int SB_STAT_OFF = 1 << 20;

// reset of the per-problem block state; runs after k_qd_init
// j_first: pivot steps already done by a launch-per-step head (hybrid stage; 0 otherwise)
__global__ void k_sb_reset(SbArgs a, int n, int j_first) {
    const int prob = blockIdx.y + a.q.prob0;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) a.inblk[prob * a.sIn + c] = -1;
    if (c == 0) {
        const int kp = a.q.state[prob].kp;
        SbInfo z = {j_first < kp ? j_first : kp, 0, -1, 0};
        a.info[prob] = z;
    }
}

constexpr int SB_UCW = 32;   // columns per workgroup of the update kernel (8 per wave)

// ---------------------------------------------------------------------------------------------
// The block update (kp <= 512): the block's s reflectors are applied to 32 outside
// columns at once as C <- C - V (T' (V' C)) on the matrix pipe — the workgroup body of the CAQR trailing
// update (gn_kernels_update_v4.hpp) with element-granular row masks (the row range starts at the
// arbitrary pivot step jb) — followed by the dlaqp2 norm downdates of those columns.
// The downdate of step t needs the entry of row jb + t after reflectors 0..t; later reflectors do not
// touch that row, so it is the FINAL entry.  A recomputation (cancellation test of dlaqp2) needs the norm
// of the rows below jb + t at that point; the later reflectors act on exactly those rows and are
// orthogonal, so it equals sqrt(sum of the final squares of rows > jb + t) — taken from the R rows of
// the block and a per-column sum of squares of the rows below them.
// ---------------------------------------------------------------------------------------------
struct SbPost {           // per-unit hook of v4_body: sums of squares below the R rows + export of unit 0
    double* acc8;         // [8] per lane: column 16 ct + lq + 4 r  <->  index 4 ct + r
    double* img;          // wave 0: [col][row] image (ld V4_LD) of the block's first 32 rows
    int w, lr, lq, s;     // s = first slot below the block's R rows (steps + alignment shift)
    template <class F>
    __device__ __forceinline__ void operator()(int g, const F& fr) const {
        const int s0 = 32 * (w + 4 * g) + 2 * lr;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double x0 = fr[0][ct][r], x1 = fr[1][ct][r];
                acc8[4 * ct + r] += ((s0 >= s) ? x0 * x0 : 0.0) + ((s0 + 1 >= s) ? x1 * x1 : 0.0);
                if (g == 0 && w == 0) *(v4_d2*)&img[(16 * ct + lq + 4 * r) * V4_LD + 2 * lr] = (v4_d2){x0, x1};
            }
    }
};

__global__ __launch_bounds__(256, 2) void k_sb_update_blk(SbArgs a) {
    __shared__ __attribute__((aligned(16))) double stage[4][V4_STAGE];
    __shared__ __attribute__((aligned(16))) double W2l[PB * PB];
    __shared__ double ssq[4][32];

    const int prob = blockIdx.y + a.q.prob0;
    const ProbState st = a.q.state[prob];
    if (a.q.n2cap > 0 && st.n2 > a.q.n2cap) return;      // wider than the launch shape: redone by the caller
    const int kp = st.kp, n2 = st.n2, ctot = n2 + 1;
    const SbInfo info = a.info[prob];
    if (info.blk != a.blkid || info.s == 0) return;     // this problem did no step in this block
    const int jb = info.pad, s = info.s;
    const int cb0 = blockIdx.x * 32;                    // position in the list of active columns
    const int nact = info.nact;
    if (cb0 >= nact) return;
    (void)ctot;
    const int ln = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int lr = ln & 15, lq = ln >> 4;
    const int* actl = a.act + prob * a.sAct;
    unsigned vmask = 0u;
    unsigned coff[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int idx = cb0 + 16 * (b >> 2) + 4 * (b & 3) + lq;
        const bool ok = idx < nact;
        vmask |= ok ? (1u << b) : 0u;
        coff[b] = (unsigned)actl[ok ? idx : cb0] * (unsigned)a.q.ldr * 8u;   // invalid slots re-read a valid column, masked out
    }
    const unsigned smask = vmask;
    V4Ctx c;
    c.Wm = a.q.Vb + prob * a.q.sVb;
    c.C = a.q.M + prob * a.q.sM;
    c.T = a.Tsb + prob * a.sTsb;
    const int d = jb & 1;                 // start one row early when jb is odd: every row pair is 16-byte aligned
    c.tile_row0 = jb - d; c.r0 = jb - d; c.gblk0 = 0; c.S = 0; c.mode = 0;
    c.T2 = nullptr; c.col02 = 0; c.bw2 = 0; c.dshift2 = 0;
    c.ldw = a.q.ldr; c.col0 = jb; c.bw = s; c.cb0 = cb0;
    c.rows_valid = kp - (jb - d);
    c.dshift = d;
    const int nvu = (c.rows_valid + 31) / 32;
    const int ngw = nvu > w ? (nvu - w + 3) / 4 : 0;

#ifdef ENLSIP_SB_STEP_STAMPS
    const bool ub_st = threadIdx.x == 0 && blockIdx.x == 1 && prob == a.q.prob0;
    long long ub_t0 = wall_clock64();
#endif
    double acc8[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc8[b] = 0.0;
    SbPost post{acc8, stage[0], w, lr, lq, s + d};
    if (nact - cb0 <= 16) v4_dispatch<8, false, false, 1, true, SbPost, true>(c, w, ngw, stage, W2l, vmask, smask, post, coff);
    else v4_dispatch<8, false, false, 2, true, SbPost, true>(c, w, ngw, stage, W2l, vmask, smask, post, coff);

    // per-column sums of squares: reduce over the 16 row-pair lanes, one writer per (wave, column)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        double x = acc8[b];
        x += dpp_f64<0xB1>(x);
        x += dpp_f64<0x4E>(x);
        x += dpp_f64<0x141>(x);
        x += dpp_f64<0x140>(x);
        if (lr == 0) ssq[w][16 * (b >> 2) + 4 * (b & 3) + lq] = x;
    }
    __syncthreads();
#ifdef ENLSIP_SB_STEP_STAMPS
    long long ub_t1 = wall_clock64();
    if (ub_st) { atomicAdd((unsigned long long*)&g_sb_phase[10], (unsigned long long)(ub_t1 - ub_t0)); atomicAdd((unsigned long long*)&g_sb_phase[12], 1ull); }
#endif
    if (w != 0 || ln >= 32 || cb0 + ln >= nact) return;
    const int cc = actl[cb0 + ln];
    if (cc >= n2) return;                               // the carried right-hand side has no norm
    double* vn1 = a.q.vn1 + prob * a.q.sVn;
    double* vn2 = a.q.vn2 + prob * a.q.sVn;
    const double tol3z = 1.4901161193847656e-08;
    const double srest = (ssq[0][ln] + ssq[1][ln]) + (ssq[2][ln] + ssq[3][ln]);
    const double* col = stage[0] + ln * V4_LD + d;        // final entries of rows jb .. jb + 31 of this column
    // The s downdates of a column in a row, on the SQUARED norm q = o1^2: dlaqp2's step
    //     temp = max(0, 1 - (|a| / o1)^2), temp2 = temp (o1 / o2)^2, o1 <- o1 sqrt(temp)  unless temp2 <= tol3z (then recompute)
    // is  q <- max(0, q - a^2)  unless  q - a^2 <= tol3z o2^2 : one multiply-add and a compare per step instead of two divisions and
    // a square root, in a chain of up to 32 steps that one wave works off while the other three of the workgroup wait (4.7 us of a
    // workgroup's 23); the norm itself is needed once, at the end.
    const double o1_in = vn1[cc];
    double q = o1_in * o1_in, o2 = vn2[cc], lim = tol3z * o2 * o2;
    for (int t = 0; t < s; ++t) {
        if (q == 0.0) continue;
        const double ajc = col[t];
        double qn = q - ajc * ajc;
        qn = qn > 0.0 ? qn : 0.0;
        if (qn <= lim) {
            double sq = srest;
            for (int t2 = s - 1; t2 > t; --t2) sq += col[t2] * col[t2];
            q = (jb + t + 1 < kp) ? sq : 0.0;
            o2 = sqrt(q);
            lim = tol3z * q;
        } else {
            q = qn;
        }
    }
    vn1[cc] = (q == o1_in * o1_in) ? o1_in : sqrt(q);      // untouched norms stay bit for bit
    vn2[cc] = o2;
#ifdef ENLSIP_SB_STEP_STAMPS
    if (ub_st) atomicAdd((unsigned long long*)&g_sb_phase[11], (unsigned long long)(wall_clock64() - ub_t1));
#endif
}

}  // namespace gn
