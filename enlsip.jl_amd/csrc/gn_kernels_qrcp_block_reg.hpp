// Select/factor kernel of the blocked pivoted QR (gn_kernels_qrcp_block.hpp), second form: the K <= 64
// candidate columns of a block live in REGISTERS (16 waves x 4 columns, lane = row), not in LDS.
//
// The first form keeps the candidates in an LDS slab and moves every candidate through LDS <-> registers once
// per pivot step: at K = 64 and 448 rows that is 459 KB of LDS traffic per step (~2.5 us at the LDS rates of
// this chip) and the slab limits K to 32 while the rows are long.  Here a step touches LDS only for the
// reflector broadcast (one column), exactly like the CAQR panel kernel:
//   * every wave finds the pivot among the candidates' norms (LDS scalars, identical result everywhere);
//   * the wave that owns the pivot column builds the reflector from its registers and broadcasts it;
//   * every wave applies it to its 4 columns with one batched transposed reduction (wave_allsum4), downdates
//     their norms (dlaqp2 rule), and the dot products with the columns retired earlier in the block are the Gram
//     entries of the block's T factor — no separate pass over the reflectors.
// Pivot rule, certainty test against the best outside norm, logical positions and the write-back are those of
// the first form (same bookkeeping arrays, no slab).
#pragma once
#include "gn_kernels_qrcp_block.hpp"

namespace gn {

#ifdef ENLSIP_SB_STEP_STAMPS     // diagnostic build only (tests/probes/sb_step_probe.py): phase times of a pivot step, summed over the steps of wave 0 of problem prob0 (g_sb_phase: gn_kernels_qrcp_block.hpp)
#define SB_PH(i) do { __builtin_amdgcn_sched_barrier(0); if (sb_st) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long t_ = wall_clock64(); sb_acc[i] += t_ - sb_t; sb_t = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SB_PH(i) do { } while (0)
#endif

struct SbRegLds {             // bookkeeping of a block: keys, positions, candidates, the block's Gram / T entries
    double valk[SB_NMAX];
    int posk[SB_NMAX];
    int pos_l[SB_NMAX];
    int colat_l[SB_NMAX];
    double cvn1[2][SB_KMAX], cvn2[SB_KMAX];
    int ccol[SB_KMAX], cpos[2][SB_KMAX];
    int rankl[SB_NMAX];
    double taul[SB_KMAX], betal[SB_KMAX];
    double gram[32 * 33];
    double tmat[32 * 33];
    int tslot[32];
    double taustep[32];       // tau of the block's steps, by step (the T factor reads them without the detour over tslot)
    double vsh[2][512];       // reflector broadcast, double-buffered by step parity
    double tau_s[2];
    int wtot[16];
    double bval;
    int bpos;
    int K;
};

// RPL: rows of the block kp - j0 <= 64 * RPL.  NWV waves hold NCW = 64 / NWV candidate columns each.  With 16 waves
// (128 registers per lane) the compiler spills inside the step loop and a step costs 7.7 us; 8 waves x 8 columns
// run spill-free.
// The row count of a block, kp - j0, shrinks as the factorisation advances, and with it what a step costs and what the
// workgroup needs: the <7 / 8> forms take the whole register file of a CU (256 registers x 512 threads), the <4> form half of
// it, the <2> form a quarter.  Every block is therefore launched in up to three forms (run_qrcp_block) and a workgroup runs
// only in the form that fits its problem's current row count: RPL_LOW = rows-per-lane of the next smaller form launched
// beside this one (0: none) — problems that fit it are left to it.
template <int RPL, int NWV, int RPL_LOW = 0>
__global__ __launch_bounds__(64 * NWV, RPL <= 2 ? 4 : (RPL <= 4 ? 3 : 2)) void k_sb_factor_reg(SbArgs a) {
    constexpr int NCW = SB_KMAX / NWV;
    constexpr int NT = 64 * NWV;
    __shared__ SbRegLds L;
    const int prob = blockIdx.x + a.q.prob0;
    const ProbState st = a.q.state[prob];
    if (a.q.n2cap > 0 && st.n2 > a.q.n2cap) return;      // wider than the launch shape: redone by the caller
    const int kp = st.kp, n2 = st.n2;
    SbInfo* info = a.info + prob;
    const int j0 = info->j0;
    if (j0 >= kp) return;
    const int tid = threadIdx.x, ln = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int rows = kp - j0;
    if (info->blk == a.blkid) return;                     // a larger form has just done this block's steps (and advanced j0)
    if (rows > 64 * RPL) return;                          // served by a larger form
    if (RPL_LOW > 0 && rows <= 64 * RPL_LOW) return;      // served by the smaller form launched beside this one
    if (tid == 0 && a.rows_stat && a.blkid < SB_STAT_BLKS) {      // row-count statistics of this block id (launch hints of the next solve)
        atomicMax(a.rows_stat + a.blkid, rows);
        atomicMax(a.rows_stat + SB_STAT_BLKS + a.blkid, SB_STAT_OFF - rows);
    }
    double* M = a.q.M + prob * a.q.sM;
    double* vn1 = a.q.vn1 + prob * a.q.sVn;
    double* vn2 = a.q.vn2 + prob * a.q.sVn;
    int* chosen = a.q.chosen + prob * a.q.sI;
    int* pos = a.q.pos + prob * 2 * a.q.sI;       // parity-0 arrays only
    int* colat = a.q.colat + prob * 2 * a.q.sI;
    const double tol3z = 1.4901161193847656e-08;

#ifdef ENLSIP_SB_STEP_STAMPS
    long long bl_t = wall_clock64();
    auto stamp = [&](int i) {        // block-level phases: time since the previous stamp -> g_sb_blk[i]
        if (tid == 0 && prob == a.q.prob0) {
            const long long t_ = wall_clock64();
            atomicAdd((unsigned long long*)&g_sb_blk[i], (unsigned long long)(t_ - bl_t));
            bl_t = t_;
        }
    };
#else
    auto stamp = [&](int i) {
        if (a.dbg && tid == 0 && prob == a.q.prob0) a.dbg[a.blkid * 8 + i] = (long long)__builtin_amdgcn_s_memrealtime();
    };
#endif
    stamp(0);
    // ---- 1. keys of the trailing columns --------------------------------------------------------
    for (int c = tid; c < n2; c += NT) {
        const bool tr = chosen[c] < 0;
        L.valk[c] = tr ? pivot_key(vn1[c]) : -1.0;      // NaN norms rank as +inf: the order stays total, every slot < K is filled
        const int p = pos[c];
        L.posk[c] = tr ? p : 0x7fffffff;
        L.pos_l[c] = p;
        L.colat_l[c] = colat[c];
        L.rankl[c] = 0;
    }
    const int ntrail = n2 - j0;
    int K = SB_KMAX < ntrail ? SB_KMAX : ntrail;
    if (tid == 0) {
        L.K = K;
        L.bval = -1.0;     // no outside column unless a rank == K exists
        L.bpos = 0x7fffffff;
    }
    __syncthreads();
    // ---- 2. order the trailing columns by (norm desc, position asc): bitonic sort of (norm, position, column) in
    // LDS (the Gram / T images are free now); rank r < K = candidate slot r, rank K = best outside column = the bound
    if (n2 <= 512 && NT == 512) {
        // Register form (at most 512 columns, one element per thread): a bitonic network over 512 records (norm, position << 16 |
        // column) — the packed word makes the order strict and total, so every element can decide for itself whether it keeps
        // its record or takes its partner's.  Partners at distance < 64 are lanes of the same wave (DPP / permlane swaps: no
        // memory, no barrier — 39 of the 45 passes), partners at distance >= 64 go through two alternating LDS images with one
        // barrier per pass.  Fully unrolled: the partner distance of a pass is a constant.  Per block (C2, phase stamps): 4.2 us
        // against 13.3 us for the LDS form below, which pays an LDS round trip per pass (with ds_bpermute exchanges 4.9 us; with
        // the loops left rolled 7.7 us, and 11.3 us with the DPP exchanges behind a run-time switch).
        double* xv = L.gram;                                                       // 2 x 512 doubles
        unsigned* xp = reinterpret_cast<unsigned*>(L.gram + 1024);                 // 2 x 512 words
        const int e = tid;
        const bool real = e < n2;
        double v = real ? L.valk[e] : -2.0;
        const int p0 = real ? L.posk[e] : 0x7fffffff;
        unsigned pc = ((unsigned)(p0 > 0xFFFF ? 0xFFFF : p0) << 16) | (unsigned)e;
        int xb = 0;
#pragma unroll
        for (int kk = 2; kk <= 512; kk <<= 1) {
            const bool up = (e & kk) == 0;                  // better-first in this half
#pragma unroll
            for (int jx = kk >> 1; jx > 0; jx >>= 1) {
                double pv;
                unsigned ppc;
                if (jx >= 64) {
                    xv[512 * xb + e] = v;
                    xp[512 * xb + e] = pc;
                    __syncthreads();
                    pv = xv[512 * xb + (e ^ jx)];
                    ppc = xp[512 * xb + (e ^ jx)];
                    xb ^= 1;
                } else {
                    int lo = __double2loint(v), hi = __double2hiint(v), pw = (int)pc;
                    switch (jx) {                           // a constant after unrolling
#define GN_SB_X(J) case J: lo = lane_xor_i32<J>(lo, ln); hi = lane_xor_i32<J>(hi, ln); pw = lane_xor_i32<J>(pw, ln); break;
                        GN_SB_X(1) GN_SB_X(2) GN_SB_X(4) GN_SB_X(8) GN_SB_X(16)
                        default: lo = lane_xor_i32<32>(lo, ln); hi = lane_xor_i32<32>(hi, ln); pw = lane_xor_i32<32>(pw, ln); break;
#undef GN_SB_X
                    }
                    ppc = (unsigned)pw;
                    pv = __hiloint2double(hi, lo);
                }
                const bool partner_better = pv > v || (pv == v && ppc < pc);
                const bool lower = (e & jx) == 0;            // this element is the first of its pair
                const bool take = lower ? (partner_better == up) : (partner_better != up);
                if (take) {
                    v = pv;
                    pc = ppc;
                }
            }
        }
        // thread r holds the record of rank r
        {
            const int r = tid;
            const int c = (int)(pc & 0xFFFFu);
            const int p = (int)(pc >> 16);
            if (c < n2 && p != 0xFFFF) {                    // not padding, a trailing column
                L.rankl[c] = r;
                if (r < K) {
                    L.ccol[r] = c;
                    L.cpos[0][r] = p;
                    L.cvn1[0][r] = v;
                    L.cvn2[r] = vn2[c];
                    a.inblk[prob * a.sIn + c] = a.blkid;
                } else if (r == K) {
                    L.bval = v;
                    L.bpos = p;
                }
            }
        }
    } else {
        int P = 64;
        while (P < n2) P <<= 1;                            // <= 1024
        double* sv = L.gram;                               // 1024 doubles
        int* sp = reinterpret_cast<int*>(L.gram + 1024);   // 1024 ints
        int* sc = sp + 1024;                               // 1024 ints  (gram + tmat = 2112 doubles >= 2048)
        for (int e = tid; e < P; e += NT) {
            const bool in = e < n2;
            sv[e] = in ? L.valk[e] : -2.0;
            sp[e] = in ? L.posk[e] : 0x7fffffff;
            sc[e] = in ? e : -1;
        }
        __syncthreads();
        // P / 2 <= NT: a thread handles one pair per pass, and in the passes with partner distance jx <= 64 the pairs of wave w
        // lie in elements [128 w, 128 w + 128) — wave-local, no workgroup barrier.  Only the passes with jx >= 128 (three of
        // the 45 at P = 512) exchange data between waves: barriers around those only.
        static_assert(NT >= SB_NMAX / 2, "one pair per thread and pass");
        for (int kk = 2; kk <= P; kk <<= 1) {
            for (int jx = kk >> 1; jx > 0; jx >>= 1) {
                if (jx >= 128) __syncthreads();
                for (int t = tid; t < (P >> 1); t += NT) {
                    const int i = ((t & ~(jx - 1)) << 1) | (t & (jx - 1));
                    const int l = i | jx;
                    const bool up = (i & kk) == 0;          // better-first in this half
                    const double va = sv[i], vb = sv[l];
                    const int pa = sp[i], pb = sp[l];
                    const bool b_better = vb > va || (vb == va && pb < pa);
                    if (b_better == up) {
                        const int ca = sc[i], cb = sc[l];
                        sv[i] = vb; sv[l] = va;
                        sp[i] = pb; sp[l] = pa;
                        sc[i] = cb; sc[l] = ca;
                    }
                }
                if (jx >= 128) __syncthreads();
                else wave_mem_sync();
            }
        }
        __syncthreads();
        for (int r = tid; r < n2; r += NT) {
            const int c = sc[r];
            const int p = sp[r];
            if (c < 0 || p == 0x7fffffff) continue;         // padding / not a trailing column
            L.rankl[c] = r;
            if (r < K) {
                L.ccol[r] = c;
                L.cpos[0][r] = p;
                L.cvn1[0][r] = sv[r];
                L.cvn2[r] = vn2[c];
                a.inblk[prob * a.sIn + c] = a.blkid;
            } else if (r == K) {
                L.bval = sv[r];
                L.bpos = p;
            }
        }
    }
    __syncthreads();
    stamp(1);
    // ---- 3. candidate columns -> registers: slot k = w + NWV cc, local row ln + 64 i (row j0 + .. of M) --------
    double x[NCW][RPL];
#pragma unroll
    for (int cc = 0; cc < NCW; ++cc) {
        const int k = w + NWV * cc;
        const int c = (k < K) ? L.ccol[k] : 0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            x[cc][i] = (k < K && r < rows) ? M[(j0 + r) + (size_t)c * a.q.ldr] : 0.0;
        }
    }
    // value of local row r < 64 (wave-uniform) of column x[cc]: a block makes at most 64 steps, so its pivot rows all
    // sit in register 0
    auto row_of = [&](const double (&xc)[RPL], int r) -> double { return wave_bcast(xc[0], r); };
    stamp(2);
    // ---- 4. pivot steps -------------------------------------------------------------------------------------
    const int smax = (a.Tsb != nullptr) ? 32 - (j0 & 1) : SB_KMAX;   // the blocked update applies <= 32 reflectors at once
    // The step loop is written in rotated form — the pivot search and the certainty test of step s + 1 close iteration s, the
    // only exit is at the bottom — because a `break` out of the middle makes the compiler (whose structurizer turns every
    // loop with lane-dependent branches inside into a single-exit loop) carry the exit copy AND the next-iteration copy of
    // the whole candidate block through the loop: two register sets, ~200 v_mov_b64 per step.
    // every wave finds the pivot among the active candidates (identical result in all waves)
    auto search = [&](int sx, int lnx) -> ArgMax {
        const int rdx = sx & 1;
        const int cp = L.cpos[rdx][lnx];               // both words unconditionally (lnx < 64 = SB_KMAX): one LDS round trip
        const double cv = L.cvn1[rdx][lnx];
        const bool in = lnx < K && cp >= 0;
        return wave_argmax(in ? pivot_key(cv) : -1.0, in ? cp : 0x7fffffff, in ? lnx : -1);
    };
    // certain iff it beats every column left outside (their current norms are <= bval); the first
    // step of a block is always certain: rank 0 is the global maximum
    auto certain = [&](const ArgMax& m, int sx) -> bool { return m.idx >= 0 && (sx == 0 || m.val > uniform_f64(L.bval)); };
    int s = 0;
    ArgMax am = {-1.0, 0x7fffffff, -1};
    bool go = j0 < kp && smax > 0;
    if (go) {
        am = search(0, ln);
        go = certain(am, 0);
    }
#ifdef ENLSIP_SB_STEP_STAMPS
#ifndef ENLSIP_SB_STAMP_WAVE
#define ENLSIP_SB_STAMP_WAVE 0
#endif
    const bool sb_st = tid == 64 * ENLSIP_SB_STAMP_WAVE && prob == a.q.prob0;
    long long sb_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long sb_t = wall_clock64();
#endif
    while (go) {
        // opaque per-iteration copies of the lane / wave ids: without them loop-invariant code motion hoists every row
        // mask and LDS address of the unrolled columns out of the loop and spills them
        int lnl = ln, wl = w;
        asm volatile("" : "+v"(lnl));
        asm volatile("" : "+s"(wl));
        const int j = j0 + s;
        const int rd = s & 1, wr = rd ^ 1;
        const int ci = am.idx, q = am.pos;
        // the owner of the pivot column builds the reflector (rows > s of the column; pivot entry in local row s).
        // It only READS its registers here: the retired column takes its reflector / beta in the apply loop below, where
        // every x[cc] is modified in place anyway — a conditional write here makes the compiler carry a second copy of the
        // whole candidate block through the step loop (~290 v_mov_b64 per step).  s < 64: rows of registers i >= 1 are
        // all below the pivot row, no masks there.
        if (wl == ci % NWV) {
#pragma unroll
            for (int cc = 0; cc < NCW; ++cc) {
                if (cc == ci / NWV) {
                    double xn2 = (lnl > s) ? x[cc][0] * x[cc][0] : 0.0;
#pragma unroll
                    for (int i = 1; i < RPL; ++i) xn2 += x[cc][i] * x[cc][i];
                    xn2 = wave_allsum(xn2);
                    const double alpha = row_of(x[cc], s);
                    const Reflector h = make_reflector(alpha, xn2);
                    L.vsh[rd][lnl] = (lnl > s) ? x[cc][0] * h.scale : (lnl == s ? 1.0 : 0.0);
#pragma unroll
                    for (int i = 1; i < RPL; ++i) L.vsh[rd][lnl + 64 * i] = x[cc][i] * h.scale;
                    if (lnl == 0) {
                        L.taul[ci] = h.tau;
                        L.betal[ci] = h.beta;
                        L.tau_s[rd] = h.tau;
                        if (s < 32) {
                            L.tslot[s] = ci;
                            L.taustep[s] = h.tau;
                        }
                    }
                }
            }
        }
        SB_PH(0);       // owner's part (wave 0 owns every 8th pivot)
        __syncthreads();
        SB_PH(1);       // wait for the owner
        double v[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = L.vsh[rd][lnl + 64 * i];
        // everything else this step reads from LDS is fetched here too, in one round trip (the column loop below stores Gram
        // entries, which would otherwise keep each column's position read behind the previous column's store): positions
        // of the wave's columns, the owner's beta, the norms of the downdate (lane u < NCW <-> column u)
        int pkv[NCW];
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) pkv[cc] = L.cpos[rd][wl + NWV * cc];
        const double beta_o = L.betal[ci];
        const int u = lnl & (NCW - 1);
        const int ku = wl + NWV * u;
        const double o1_in = L.cvn1[rd][ku], o2_in = L.cvn2[ku];
        const int pk_u = L.cpos[rd][ku];
        const double tj = uniform_f64(L.tau_s[rd]);
        double dot[NCW], ds[NCW];
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) {
            dot[cc] = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) dot[cc] += x[cc][i] * v[i];
        }
        wave_allsumN(dot, ds);
        SB_PH(2);       // LDS reads, dot products, reduction
        // apply the reflector to the active candidates, WITHOUT branches per column: a column that is not live (slot beyond K,
        // the pivot itself, retired earlier) gets the factor 0 — 56 independent FMAs the scheduler can interleave instead of
        // eight compare / branch / multiply / update / broadcast sequences one behind the other (the column loop was the
        // longest phase of a step: 0.93 of 3.1 us)
        double wdv[NCW];
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) {
            const int k = wl + NWV * cc;
            const bool live = (k < K) && (k != ci) && (pkv[cc] >= 0) && (tj != 0.0);
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
        // lane u < NCW <-> column u of this wave: its entry in the pivot row (norm downdate) and its dot product — for a column
        // retired earlier in this block (step aa = -1 - position - j0 < s) that is v_aa' v_s, a Gram entry of the T factor
        double ajc = 0.0, dsu = 0.0;
#pragma unroll
        for (int cc = 0; cc < NCW; ++cc) {
            const double rv = row_of(x[cc], s);
            ajc = (u == cc) ? rv : ajc;
            dsu = (u == cc) ? ds[cc] : dsu;
        }
        const bool mine_col = (lnl < NCW) && (ku < K) && (ku != ci);
        if (mine_col && pk_u < 0) {
            const int aa = (-1 - pk_u) - j0;
            if (aa >= 0 && aa < 32 && s < 32) L.gram[aa * 33 + s] = dsu;
        }
        SB_PH(3);       // column loop: updates, pivot-row entries, Gram entries
        // dlaqp2 norm downdate of the wave's active columns as ONE instruction stream: lane u < NCW <-> column u
        {
            const bool mine = mine_col && pk_u >= 0;
            double o1 = mine ? o1_in : 0.0;
            const double o2 = mine ? o2_in : 1.0;
            bool need = false;
            if (mine && o1 != 0.0) {
                const NormDown nd = norm_downdate(o1, o2, ajc, tol3z);
                if (nd.need) need = true;
                else o1 = nd.vn1;
            }
            unsigned nm = (unsigned)(__ballot(need) & ((1ull << NCW) - 1ull));
            while (nm) {                                   // rare: recompute the partial norm from the column
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
                    o1 = (j + 1 < kp) ? sqrt(sq) : 0.0;
                    L.cvn2[ku] = o1;
                }
            }
            if (mine) L.cvn1[wr][ku] = o1;
        }
        SB_PH(4);       // norm downdate
        // position bookkeeping (wave 0): pivot at position q <-> column cj that sat at position j.  Wave 0 has the time: the two
        // waves of a SIMD share its issue slots and the older one (waves 0..3) has priority, so waves 4..7 reach the barrier
        // ~0.5 us later anyway (phase stamps per wave, tests/probes/sb_step_probe.py); done by the owner's neighbour under the owner's
        // reflector construction instead, the step takes the same time.
        if (wl == 0) {
            const int pc = L.ccol[ci];
            const int cj = L.colat_l[j];
            const bool match = (lnl < K) && (L.ccol[lnl] == cj) && (cj != pc);
            const unsigned long long mb = __ballot(match);
            const int slot = mb ? (__ffsll((long long)mb) - 1) : -1;
            if (lnl < K) {
                const int old = L.cpos[rd][lnl];
                L.cpos[wr][lnl] = (lnl == ci) ? (-1 - j) : ((lnl == slot) ? q : old);
            }
            if (lnl == 0) {
                L.colat_l[j] = pc;
                L.colat_l[q] = cj;
                L.pos_l[cj] = q;
                L.pos_l[pc] = j;
            }
        }
        SB_PH(5);       // bookkeeping
        __syncthreads();
        SB_PH(6);       // wait for the slowest wave
        ++s;
        go = (j0 + s < kp) && (s < smax);
        if (go) {
            am = search(s, lnl);
            go = certain(am, s);
        }
        SB_PH(7);       // pivot search of the next step
    }
#ifdef ENLSIP_SB_STEP_STAMPS
    if (sb_st) {
        for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&g_sb_phase[i], (unsigned long long)sb_acc[i]);
        atomicAdd((unsigned long long*)&g_sb_phase[8], (unsigned long long)s);
    }
#endif
    const int fin = s & 1;   // buffer holding the state after the last completed step
    stamp(3);
    // ---- 4b. T factor (dlarft, forward / columnwise) from the Gram entries gathered during the steps -------------
    if (a.Tsb != nullptr && w == 0) {
        double* T = a.Tsb + prob * a.sTsb;
        if (ln < 32) {
            // row ln of T in registers, fully unrolled and branch-free (T upper triangular: trow[l] = 0 for l < ln).
            // Gram entries exist only for pairs below s; everything else is discarded by the selects (the Gram image
            // was also the scratch of the ranking sort).
            double trow[32];
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                const bool on = b < s;
                const double tb = on ? L.taustep[b] : 0.0;
                double ac[4] = {0.0, 0.0, 0.0, 0.0};         // four chains: the sum is latency bound, not FMA bound
#pragma unroll
                for (int l = 0; l < b; ++l) ac[l & 3] += trow[l] * L.gram[l * 33 + b];
                const double acc = (ac[0] + ac[1]) + (ac[2] + ac[3]);
                const double tv = on ? ((ln == b) ? tb : ((ln < b) ? -tb * acc : 0.0)) : 0.0;
                trow[b] = tv;
                T[ln + b * 32] = tv;
            }
        }
    }
    stamp(4);
    if (a.dbg && tid == 0 && prob == a.q.prob0) a.dbg[a.blkid * 8 + 5] = s;
    // ---- 5. write back --------------------------------------------------------------------------------------
#pragma unroll
    for (int cc = 0; cc < NCW; ++cc) {
        const int k = w + NWV * cc;
        if (k >= K) continue;
        const int c = L.ccol[k];
        const int pk = L.cpos[fin][k];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            if (r < rows) M[(j0 + r) + (size_t)c * a.q.ldr] = x[cc][i];
        }
        if (pk < 0) {   // retired at step jr: rows below the diagonal hold the reflector
            const int jr = -1 - pk;
            double* Vb = a.q.Vb + prob * a.q.sVb;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = ln + 64 * i;
                if (r > jr - j0 && r < rows) Vb[(j0 + r) + (size_t)jr * a.q.ldr] = x[cc][i];
            }
            if (ln == 0) {
                a.q.diag[prob * a.q.sDiag + jr] = L.betal[k];
                a.q.tau[prob * a.q.sTau + jr] = L.taul[k];
                chosen[c] = jr;
            }
        } else if (ln == 0) {
            vn1[c] = L.cvn1[fin][k];
            vn2[c] = L.cvn2[k];
        }
    }
    for (int c = tid; c < n2; c += NT) {
        pos[c] = L.pos_l[c];
        colat[c] = L.colat_l[c];
    }
    // compact list of the columns the block update has to touch (see the first form)
    int nact = 0;
    if (a.act != nullptr) {
        int* act = a.act + prob * a.sAct;
        const int per = (n2 + NWV - 1) / NWV;
        int cnt = 0;
        for (int c0 = w * per; c0 < (w + 1) * per && c0 < n2; c0 += WAVE) {
            const int c = c0 + ln;
            const bool f = (c < (w + 1) * per) && (c < n2) && (L.posk[c] != 0x7fffffff) && (L.rankl[c] >= K);
            cnt += __popcll(__ballot(f));
        }
        if (ln == 0) L.wtot[w] = cnt;
        __syncthreads();
        int off = 0;
        for (int qq = 0; qq < w; ++qq) off += L.wtot[qq];
        for (int qq = 0; qq < NWV; ++qq) nact += L.wtot[qq];
        for (int c0 = w * per; c0 < (w + 1) * per && c0 < n2; c0 += WAVE) {
            const int c = c0 + ln;
            const bool f = (c < (w + 1) * per) && (c < n2) && (L.posk[c] != 0x7fffffff) && (L.rankl[c] >= K);
            const unsigned long long bal = __ballot(f);
            if (f) act[off + __popcll(bal & ((1ull << ln) - 1ull))] = c;
            off += __popcll(bal);
        }
        if (tid == 0) act[nact] = n2;
        nact += 1;
    }
    if (tid == 0) {
        SbInfo o = {j0 + s, s, a.blkid, j0, nact, {0, 0, 0}};
        *info = o;
    }
#ifdef ENLSIP_SB_STEP_STAMPS
    stamp(5);
    if (tid == 0 && prob == a.q.prob0) atomicAdd((unsigned long long*)&g_sb_blk[7], 1ull);
#endif
}

}  // namespace gn
