// Column-pivoted QR (dgeqp3 pivot rule, dlaqp2 arithmetic) of the kp x n2 factor R0 in BLOCKS
// of pivot steps with exact pivots ("speculate on the candidates, verify against a bound").
//
// A pivot step needs the partial norms of ALL trailing columns, which is what makes dgeqp3 half
// BLAS-2.  But partial norms only decrease.  So at the start of a block (all norms exact):
//   * the K columns with the largest norms (ties -> lowest LAPACK position) are pulled into the
//     LDS of ONE workgroup, and B := the best norm among the columns left outside is recorded;
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
constexpr int SB_SLAB_DOUBLES = 14336;       // 112 KB of candidate columns
constexpr int SB_NMAX = 1024;

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
};

// LDS carve of the select/factor kernel
struct SbLds {
    double slab[SB_SLAB_DOUBLES];
    double valk[SB_NMAX];
    int posk[SB_NMAX];
    int pos_l[SB_NMAX];
    int colat_l[SB_NMAX];
    double cvn1[2][SB_KMAX], cvn2[SB_KMAX];   // cvn1 / cpos are double-buffered by step parity: a step
    int ccol[SB_KMAX], cpos[2][SB_KMAX];      // reads [s & 1] and writes [(s + 1) & 1] -> one barrier per step
    int rankl[SB_NMAX];
    double taul[SB_KMAX], betal[SB_KMAX];     // per candidate slot: tau / beta of the step that retired it
    double gram[32 * 33];                     // v_a' v_b of the block's reflectors (a < b)
    double tmat[32 * 33];                     // T factor, row a in lane a
    int tslot[32];                            // candidate slot of the block's t-th reflector
    double bval;
    int bpos;
    int K;
};

// reset of the per-problem block state; runs after k_qd_init
__global__ void k_sb_reset(SbArgs a, int n) {
    const int prob = blockIdx.y + a.q.prob0;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) a.inblk[prob * a.sIn + c] = -1;
    if (c == 0) {
        SbInfo z = {0, 0, -1, 0};
        a.info[prob] = z;
    }
}

// ---------------------------------------------------------------------------------------------
// select the candidates of the block and run pivot steps on them while the choice is certain
// ---------------------------------------------------------------------------------------------
template <int RPL>
__global__ __launch_bounds__(1024) void k_sb_factor(SbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_raw[];
    SbLds& L = *reinterpret_cast<SbLds*>(sb_raw);
    const int prob = blockIdx.x + a.q.prob0;
    const ProbState st = a.q.state[prob];
    if (a.q.n2cap > 0 && st.n2 > a.q.n2cap) return;      // wider than the launch shape: redone by the caller
    const int kp = st.kp, n2 = st.n2;
    SbInfo* info = a.info + prob;
    const int j0 = info->j0;
    if (j0 >= kp) return;
    const int tid = threadIdx.x, ln = lane_id(), w = wave_id();
    const int rows = kp - j0;
    const int ldk = (rows + 1) & ~1;
    double* M = a.q.M + prob * a.q.sM;
    double* vn1 = a.q.vn1 + prob * a.q.sVn;
    double* vn2 = a.q.vn2 + prob * a.q.sVn;
    int* chosen = a.q.chosen + prob * a.q.sI;
    int* pos = a.q.pos + prob * 2 * a.q.sI;       // parity-0 arrays only
    int* colat = a.q.colat + prob * 2 * a.q.sI;
    const double tol3z = 1.4901161193847656e-08;

    auto stamp = [&](int i) {
        if (a.dbg && tid == 0 && prob == a.q.prob0) a.dbg[a.blkid * 8 + i] = (long long)__builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    // ---- 1. keys of the trailing columns --------------------------------------------------------
    for (int c = tid; c < n2; c += 1024) {
        const bool tr = chosen[c] < 0;
        L.valk[c] = tr ? pivot_key(vn1[c]) : -1.0;      // NaN norms rank as +inf: the order stays total, every slot < K is filled
        const int p = pos[c];
        L.posk[c] = tr ? p : 0x7fffffff;
        L.pos_l[c] = p;
        L.colat_l[c] = colat[c];
    }
    const int ntrail = n2 - j0;
    int K = SB_SLAB_DOUBLES / ldk;
    K = K < SB_KMAX ? K : SB_KMAX;
    K = K < ntrail ? K : ntrail;
    if (tid == 0) {
        L.K = K;
        L.bval = -1.0;     // no outside column unless a rank == K exists
        L.bpos = 0x7fffffff;
    }
    __syncthreads();
    // ---- 2. rank by (norm desc, position asc); ranks < K are the candidates, rank K is the bound
    // (n2 <= 512: two threads per column, each counting over half of the columns)
    for (int c = tid; c < n2; c += 1024) L.rankl[c] = 0;
    __syncthreads();
    {
        const int parts = (n2 <= 512) ? 2 : 1;
        const int per = (n2 + parts - 1) / parts;
        for (int e = tid; e < n2 * parts; e += 1024) {
            const int c = (parts == 2) ? (e >> 1) : e;
            const int part = (parts == 2) ? (e & 1) : 0;
            const double v = L.valk[c];
            const int p = L.posk[c];
            if (p == 0x7fffffff) continue;
            int rank = 0;
            const int k1 = (part + 1) * per < n2 ? (part + 1) * per : n2;
            for (int k = part * per; k < k1; ++k) {
                const double vk = L.valk[k];
                const int pk = L.posk[k];
                rank += (pk != 0x7fffffff && (vk > v || (vk == v && pk < p))) ? 1 : 0;
            }
            if (rank) atomicAdd(&L.rankl[c], rank);
        }
    }
    __syncthreads();
    for (int c = tid; c < n2; c += 1024) {
        const int p = L.posk[c];
        if (p == 0x7fffffff) continue;
        const int rank = L.rankl[c];
        if (rank < K) {
            L.ccol[rank] = c;
            L.cpos[0][rank] = p;
            L.cvn1[0][rank] = L.valk[c];
            L.cvn2[rank] = vn2[c];
            a.inblk[prob * a.sIn + c] = a.blkid;
        } else if (rank == K) {
            L.bval = L.valk[c];
            L.bpos = p;
        }
    }
    __syncthreads();
    stamp(1);
    // ---- 3. candidate columns -> LDS (rows j0 .. kp-1) ---------------------------------------------
    for (int k = w; k < K; k += 16) {
        const int c = L.ccol[k];
        for (int r = ln; r < rows; r += WAVE) L.slab[(size_t)k * ldk + r] = M[(j0 + r) + (size_t)c * a.q.ldr];
    }
    __syncthreads();
    stamp(2);
    // ---- 4. pivot steps (one barrier per step) -------------------------------------------------------
    int s = 0;
    int pend = -1;        // candidate slot retired by this wave in the previous step
    double v[RPL];
    for (;; ++s) {
        const int j = j0 + s;
        if (j >= kp) break;
        if (a.Tsb != nullptr && s >= 32 - (j0 & 1)) break;   // the blocked update applies <= 32 reflectors whose R rows fit its first (16-byte aligned) 32-row unit
        const int rd = s & 1, wr = rd ^ 1;
        if (pend >= 0) {   // reflector of the previous step -> LDS column of the candidate it retired
            double* colp = L.slab + (size_t)pend * ldk;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = (s - 1) + ln + 64 * i;
                if (r < rows && ln + 64 * i > 0) colp[r] = v[i];
            }
            pend = -1;
        }
        // every wave finds the pivot among the active candidates (identical result in all waves)
        double bv = -1.0;
        int bp = 0x7fffffff, bk = -1;
        if (ln < K && L.cpos[rd][ln] >= 0) {
            bv = L.cvn1[rd][ln];
            bp = L.cpos[rd][ln];
            bk = ln;
        }
        const ArgMax am = wave_argmax(bv, bp, bk);
        // certain iff it beats every column left outside (their current norms are <= bval); the first
        // step of a block is always certain: rank 0 is the global maximum
        if (!(am.idx >= 0 && (s == 0 || am.val > L.bval))) break;
        const int ci = am.idx, q = am.pos;
        // reflector from the pivot candidate (every wave, identical bits)
        double xn2 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = s + ln + 64 * i;
            v[i] = (r < rows) ? L.slab[(size_t)ci * ldk + r] : 0.0;
            if (ln + 64 * i > 0) xn2 += v[i] * v[i];
        }
        xn2 = wave_allsum(xn2);
        const double alpha = wave_bcast(v[0], 0);
        const Reflector h = make_reflector(alpha, xn2);
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = (ln + 64 * i == 0) ? 1.0 : v[i] * h.scale;
        for (int k = w; k < K; k += 16) {
            double* col = L.slab + (size_t)k * ldk;
            if (k == ci) {
                // retire.  No global store inside the step loop (a pending store would be drained by
                // every barrier): the reflector stays in this wave's v[] registers and is written
                // over the column's LDS copy after the barrier, when nobody reads that column any more.
                if (ln == 0) {
                    L.taul[k] = h.tau;
                    L.betal[k] = h.beta;
                }
                pend = k;
            } else if (L.cpos[rd][k] >= 0) {
                double x[RPL];
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const int r = s + ln + 64 * i;
                    x[i] = (r < rows) ? col[r] : 0.0;
                    dot += x[i] * v[i];
                }
                dot = wave_allsum(dot);
                if (h.tau != 0.0) {
                    const double wd = h.tau * dot;
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        const int r = s + ln + 64 * i;
                        x[i] -= wd * v[i];
                        if (r < rows) col[r] = x[i];
                    }
                }
                const double ajc = wave_bcast(x[0], 0);
                double o1 = L.cvn1[rd][k];
                const double o2 = L.cvn2[k];
                if (o1 != 0.0) {
                    double temp = 1.0 - (fabs(ajc) / o1) * (fabs(ajc) / o1);
                    temp = temp > 0.0 ? temp : 0.0;
                    const double qq = o1 / o2;
                    const double temp2 = temp * qq * qq;
                    if (temp2 <= tol3z) {
                        double sq = 0.0;
#pragma unroll
                        for (int i = 0; i < RPL; ++i)
                            if (ln + 64 * i > 0) sq += x[i] * x[i];
                        sq = wave_allsum(sq);
                        o1 = (j + 1 < kp) ? sqrt(sq) : 0.0;
                        if (ln == 0) L.cvn2[k] = o1;
                    } else {
                        o1 = o1 * sqrt(temp);
                    }
                }
                if (ln == 0) L.cvn1[wr][k] = o1;
            }
        }
        // position bookkeeping (wave 0): pivot at position q <-> column cj that sat at position j
        if (w == 0) {
            const int pc = L.ccol[ci];
            const int cj = L.colat_l[j];
            const bool match = (ln < K) && (L.ccol[ln] == cj) && (cj != pc);
            const unsigned long long mb = __ballot(match);
            const int slot = mb ? (__ffsll((long long)mb) - 1) : -1;
            if (ln < K) {
                const int old = L.cpos[rd][ln];
                L.cpos[wr][ln] = (ln == ci) ? (-1 - j) : ((ln == slot) ? q : old);
            }
            if (ln == 0) {
                L.colat_l[j] = pc;
                L.colat_l[q] = cj;
                L.pos_l[cj] = q;
                L.pos_l[pc] = j;
            }
        }
        __syncthreads();
    }
    const int fin = s & 1;   // buffer holding the state after the last completed step
    stamp(3);
    if (pend >= 0) {
        double* colp = L.slab + (size_t)pend * ldk;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = (s - 1) + ln + 64 * i;
            if (r < rows && ln + 64 * i > 0) colp[r] = v[i];
        }
    }
    __syncthreads();
    // ---- 4b. T factor (dlarft, forward / columnwise) of the block's s <= 32 reflectors, for the blocked
    // MFMA update of the outside columns (k_sb_update_blk).  Reflector t lives in the LDS column of the
    // candidate it retired: rows t + 1 .. rows - 1 (block-local), implicit one at row t.
    if (a.Tsb != nullptr && s <= 32) {
        if (tid < K) {
            const int pk = L.cpos[fin][tid];
            if (pk < 0) {
                const int t = (-1 - pk) - j0;
                if (t >= 0 && t < s) L.tslot[t] = tid;
            }
        }
        __syncthreads();
        for (int b = w; b < s; b += 16) {
            const double* cb = L.slab + (size_t)L.tslot[b] * ldk;
            for (int aa = 0; aa < b; ++aa) {
                const double* ca = L.slab + (size_t)L.tslot[aa] * ldk;
                double dot = 0.0;
                for (int r = b + 1 + ln; r < rows; r += WAVE) dot += ca[r] * cb[r];
                dot = wave_allsum(dot);
                if (ln == 0) L.gram[aa * 33 + b] = dot + ca[b];
            }
        }
        __syncthreads();
        if (w == 0) {
            double* T = a.Tsb + prob * a.sTsb;
            if (ln < 32) {
                double* trow = L.tmat + ln * 33;
                for (int b = 0; b < 32; ++b) {
                    double tv = 0.0;
                    if (b < s && ln <= b) {
                        const double tb = L.taul[L.tslot[b]];
                        if (ln == b) tv = tb;
                        else {
                            double acc = 0.0;
                            for (int l = ln; l < b; ++l) acc += trow[l] * L.gram[l * 33 + b];
                            tv = -tb * acc;
                        }
                    }
                    trow[b] = tv;
                    T[ln + b * 32] = tv;
                }
            }
        }
    }
    // ---- 5. write back --------------------------------------------------------------------------------
    for (int k = w; k < K; k += 16) {
        const int c = L.ccol[k];
        const int pk = L.cpos[fin][k];
        for (int r = ln; r < rows; r += WAVE) M[(j0 + r) + (size_t)c * a.q.ldr] = L.slab[(size_t)k * ldk + r];
        if (pk < 0) {   // retired at step jr: rows below the diagonal hold the reflector
            const int jr = -1 - pk;
            double* Vb = a.q.Vb + prob * a.q.sVb;
            for (int r = (jr - j0) + 1 + ln; r < rows; r += WAVE) Vb[(j0 + r) + (size_t)jr * a.q.ldr] = L.slab[(size_t)k * ldk + r];
            if (ln == 0) {
                a.q.diag[prob * a.q.sDiag + jr] = L.betal[k];
                a.q.tau[prob * a.q.sTau + jr] = L.taul[k];
            }
        }
        if (ln == 0) {
            const int p = pk;
            if (p < 0) {
                chosen[c] = -1 - p;
            } else {
                vn1[c] = L.cvn1[fin][k];
                vn2[c] = L.cvn2[k];
            }
        }
    }
    for (int c = tid; c < n2; c += 1024) {
        pos[c] = L.pos_l[c];
        colat[c] = L.colat_l[c];
    }
    stamp(4);
    if (a.dbg && tid == 0 && prob == a.q.prob0) a.dbg[a.blkid * 8 + 5] = s;
    // compact list of the columns the block update has to touch: trailing at block start and not a candidate of
    // this block (candidates received every reflector here), followed by the carried right-hand side
    int nact = 0;
    if (a.act != nullptr) {
        int* act = a.act + prob * a.sAct;
        __shared__ int wtot[16];
        const int per = (n2 + 15) / 16;                     // columns per wave, processed 64 at a time
        int cnt = 0;
        for (int c0 = w * per; c0 < (w + 1) * per && c0 < n2; c0 += WAVE) {
            const int c = c0 + ln;
            const bool f = (c < (w + 1) * per) && (c < n2) && (L.posk[c] != 0x7fffffff) && (L.rankl[c] >= K);
            cnt += __popcll(__ballot(f));
        }
        if (ln == 0) wtot[w] = cnt;
        __syncthreads();
        int off = 0;
        for (int q = 0; q < w; ++q) off += wtot[q];
        for (int q = 0; q < 16; ++q) nact += wtot[q];
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
        SbInfo o = {j0 + s, s, a.blkid, j0, nact, {0, 0, 0}};   // pad carries the block's first step for the update kernel
        *info = o;
    }
}

// ---------------------------------------------------------------------------------------------
// apply the block's reflectors to every column that was not a candidate (+ the right-hand side).
// A wave owns 8 columns: the 8 dot products share one transposed reduction, and the scalar norm
// downdates of the 8 columns run as ONE instruction stream in lanes 0..7 (lane u = column u).
// ---------------------------------------------------------------------------------------------
constexpr int SB_UCW = 32;   // columns per workgroup of the update kernel (8 per wave)

template <int RPL>
__global__ __launch_bounds__(256) void k_sb_update(SbArgs a) {
    const int prob = blockIdx.y + a.q.prob0;
    const ProbState st = a.q.state[prob];
    if (a.q.n2cap > 0 && st.n2 > a.q.n2cap) return;      // wider than the launch shape: redone by the caller
    const int kp = st.kp, n2 = st.n2, ctot = n2 + 1;
    const SbInfo info = a.info[prob];
    if (info.blk != a.blkid || info.s == 0) return;     // this problem did no step in this block
    const int jb = info.pad, s = info.s;
    const int g = blockIdx.x;
    if (g * SB_UCW >= ctot) return;
    const int ln = lane_id(), w = wave_id();
    double* M = a.q.M + prob * a.q.sM;
    const double* Vb = a.q.Vb + prob * a.q.sVb;
    const double* tau = a.q.tau + prob * a.q.sTau;
    double* vn1 = a.q.vn1 + prob * a.q.sVn;
    double* vn2 = a.q.vn2 + prob * a.q.sVn;
    const int* chosen = a.q.chosen + prob * a.q.sI;
    const int* inblk = a.inblk + prob * a.sIn;
    const double tol3z = 1.4901161193847656e-08;
    const int rows = kp - jb;
    const int cbase = g * SB_UCW + 8 * w;

    // lane u (< 8) carries the scalar state of column cbase + u
    const int myc = cbase + (ln & 7);
    const bool my_act = (myc == n2) || (myc < n2 && chosen[myc] < 0 && inblk[myc] != a.blkid);
    double my_o1 = (my_act && myc < n2) ? vn1[myc] : 0.0;
    double my_o2 = (my_act && myc < n2) ? vn2[myc] : 1.0;
    const unsigned actmask = (unsigned)(__ballot(my_act && ln < 8) & 0xffull);
    if (actmask == 0u) return;

    double x[8][RPL];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const bool act = (actmask >> u) & 1u;
        const int c = cbase + u;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            x[u][i] = (act && r < rows) ? M[(jb + r) + (size_t)c * a.q.ldr] : 0.0;
        }
    }
    auto load_v = [&](int t, double* v) {
        // reflector of step jb + t in block-local rows: zero above t, one at t
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            v[i] = (r > t && r < rows) ? Vb[(jb + r) + (size_t)(jb + t) * a.q.ldr] : (r == t ? 1.0 : 0.0);
        }
    };
    double v[RPL], vn[RPL];
    load_v(0, v);
    for (int t = 0; t < s; ++t) {
        if (t + 1 < s) load_v(t + 1, vn);
        const double tj = tau[jb + t];
        double dot[8], ds[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            dot[u] = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) dot[u] += x[u][i] * v[i];
        }
        wave_allsum8(dot, ds);
        if (tj != 0.0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double wd = tj * ds[u];
#pragma unroll
                for (int i = 0; i < RPL; ++i) x[u][i] -= wd * v[i];
            }
        }
        // entry of row jb + t of "my" column: local row t lives in lane t & 63 of register t >> 6
        // (one wave-uniform branch selects the register, then 8 readlanes)
        double ajc = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            if ((t >> 6) == i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double au = wave_bcast(x[u][i], t & 63);
                    if ((ln & 7) == u) ajc = au;
                }
            }
        }
        // norm downdate of the 8 columns, one column per lane (dlaqp2 rule)
        bool need = false;
        if (my_o1 != 0.0) {
            double temp = 1.0 - (fabs(ajc) / my_o1) * (fabs(ajc) / my_o1);
            temp = temp > 0.0 ? temp : 0.0;
            const double qq = my_o1 / my_o2;
            const double temp2 = temp * qq * qq;
            if (temp2 <= tol3z) need = true;
            else my_o1 = my_o1 * sqrt(temp);
        }
        unsigned nm = (unsigned)(__ballot(need && ln < 8) & 0xffull);
        while (nm) {   // rare: recompute the partial norm of a column from its entries
            const int u = __ffs((int)nm) - 1;
            nm &= nm - 1;
            double sq = 0.0;
#pragma unroll
            for (int uu = 0; uu < 8; ++uu)
                if (uu == u) {
#pragma unroll
                    for (int i = 0; i < RPL; ++i)
                        if (ln + 64 * i > t) sq += x[uu][i] * x[uu][i];
                }
            sq = wave_allsum(sq);
            if ((ln & 7) == u) {
                my_o1 = (jb + t + 1 < kp) ? sqrt(sq) : 0.0;
                my_o2 = my_o1;
            }
        }
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = vn[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (!((actmask >> u) & 1u)) continue;
        const int c = cbase + u;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            if (r < rows) M[(jb + r) + (size_t)c * a.q.ldr] = x[u][i];
        }
    }
    if (ln < 8 && my_act && myc < n2) {
        vn1[myc] = my_o1;
        vn2[myc] = my_o2;
    }
}

// ---------------------------------------------------------------------------------------------
// Blocked form of the update above (kp <= 512): the block's s reflectors are applied to 32 outside
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
    c.tile_row0 = jb - d; c.r0 = jb - d; c.gblk0 = 0; c.S = 0;
    c.ldw = a.q.ldr; c.col0 = jb; c.bw = s; c.cb0 = cb0;
    c.rows_valid = kp - (jb - d);
    c.dshift = d;
    const int nvu = (c.rows_valid + 31) / 32;
    const int ngw = nvu > w ? (nvu - w + 3) / 4 : 0;

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
    if (w != 0 || ln >= 32 || cb0 + ln >= nact) return;
    const int cc = actl[cb0 + ln];
    if (cc >= n2) return;                               // the carried right-hand side has no norm
    double* vn1 = a.q.vn1 + prob * a.q.sVn;
    double* vn2 = a.q.vn2 + prob * a.q.sVn;
    const double tol3z = 1.4901161193847656e-08;
    const double srest = (ssq[0][ln] + ssq[1][ln]) + (ssq[2][ln] + ssq[3][ln]);
    const double* col = stage[0] + ln * V4_LD + d;        // final entries of rows jb .. jb + 31 of this column
    double o1 = vn1[cc], o2 = vn2[cc];
    for (int t = 0; t < s; ++t) {
        if (o1 == 0.0) continue;
        const double ajc = col[t];
        double temp = 1.0 - (fabs(ajc) / o1) * (fabs(ajc) / o1);
        temp = temp > 0.0 ? temp : 0.0;
        const double qq = o1 / o2;
        const double temp2 = temp * qq * qq;
        if (temp2 <= tol3z) {
            double sq = srest;
            for (int t2 = s - 1; t2 > t; --t2) sq += col[t2] * col[t2];
            o1 = (jb + t + 1 < kp) ? sqrt(sq) : 0.0;
            o2 = o1;
        } else {
            o1 = o1 * sqrt(temp);
        }
    }
    vn1[cc] = o1;
    vn2[cc] = o2;
}

}  // namespace gn
