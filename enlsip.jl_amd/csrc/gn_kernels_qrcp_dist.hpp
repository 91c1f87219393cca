// Column-pivoted QR (dgeqp3 semantics) of the kp x n2 factor R0 when it is too large for one
// workgroup's LDS: the columns are distributed over G workgroups (QD_CPW physical columns each)
// and every pivot step is one launch over (G, batch).  Nothing is ever swapped physically:
//   * a column keeps its physical slot in M; its logical LAPACK position is tracked in pos[] /
//     colat[] (double-buffered by step parity, so a step only reads what the previous launch wrote);
//   * the reflector of step j is written to Vb[:, j] (position-indexed), never into M, so every
//     workgroup can read the pivot column while its owner retires it;
//   * each workgroup publishes its best remaining column (max partial norm, ties -> lowest
//     position = LAPACK's idamax rule) and the next launch reduces the G candidates redundantly.
// k_qd_assemble finally gathers Rt = [R above / beta on / V below the diagonal] in pivoted order.
// Norm downdate, recompute rule and reflector generation are those of oracle/lapack_semantics.py.
#pragma once
#include "gn_device_utils.hpp"

namespace gn {

constexpr int QD_CPW = 8;   // physical columns per workgroup (2 per wave)

struct QdCand {
    double val;
    int pos;
    int col;
};

struct QdArgs {
    int n, ldw, ldr, step;
    int prob0;
    const double* W;  long long sW;      // CAQR result: R0 in the upper triangle of the J2 columns, d in column n
    double* M;        long long sM;      // ldr x (n + 1) physical columns (+ rhs at index n2)
    double* Vb;       long long sVb;     // ldr x kpmax reflectors by position
    double* Rt;       long long sRt;     // ldr x (n + 1) assembled factors
    double* tau;      long long sTau;
    double* diag;     long long sDiag;
    double* vn1;      long long sVn;     // n each
    double* vn2;
    int* chosen;      long long sI;      // n
    int* pos;                            // 2 x n   (parity-major)
    int* colat;                          // 2 x n
    QdCand* cand;     long long sCand;   // 2 x Gmax
    int Gmax;
    long long* jpvt;  long long sJ;
    const ProbState* state;
    // General (tall) inputs — the constraint stage with many constraints (run_constraint_dist): the matrix has `rows` rows,
    // state.n2 columns (+ one carried column) and min(rows, n2) pivot steps.  rows = 0: the R0 case, rows = steps = state.kp.
    int n2cap;         // > 0: launch shape; a problem whose J2 turned out WIDER (rank-deficient A, state.n2 > n2cap) is skipped by
                       // every kernel of the stage: the grids do not cover its columns, so its maps would hold stale entries of
                       // earlier solves that the single-workgroup kernels would use as indices.  The caller redoes the batch.
    int rows;
    int in_mode;       // k_qd_init: 0 = R0 out of the CAQR storage W; 1 = dense Ain (rows x n2, ld ldain), carried column zero;
                       // 2 = lower trapezoid L[i][j] = (j <= i) ? Ain[j + i * ldain] : 0 (the transposed R of a factor),
                       //     carried column rin; the unfactored L is also kept in Lout (ld ldr)
    const double* Ain; long long ldain, sAin;
    const double* rin; long long sRin;
    double* Lout;      long long sLout;
    int hyb;           // k_qd_assemble after a hybrid stage (run_qrcp_block with a launch-per-step head of `hyb` steps, hyb even): a
                       // problem that finished inside the head keeps its maps in the parity of its step count, the others in 0
};

__device__ __forceinline__ int qd_rows(const QdArgs& a, int kp) { return a.rows > 0 ? a.rows : kp; }
__device__ __forceinline__ int qd_steps(const QdArgs& a, int kp, int n2) {
    return a.rows > 0 ? (a.rows < n2 ? a.rows : n2) : kp;
}

__device__ __forceinline__ bool qd_better(double v, int p, double bv, int bp) {
    return v > bv || (v == bv && p < bp);
}

// extract own columns, initial norms / permutation / candidate
template <int RPL>
__global__ __launch_bounds__(256) void k_qd_init(QdArgs a) {
    __shared__ double cval[QD_CPW];
    __shared__ int cpos[QD_CPW];
    const int prob = blockIdx.y + a.prob0;
    const ProbState st = a.state[prob];
    if (a.n2cap > 0 && st.n2 > a.n2cap) return;
    const int n2 = st.n2, ctot = n2 + 1;
    const int kp = qd_rows(a, st.kp);          // rows of the matrix
    const int g = blockIdx.x;
    if (kp == 0 || g * QD_CPW >= ctot) return;
    const int ln = lane_id(), w = wave_id();
    const double* W = a.W + prob * a.sW;
    double* M = a.M + prob * a.sM;
    for (int u = 0; u < 2; ++u) {
        const int slot = 2 * w + u;
        const int c = g * QD_CPW + slot;
        double s = 0.0;
        if (c < ctot && a.in_mode == 0) {
            // R0 out of the CAQR storage: UNCONDITIONAL loads (row clamped into the matrix) masked with AND afterwards — a guarded
            // load is compiled as a branch around the load with a wait right behind it, and the RPL loads of a column then are RPL
            // HBM round trips in a row (the copy ran at half the stream rate)
            const double* src = (c < n2) ? W + (size_t)(st.rankA + c) * a.ldw : W + (size_t)a.n * a.ldw;      // uniform
            double v[RPL];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = ln + 64 * i;
                // 64-row pieces that lie entirely below the diagonal of R0 are not fetched: their load re-reads piece 0 (a cache
                // hit; the choice is wave-uniform) and the mask discards it
                const int rs = (64 * i <= c || c >= n2) ? r : ln;
                v[i] = src[rs < kp ? rs : 0];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = ln + 64 * i;
                const bool keep = (r < kp) && (c >= n2 || r <= c);
                const unsigned long long mk = keep ? ~0ull : 0ull;
                v[i] = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v[i]) & mk));
                if (r < kp) M[r + (size_t)c * a.ldr] = v[i];
                s += v[i] * v[i];
            }
        } else if (c < ctot) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = ln + 64 * i;
                if (r < kp) {
                    double v;
                    if (a.in_mode == 1) {
                        v = (c < n2) ? a.Ain[prob * a.sAin + r + (size_t)c * a.ldain] : 0.0;
                    } else if (a.in_mode == 2) {
                        if (c < n2) {
                            v = (c <= r) ? a.Ain[prob * a.sAin + c + (size_t)r * a.ldain] : 0.0;
                            a.Lout[prob * a.sLout + r + (size_t)c * a.ldr] = v;
                        } else {
                            v = a.rin[prob * a.sRin + r];
                        }
                    } else if (c < n2) v = (r <= c) ? W[r + (size_t)(st.rankA + c) * a.ldw] : 0.0;
                    else v = W[r + (size_t)a.n * a.ldw];
                    M[r + (size_t)c * a.ldr] = v;
                    s += v * v;
                }
            }
        }
        s = wave_allsum(s);
        if (ln == 0) {
            if (c < n2) {
                const double nv = sqrt(s);
                a.vn1[prob * a.sVn + c] = nv;
                a.vn2[prob * a.sVn + c] = nv;
                a.chosen[prob * a.sI + c] = -1;
                a.pos[prob * 2 * a.sI + c] = c;
                a.colat[prob * 2 * a.sI + c] = c;
                cval[slot] = pivot_key(nv);
                cpos[slot] = c;
            } else {
                cval[slot] = -1.0;
                cpos[slot] = 0x7fffffff;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = -1.0;
        int bp = 0x7fffffff, bc = -1;
        for (int s = 0; s < QD_CPW; ++s)
            if (qd_better(cval[s], cpos[s], bv, bp)) {
                bv = cval[s];
                bp = cpos[s];
                bc = g * QD_CPW + s;
            }
        QdCand cd = {bv, bp, bc};
        a.cand[prob * a.sCand + g] = cd;   // parity 0
    }
}

template <int RPL>
__global__ __launch_bounds__(256) void k_qd_step(QdArgs a) {
    __shared__ double cval[QD_CPW];
    __shared__ int cpos[QD_CPW];
    const int prob = blockIdx.y + a.prob0;
    const ProbState st = a.state[prob];
    if (a.n2cap > 0 && st.n2 > a.n2cap) return;
    const int n2 = st.n2, ctot = n2 + 1;
    const int kp = qd_rows(a, st.kp);          // rows of the matrix (= number of steps in the R0 case)
    const int j = a.step;
    const int g = blockIdx.x;
    if (j >= qd_steps(a, st.kp, n2) || g * QD_CPW >= ctot) return;
    const int gact = (ctot + QD_CPW - 1) / QD_CPW;
    const int ln = lane_id(), w = wave_id();
    const int par = j & 1;
    double* M = a.M + prob * a.sM;
    double* vn1 = a.vn1 + prob * a.sVn;
    double* vn2 = a.vn2 + prob * a.sVn;
    int* chosen = a.chosen + prob * a.sI;
    const int* pos_old = a.pos + prob * 2 * a.sI + par * a.sI;
    int* pos_new = a.pos + prob * 2 * a.sI + (par ^ 1) * a.sI;
    const int* colat_old = a.colat + prob * 2 * a.sI + par * a.sI;
    int* colat_new = a.colat + prob * 2 * a.sI + (par ^ 1) * a.sI;
    const QdCand* cand_old = a.cand + prob * a.sCand + par * a.Gmax;
    QdCand* cand_new = a.cand + prob * a.sCand + (par ^ 1) * a.Gmax;
    const double tol3z = 1.4901161193847656e-08;

    // Everything that does not depend on the pivot is requested first, so the step costs two
    // dependent memory round trips (candidates -> pivot column) instead of four.
    const int cbase = g * QD_CPW + 2 * w;
    int cst[2], cps[2];
    double cv1[2], cv2[2];
    double x[2][RPL];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = cbase + u;
        const bool real = c < n2;
        cst[u] = real ? chosen[c] : -1;
        cps[u] = real ? pos_old[c] : 0x7fffffff;
        cv1[u] = real ? vn1[c] : 0.0;
        cv2[u] = real ? vn2[c] : 1.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = j + ln + 64 * i;
            x[u][i] = (c < ctot && r < kp) ? M[r + (size_t)c * a.ldr] : 0.0;
        }
    }
    const int cj = colat_old[j];
    // (a) winner of the previous launch's candidates (every wave, identical result)
    double bv = -1.0;
    int bp = 0x7fffffff, bc = -1;
    for (int e = ln; e < gact; e += WAVE) {
        const QdCand cd = cand_old[e];
        if (qd_better(cd.val, cd.pos, bv, bp)) {
            bv = cd.val;
            bp = cd.pos;
            bc = cd.col;
        }
    }
    {   // DPP butterflies + readlanes (wave_argmax: same (value, position) order as qd_better) instead of six ds_bpermute rounds
        const ArgMax am = wave_argmax(bv, bp, bc);
        bv = am.val;
        bp = am.pos;
        bc = am.idx;
    }
    if (bc < 0) return;        // no candidate (cannot happen while unchosen columns exist: NaN norms rank as +inf); every
                               // workgroup takes the same decision, so nobody indexes with -1
    const int p = bc;          // physical pivot column
    const int q = bp;          // its logical position
    // (b) reflector of the pivot column (rows j..kp-1)
    double v[RPL];
    double xn2 = 0.0;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int r = j + ln + 64 * i;
        v[i] = (r < kp) ? M[r + (size_t)p * a.ldr] : 0.0;
        if (ln + 64 * i > 0) xn2 += v[i] * v[i];
    }
    xn2 = wave_allsum(xn2);
    const double alpha = wave_bcast(v[0], 0);
    const Reflector h = make_reflector(alpha, xn2);
#pragma unroll
    for (int i = 0; i < RPL; ++i) v[i] = (ln + 64 * i == 0) ? 1.0 : v[i] * h.scale;

    // (c) own columns (both at once: the two reductions overlap)
    double dot[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        dot[u] = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) dot[u] += x[u][i] * v[i];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) dot[u] = wave_allsum(dot[u]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int slot = 2 * w + u;
        const int c = cbase + u;
        double candv = -1.0;
        int candp = 0x7fffffff;
        if (c < ctot) {
            if (c == p) {
                // retire the pivot column: reflector by position, beta, tau
                double* Vb = a.Vb + prob * a.sVb;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const int r = j + ln + 64 * i;
                    if (r < kp && ln + 64 * i > 0) Vb[r + (size_t)j * a.ldr] = v[i];
                }
                if (ln == 0) {
                    a.diag[prob * a.sDiag + j] = h.beta;
                    a.tau[prob * a.sTau + j] = h.tau;
                    chosen[p] = j;
                    pos_new[p] = j;
                }
            } else if (c == n2 || cst[u] < 0) {
                if (h.tau != 0.0) {
                    const double wd = h.tau * dot[u];
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        const int r = j + ln + 64 * i;
                        x[u][i] -= wd * v[i];
                        if (r < kp) M[r + (size_t)c * a.ldr] = x[u][i];
                    }
                }
                if (c < n2) {
                    const double ajc = wave_bcast(x[u][0], 0);
                    double o1 = cv1[u];
                    const double o2 = cv2[u];
                    if (o1 != 0.0) {
                        const NormDown nd = norm_downdate(o1, o2, ajc, tol3z);
                        if (nd.need) {
                            double s = 0.0;
#pragma unroll
                            for (int i = 0; i < RPL; ++i)
                                if (ln + 64 * i > 0) s += x[u][i] * x[u][i];
                            s = wave_allsum(s);
                            o1 = (j + 1 < kp) ? sqrt(s) : 0.0;
                            if (ln == 0) {
                                vn1[c] = o1;
                                vn2[c] = o1;
                            }
                        } else {
                            o1 = nd.vn1;
                            if (ln == 0) vn1[c] = o1;
                        }
                    }
                    const int np = (c == cj) ? q : cps[u];
                    if (ln == 0) pos_new[c] = np;
                    candv = pivot_key(o1);
                    candp = np;
                }
            } else if (ln == 0) {
                pos_new[c] = cps[u];   // already retired: keeps its position
            }
        }
        if (ln == 0) {
            cval[slot] = candv;
            cpos[slot] = candp;
        }
    }
    // (d) position -> column map for the positions this workgroup administers
    if (threadIdx.x < QD_CPW) {
        const int k = g * QD_CPW + threadIdx.x;
        if (k < n2) colat_new[k] = (k == j) ? p : ((k == q) ? cj : colat_old[k]);
    }
    __syncthreads();
    // (e) candidate for the next step: lanes 0..7 of wave 0 hold the eight slots, three DPP levels inside the group of 8 (the
    // kernel cannot end before this: a serial loop of one thread over the slots cost eight dependent LDS round trips per step)
    if (w == 0) {
        static_assert(QD_CPW == 8, "eight slots <-> lanes 0..7");
        const bool in = ln < QD_CPW;
        const double sv = cval[ln & (QD_CPW - 1)];
        const int sp = cpos[ln & (QD_CPW - 1)];
        const bool has = in && qd_better(sv, sp, -1.0, 0x7fffffff);          // an empty slot never wins (as in the serial loop)
        ArgMax am = {has ? sv : -1.0, has ? sp : 0x7fffffff, has ? g * QD_CPW + ln : -1};
        am_step<0xB1>(am);     // quad_perm [1,0,3,2]
        am_step<0x4E>(am);     // quad_perm [2,3,0,1]
        am_step<0x141>(am);    // row_half_mirror: the other quad of the group of 8
        if (ln == 0) {
            QdCand cd = {am.val, am.pos, am.idx};
            cand_new[g] = cd;
        }
    }
}

// gather the LAPACK-style compact factors in pivoted order + jpvt
__global__ __launch_bounds__(256) void k_qd_assemble(QdArgs a) {
    const int prob = blockIdx.y + a.prob0;
    const ProbState st = a.state[prob];
    if (a.n2cap > 0 && st.n2 > a.n2cap) return;
    const int n2 = st.n2, ctot = n2 + 1;
    const int kp = qd_rows(a, st.kp);
    const int nst = qd_steps(a, st.kp, n2);
    const int g = blockIdx.x;
    if (kp == 0 || g * QD_CPW >= ctot) return;
    const int par = (a.step < 0) ? ((a.hyb > 0 && nst <= a.hyb) ? (nst & 1) : 0) : (nst & 1);   // block form keeps a single (parity 0) map
    const int* colat = a.colat + prob * 2 * a.sI + par * a.sI;
    const double* M = a.M + prob * a.sM;
    const double* Vb = a.Vb + prob * a.sVb;
    double* Rt = a.Rt + prob * a.sRt;
    const int w = wave_id(), ln = lane_id();
    for (int u = 0; u < 2; ++u) {
        const int k = g * QD_CPW + 2 * w + u;   // position
        if (k >= ctot) continue;
        const int c = (k < n2) ? colat[k] : n2;
        // eight rows per lane and round, every load unconditional from a per-lane choice of the source (R part of the column in M,
        // reflector in Vb): with the choice as a branch around each load the copy paid one memory round trip per 64 rows
        const double* mc = M + (size_t)c * a.ldr;
        const double* vc = Vb + (size_t)k * a.ldr;
        const double dk = (k < nst) ? a.diag[prob * a.sDiag + k] : 0.0;
        for (int r0 = 0; r0 < kp; r0 += 8 * WAVE) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + ln + WAVE * i;
                const int rc = r < kp ? r : 0;
                const double* src = (k < nst && rc > k) ? vc + rc : mc + rc;
                v[i] = *src;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + ln + WAVE * i;
                if (r < kp) Rt[r + (size_t)k * a.ldr] = (k < nst && r == k) ? dk : v[i];
            }
        }
        if (ln == 0 && k < n2) a.jpvt[prob * a.sJ + k] = c + 1;
    }
}

}  // namespace gn
