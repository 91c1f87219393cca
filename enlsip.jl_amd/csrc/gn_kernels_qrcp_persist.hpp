// Column-pivoted QR (dgeqp3 semantics) of the kp x n2 factor R0, persistent form: the columns of
// one problem are distributed over G co-resident workgroups which keep their column slabs in LDS
// for the whole factorisation (no per-step HBM traffic); the only inter-workgroup traffic per
// pivot step is one hand-off of <= kp doubles per workgroup:
//   every workgroup publishes its best remaining column (max partial norm, ties -> lowest LAPACK
//   position) together with the column's current entries; every workgroup reads the G headers,
//   picks the same winner, reads the winner's entries, forms the same reflector and applies it to
//   its own slab.
// Hand-off protocol (cdna_hip_programming.md Guideline 16, form R1 with write-through stores):
//   payload and header words are agent-scope relaxed atomics (global_store/load ... sc1); every
//   storing wave drains with s_waitcnt vmcnt(0), the workgroup barriers, ONE lane stores the tag;
//   ONE wave polls the tags (relaxed), the others wait at a workgroup barrier.  Buffers are
//   double-buffered by step parity; a workgroup can only be one step ahead of the slowest one.
//   The grid never exceeds the CU count with one workgroup per CU, every spin is bounded and
//   an abort word releases all workgroups if a bound is hit (status bit 2).
// Arithmetic (reflector, norm downdate, recompute rule) follows oracle/lapack_semantics.py.
#pragma once
#include "gn_device_utils.hpp"

namespace gn {

struct QpHeader {            // 16 bytes
    unsigned long long val_bits;   // double: best partial norm (or -1)
    unsigned long long pos_tag;    // low 32: LAPACK position of that column, high 32: step tag (j + 1)
};

struct QpArgs {
    int n, ldw, ldr;
    int G;                 // workgroups per problem
    int CW;                // physical columns per workgroup
    int kp_pad;            // slab column stride (doubles)
    int prob0;
    const double* W;   long long sW;
    double* Rt;        long long sRt;
    double* tau;       long long sTau;
    long long* jpvt;   long long sJ;
    QpHeader* hdr;     long long sHdr;     // per problem: 2 x G headers
    double* xbuf;      long long sX;       // per problem: 2 x G x ldr doubles
    unsigned* abort_word;                  // one word, zeroed per launch
    ProbState* state;
    int spin_limit;
    int n2cap;             // > 0: launch shape; wider problems are skipped (see QdArgs::n2cap)
};

__device__ __forceinline__ void qp_store_f64(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double qp_load_f64(const double* p) {
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)b);
}

template <int RPL>
__global__ __launch_bounds__(256) void k_qrcp_persist(QpArgs a) {
    extern __shared__ __attribute__((aligned(16))) double qp_smem[];
    const int prob = blockIdx.y + a.prob0;
    ProbState* stp = a.state + prob;
    const int kp = stp->kp, n2 = stp->n2, rankA = stp->rankA, ctot = n2 + 1;
    if (kp == 0 || (a.n2cap > 0 && n2 > a.n2cap)) return;   // uniform over the problem's workgroups
    const int g = blockIdx.x, G = a.G, CW = a.CW, ldk = a.kp_pad;
    const int ln = lane_id(), w = wave_id(), tid = threadIdx.x;
    double* slab = qp_smem;                               // [lc * ldk + r]
    double* vn1 = slab + (size_t)CW * ldk;                // [CW]
    double* vn2 = vn1 + CW;
    int* cpos = reinterpret_cast<int*>(vn2 + CW);         // [CW]  LAPACK position, or -1 - step once retired
    int* shi = cpos + CW;                                 // [8] scratch: 0 winner g, 1 winner pos, 2 abort, 3 my candidate lc
    const double* Wm = a.W + prob * a.sW;
    QpHeader* hdr = a.hdr + prob * a.sHdr;
    double* xbuf = a.xbuf + prob * a.sX;
    const double tol3z = 1.4901161193847656e-08;
    const int c_first = g * CW;
    const int ncol = (ctot - c_first) < CW ? (ctot - c_first) : CW;   // may be <= 0 for trailing workgroups

    // ---- init: own columns into LDS, norms, positions ------------------------------------------
    for (int lc = w; lc < CW; lc += 4) {
        const int c = c_first + lc;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            if (r < kp) {
                double v = 0.0;
                if (c < n2) v = (r <= c) ? Wm[r + (size_t)(rankA + c) * a.ldw] : 0.0;
                else if (c == n2) v = Wm[r + (size_t)a.n * a.ldw];
                slab[(size_t)lc * ldk + r] = v;
                s += v * v;
            }
        }
        s = wave_allsum(s);
        if (ln == 0) {
            const double nv = sqrt(s);
            vn1[lc] = (c < n2) ? nv : -1.0;
            vn2[lc] = nv;
            cpos[lc] = (c < n2) ? c : 0x7fffffff;   // rhs / padding: never a candidate
        }
    }
    if (tid == 0) shi[2] = 0;
    __syncthreads();

    int par = 0;
    for (int j = 0; j <= kp; ++j) {
        // ---- publish this workgroup's candidate for step j (skipped after the last step) ---------
        if (j < kp) {
            if (w == 0) {
                double bv = -1.0;
                int bp = 0x7fffffff, bl = -1;
                for (int lc = ln; lc < CW; lc += WAVE) {
                    const int p = cpos[lc];
                    if (p >= 0 && p != 0x7fffffff) {
                        const double v = vn1[lc];
                        if (v > bv || (v == bv && p < bp)) {
                            bv = v;
                            bp = p;
                            bl = lc;
                        }
                    }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const double ov = __shfl_xor(bv, off, WAVE);
                    const int op = __shfl_xor(bp, off, WAVE);
                    const int ol = __shfl_xor(bl, off, WAVE);
                    if (ov > bv || (ov == bv && op < bp)) {
                        bv = ov;
                        bp = op;
                        bl = ol;
                    }
                }
                if (ln == 0) {
                    shi[3] = bl;
                    shi[4] = bp;
                    reinterpret_cast<double*>(shi + 6)[0] = bv;   // shi[6..7]
                }
            }
            __syncthreads();
            const int bl = shi[3];
            double* xb = xbuf + ((size_t)par * G + g) * a.ldr;
            if (bl >= 0) {
                for (int r = j + tid; r < kp; r += 256) qp_store_f64(xb + r, slab[(size_t)bl * ldk + r]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                QpHeader* hme = hdr + (size_t)par * G + g;
                const double bv = reinterpret_cast<double*>(shi + 6)[0];
                __hip_atomic_store(&hme->val_bits, (unsigned long long)__double_as_longlong(bv), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long pt = ((unsigned long long)(unsigned)(j + 1) << 32) | (unsigned)shi[4];
                __hip_atomic_store(&hme->pos_tag, pt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (j == kp) break;
        // ---- gather the G candidates of step j ------------------------------------------------------
        if (w == 0) {
            double bv = -1.0;
            int bp = 0x7fffffff, bg = -1;
            bool ok = true;
            if (ln < G) {
                const QpHeader* he = hdr + (size_t)par * G + ln;
                unsigned long long pt = 0;
                int spins = 0;
                while (true) {
                    pt = __hip_atomic_load(&he->pos_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(pt >> 32) == (unsigned)(j + 1)) break;
                    if (++spins > a.spin_limit ||
                        __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (ok) {
                    bp = (int)(unsigned)(pt & 0xffffffffull);
                    const unsigned long long vb = __hip_atomic_load(&he->val_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bv = __longlong_as_double((long long)vb);
                    bg = ln;
                }
            }
            const bool all_ok = __all(ok);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(bv, off, WAVE);
                const int op = __shfl_xor(bp, off, WAVE);
                const int og = __shfl_xor(bg, off, WAVE);
                if (ov > bv || (ov == bv && op < bp)) {
                    bv = ov;
                    bp = op;
                    bg = og;
                }
            }
            if (ln == 0) {
                shi[0] = bg;
                shi[1] = bp;
                if (!all_ok || bg < 0) {
                    shi[2] = 1;
                    __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
        if (shi[2]) break;   // protocol failure: every workgroup leaves through the abort word
        const int gw = shi[0], q = shi[1];
        // ---- reflector from the winner's published entries (every wave, identical bits) -----------
        const double* xw = xbuf + ((size_t)par * G + gw) * a.ldr;
        double v[RPL];
        double xn2 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = j + ln + 64 * i;
            v[i] = (r < kp) ? qp_load_f64(xw + r) : 0.0;
            if (ln + 64 * i > 0) xn2 += v[i] * v[i];
        }
        xn2 = wave_allsum(xn2);
        const double alpha = wave_bcast(v[0], 0);
        const Reflector h = make_reflector(alpha, xn2);
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = (ln + 64 * i == 0) ? 1.0 : v[i] * h.scale;
        const int my_cand = shi[3];
        // ---- own columns ---------------------------------------------------------------------------------
        for (int lc = w; lc < ncol; lc += 4) {
            const int c = c_first + lc;
            const int p = cpos[lc];
            double* col = slab + (size_t)lc * ldk;
            if (g == gw && lc == my_cand) {
                // retire: V below, beta on the diagonal, R above stays
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const int r = j + ln + 64 * i;
                    if (r < kp) col[r] = (ln + 64 * i == 0) ? h.beta : v[i];
                }
                if (ln == 0) {
                    a.tau[prob * a.sTau + j] = h.tau;
                    cpos[lc] = -1 - j;
                    vn1[lc] = -1.0;
                }
            } else if (p >= 0) {   // trailing column or the carried right-hand side
                double x[RPL];
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const int r = j + ln + 64 * i;
                    x[i] = (r < kp) ? col[r] : 0.0;
                    dot += x[i] * v[i];
                }
                dot = wave_allsum(dot);
                if (h.tau != 0.0) {
                    const double wd = h.tau * dot;
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        const int r = j + ln + 64 * i;
                        x[i] -= wd * v[i];
                        if (r < kp) col[r] = x[i];
                    }
                }
                if (c < n2) {
                    const double ajc = wave_bcast(x[0], 0);
                    double o1 = vn1[lc];
                    const double o2 = vn2[lc];
                    if (o1 != 0.0) {
                        double temp = 1.0 - (fabs(ajc) / o1) * (fabs(ajc) / o1);
                        temp = temp > 0.0 ? temp : 0.0;
                        const double qq = o1 / o2;
                        const double temp2 = temp * qq * qq;
                        if (temp2 <= tol3z) {
                            double s = 0.0;
#pragma unroll
                            for (int i = 0; i < RPL; ++i)
                                if (ln + 64 * i > 0) s += x[i] * x[i];
                            s = wave_allsum(s);
                            o1 = (j + 1 < kp) ? sqrt(s) : 0.0;
                            if (ln == 0) {
                                vn1[lc] = o1;
                                vn2[lc] = o1;
                            }
                        } else if (ln == 0) {
                            vn1[lc] = o1 * sqrt(temp);
                        }
                    }
                    if (p == j && ln == 0) cpos[lc] = q;   // this column sat at position j: swapped to q
                }
            }
        }
        __syncthreads();
        par ^= 1;
    }
    // ---- write the factors in pivoted order -----------------------------------------------------------
    const bool failed = shi[2] != 0;
    double* Rt = a.Rt + prob * a.sRt;
    for (int lc = w; lc < ncol; lc += 4) {
        const int c = c_first + lc;
        const int p = cpos[lc];
        int k;
        if (c == n2) k = n2;
        else k = (p < 0) ? (-1 - p) : p;
        for (int r = ln; r < kp; r += WAVE) Rt[r + (size_t)k * a.ldr] = slab[(size_t)lc * ldk + r];
        if (ln == 0 && c < n2) a.jpvt[prob * a.sJ + k] = c + 1;
    }
    if (failed && tid == 0 && g == 0) atomicOr(&stp->status, 4);
}

}  // namespace gn
