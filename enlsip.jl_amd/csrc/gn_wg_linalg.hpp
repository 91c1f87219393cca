// Workgroup-cooperative dense kernels on small column-major matrices (global/L2 or LDS resident):
// column-pivoted Householder QR with LAPACK dgeqp3 semantics, triangular solves, reflector
// application to a vector.  Follows oracle/lapack_semantics.py step by step.
//
// Mapping: lanes run along ROWS (coalesced for column-major data), waves take columns.  A dot
// product over a column is therefore a wave butterfly reduction; every lane of every wave sees
// bitwise identical reflector scalars, so no broadcast through memory is needed.
#pragma once
#include "gn_device_utils.hpp"

namespace gn {

// ---------------------------------------------------------------------------------------------
// wg_geqp2: A is rows x (cols + nrhs), leading dimension ld.  The first `cols` columns are
// factored with column pivoting (dlaqp2: pivot = FIRST maximal partial norm, Householder
// reflector by dlarfg, norm downdate with the sqrt(eps) recompute rule); the last `nrhs`
// columns receive the reflectors but never pivot.  tau: min(rows, cols); jpvt: cols, 1-based.
// vn1/vn2: LDS scratch, >= cols doubles each.  sh_i: LDS scratch, >= 4 ints.
// Requires rows <= 64 * RPL.  Ends with a __syncthreads().
// ---------------------------------------------------------------------------------------------
template <int RPL, int G>
__device__ void wg_geqp2(double* __restrict__ A, int ld, int rows, int cols, int nrhs,
                         double* __restrict__ tau, long long* __restrict__ jpvt,
                         double* vn1, double* vn2, int* sh_i) {
    const int ln = lane_id();
    const int w = wave_id();
    const int nw = blockDim.x >> 6;
    const int ctot = cols + nrhs;
    const int k = rows < cols ? rows : cols;
    const double tol3z = 1.4901161193847656e-08;  // sqrt(eps), dlaqp2

    // initial partial column norms + identity permutation
    for (int c = w; c < cols; c += nw) {
        double s = 0.0;
        for (int r = ln; r < rows; r += WAVE) {
            double x = A[r + (size_t)c * ld];
            s += x * x;
        }
        s = wave_allsum(s);
        if (ln == 0) {
            vn1[c] = sqrt(s);
            vn2[c] = vn1[c];
            jpvt[c] = c + 1;
        }
    }
    __syncthreads();

    for (int j = 0; j < k; ++j) {
        // (a) pivot = first index of max vn1[j:cols)
        if (w == 0) {
            double best = -1.0;
            int bidx = j;
            for (int c = j + ln; c < cols; c += WAVE) {
                double v = vn1[c];
                if (v > best) {
                    best = v;
                    bidx = c;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                double ob = __shfl_xor(best, off, WAVE);
                int oi = __shfl_xor(bidx, off, WAVE);
                if (ob > best || (ob == best && oi < bidx)) {
                    best = ob;
                    bidx = oi;
                }
            }
            if (ln == 0) sh_i[0] = bidx;
        }
        __syncthreads();
        const int pvt = sh_i[0];
        // (b) swap columns j <-> pvt (whole columns, as LAPACK does)
        if (pvt != j) {
            for (int r = threadIdx.x; r < rows; r += blockDim.x) {
                double a = A[r + (size_t)j * ld];
                double b = A[r + (size_t)pvt * ld];
                A[r + (size_t)j * ld] = b;
                A[r + (size_t)pvt * ld] = a;
            }
            if (threadIdx.x == 0) {
                long long tj = jpvt[j];
                jpvt[j] = jpvt[pvt];
                jpvt[pvt] = tj;
                vn1[pvt] = vn1[j];
                vn2[pvt] = vn2[j];
            }
        }
        __syncthreads();
        // (c) every wave builds the reflector of column j redundantly (identical bits)
        double v[RPL];
        double xn2 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = j + ln + WAVE * i;
            v[i] = (r < rows) ? A[r + (size_t)j * ld] : 0.0;
            if (ln + WAVE * i > 0) xn2 += v[i] * v[i];
        }
        xn2 = wave_allsum(xn2);
        const double alpha = wave_bcast(v[0], 0);
        const Reflector h = make_reflector(alpha, xn2);
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = (ln + WAVE * i == 0) ? 1.0 : v[i] * h.scale;
        __syncthreads();  // all waves have read column j before wave 0 overwrites it
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = j + ln + WAVE * i;
                if (r < rows) A[r + (size_t)j * ld] = (ln + WAVE * i == 0) ? h.beta : v[i];
            }
            if (ln == 0) tau[j] = h.tau;
        }
        // (d) apply H_j to the trailing columns, G columns per wave at a time
        for (int cbase = j + 1 + w * G; cbase < ctot; cbase += nw * G) {
            double a[G][RPL];
            double dot[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int c = cbase + g;
                dot[g] = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const int r = j + ln + WAVE * i;
                    a[g][i] = (c < ctot && r < rows) ? A[r + (size_t)c * ld] : 0.0;
                    dot[g] += a[g][i] * v[i];
                }
            }
            if constexpr (G == 8 || G == 4) {
                double red[G];
                wave_allsumN(dot, red);      // G reductions in one transposed butterfly
#pragma unroll
                for (int g = 0; g < G; ++g) dot[g] = red[g];
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) dot[g] = wave_allsum(dot[g]);
            }
            if (h.tau != 0.0) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int c = cbase + g;
                    const double wd = h.tau * dot[g];
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        const int r = j + ln + WAVE * i;
                        a[g][i] -= wd * v[i];
                        if (c < ctot && r < rows) A[r + (size_t)c * ld] = a[g][i];
                    }
                }
            }
            // norm downdate (dlaqp2) of the G columns at once: lane g owns column cbase + g (the divisions and the
            // square root are issued once per sweep instead of once per column)
            double ajl = 0.0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double x = wave_bcast(a[g][0], 0);
                ajl = (ln == g) ? x : ajl;
            }
            bool need = false;
            const int cl = cbase + ln;
            if (ln < G && cl < cols) {
                const double o1 = vn1[cl], o2 = vn2[cl];
                if (o1 != 0.0) {
                    const NormDown nd = norm_downdate(o1, o2, ajl, tol3z);
                    if (nd.need) need = true;
                    else vn1[cl] = nd.vn1;
                }
            }
            const unsigned long long redo = __ballot(need);
            if (redo) {   // rare: recompute the partial norm from the updated column
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (!((redo >> g) & 1ull)) continue;
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < RPL; ++i)
                        if (ln + WAVE * i > 0) s += a[g][i] * a[g][i];
                    s = wave_allsum(s);
                    const double nv = (j + 1 < rows) ? sqrt(s) : 0.0;
                    if (ln == 0) {
                        vn1[cbase + g] = nv;
                        vn2[cbase + g] = nv;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// wg_trsv: solve T x = y in place, T dim x dim triangular, column-major (ld), upper or lower.
// y may live in LDS or global.  blk: LDS scratch of 64*65 doubles.  A zero diagonal sets
// *status |= 1 (Julia's `\` would throw SingularException; we flag and keep IEEE semantics).
// Blocked by 64: one wave solves the diagonal block out of LDS, the workgroup applies the
// off-diagonal GEMV.
// ---------------------------------------------------------------------------------------------
template <bool LOWER>
__device__ void wg_trsv(const double* __restrict__ T, int ld, int dim, double* y, double* blk,
                        int* status) {
    const int ln = lane_id();
    const int w = wave_id();
    const int nblk = (dim + 63) / 64;
    for (int bb = 0; bb < nblk; ++bb) {
        const int bi = LOWER ? bb : nblk - 1 - bb;
        const int i0 = bi * 64;
        const int nb = (dim - i0) < 64 ? (dim - i0) : 64;
        // stage the diagonal block
        for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) {
            const int r = e % nb, c = e / nb;
            blk[r + c * 65] = T[(i0 + r) + (size_t)(i0 + c) * ld];
        }
        __syncthreads();
        if (w == 0) {
            // Row-scaled form: lane i divides its right-hand side by its diagonal entry ONCE and multiplies its row by the
            // reciprocal, so the chain of a step is broadcast -> multiply-add instead of broadcast -> division -> multiply-add
            // (a 64-step block: ~10 us -> ~1.5 us; the division was on the critical path of every step).  The scaling perturbs
            // every entry of a row by one rounding: as backward stable as the plain recurrence.  A zero diagonal is flagged and
            // yields the IEEE infinities / NaNs of a division by zero, as before.
            const double di = (ln < nb) ? blk[ln + ln * 65] : 1.0;
            if (di == 0.0) atomicOr(status, 1);
            const double ri = 1.0 / di;
            double yi = (ln < nb) ? y[i0 + ln] / di : 0.0;
            if (LOWER) {
                for (int kk = 0; kk < nb; ++kk) {
                    const double xk = wave_bcast(yi, kk);
                    if (ln > kk && ln < nb) yi -= (blk[ln + kk * 65] * ri) * xk;
                }
            } else {
                for (int kk = nb - 1; kk >= 0; --kk) {
                    const double xk = wave_bcast(yi, kk);
                    if (ln < kk) yi -= (blk[ln + kk * 65] * ri) * xk;
                }
            }
            if (ln < nb) y[i0 + ln] = yi;
        }
        __syncthreads();
        // off-diagonal update with the freshly solved block
        if (LOWER) {
            for (int r = i0 + nb + threadIdx.x; r < dim; r += blockDim.x) {
                double s = 0.0;
#pragma unroll 8
                for (int c = 0; c < nb; ++c) s += T[r + (size_t)(i0 + c) * ld] * y[i0 + c];   // 8 independent loads in flight
                y[r] -= s;
            }
        } else {
            for (int r = threadIdx.x; r < i0; r += blockDim.x) {
                double s = 0.0;
#pragma unroll 8
                for (int c = 0; c < nb; ++c) s += T[r + (size_t)(i0 + c) * ld] * y[i0 + c];
                y[r] -= s;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// wave_apply_reflectors: one wave applies H_0..H_{k-1} (TRANS: in that order = Q' x) or
// H_{k-1}..H_0 (Q x) of a compact factorisation (factors ld, tau) to the vector x (length len,
// LDS or global).  Called by a single wave; no barriers inside.
// ---------------------------------------------------------------------------------------------
template <bool TRANS>
__device__ void wave_apply_reflectors(const double* __restrict__ F, int ld, const double* __restrict__ tau,
                                      int k, int len, double* x) {
    const int ln = lane_id();
    for (int s = 0; s < k; ++s) {
        const int j = TRANS ? s : k - 1 - s;
        const double tj = tau[j];
        if (tj == 0.0) continue;
        double dot = 0.0;
        for (int r = j + ln; r < len; r += WAVE) {
            const double vr = (r == j) ? 1.0 : F[r + (size_t)j * ld];
            dot += vr * x[r];
        }
        dot = wave_allsum(dot) * tj;
        for (int r = j + ln; r < len; r += WAVE) {
            const double vr = (r == j) ? 1.0 : F[r + (size_t)j * ld];
            x[r] -= dot * vr;
        }
    }
}

// Register form of wave_apply_reflectors for len <= 64 * RPL: the reflector is fetched ONCE (the generic form reads it for
// the dot product and again for the update, two dependent L2 round trips per reflector) and the next one travels while the
// current one is applied.  x in LDS, lane ln owns entries ln + 64 i.
template <bool TRANS, int RPL>
__device__ void wave_apply_reflectors_reg(const double* __restrict__ F, int ld, const double* __restrict__ tau,
                                          int k, int len, double* x) {
    const int ln = lane_id();
    double xr[RPL], vn[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) xr[i] = (ln + 64 * i < len) ? x[ln + 64 * i] : 0.0;
    auto fetch = [&](int j, double (&v)[RPL]) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = ln + 64 * i;
            v[i] = (r > j && r < len) ? F[r + (size_t)j * ld] : 0.0;
        }
    };
    if (k > 0) fetch(TRANS ? 0 : k - 1, vn);
    // tau of up to 64 reflectors in one request (lane j holds tau[j]); a load per reflector inside the loop is an exposed L2 round
    // trip each (the one-wave-per-problem kernels apply Q1 to one vector: their tail is these chains)
    const double tl = (ln < k && ln < 64) ? tau[ln] : 0.0;
    for (int s = 0; s < k; ++s) {
        const int j = TRANS ? s : k - 1 - s;
        double v[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) v[i] = (ln + 64 * i == j) ? 1.0 : vn[i];
        if (s + 1 < k) fetch(TRANS ? s + 1 : k - 2 - s, vn);
        const double tj = (j < 64) ? readlane_f64(tl, j) : tau[j];
        if (tj == 0.0) continue;
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) dot += v[i] * xr[i];
        dot = wave_allsum(dot) * tj;
#pragma unroll
        for (int i = 0; i < RPL; ++i) xr[i] -= dot * v[i];
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i)
        if (ln + 64 * i < len) x[ln + 64 * i] = xr[i];
}

}  // namespace gn
