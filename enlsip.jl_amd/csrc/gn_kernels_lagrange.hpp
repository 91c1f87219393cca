// Lagrange-multiplier estimates on the resident factors: the two consumers the reference runs
// right after the subproblem — first_lagrange_mult_estimate! (src/enlsip_functions.jl:461-508) and
// second_lagrange_mult_estimate! (:514-537) — plus the two matrix-vector products they need,
// grad = J' rx (:2690) and J1' (rx + J p) with J1 = (J*F_A.Q)[:, 1:t] taken from the workspace instead of being
// recomputed (quirk Q6 of the reference).
#pragma once
#include "gn_wg_linalg.hpp"

namespace gn {

// y[c] = sum_r A[r + c*ld] * x[r], c < ncols: one wave per column, lanes along rows (coalesced)
__global__ __launch_bounds__(256) void k_gemv_t(const double* __restrict__ A, long long ld, int rows, int ncols,
                                                const double* __restrict__ x, double* __restrict__ y) {
    const int c = blockIdx.x * 4 + wave_id();
    if (c >= ncols) return;
    const int ln = lane_id();
    const double* col = A + (size_t)c * ld;
    double s = 0.0;
    for (int r = ln; r < rows; r += WAVE) s += col[r] * x[r];
    s = wave_allsum(s);
    if (ln == 0) y[c] = s;
}

// y[r] = x0[r] + sum_c A[r + c*ld] * p[c], r < rows: one thread per row (coalesced across rows)
__global__ __launch_bounds__(256) void k_gemv_n_add(const double* __restrict__ A, long long ld, int rows, int ncols,
                                                    const double* __restrict__ p, const double* __restrict__ x0,
                                                    double* __restrict__ y) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    double s = x0 ? x0[r] : 0.0;
    for (int c = 0; c < ncols; ++c) s += A[r + (size_t)c * ld] * p[c];
    y[r] = s;
}

struct LagrangeArgs {
    int mode;              // 1: first estimate (vec = grad, n entries), 2: second estimate (vec = J1'(rx + J p), t entries)
    int n, t, kA, rank_solve;
    const double* FA;      // n x t compact factors of A' (ld n)
    const double* tauA;
    const long long* jpvtA;
    const double* vec;
    const double* cx;      // active constraint values (mode 1)
    const double* diag_scale;   // may be null
    double eps_rank;
    double* lambda;        // t
    double* scal;          // [0] grad_res (mode 1), [1] status (1: singular diagonal, 2: rank beyond the resident J1)
};

// one workgroup; LDS vectors of up to 1024 entries (n, t <= 1024 in this build)
__global__ __launch_bounds__(256) void k_lagrange(LagrangeArgs a) {
    __shared__ double bq[1024], v[1024], u[1024];
    __shared__ double blk[64 * 65];
    __shared__ double red[4];
    __shared__ int sh[2];
    const int tid = threadIdx.x, ln = lane_id(), w = wave_id();
    const int n = a.n, t = a.t, kA = a.kA;
    if (tid == 0) {
        sh[0] = pseudo_rank_serial(kA, a.eps_rank, [&](int i) { return a.FA[i + (size_t)i * n]; });
        sh[1] = 0;
    }
    if (a.mode == 1) {
        for (int i = tid; i < n; i += 256) bq[i] = a.vec[i];
    } else {
        for (int i = tid; i < t; i += 256) bq[i] = a.vec[i];
    }
    __syncthreads();
    const int pr = sh[0];
    if (a.mode == 1) {
        if (w == 0) wave_apply_reflectors<true>(a.FA, n, a.tauA, kA, n, bq);     // b = F.Q' * grad
        __syncthreads();
        // grad_res = ||b[prankA+1 : n]||
        double s = 0.0;
        for (int i = pr + tid; i < n; i += 256) s += bq[i] * bq[i];
        s = wave_allsum(s);
        if (ln == 0) red[w] = s;
        __syncthreads();
        if (tid == 0) a.scal[0] = (n > pr) ? sqrt((red[0] + red[1]) + (red[2] + red[3])) : 0.0;
    } else if (pr > a.rank_solve && tid == 0) {
        sh[1] |= 2;       // columns rank_solve .. pr-1 of J1 were overwritten by the factorisation of J2
    }
    for (int i = tid; i < t; i += 256) {
        v[i] = (i < pr) ? bq[i] : 0.0;
        u[i] = 0.0;
    }
    __syncthreads();
    if (pr > 0) wg_trsv<false>(a.FA, n, pr, v, blk, &sh[1]);                   // v = U(R[1:pr,1:pr]) \ b[1:pr]
    if (a.mode == 1 && pr > 0) {
        // y = LowerTriangular(R'[1:pr,1:pr]) \ (-cx[p]) by forward substitution on the columns of R (one wave),
        // then u = U(R) \ y
        if (w == 0) {
            for (int i = 0; i < pr; ++i) {
                double s = 0.0;
                for (int j = ln; j < i; j += WAVE) s += a.FA[j + (size_t)i * n] * u[j];
                s = wave_allsum(s);
                const double dii = a.FA[i + (size_t)i * n];
                if (dii == 0.0 && ln == 0) sh[1] |= 1;
                if (ln == 0) u[i] = (-a.cx[a.jpvtA[i] - 1] - s) / dii;
            }
        }
        __syncthreads();
        wg_trsv<false>(a.FA, n, pr, u, blk, &sh[1]);
    }
    __syncthreads();
    // lambda = (v + u)[invperm(p)], then the row-scaling back-transform
    for (int i = tid; i < t; i += 256) {
        const int dst = (int)a.jpvtA[i] - 1;
        double lam = v[i] + u[i];
        if (a.diag_scale) lam *= a.diag_scale[dst];
        a.lambda[dst] = lam;
    }
    if (tid == 0) a.scal[1] = (double)sh[1];
}

}  // namespace gn
