// Device part of newton_search_direction (src/enlsip_functions.jl:348-423): everything after the two Hessian sums
// (:391-396, callback-bound, they stay with the caller and arrive as Gamma = r_mat - c_mat):
//     E   = F_A.Q' * Gamma * F_A.Q                              :398
//     W22 = E22 + J2' J2 ;  W21 = E21 + J2' J1                  :405-409
//     d   = -W21 p1 - J2' rx                                    :411
//     sW22 = (W22 + W22') / 2 ; isposdef -> cholesky ; y = L \ d ; p2 = U \ y ; p = F_A.Q [p1; p2]     :414-421
// on the resident F_A, p1 and J (J * F_A.Q is recomputed into scratch, as the reference recomputes it at :384).
// This branch only runs for small problems (the reference switches second derivatives off at n + m >= 1000, :2658), so the
// kernels are plain: LDS-tiled products, one-workgroup Cholesky and triangular solves.
#pragma once
#include "gn_wg_linalg.hpp"

namespace gn {

// Q[:, j] = F_A.Q e_j: one wave per column
__global__ __launch_bounds__(64) void k_newton_q1(const double* FA, int n, const double* tauA, int kA, double* Q) {
    __shared__ double x[1024];
    const int j = blockIdx.x, ln = threadIdx.x;
    for (int i = ln; i < n; i += 64) x[i] = (i == j) ? 1.0 : 0.0;
    wave_mem_sync();
    wave_apply_reflectors<false>(FA, n, tauA, kA, n, x);
    wave_mem_sync();
    for (int i = ln; i < n; i += 64) Q[i + (size_t)j * n] = x[i];
}

// C (M x N, ldc) = op(A) B + C0:  op(A) = A' (TA, A is K x M, lda) or A (M x K);  B is K x N (ldb);  C0 may be null.
// 16 x 16 outputs per workgroup, K walked in tiles of 16 through LDS.
template <bool TA>
__global__ __launch_bounds__(256) void k_newton_gemm(int M, int N, int K, const double* A, long long lda, const double* B, long long ldb,
                                                     const double* C0, long long ldc0, double* C, long long ldc) {
    __shared__ double As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row = blockIdx.x * 16 + tx, col = blockIdx.y * 16 + ty;
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += 16) {
        // As[i][kk] = op(A)[row0 + i][k0 + kk], Bs[kk][j] = B[k0 + kk][col0 + j]
        {
            const int i = TA ? ty : tx, kk = TA ? tx : ty;           // lanes along the contiguous index of A
            const int r = blockIdx.x * 16 + i, k = k0 + kk;
            double v = 0.0;
            if (r < M && k < K) v = TA ? A[k + (size_t)r * lda] : A[r + (size_t)k * lda];
            As[i][kk] = v;
        }
        {
            const int k = k0 + tx, c = blockIdx.y * 16 + ty;
            Bs[tx][ty] = (k < K && c < N) ? B[k + (size_t)c * ldb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc += As[tx][kk] * Bs[kk][ty];
        __syncthreads();
    }
    if (row < M && col < N) C[row + (size_t)col * ldc] = acc + (C0 ? C0[row + (size_t)col * ldc0] : 0.0);
}

// d[i] = -(W21 p1)[i] - g[i]   (g = J2' rx), i < n2
__global__ __launch_bounds__(256) void k_newton_rhs(int n2, int rankA, const double* W21, long long ld, const double* p1, const double* g,
                                                    double* d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    double s = 0.0;
    for (int c = 0; c < rankA; ++c) s += W21[i + (size_t)c * ld] * p1[c];
    d[i] = -s - g[i];
}

// sW = (W + W') / 2 in place, then its Cholesky factor (LAPACK dpotrf semantics: a pivot <= 0 or NaN means "not positive
// definite", which is what Julia's isposdef / cholesky report): L in the lower triangle of W, U = L' written to Ut.
// flag[0] = 1 when not positive definite.  One workgroup.
__global__ __launch_bounds__(1024) void k_newton_chol(int n2, double* W, long long ld, double* Ut, int* flag) {
    __shared__ int bad;
    __shared__ double piv;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (long long e = tid; e < (long long)n2 * n2; e += 1024) {
        const int r = (int)(e % n2), c = (int)(e / n2);
        if (r > c) {
            const double s = (W[r + (size_t)c * ld] + W[c + (size_t)r * ld]) * 0.5;
            W[r + (size_t)c * ld] = s;
            W[c + (size_t)r * ld] = s;
        }
    }
    __syncthreads();
    for (int j = 0; j < n2; ++j) {
        if (tid == 0) {
            const double ajj = W[j + (size_t)j * ld];
            if (!(ajj > 0.0)) bad = 1;
            piv = sqrt(ajj);
        }
        __syncthreads();
        if (bad) break;
        const double ljj = piv;
        for (int r = j + tid; r < n2; r += 1024) W[r + (size_t)j * ld] = (r == j) ? ljj : W[r + (size_t)j * ld] / ljj;
        __syncthreads();
        // trailing update of the lower triangle: W[r][c] -= L[r][j] L[c][j], j < c <= r
        const int nt = n2 - j - 1;
        for (long long e = tid; e < (long long)nt * nt; e += 1024) {
            const int r = j + 1 + (int)(e % nt), c = j + 1 + (int)(e / nt);
            if (r >= c) W[r + (size_t)c * ld] -= W[r + (size_t)j * ld] * W[c + (size_t)j * ld];
        }
        __syncthreads();
    }
    if (tid == 0) flag[0] = bad;
    if (bad) return;
    for (long long e = tid; e < (long long)n2 * n2; e += 1024) {
        const int r = (int)(e % n2), c = (int)(e / n2);
        Ut[r + (size_t)c * n2] = (r <= c) ? W[c + (size_t)r * ld] : 0.0;        // U[r][c] = L[c][r]
    }
}

// y = L \ d ; p2 = U \ y   (one workgroup, the vector in LDS)
__global__ __launch_bounds__(256) void k_newton_solve(int n2, const double* L, long long ld, const double* Ut, const double* d, double* p2,
                                                      int* flag) {
    __shared__ double y[1024];
    __shared__ double blk[64 * 65];
    __shared__ int st;
    if (threadIdx.x == 0) st = 0;
    for (int i = threadIdx.x; i < n2; i += 256) y[i] = d[i];
    __syncthreads();
    wg_trsv<true>(L, (int)ld, n2, y, blk, &st);
    __syncthreads();
    wg_trsv<false>(Ut, n2, n2, y, blk, &st);
    __syncthreads();
    for (int i = threadIdx.x; i < n2; i += 256) p2[i] = y[i];
    if (threadIdx.x == 0 && st) flag[1] = st;
}

// Ep = E[p, p] (1-based permutation p of 1..n): newton_search_direction :396-399
__global__ void k_newton_perm(int n, const double* E, const long long* p, double* Ep) {
    const int j = blockIdx.x;
    const long long cj = p[j] - 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) Ep[i + (size_t)j * n] = E[(p[i] - 1) + (size_t)cj * n];
}

}  // namespace gn
