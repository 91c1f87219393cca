// Helpers of the constraint stage with MANY constraints (t > 64, matrices beyond one workgroup's LDS): F_A = qr(C.A', ColumnNorm())
// and F_L11 = qr(F_A.R', ColumnNorm()) (src/enlsip_functions.jl:700, :769) run through the distributed pivoted QR of
// gn_kernels_qrcp_dist.hpp (one launch per pivot step over all column groups and problems) instead of one workgroup walking
// the whole matrix in L2 (measured: 293 ms for the 1000 x 998 matrix of the reference's chained-Rosenbrock test).
#pragma once
#include "gn_device_utils.hpp"

namespace gn {

// dst[r + c * ldd] = src[r + c * lds], r < rows, c < cols, per problem
__global__ __launch_bounds__(256) void k_copy_cols(double* __restrict__ dst, long long ldd, long long sD, const double* __restrict__ src,
                                                   long long lds, long long sS, int rows, int cols) {
    const int c = blockIdx.x;
    const int prob = blockIdx.y;
    if (c >= cols) return;
    double* d = dst + prob * sD + (size_t)c * ldd;
    const double* s = src + prob * sS + (size_t)c * lds;
    for (int r = threadIdx.x; r < rows; r += 256) d[r] = s[r];
}

// b_buff[i] = -cx[F_A.p[i]]   (src/enlsip_functions.jl:131 / :141)
__global__ __launch_bounds__(256) void k_bbuff(double* __restrict__ out, long long sOut, const double* __restrict__ cx, long long sCx,
                                               const long long* __restrict__ jpvt, long long sJ, int t) {
    const int prob = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < t) out[prob * sOut + i] = -cx[prob * sCx + jpvt[prob * sJ + i] - 1];
}

// per-problem stand-in records for the distributed QR (it reads kp = steps and n2 = columns from a ProbState)
__global__ __launch_bounds__(256) void k_fake_state(ProbState* st, int batch, int kp, int n2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < batch) {
        ProbState s{};
        s.rankA = 0; s.n2 = n2; s.kp = kp;
        st[i] = s;
    }
}

}  // namespace gn
