// Small helper kernels behind the factor accessors and the re-solve path.
#pragma once
#include "gn_wg_linalg.hpp"

namespace gn {

// x <- Q' x (TRANS) or Q x for a compact factorisation (F ld, tau, k reflectors), one wave.
template <bool TRANS>
__global__ __launch_bounds__(64) void k_vec_reflectors(const double* F, int ld, const double* tau, int k, int len,
                                                        double* x) {
    wave_apply_reflectors<TRANS>(F, ld, tau, k, len, x);
}

// dvec[0:m] = -J1 * p1 - rx ; rows m..ldw-1 zero.   (src/enlsip_functions.jl:134 / :145)
__global__ __launch_bounds__(256) void k_dtemp(const double* W, int ldw, int m, int rankA, const double* p1,
                                                const double* rx, double* dvec) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= ldw) return;
    double s = 0.0;
    if (row < m) {
        for (int c = 0; c < rankA; ++c) s += W[row + (size_t)c * ldw] * p1[c];
        s = -s - rx[row];
    }
    dvec[row] = s;
}

// zero-padded copy of a length-len vector into a length-ldw device vector
__global__ void k_pad_copy(const double* src, int len, int ldw, double* dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ldw) dst[i] = (i < len) ? src[i] : 0.0;
}

}  // namespace gn
