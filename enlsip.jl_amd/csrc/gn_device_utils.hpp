// Device-side helpers shared by every kernel of libenlsip_gn (gfx950 only, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gn {

constexpr int WAVE = 64;

// Per-problem device record: written by the constraint stage, read by every later kernel, so no
// host round trip is needed in the middle of a solve.
struct ProbState {
    int rankA;     // pseudo_rank(diag(R_A))
    int n2;        // n - rankA           (columns of J2)
    int kp;        // min(m, n2)          (reflectors of the unpivoted QR of J2)
    int rankJ2;    // pseudo_rank(diag(R_J2))
    int code;      // 1 / -1
    int dimA;
    int dimJ2;
    int status;    // bit0: exactly singular triangular diagonal met
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// butterfly all-reduce: every lane ends with the same bits (same association order in all lanes)
__device__ __forceinline__ double wave_allsum(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}

__device__ __forceinline__ double wave_bcast(double x, int src_lane) {
    return __shfl(x, src_lane, WAVE);
}

// LAPACK dlarfg without the safmin rescaling loop:
// beta = -sign(alpha) * hypot(alpha, xnorm), tau = (beta - alpha) / beta, v = x / (alpha - beta)
struct Reflector {
    double beta, tau, scale;
};
__device__ __forceinline__ Reflector make_reflector(double alpha, double xnorm2) {
    Reflector r;
    if (xnorm2 == 0.0) {
        r.beta = alpha;
        r.tau = 0.0;
        r.scale = 0.0;
    } else {
        double nrm = sqrt(alpha * alpha + xnorm2);
        r.beta = -copysign(nrm, alpha);
        r.tau = (r.beta - alpha) / r.beta;
        r.scale = 1.0 / (alpha - r.beta);
    }
    return r;
}

// pseudo_rank (src/enlsip_functions.jl:17-31), diag accessed through a functor; serial, call
// from one lane.
template <class DiagFn>
__device__ inline int pseudo_rank_serial(int len, double eps_rank, DiagFn diag) {
    if (len == 0 || fabs(diag(0)) < eps_rank) return 0;
    const double tol = fabs(diag(0)) * sqrt((double)len) * eps_rank;
    int r = 1;
    while (r < len && fabs(diag(r - 1)) > tol) ++r;
    return r - ((r == len && fabs(diag(r - 1)) > tol) ? 0 : 1);
}

}  // namespace gn
