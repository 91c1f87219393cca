// Magnitudes beyond the range of plain sums of squares (entries above ~2^500 or below ~2^-500): LAPACK's behaviour on the same
// inputs.  The reference's qr(., ColumnNorm()) (src/enlsip_functions.jl:223, :700, :769) runs dgeqp3, whose column norms (dnrm2)
// and reflectors (dlarfg with its safmin loop) scale internally and never overflow or underflow; the kernels of this library
// square plainly.  Instead of a scaled norm at every one of their norm sites, the rare case is DETECTED on the result and the
// problem is solved again on a copy of its inputs scaled by a power of two — exact, so every factor comes out as
// 2^shift x (what LAPACK computes on the unscaled data), the reflectors and pivots as they are — and the resident factors and the
// outputs are scaled back (enlsip_gn.hip: "rescale").  The absolute first test of pseudo_rank (:17-31) is taken against
// 2^shift eps_rank (pseudo_rank_abs_threshold).  Costs one tiny launch per solve on the normal path (k_extreme_flags).
#pragma once
#include "gn_device_utils.hpp"

namespace gn {

constexpr int GN_FLAG_NONFINITE = 1 << 29;     // ProbState::status bits, host-internal: cleared before the caller sees status
constexpr int GN_FLAG_TINY = 1 << 30;
constexpr int GN_RESCALE_BAND = 400;           // inputs with 2^-400 <= max |entry| <= 2^400 are never rescaled

// first diagonal entries of F_A.R and F_J2.R: the largest column norm of each factorisation (pivoting puts it first).  Not
// finite: a sum of squares overflowed (or the inputs held NaN / Inf).  Below 2^-440 (zero included): the squares of the largest
// column sit at the bottom of the exponent range (or J / A is zero).  Both only nominate the problem: the host looks at the
// magnitudes of its inputs before anything is redone.
__global__ __launch_bounds__(256) void k_extreme_flags(ProbState* state, const double* Rt, long long sRt, const double* FA, long long sFA,
                                                       int kA, int n2cap, int batch) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= batch) return;
    const ProbState st = state[k];
    if (n2cap > 0 && st.n2 > n2cap) return;            // redone by the caller (second attempt): nothing of it is final yet
    const double tiny = 0x1p-440;
    int f = 0;
    if (kA > 0) {
        const double a = fabs(FA[k * sFA]);
        if (!(a <= __builtin_huge_val()) || a == __builtin_huge_val()) f |= GN_FLAG_NONFINITE;
        else if (a < tiny) f |= GN_FLAG_TINY;
    }
    if (Rt && st.kp > 0) {
        const double r = fabs(Rt[k * sRt]);
        if (!(r <= __builtin_huge_val()) || r == __builtin_huge_val()) f |= GN_FLAG_NONFINITE;
        else if (r < tiny) f |= GN_FLAG_TINY;
    }
    if (f) state[k].status = st.status | f;
}

__global__ __launch_bounds__(256) void k_clear_status_bits(ProbState* state, int bits, int batch) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < batch) state[k].status &= ~bits;
}

// bits of max |x| over a rows x cols matrix (non-negative doubles order like their bit patterns; a NaN outranks everything, so
// the host sees it); one workgroup per column
__global__ __launch_bounds__(256) void k_amax_bits(const double* A, long long ld, int rows, int cols, unsigned long long* out) {
    const int c = blockIdx.x;
    if (c >= cols) return;
    const double* col = A + (size_t)c * ld;
    unsigned long long mx = 0ull;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(col[r]));
        mx = b > mx ? b : mx;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = (unsigned long long)__shfl_xor((long long)mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, mx);
}

// dst = src * 2^shift (exact unless an entry leaves the exponent range: such entries are below 2^-1000 of the largest one)
__global__ __launch_bounds__(256) void k_scale_copy(double* dst, long long ldd, const double* src, long long lds, int rows, int cols, int shift) {
    const int c = blockIdx.y;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) dst[r + (size_t)c * ldd] = __builtin_ldexp(src[r + (size_t)c * lds], shift);
}

// in place: X *= 2^shift on a rows x cols region; upper = 1: only the entries on and above the diagonal (the R part of compact
// LAPACK factors: the reflector vectors below it do not depend on the scale of the data)
__global__ __launch_bounds__(256) void k_scale_region(double* X, long long ld, int rows, int cols, int shift, int upper) {
    const int c = blockIdx.y;
    if (c >= cols) return;
    const int rmax = upper ? (c + 1 < rows ? c + 1 : rows) : rows;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rmax; r += gridDim.x * 256) X[r + (size_t)c * ld] = __builtin_ldexp(X[r + (size_t)c * ld], shift);
}

}  // namespace gn
