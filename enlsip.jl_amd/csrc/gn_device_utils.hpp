// Device-side helpers shared by every kernel of libenlsip_gn (gfx950 only, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/enlsip_gn.h"

namespace gn {

constexpr int WAVE = 64;

// ENLSIP_GN_LAB (not set by build.sh): the laboratory build.  Only with it do the timing ablations (ENLSIP_V4_ABLATE,
// ENLSIP_FACTOR_ABLATE: kernels whose RESULTS ARE WRONG by design), the phase-stamp macros (ENLSIP_V4_STAMPS,
// ENLSIP_FACTOR_STAMPS, ENLSIP_SB_STEP_STAMPS, GN_PS_STAMPS) and the A/B run-time switches that no test of the suite uses
// (ENLSIP_GN_FACTOR_NW4, ENLSIP_GN_DEBUG_MAXPAN, ENLSIP_GN_DEBUG_STAGE, ENLSIP_GN_JQ1_ROWS2) exist; the product build refuses the
// macros and never reads those variables.  tests/microbench/*.hip and tests/probes/*.sh build with -DENLSIP_GN_LAB.
#ifndef ENLSIP_GN_LAB
#if (defined(ENLSIP_V4_ABLATE) && ENLSIP_V4_ABLATE != 0) || (defined(ENLSIP_FACTOR_ABLATE) && ENLSIP_FACTOR_ABLATE != 0) || \
    defined(ENLSIP_V4_STAMPS) || defined(ENLSIP_FACTOR_STAMPS) || defined(ENLSIP_SB_STEP_STAMPS) || defined(GN_PS_STAMPS)
#error "timing ablations and phase stamps are laboratory code: build with -DENLSIP_GN_LAB"
#endif
#endif

// Host side: kernel-selection record of the solve that runs on this thread (include/enlsip_gn.h: ENLSIP_GN_ROUTE_*); the launch
// helpers OR their bit in, solve_dev folds it into the handle.  A pipelined solve runs its second half on a thread of its own.
inline thread_local unsigned long long gn_route_acc = 0;
#define GN_ROUTE(bit) (::gn::gn_route_acc |= (1ull << (bit)))

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

// Orders memory traffic between the LANES OF ONE WAVE: loads that follow may read what other lanes of the wave stored
// before.  Without it the compiler, which reasons per lane, may hoist a load above another lane's (conditional) store of
// the same address and forward the stored value only inside the storing lane.  Costs a waitcnt, no barrier instruction.
__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one DPP lane permutation of a double (two 32-bit v_mov_dpp)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// butterfly partner sums across the 16-lane rows with gfx950's v_permlane16_swap / v_permlane32_swap
// (VALU, no LDS crossbar): with vdst = vsrc = x the swap leaves {r0,r0,r2,r2} / {r1,r1,r3,r3}
// (rows) resp. {lo,lo} / {hi,hi} (32-lane halves), whose sum is the xor-16 / xor-32 butterfly.
__device__ __forceinline__ double xor16_sum(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double xor32_sum(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// Wave all-reduce without LDS traffic: four DPP butterflies inside each row of 16 lanes
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the xor-16 / xor-32 row butterflies.
// Every level adds the two partners' partial sums (a + b == b + a), so all lanes end with the
// same bits.
__device__ __forceinline__ double wave_allsum(double x) {
    x += dpp_f64<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_f64<0x141>(x);   // row_half_mirror
    x += dpp_f64<0x140>(x);   // row_mirror
    x = xor16_sum(x);
    return xor32_sum(x);
}

// All-reduce inside each 32-lane half of the wave (two problems per wave: gn_kernels_final_small.hpp): wave_allsum without its
// last level.
__device__ __forceinline__ double half_allsum(double x) {
    x += dpp_f64<0xB1>(x);
    x += dpp_f64<0x4E>(x);
    x += dpp_f64<0x141>(x);
    x += dpp_f64<0x140>(x);
    return xor16_sum(x);
}

// Eight simultaneous wave all-reduces ("transposed" butterfly): at every DPP level a lane hands
// half of its partial sums to its partner and keeps the other half, so 8 -> 4 -> 2 -> 1 values are
// moved instead of 8 each time.  Partner order row_mirror (i^15), row_half_mirror (i^7),
// quad_perm [3,2,1,0] (i^3), quad_perm [1,0,3,2] (i^1): each partner keeps the same column set as
// the lane (the set is decided by the lane bits the partner shares).  After the row levels lane l
// holds column 4*bit3 + 2*bit2 + bit1; two row butterflies finish the sum and 8 readlanes
// broadcast it.  out[c] is identical in all lanes.
__device__ __forceinline__ void wave_allsum8(const double (&d)[8], double (&out)[8]) {
    const int ln = threadIdx.x & 63;
    const bool b3 = (ln & 8) != 0, b2 = (ln & 4) != 0, b1 = (ln & 2) != 0;
    double k4[4], k2[2], k1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double mine = b3 ? d[c + 4] : d[c];
        const double give = b3 ? d[c] : d[c + 4];
        k4[c] = mine + dpp_f64<0x140>(give);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double mine = b2 ? k4[c + 2] : k4[c];
        const double give = b2 ? k4[c] : k4[c + 2];
        k2[c] = mine + dpp_f64<0x141>(give);
    }
    {
        const double mine = b1 ? k2[1] : k2[0];
        const double give = b1 ? k2[0] : k2[1];
        k1 = mine + dpp_f64<0x1B>(give);     // quad_perm [3,2,1,0]
    }
    k1 += dpp_f64<0xB1>(k1);                 // quad_perm [1,0,3,2]
    k1 = xor16_sum(k1);
    k1 = xor32_sum(k1);
#pragma unroll
    for (int c = 0; c < 8; ++c) out[c] = readlane_f64(k1, 8 * (c >> 2) + 4 * ((c >> 1) & 1) + 2 * (c & 1));
}

// Four-column version of wave_allsum8 (lane l ends with column 2*bit2 + bit1).
__device__ __forceinline__ void wave_allsum4(const double (&d)[4], double (&out)[4]) {
    const int ln = threadIdx.x & 63;
    const bool b2 = (ln & 4) != 0, b1 = (ln & 2) != 0;
    double k2[2], k1;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double mine = b2 ? d[c + 2] : d[c];
        const double give = b2 ? d[c] : d[c + 2];
        k2[c] = mine + dpp_f64<0x141>(give);
    }
    {
        const double mine = b1 ? k2[1] : k2[0];
        const double give = b1 ? k2[0] : k2[1];
        k1 = mine + dpp_f64<0x1B>(give);     // quad_perm [3,2,1,0]
    }
    k1 += dpp_f64<0xB1>(k1);                 // quad_perm [1,0,3,2]
    k1 += dpp_f64<0x128>(k1);                // row_ror:8 (the other half of the row holds the same column)
    k1 = xor16_sum(k1);
    k1 = xor32_sum(k1);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = readlane_f64(k1, 4 * (c >> 1) + 2 * (c & 1));
}
__device__ __forceinline__ void wave_allsumN(const double (&d)[8], double (&out)[8]) { wave_allsum8(d, out); }
// two columns: lane bit 1 decides which column a lane keeps after the first level
__device__ __forceinline__ void wave_allsumN(const double (&d)[2], double (&out)[2]) {
    const int ln = threadIdx.x & 63;
    const bool b1 = (ln & 2) != 0;
    const double mine = b1 ? d[1] : d[0];
    const double give = b1 ? d[0] : d[1];
    double k1 = mine + dpp_f64<0x1B>(give);    // quad_perm [3,2,1,0]: partner has the other bit 1
    k1 += dpp_f64<0xB1>(k1);                   // quad_perm [1,0,3,2]
    k1 += dpp_f64<0x124>(k1);                  // row_ror:4  (partners share bit 1)
    k1 += dpp_f64<0x128>(k1);                  // row_ror:8
    k1 = xor16_sum(k1);
    k1 = xor32_sum(k1);
    out[0] = readlane_f64(k1, 0);
    out[1] = readlane_f64(k1, 2);
}
__device__ __forceinline__ void wave_allsumN(const double (&d)[4], double (&out)[4]) { wave_allsum4(d, out); }

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}

// The dword of lane (l ^ JX), JX = 1 .. 32, without LDS: DPP inside the rows of 16 lanes, permlane swaps across them.
template <int JX>
__device__ __forceinline__ int lane_xor_i32(int x, int ln) {
    if constexpr (JX == 1) return dpp_i32<0xB1>(x);            // quad_perm [1,0,3,2]
    else if constexpr (JX == 2) return dpp_i32<0x4E>(x);       // quad_perm [2,3,0,1]
    else if constexpr (JX == 4) {                              // row_ror:12 brings lane i + 4, row_ror:4 lane i - 4
        const int up = dpp_i32<0x12C>(x), dn = dpp_i32<0x124>(x);
        return (ln & 4) ? dn : up;
    } else if constexpr (JX == 8) return dpp_i32<0x128>(x);    // row_ror:8
    else if constexpr (JX == 16) {
        const auto a = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);   // {r0,r0,r2,r2} / {r1,r1,r3,r3}
        return (ln & 16) ? (int)a[0] : (int)a[1];
    } else {
        const auto a = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);   // {lo,lo} / {hi,hi}
        return (ln & 32) ? (int)a[0] : (int)a[1];
    }
}

// Pivot key of a partial column norm: a NaN norm (NaN / Inf in the input) ranks as +infinity, so that a pivot search always
// returns one of its candidates — with plain comparisons a NaN never wins and the search comes back empty (index -1), which
// the callers would use as a column index.
__device__ __forceinline__ double pivot_key(double norm) { return (norm != norm) ? __builtin_huge_val() : norm; }

// Wave arg-max with LAPACK's idamax tie rule on (value, position): larger value wins, equal values
// -> lower position wins.  Returns the same (val, pos, idx) in every lane.  DPP butterflies inside
// the rows of 16 lanes, then the four row winners are combined through SGPRs.
struct ArgMax {
    double val;
    int pos, idx;
};
__device__ __forceinline__ bool am_better(double v, int p, double bv, int bp) {
    return v > bv || (v == bv && p < bp);
}
template <int CTRL>
__device__ __forceinline__ void am_step(ArgMax& a) {
    const double ov = dpp_f64<CTRL>(a.val);
    const int op = dpp_i32<CTRL>(a.pos);
    const int oi = dpp_i32<CTRL>(a.idx);
    if (am_better(ov, op, a.val, a.pos)) {
        a.val = ov;
        a.pos = op;
        a.idx = oi;
    }
}
// max over the rows of 16 lanes by DPP, the four row maxima combined through scalar registers: the same value in every lane
__device__ __forceinline__ double wave_allmax(double x) {
    x = fmax(x, dpp_f64<0xB1>(x));
    x = fmax(x, dpp_f64<0x4E>(x));
    x = fmax(x, dpp_f64<0x141>(x));
    x = fmax(x, dpp_f64<0x140>(x));
    return fmax(fmax(readlane_f64(x, 0), readlane_f64(x, 16)), fmax(readlane_f64(x, 32), readlane_f64(x, 48)));
}
// The exact (value, position) tournament below costs ~80 dependent operations and stands between the norm downdates of a pivot
// step and the reflector of the next.  Almost always ONE lane holds the largest value: then a plain max tournament (3 operations
// per level) and a ballot name the winner; an exact tie (or no candidate at all) takes the full tournament, so the result is
// the same in every case.
__device__ __forceinline__ ArgMax wave_argmax_exact(double val, int pos, int idx);
__device__ __forceinline__ ArgMax wave_argmax(double val, int pos, int idx) {
    const double m = wave_allmax(val);
    const unsigned long long mk = __ballot(val == m);
    if (__popcll(mk) == 1) {
        const int l = (int)__builtin_ctzll(mk);
        ArgMax r = {m, __builtin_amdgcn_readlane(pos, l), __builtin_amdgcn_readlane(idx, l)};
        return r;
    }
    return wave_argmax_exact(val, pos, idx);
}
__device__ __forceinline__ ArgMax wave_argmax_exact(double val, int pos, int idx) {
    ArgMax a = {val, pos, idx};
    am_step<0xB1>(a);
    am_step<0x4E>(a);
    am_step<0x141>(a);
    am_step<0x140>(a);
    ArgMax r = {readlane_f64(a.val, 0), __builtin_amdgcn_readlane(a.pos, 0), __builtin_amdgcn_readlane(a.idx, 0)};
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const double ov = readlane_f64(a.val, 16 * row);
        const int op = __builtin_amdgcn_readlane(a.pos, 16 * row);
        const int oi = __builtin_amdgcn_readlane(a.idx, 16 * row);
        if (am_better(ov, op, r.val, r.pos)) {
            r.val = ov;
            r.pos = op;
            r.idx = oi;
        }
    }
    return r;
}

// A value every lane holds identically (an LDS word read at a wave-uniform address) moved to scalar registers: tells the
// compiler that a branch on it is wave-uniform.  Without it such a branch is compiled as a divergent one (exec masks, flow
// blocks whose phi nodes copy every live register array: k_sb_factor_reg carried two copies of its candidate block).
__device__ __forceinline__ int uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double uniform_f64(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// src_lane must be wave-uniform (it is a loop counter everywhere it is used): v_readlane
__device__ __forceinline__ double wave_bcast(double x, int src_lane) {
    return readlane_f64(x, src_lane);
}

// LAPACK dlarfg without the safmin rescaling loop:
// beta = -sign(alpha) * hypot(alpha, xnorm), tau = (beta - alpha) / beta, v = x / (alpha - beta)
// The norm is taken from a plain sum of squares, so a column of norm below ~1e-140 has squares in (or under) the denormal
// range: tau and v would no longer describe an orthogonal reflector (measured: v'v - 2 = 5e-8 at norm 1e-159).  Such columns
// do occur: a tile whose content has just been rotated out exactly by the previous panel holds rounding dust, and the dust of
// the dust of ... falls by 1e-16 per column (a stack of upper triangles — the TSQR combine stage — does this).  They are
// treated as zero columns: H = I, the dust below the diagonal is dropped (callers store v = x * scale = 0), a perturbation of
// 1e-140 in absolute terms, inside the magnitude range DESIGN.md section 2 states.
constexpr double GN_TINY_NORM2 = 1e-280;
// dlarfg's three scalars from one reciprocal square root and one reciprocal, each refined by two Newton steps (and one correction
// of the square root), instead of the IEEE sequences of sqrt and two divisions: ~35 -> ~16 dependent operations on the critical
// path of EVERY reflector step of every factorisation kernel (panel kernel 351 -> 337 us per C2 launch).  The scalars agree with the
// IEEE forms to 1-2 ulp — the order of what the reduction trees of the dot products differ by anyway.  -DENLSIP_FAST_REFLECTOR=0
// builds the IEEE forms.
#ifndef ENLSIP_FAST_REFLECTOR
#define ENLSIP_FAST_REFLECTOR 1
#endif
struct Reflector {
    double beta, tau, scale;
};
__device__ __forceinline__ Reflector make_reflector(double alpha, double xnorm2) {
    Reflector r;
    if (xnorm2 == 0.0 || alpha * alpha + xnorm2 < GN_TINY_NORM2) {
        r.beta = alpha;
        r.tau = 0.0;
        r.scale = 0.0;
    } else {
#if ENLSIP_FAST_REFLECTOR
        const double sq = alpha * alpha + xnorm2;
        double y = __builtin_amdgcn_rsq(sq);                 // ~2^-26
        y = y * (1.5 - 0.5 * sq * y * y);
        y = y * (1.5 - 0.5 * sq * y * y);
        double nrm = sq * y;
        nrm = nrm + 0.5 * y * (sq - nrm * nrm);              // one correction of the square root
        r.beta = -copysign(nrm, alpha);
        r.tau = 1.0 + fabs(alpha) * y;                       // (beta - alpha) / beta = 1 + |alpha| / nrm
        const double dd = fabs(alpha) + nrm;                 // alpha - beta = sign(alpha) (|alpha| + nrm)
        double rc = __builtin_amdgcn_rcp(dd);
        rc = rc * (2.0 - dd * rc);
        rc = rc * (2.0 - dd * rc);
        r.scale = copysign(rc, alpha);
#else
        double nrm = sqrt(alpha * alpha + xnorm2);
        r.beta = -copysign(nrm, alpha);
        r.tau = (r.beta - alpha) / r.beta;
        r.scale = 1.0 / (alpha - r.beta);
#endif
    }
    return r;
}

// dlaqp2's downdate of a partial column norm o1 (o2 = the norm at its last recomputation) after a pivot step in which the column's
// entry in the pivot row became ajc:  temp = max(0, 1 - (|ajc| / o1)^2), temp2 = temp (o1 / o2)^2;  temp2 <= tol3z: the norm has to be
// recomputed (need), otherwise it becomes o1 sqrt(temp).  Caller guarantees o1 != 0.  Like make_reflector this sits on the dependent
// chain of every pivot step (the next arg-max waits for it): reciprocals and the root by rcp / rsq with two Newton steps.
struct NormDown {
    double vn1;        // the downdated norm (meaningless when need)
    bool need;
};
__device__ __forceinline__ NormDown norm_downdate(const double o1, const double o2, const double ajc, const double tol3z) {
    NormDown r;
#if ENLSIP_FAST_REFLECTOR
    double r1 = __builtin_amdgcn_rcp(o1), r2 = __builtin_amdgcn_rcp(o2);
    r1 = r1 * (2.0 - o1 * r1); r2 = r2 * (2.0 - o2 * r2);
    r1 = r1 * (2.0 - o1 * r1); r2 = r2 * (2.0 - o2 * r2);
    const double tq = fabs(ajc) * r1;
    double temp = 1.0 - tq * tq;
    temp = temp > 0.0 ? temp : 0.0;
    const double qq = o1 * r2;
    const double temp2 = temp * qq * qq;
    r.need = temp2 <= tol3z;
    double y = __builtin_amdgcn_rsq(temp);                   // temp == 0: need is set and the value below is not used
    y = y * (1.5 - 0.5 * temp * y * y);
    y = y * (1.5 - 0.5 * temp * y * y);
    double sr = temp * y;
    sr = sr + 0.5 * y * (temp - sr * sr);
    r.vn1 = o1 * sr;
#else
    double temp = 1.0 - (fabs(ajc) / o1) * (fabs(ajc) / o1);
    temp = temp > 0.0 ? temp : 0.0;
    const double qq = o1 / o2;
    const double temp2 = temp * qq * qq;
    r.need = temp2 <= tol3z;
    r.vn1 = o1 * sqrt(temp);
#endif
    return r;
}

// pseudo_rank (src/enlsip_functions.jl:17-31), diag accessed through a functor; serial, call
// from one lane.
// eps_abs: the threshold of the FIRST test, which is absolute.  It equals eps_rank except when the factor at hand belongs to
// inputs that the host scaled by a power of two (magnitudes beyond the range of plain sums of squares, enlsip_gn.hip
// "rescale"): the diagonal is then 2^shift times the true one and the test compares it with 2^shift eps_rank — the decision
// LAPACK's scaled norms would lead to on the unscaled data.  The second test is relative and needs nothing.
__device__ __forceinline__ double pseudo_rank_abs_threshold(double eps_rank, int abs_shift) {
    return abs_shift ? __builtin_ldexp(eps_rank, abs_shift) : eps_rank;
}
template <class DiagFn>
__device__ inline int pseudo_rank_serial(int len, double eps_rank, DiagFn diag, double eps_abs) {
    if (len == 0 || fabs(diag(0)) < eps_abs) return 0;
    const double tol = fabs(diag(0)) * sqrt((double)len) * eps_rank;
    int r = 1;
    while (r < len && fabs(diag(r - 1)) > tol) ++r;
    return r - ((r == len && fabs(diag(r - 1)) > tol) ? 0 : 1);
}
template <class DiagFn>
__device__ inline int pseudo_rank_serial(int len, double eps_rank, DiagFn diag) {
    return pseudo_rank_serial(len, eps_rank, diag, eps_rank);
}

}  // namespace gn
