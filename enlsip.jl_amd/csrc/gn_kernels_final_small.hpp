// Pivot + solve stage for SMALL factors (kp <= 64 rows, n2 + 1 <= 64 columns): ONE WAVE per problem, no barriers.
// Same contract as k_pivot_solve (gn_kernels_final.hpp, refactor == 1):
//   R0 P = Qt Rt, rankJ2 = pseudo_rank(diag Rt)         src/enlsip_functions.jl:223-224
//   dp2 = U(Rt) \ d[1:dimJ2], p2 = [dp2;0][invperm]     :136-137 / :147-148
//   p = F_A.Q [p1; p2]                                   :151
//
// Layout: lane c owns COLUMN c of [R0 | z] (lane n2 = the carried right-hand side), rows in registers: gn_wave_qrcp.hpp.
#pragma once
#include "gn_kernels_final.hpp"
#include "gn_wave_qrcp.hpp"

namespace gn {

// LDS (doubles): tmp[kpm * 65] vbuf[64] dg[64] pbuf[nv]
inline size_t final_small_lds_bytes(int kpm, int nv) { return (size_t)(kpm * 65 + 64 + 64 + nv + 8) * 8; }

#ifdef GN_PS_STAMPS   // timing-only build: phase stamps (100 MHz wall clock) of problem 0 in the last entries of d_out
#define GN_PS_STAMP(i) do { if (prob == 0 && ln == 0 && a.d_out) stamps[i] = (double)wall_clock64(); } while (0)
#else
#define GN_PS_STAMP(i) do { } while (0)
#endif

template <int NR>
#ifndef ENLSIP_PS_OCC
#define ENLSIP_PS_OCC 0           // measured: 4 waves per SIMD (128 registers, 120 B scratch) more than doubles the kernel's time
#endif
#ifndef ENLSIP_PS64_OCC
#define ENLSIP_PS64_OCC 0         // measured: 2 waves per SIMD = 152 spilled registers, C3 pivot stage 0.19 -> 0.36 ms
#endif
__global__ __launch_bounds__(64, (NR == 32 && ENLSIP_PS_OCC) ? ENLSIP_PS_OCC : ((NR == 64 && ENLSIP_PS64_OCC) ? ENLSIP_PS64_OCC : 1)) void k_pivot_small(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int prob = blockIdx.x + a.prob0;
    ProbState* stp = a.state + prob;
    const int rankA = stp->rankA, n2 = stp->n2, kp = stp->kp;
    // a problem whose J2 is wider than the launch shape (rank-deficient A) is redone by the caller with the true width
    if (kp > NR || n2 + 1 > 64 || kp * 65 > a.matd || (a.n2cap > 0 && n2 > a.n2cap)) return;
    double* tmp = smem;
    double* vbuf = tmp + a.matd;
    double* dg = vbuf + 64;
    double* pbuf = dg + 64;
    const int n = a.n, m = a.m, t = a.t, kA = a.kA, ldr = a.ldr, ldw = a.ldw;
    const double* W = a.W + prob * a.sW;
    double* Rt = a.Rt + prob * a.sRt;
    double* tauJ = a.tauJ + prob * a.sTauJ;
    long long* jpvtJ = a.jpvtJ + prob * a.sJJ;
    const double* FA = a.FA + prob * a.sFA;
    const double* tauA = a.tauA + prob * a.sTauA;
    const double* p1 = a.p1 + prob * a.sP1;
    const double* bvec = a.bvec + prob * a.sB;
    const int ln = threadIdx.x;
    int status = 0;

    GN_PS_STAMP(0);
    // ---- R0 (upper trapezoid) and z: columns through LDS so that the global loads run along rows ---------------
    // 32 independent loads in flight, then 32 LDS writes (the column registers are not live yet): with eight per round the 57
    // columns of a C3 problem were eight HBM round trips in a row, 21 us of the kernel's 119
    // The loads are UNCONDITIONAL (row and column clamped into the matrix) and masked with AND afterwards: a guarded load
    // (`ok ? W[i] : 0`) is compiled as a branch around the load with a wait right behind it, so that the 57 columns of a C3 problem
    // were 57 HBM round trips in a row.
    constexpr int EXB = 32;
    const int lnc = ln < kp ? ln : 0;
    for (int c0 = 0; c0 <= n2; c0 += EXB) {
        double v[EXB];
#pragma unroll
        for (int u = 0; u < EXB; ++u) {
            const int c = c0 + u;
            const int col = (c < n2) ? rankA + c : n;          // uniform; columns past n2 read the carried right-hand side again
            v[u] = W[lnc + (size_t)col * ldw];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < EXB; ++u) {
            const int c = c0 + u;
            const bool keep = (ln < kp) && ((c < n2) ? (ln <= c) : (c == n2));
            const unsigned long long mk = keep ? ~0ull : 0ull;
            v[u] = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v[u]) & mk));
        }
#pragma unroll
        for (int u = 0; u < EXB; ++u)
            if (ln < kp && c0 + u <= n2) tmp[ln * 65 + c0 + u] = v[u];
    }
    if (a.zsave && ln < kp) a.zsave[prob * a.sZ + ln] = W[ln + (size_t)n * ldw];
    wave_mem_sync();
    double x[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) x[r] = (r < kp && ln <= n2) ? tmp[r * 65 + ln] : 0.0;
    WaveQrcp q{};
    q.rows = kp; q.k = kp; q.ncand = n2; q.npart = n2 + 1;
    q.tmp = tmp; q.vbuf = vbuf; q.dg = dg; q.F = Rt; q.ldf = ldr; q.tau = tauJ;
    int mypos = ln;
    GN_PS_STAMP(1);
    if (kp > 0) wave_qrcp<NR, false>(x, q, ln, mypos);
    GN_PS_STAMP(2);

    // ---- permutation, upper parts of the columns, rank ---------------------------------------------------------------
    if (ln < n2) jpvtJ[mypos] = ln + 1;
    // lane-of-position table (positions 0..n2): lp of lane i = lane that sits at position i
    int* lpos = reinterpret_cast<int*>(vbuf);
    wave_mem_sync();
    if (ln <= n2) lpos[mypos] = ln;
    wave_mem_sync();
    const int lp = (ln <= n2) ? lpos[ln] : 0;
    wave_qrcp_store_upper(q, ln, lp, n2 + 1, Rt, ldr);
    const int rankJ2 = wave_pseudo_rank(dg, kp, a.eps_rank, ln, pseudo_rank_abs_threshold(a.eps_rank, a.abs_shift));
    int dimJ2 = (a.dimJ2_override >= 0) ? a.dimJ2_override : rankJ2;
    dimJ2 = dimJ2 < kp ? dimJ2 : kp;      // as k_pivot_solve: never past the factor
    GN_PS_STAMP(3);

    // ---- dp2 = U(Rt[1:dimJ2,1:dimJ2]) \ d[1:dimJ2]: lane r carries row r of the right-hand side ------------------------
    double zw = (ln < dimJ2 && ln < kp) ? tmp[ln * 65 + n2] : 0.0;
    // lane i divides ITS diagonal entry once, before the loop: a division on the critical path of every step otherwise
    const double dmine = (ln < kp && ln <= n2) ? tmp[ln * 65 + lp] : 1.0;
    const double rinv = 1.0 / dmine;
    if ((ln < dimJ2 && ln < kp) && dmine == 0.0) status |= 1;
    for (int i = (dimJ2 < kp ? dimJ2 : kp) - 1; i >= 0; --i) {
        const int li = __builtin_amdgcn_readlane(lp, i);
        const double yi = wave_bcast(zw, i) * wave_bcast(rinv, i);
        if (ln == i) zw = yi;
        if (ln < i) zw -= tmp[ln * 65 + li] * yi;
    }
    status = __any(status) ? 1 : 0;
    GN_PS_STAMP(4);
    // y = [p1 ; p2],  p2[pJ[i]-1] = (i < dimJ2 ? dp2[i] : 0)
    for (int i = ln; i < rankA; i += WAVE) pbuf[i] = p1[i];
    if (ln < n2) pbuf[rankA + lp] = (ln < dimJ2) ? zw : 0.0;
    // p = F_A.Q * y
    wave_mem_sync();
    if (n <= 64) wave_apply_reflectors_reg<false, 1>(FA, n, tauA, kA, n, pbuf);      // reflectors prefetched, tau in one request
    else if (n <= 128) wave_apply_reflectors_reg<false, 2>(FA, n, tauA, kA, n, pbuf);
    else wave_apply_reflectors<false>(FA, n, tauA, kA, n, pbuf);
    wave_mem_sync();
    GN_PS_STAMP(5);
    if (a.p_out)
        for (int i = ln; i < n; i += WAVE) a.p_out[prob * a.sPo + i] = pbuf[i];
    if (a.b_out)
        for (int i = ln; i < t; i += WAVE) a.b_out[prob * a.sBo + i] = bvec[i];
    if (a.d_out) {
        for (int i = ln; i < m; i += WAVE)
            a.d_out[prob * a.sDo + i] = (i < kp) ? tmp[i * 65 + n2] : W[i + (size_t)n * ldw];
    }
    if (a.jA_out)
        for (int i = ln; i < t; i += WAVE) a.jA_out[prob * a.sJAo + i] = a.jpvtA[prob * a.sJA + i];
    if (a.jL_out)
        for (int i = ln; i < kA; i += WAVE) a.jL_out[prob * a.sJLo + i] = a.jpvtL[prob * a.sJL + i];
    if (a.jJ_out && ln < n2) a.jJ_out[prob * a.sJJo + mypos] = ln + 1;
    if (ln == 0) {
        stp->rankJ2 = rankJ2;
        stp->dimJ2 = dimJ2;
        stp->status |= status;
    }
#ifdef GN_PS_STAMPS
    GN_PS_STAMP(6);
    if (prob == 0 && ln == 0 && a.d_out)
        for (int i = 0; i < 7; ++i) a.d_out[m - 8 + i] = stamps[i];
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same stage with TWO problems per wave (gn_wave_qrcp.hpp, second part): at most 32 rows of R0 and n2 + 1 <= 32 columns each —
// the C5 shape (kp = 28, 29 columns), where the one-problem form leaves 35 lanes of every wave idle.  Lanes 0..31 / 32..63 carry
// the columns of problem 2 b / 2 b + 1 of the launch; after the factorisation the wave finishes the two problems one after the
// other with all of its lanes as workers (back substitution, p = F_A.Q [p1; p2], output records), exactly as k_pivot_small does.
// LDS (doubles): tmp[kpm * 65] vbuf[128] dg[128] lpos[64 ints] pbuf[nv]
// ---------------------------------------------------------------------------------------------------------------------------------
inline size_t final_small2_lds_bytes(int kpm, int nv) { return (size_t)(kpm * 65 + 128 + 128 + 32 + 2 * nv + 8) * 8; }   // pbuf: one per half

__global__ __launch_bounds__(64, 3) void k_pivot_small2(FinalArgs a, int nprob) {      // 3 waves per SIMD: 168 registers
    constexpr int NR = 32;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int kpm = a.matd;                            // rows of the finished-row image (launch-wide bound on kp, <= 32)
    double* tmp = smem;
    double* vbuf = tmp + kpm * 65;
    double* dg = vbuf + 128;
    int* lpos = reinterpret_cast<int*>(dg + 128);
    double* pbuf = dg + 128 + 32;
    const int ln = threadIdx.x, lh = ln & 31, hb = ln & 32, half = ln >> 5;
    const int pidx = 2 * blockIdx.x + half;            // the lane's problem inside the launch
    const bool have = pidx < nprob;
    const int prob_l = a.prob0 + (have ? pidx : 0);
    const int n = a.n, m = a.m, t = a.t, kA = a.kA, ldr = a.ldr, ldw = a.ldw;
    int rankA = 0, n2 = 0, kp = 0;
    bool ok = false;
    if (have) {
        const ProbState st = a.state[prob_l];
        rankA = st.rankA; n2 = st.n2; kp = st.kp;
        // a problem whose J2 is wider than the launch shape (rank-deficient A) is redone by the caller with the true width
        ok = kp <= NR && kp <= kpm && n2 + 1 <= 32 && !(a.n2cap > 0 && n2 > a.n2cap);
        if (!ok) kp = 0;
    }
    const double* W = a.W + (size_t)prob_l * a.sW;
    double* Rt = a.Rt + (size_t)prob_l * a.sRt;
    // ---- R0 (upper trapezoid) and z: lane = row for the loads, columns through the LDS image ------------------------------
    for (int c0 = 0; c0 < 32; c0 += 16) {        // 16 independent loads in flight (see k_pivot_small)
        double v[16];
        // unconditional loads from clamped addresses, masked with AND (see k_pivot_small)
        const int lhc = (ok && lh < kp) ? lh : 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = c0 + u;
            const int col = (c < n2) ? rankA + c : n;
            v[u] = W[lhc + (size_t)col * ldw];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = c0 + u;
            const bool keep = ok && (lh < kp) && ((c < n2) ? (lh <= c) : (c == n2));
            const unsigned long long mk = keep ? ~0ull : 0ull;
            v[u] = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v[u]) & mk));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lh < kpm) tmp[lh * 65 + hb + c0 + u] = v[u];
    }
    if (a.zsave && ok && lh < kp) a.zsave[(size_t)prob_l * a.sZ + lh] = W[lh + (size_t)n * ldw];
    wave_mem_sync();
    double x[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) x[r] = (r < kp && lh <= n2) ? tmp[r * 65 + ln] : 0.0;
    wave_mem_sync();                                   // every lane has taken its column before the image receives finished rows
    WaveQrcp2 q{};
    q.rows = kp; q.k = kp; q.ncand = ok ? n2 : 0; q.npart = ok ? n2 + 1 : 0;
    q.tmp = tmp; q.vbuf = vbuf; q.dg = dg; q.F = Rt; q.ldf = ldr; q.tau = a.tauJ + (size_t)prob_l * a.sTauJ;
    const int k0 = __builtin_amdgcn_readlane(kp, 0), k1 = __builtin_amdgcn_readlane(kp, 32);
    const int kmax = k0 > k1 ? k0 : k1;
    int mypos = lh;
    if (kmax > 0) wave_qrcp2<NR>(x, q, kmax, ln, mypos);
    wave_mem_sync();

    // ---- n <= 128: both problems finish SIDE BY SIDE, each on its own 32 lanes (the serial form below ran this tail — upper parts,
    //      rank, back substitution, Q1 y, outputs: about as long as the 28 pivot steps of a C5 problem — twice per wave) ------------
    if (n <= 128) {
        const bool act = have && ok;
        const size_t pb = (size_t)prob_l;
        ProbState* stp = a.state + pb;
        long long* jpvtJ = a.jpvtJ + pb * a.sJJ;
        const double* FA = a.FA + pb * a.sFA;
        const double* tauA = a.tauA + pb * a.sTauA;
        const double* p1 = a.p1 + pb * a.sP1;
        const double* bvec = a.bvec + pb * a.sB;
        const double* dgh = dg + 2 * hb;
        int* lposh = lpos + hb;
        double* pbufh = pbuf + half * a.nv;
        int status = 0;
        if (act && lh < n2) jpvtJ[mypos] = lh + 1;
        wave_mem_sync();
        if (act && lh <= n2) lposh[mypos] = lh;
        wave_mem_sync();
        const int lp = (act && lh <= n2) ? lposh[lh] : 0;        // lane i of the half: the column that sits at position i
        const int n2a = act ? n2 : -1;
        const int n2m = max(__builtin_amdgcn_readlane(n2a, 0), __builtin_amdgcn_readlane(n2a, 32));
        for (int P = 0; P <= n2m; ++P) {                           // upper parts of the factor columns (and the carried one)
            const bool on = P <= n2a;
            const int src = on ? lposh[P] : 0;
            if (on && lh < kp && lh <= P) Rt[lh + (size_t)P * ldr] = tmp[lh * 65 + hb + src];
        }
        int rankJ2 = 0;
        {
            const double d0 = (kp > 0) ? fabs(dgh[0]) : 0.0;
            const double tol = d0 * sqrt((double)(kp > 0 ? kp : 1)) * a.eps_rank;
            const bool fail = (lh < kp) && !(fabs(dgh[lh < kp ? lh : 0]) > tol);
            const unsigned mh = (unsigned)(__ballot(fail) >> hb);
            if (kp > 0 && !(d0 < pseudo_rank_abs_threshold(a.eps_rank, a.abs_shift))) rankJ2 = mh ? (int)__builtin_ctz(mh) : kp;
        }
        int dimJ2 = (a.dimJ2_override >= 0) ? a.dimJ2_override : rankJ2;
        dimJ2 = dimJ2 < kp ? dimJ2 : kp;
        if (!act) dimJ2 = 0;
        // dp2 = U(Rt[1:dimJ2,1:dimJ2]) \ d[1:dimJ2]: lane r of the half carries row r of the right-hand side
        double zw = (lh < dimJ2) ? tmp[lh * 65 + hb + n2] : 0.0;
        const double dmine = (lh < kp && lh <= n2 && act) ? tmp[lh * 65 + hb + lp] : 1.0;     // one division per row, before the loop
        const double rinv = 1.0 / dmine;
        if (lh < dimJ2 && dmine == 0.0) status |= 1;
        const int dmx = max(__builtin_amdgcn_readlane(dimJ2, 0), __builtin_amdgcn_readlane(dimJ2, 32));
        for (int i = dmx - 1; i >= 0; --i) {
            const bool on = i < dimJ2;
            const int li = on ? lposh[i] : 0;
            const double y0 = readlane_f64(zw, i) * readlane_f64(rinv, i);
            const double y1 = readlane_f64(zw, 32 + i) * readlane_f64(rinv, 32 + i);
            const double yi = half ? y1 : y0;
            if (on && lh == i) zw = yi;
            if (on && lh < i) zw -= tmp[lh * 65 + hb + li] * yi;
        }
        // y = [p1 ; p2],  p2[pJ[i]-1] = (i < dimJ2 ? dp2[i] : 0)
        const int rAa = act ? rankA : 0;
        const int rAm = max(__builtin_amdgcn_readlane(rAa, 0), __builtin_amdgcn_readlane(rAa, 32));
        for (int i = lh; i < rAm; i += 32)
            if (i < rAa) pbufh[i] = p1[i];
        if (act && lh < n2) pbufh[rankA + lp] = (lh < dimJ2) ? zw : 0.0;
        wave_mem_sync();
        {   // p = F_A.Q * y, per half: lane lh owns entries lh + 32 i of its problem's vector; reflectors prefetched one ahead
            constexpr int RH = 4;
            double xr[RH], vn[RH];
#pragma unroll
            for (int i = 0; i < RH; ++i) xr[i] = (act && lh + 32 * i < n) ? pbufh[lh + 32 * i] : 0.0;
            auto fetch = [&](int j, double (&v)[RH]) {
#pragma unroll
                for (int i = 0; i < RH; ++i) {
                    const int r = lh + 32 * i;
                    v[i] = (act && r > j && r < n) ? FA[r + (size_t)j * n] : 0.0;
                }
            };
            // every tau in one request: lane lh of a half holds tau[lh + 32 q], q = 0..3 (kA <= n <= 128 on this path)
            double tlq[RH];
#pragma unroll
            for (int qq = 0; qq < RH; ++qq) tlq[qq] = (act && lh + 32 * qq < kA) ? tauA[lh + 32 * qq] : 0.0;
            if (kA > 0) fetch(kA - 1, vn);
            for (int sidx = 0; sidx < kA; ++sidx) {
                const int j = kA - 1 - sidx;
                double v[RH];
#pragma unroll
                for (int i = 0; i < RH; ++i) v[i] = (lh + 32 * i == j) ? 1.0 : vn[i];
                if (sidx + 1 < kA) fetch(j - 1, vn);
                const int jq = j >> 5, jl = j & 31;                       // uniform
                double tsel = tlq[0];
#pragma unroll
                for (int qq = 1; qq < RH; ++qq) tsel = (jq == qq) ? tlq[qq] : tsel;
                const double t0 = readlane_f64(tsel, jl), t1 = readlane_f64(tsel, 32 + jl);
                const double tj = half ? t1 : t0;
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RH; ++i) dot += v[i] * xr[i];
                dot = (tj != 0.0) ? half_allsum(dot) * tj : 0.0;               // tj = 0 (no reflector / inactive half): nothing happens
#pragma unroll
                for (int i = 0; i < RH; ++i) xr[i] -= dot * v[i];
            }
#pragma unroll
            for (int i = 0; i < RH; ++i)
                if (act && lh + 32 * i < n) pbufh[lh + 32 * i] = xr[i];
        }
        wave_mem_sync();
        if (act) {
            if (a.p_out)
                for (int i = lh; i < n; i += 32) a.p_out[pb * a.sPo + i] = pbufh[i];
            if (a.b_out)
                for (int i = lh; i < t; i += 32) a.b_out[pb * a.sBo + i] = bvec[i];
            if (a.d_out)
                for (int i = lh; i < m; i += 32)
                    a.d_out[pb * a.sDo + i] = (i < kp) ? tmp[i * 65 + hb + n2] : W[i + (size_t)n * ldw];
            if (a.jA_out)
                for (int i = lh; i < t; i += 32) a.jA_out[pb * a.sJAo + i] = a.jpvtA[pb * a.sJA + i];
            if (a.jL_out)
                for (int i = lh; i < kA; i += 32) a.jL_out[pb * a.sJLo + i] = a.jpvtL[pb * a.sJL + i];
            if (a.jJ_out && lh < n2) a.jJ_out[pb * a.sJJo + mypos] = lh + 1;
        }
        const int st0 = __any(status && !half) ? 1 : 0, st1 = __any(status && half) ? 1 : 0;
        if (act && lh == 0) {
            stp->rankJ2 = rankJ2;
            stp->dimJ2 = dimJ2;
            stp->status |= half ? st1 : st0;
        }
        return;
    }
    // ---- the wave finishes problem 0, then problem 1 -----------------------------------------------------------------------------
    for (int hsel = 0; hsel < 2; ++hsel) {
        const int hbu = 32 * hsel;
        if (2 * blockIdx.x + hsel >= nprob) continue;
        if (!__builtin_amdgcn_readlane((int)ok, hbu)) continue;
        const int prob = a.prob0 + 2 * blockIdx.x + hsel;
        const int n2u = __builtin_amdgcn_readlane(n2, hbu), kpu = __builtin_amdgcn_readlane(kp, hbu), rAu = __builtin_amdgcn_readlane(rankA, hbu);
        const int myposh = __shfl(mypos, hbu + lh);     // the half's positions in lanes 0..31
        ProbState* stp = a.state + prob;
        double* Rtu = a.Rt + (size_t)prob * a.sRt;
        long long* jpvtJ = a.jpvtJ + (size_t)prob * a.sJJ;
        const double* FA = a.FA + (size_t)prob * a.sFA;
        const double* tauA = a.tauA + (size_t)prob * a.sTauA;
        const double* p1 = a.p1 + (size_t)prob * a.sP1;
        const double* bvec = a.bvec + (size_t)prob * a.sB;
        const double* Wu = a.W + (size_t)prob * a.sW;
        const double* dgh = dg + 2 * hbu;
        const bool inh = ln < 32;
        int status = 0;
        if (inh && ln < n2u) jpvtJ[myposh] = ln + 1;
        wave_mem_sync();
        if (inh && ln <= n2u) lpos[myposh] = ln;
        wave_mem_sync();
        const int lp = (inh && ln <= n2u) ? lpos[ln] : 0;       // lane i: the column that sits at position i
        for (int P = 0; P <= n2u; ++P) {                          // upper parts of the factor columns (and the carried one)
            const int src = lpos[P];
            if (ln < kpu && ln <= P) Rtu[ln + (size_t)P * ldr] = tmp[ln * 65 + hbu + src];
        }
        int rankJ2 = 0;
        if (kpu > 0) {
            const double d0 = fabs(dgh[0]);
            if (!(d0 < pseudo_rank_abs_threshold(a.eps_rank, a.abs_shift))) {
                const double tol = d0 * sqrt((double)kpu) * a.eps_rank;
                const bool fail = (ln < kpu) && !(fabs(dgh[ln < kpu ? ln : 0]) > tol);
                const unsigned long long mk = __ballot(fail);
                rankJ2 = mk ? (int)__builtin_ctzll(mk) : kpu;
            }
        }
        int dimJ2 = (a.dimJ2_override >= 0) ? a.dimJ2_override : rankJ2;
        dimJ2 = dimJ2 < kpu ? dimJ2 : kpu;
        // dp2 = U(Rt[1:dimJ2,1:dimJ2]) \ d[1:dimJ2]: lane r carries row r of the right-hand side
        double zw = (ln < dimJ2) ? tmp[ln * 65 + hbu + n2u] : 0.0;
        const double dmine = (ln < kpu && ln <= n2u) ? tmp[ln * 65 + hbu + lp] : 1.0;     // one division per row, before the loop
        const double rinv = 1.0 / dmine;
        if (ln < dimJ2 && dmine == 0.0) status |= 1;
        for (int i = dimJ2 - 1; i >= 0; --i) {
            const int li = lpos[i];
            const double yi = wave_bcast(zw, i) * wave_bcast(rinv, i);
            if (ln == i) zw = yi;
            if (ln < i) zw -= tmp[ln * 65 + hbu + li] * yi;
        }
        status = __any(status) ? 1 : 0;
        // y = [p1 ; p2],  p2[pJ[i]-1] = (i < dimJ2 ? dp2[i] : 0)
        for (int i = ln; i < rAu; i += WAVE) pbuf[i] = p1[i];
        if (inh && ln < n2u) pbuf[rAu + lp] = (ln < dimJ2) ? zw : 0.0;
        wave_mem_sync();
        if (n <= 64) wave_apply_reflectors_reg<false, 1>(FA, n, tauA, kA, n, pbuf);      // p = F_A.Q * y
        else if (n <= 128) wave_apply_reflectors_reg<false, 2>(FA, n, tauA, kA, n, pbuf);
        else wave_apply_reflectors<false>(FA, n, tauA, kA, n, pbuf);
        wave_mem_sync();
        if (a.p_out)
            for (int i = ln; i < n; i += WAVE) a.p_out[(size_t)prob * a.sPo + i] = pbuf[i];
        if (a.b_out)
            for (int i = ln; i < t; i += WAVE) a.b_out[(size_t)prob * a.sBo + i] = bvec[i];
        if (a.d_out)
            for (int i = ln; i < m; i += WAVE)
                a.d_out[(size_t)prob * a.sDo + i] = (i < kpu) ? tmp[i * 65 + hbu + n2u] : Wu[i + (size_t)n * ldw];
        if (a.jA_out)
            for (int i = ln; i < t; i += WAVE) a.jA_out[(size_t)prob * a.sJAo + i] = a.jpvtA[(size_t)prob * a.sJA + i];
        if (a.jL_out)
            for (int i = ln; i < kA; i += WAVE) a.jL_out[(size_t)prob * a.sJLo + i] = a.jpvtL[(size_t)prob * a.sJL + i];
        if (a.jJ_out && inh && ln < n2u) a.jJ_out[(size_t)prob * a.sJJo + myposh] = ln + 1;
        if (ln == 0) {
            stp->rankJ2 = rankJ2;
            stp->dimJ2 = dimJ2;
            stp->status |= status;
        }
        wave_mem_sync();                               // lpos / pbuf are reused by the second problem
    }
}

// Returns false when the launch shape is outside the kernel's range (the caller uses k_pivot_solve).
inline bool launch_pivot_small(int kp_launch, int n2_launch, int batch, hipStream_t s, FinalArgs a) {
    if (kp_launch > 64 || n2_launch + 1 > 64 || a.refactor != 1 || a.dsrc) return false;
    const long long mx = a.n > a.t ? a.n : a.t;
    a.nv = (int)((mx + 7) / 8 * 8);
    a.matd = (kp_launch > 0 ? kp_launch : 1) * 65 + 1;
    a.matd = (a.matd + 1) / 2 * 2;
    if (kp_launch <= 32 && n2_launch + 1 <= 32 && batch > 1) {      // two problems per wave
        a.matd = kp_launch > 0 ? kp_launch : 1;
        GN_ROUTE(ENLSIP_GN_ROUTE_PIVOT_WAVE2);
        hipLaunchKernelGGL(k_pivot_small2, dim3((batch + 1) / 2), dim3(64), final_small2_lds_bytes(a.matd, a.nv), s, a, batch);
        return true;
    }
    const size_t lds = final_small_lds_bytes(0, a.nv) + (size_t)a.matd * 8;
    if (kp_launch <= 32) { GN_ROUTE(ENLSIP_GN_ROUTE_PIVOT_WAVE32); hipLaunchKernelGGL(k_pivot_small<32>, dim3(batch), dim3(64), lds, s, a); }
    else { GN_ROUTE(ENLSIP_GN_ROUTE_PIVOT_WAVE64); hipLaunchKernelGGL(k_pivot_small<64>, dim3(batch), dim3(64), lds, s, a); }
    return true;
}

}  // namespace gn
