// Stand-alone timing + check harness for the trailing-update kernel of the CAQR
// (gn_kernels_update_v4.hpp) on synthetic data with the C2 geometry; the reflector-by-reflector kernel
// of gn_kernels_caqr.hpp is timed beside it.  The check recomputes C - V (T' (V' C)) in plain host loops
// for sampled columns of problem 0.
// Build: hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -DENLSIP_GN_LAB -I enlsip.jl_amd/csrc -o tests/microbench/update_bench tests/microbench/update_bench.hip
//        (-DENLSIP_V4_ABLATE=2|3|4|5 for the timing-only ablations documented in the kernel header, -DENLSIP_V4_STAMPS for
//        per-phase wall-clock stamps of sample workgroups; the stamps themselves cost ~25 %)
// Run  : update_bench [batch=256] [panel=0] [level=0]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "gn_kernels_update_v4.hpp"

using namespace gn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static void launch_refl(const CaqrArgs& a, int groups, int ncols, int batch) {
    dim3 grid(groups, (ncols + 31) / 32, batch);
    hipLaunchKernelGGL(k_caqr_update_refl<8>, grid, dim3(256), 0, 0, a);
}

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 256;
    const int panel = argc > 2 ? atoi(argv[2]) : 0;
    const int level = argc > 3 ? atoi(argv[3]) : 0;
    // UB_M: row count (default 4096; ldw follows make_plan's rule); UB_PAIR_ONLY=1: time only the pair's far update
    const int m = getenv("UB_M") ? atoi(getenv("UB_M")) : 4096;
    const bool pair_only = getenv("UB_PAIR_ONLY") != nullptr;
    const int n = 512, t = 64, RPL = 8, F = 16;
    const int ldw = (m % 512 == 0) ? m + 32 : m;
    const long long sW = (long long)ldw * (n + 1 + 32);   // 32 spare columns (make_plan does the same)
    const int n2 = n - t, kp = n2;
    int nblocks = m / 32 - panel;           // level 0
    long long S = 32;
    for (int l = 0; l < level; ++l) { nblocks = (nblocks + F - 1) / F; S *= F; }
    const int groups = (nblocks + F - 1) / F;
    const int ntrail = n2 + 1 - (panel * 32 + 32);
    const long long sT = 64 * 32 * 32;
    printf("batch %d panel %d level %d: nblocks %d groups %d S %lld ntrail %d\n", batch, panel, level, nblocks, groups, S, ntrail);

    std::vector<double> hW((size_t)sW * 2), hT((size_t)sT);
    uint64_t seed = 12345;
    for (auto& x : hW) x = (double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    for (int b = 0; b < 64; ++b)
        for (int i = 0; i < 32; ++i)
            for (int l = 0; l < 32; ++l) hT[(size_t)b * 1024 + l + i * 32] = (l <= i) ? 0.05 * ((double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5) : 0.0;
    double *dW, *dT;
    ProbState* dS;
    CK(hipMalloc(&dW, (size_t)sW * batch * 8));
    CK(hipMalloc(&dT, (size_t)sT * batch * 8));
    CK(hipMalloc(&dS, sizeof(ProbState) * batch));
    std::vector<ProbState> hs(batch);
    for (auto& s : hs) { s = ProbState{}; s.rankA = t; s.n2 = n2; s.kp = kp; }
    CK(hipMemcpy(dS, hs.data(), sizeof(ProbState) * batch, hipMemcpyHostToDevice));
    for (int b = 0; b < batch; ++b) {
        CK(hipMemcpy(dW + (size_t)b * sW, hW.data() + (size_t)(b & 1) * sW, (size_t)sW * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dT + (size_t)b * sT, hT.data(), (size_t)sT * 8, hipMemcpyHostToDevice));
    }

    CaqrArgs a{};
    a.m = m; a.n = n; a.ldw = ldw; a.panel = panel; a.level = level; a.F = F; a.nblocks = nblocks; a.S = S; a.tOff = 0;
    a.W = dW; a.sW = sW; a.Tbuf = dT; a.sT = sT; a.state = dS; a.prob0 = 0;
    a.mode = level == 0 ? 0 : 1; a.base = 32 * panel; a.skip = 0; a.win = 0; a.pair = 0; a.tOff2 = 0;

    // ---- host check of problem 0: C - V (T' (V' C)) in plain loops on sampled columns ---------------------
    if (!pair_only) {
        launch_update_v4(RPL, a, groups, ntrail, 1, 0);
        CK(hipDeviceSynchronize());
        std::vector<double> r1((size_t)sW);
        CK(hipMemcpy(r1.data(), dW, r1.size() * 8, hipMemcpyDeviceToHost));
        const int r0 = 32 * panel, col0 = t + r0, first = r0 + 32;
        double maxd = 0, maxc = 0;
        for (int g = 0; g < groups; ++g) {
            const int nb = std::min(F, nblocks - g * F);
            const int rows = nb * 32;
            auto rowof = [&](int s) { return level == 0 ? (long long)r0 + (long long)g * F * 32 + s : (long long)r0 + ((long long)g * F + (s >> 5)) * S + (s & 31); };
            std::vector<double> V((size_t)rows * 32);
            for (int s = 0; s < rows; ++s)
                for (int j = 0; j < 32; ++j) {
                    double v;
                    if (level == 0) v = s > j ? hW[rowof(s) + (size_t)(col0 + j) * ldw] : (s == j ? 1.0 : 0.0);
                    else v = s < 32 ? (s == j ? 1.0 : 0.0) : ((s & 31) <= j ? hW[rowof(s) + (size_t)(col0 + j) * ldw] : 0.0);
                    V[(size_t)s * 32 + j] = v;
                }
            for (int cc = 0; cc < ntrail; cc += std::max(1, ntrail / 7)) {
                const size_t co = (size_t)(t + first + cc) * ldw;
                double w1[32], w2[32];
                for (int j = 0; j < 32; ++j) { w1[j] = 0; for (int s = 0; s < rows; ++s) w1[j] += V[(size_t)s * 32 + j] * hW[rowof(s) + co]; }
                for (int k = 0; k < 32; ++k) { w2[k] = 0; for (int l = 0; l <= k; ++l) w2[k] += hT[(size_t)g * 1024 + l + k * 32] * w1[l]; }
                for (int s = 0; s < rows; ++s) {
                    double x = hW[rowof(s) + co];
                    for (int k = 0; k < 32; ++k) x -= V[(size_t)s * 32 + k] * w2[k];
                    maxd = fmax(maxd, fabs(x - r1[rowof(s) + co]));
                    maxc = fmax(maxc, fabs(x - hW[rowof(s) + co]));
                }
            }
        }
        printf("host check (problem 0, sampled columns): max |kernel - loops| = %.3e   (max change by the update %.3e)\n", maxd, maxc);
        CK(hipMemcpy(dW, hW.data(), (size_t)sW * 8, hipMemcpyHostToDevice));
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (level == 0) {
        // ---- panel pair (panel, panel + 1): far columns get both level-0 reflectors in one pass; checked against two host passes ----
        CaqrArgs ap = a;
        ap.pair = 1; ap.win = 2; ap.tOff2 = 16;          // T blocks of the second panel: 16 .. 16 + groups - 1
        const int nfar = ntrail - 32;
        ap.skip_rhs = ((nfar - 1) % 32 == 0) ? 1 : 0;
        launch_update_v4(RPL, ap, groups, ntrail - ap.skip_rhs, 1, 0);
        CK(hipDeviceSynchronize());
        std::vector<double> r1((size_t)sW);
        CK(hipMemcpy(r1.data(), dW, r1.size() * 8, hipMemcpyDeviceToHost));
        const int r0 = 32 * panel, col0 = t + r0, first = r0 + 64;
        double maxd = 0, maxc = 0;
        for (int g = 0; g < groups; ++g) {
            const int nb = std::min(F, nblocks - g * F);
            const int rows = nb * 32;
            auto rowof = [&](int s) { return (long long)r0 + (long long)g * F * 32 + s; };
            std::vector<double> x(rows);
            for (int cc = 0; cc < nfar; cc += std::max(1, nfar / 9)) {
                const int ccol = (cc == 0) ? nfar - 1 : cc;       // include the carried right-hand side (last column)
                const size_t co = (size_t)(t + first + ccol) * ldw;
                for (int s = 0; s < rows; ++s) x[s] = hW[rowof(s) + co];
                for (int ai = 0; ai < 2; ++ai) {
                    const int dsh = 32 * ai;
                    double w1[32], w2[32];
                    auto V = [&](int s, int j) { return s > j + dsh ? hW[rowof(s) + (size_t)(col0 + 32 * ai + j) * ldw] : (s == j + dsh ? 1.0 : 0.0); };
                    for (int j = 0; j < 32; ++j) { w1[j] = 0; for (int s = 0; s < rows; ++s) w1[j] += V(s, j) * x[s]; }
                    for (int k = 0; k < 32; ++k) { w2[k] = 0; for (int l = 0; l <= k; ++l) w2[k] += hT[(size_t)(g + 16 * ai) * 1024 + l + k * 32] * w1[l]; }
                    for (int s = 0; s < rows; ++s) for (int k = 0; k < 32; ++k) x[s] -= V(s, k) * w2[k];
                }
                for (int s = 0; s < rows; ++s) {
                    maxd = fmax(maxd, fabs(x[s] - r1[rowof(s) + co]));
                    maxc = fmax(maxc, fabs(x[s] - hW[rowof(s) + co]));
                }
            }
        }
        // the pair's own 32 + 32 columns and everything left of them must be untouched
        double maxu = 0;
        for (size_t e = 0; e < (size_t)(t + first) * ldw; ++e) maxu = fmax(maxu, fabs(r1[e] - hW[e]));
        printf("PAIR host check (problem 0, sampled far columns + rhs): max |kernel - two host passes| = %.3e (change %.3e); untouched part differs by %.3e\n", maxd, maxc, maxu);
        CK(hipMemcpy(dW, hW.data(), (size_t)sW * 8, hipMemcpyHostToDevice));
        // timing: two plain passes (panel, then panel + 1 on its own trailing columns) against narrow + pair
        CaqrArgs a1 = a; a1.skip_rhs = ((ntrail - 1) % 32 == 0);
        CaqrArgs a2 = a; a2.panel = panel + 1; a2.base = 32 * (panel + 1); a2.nblocks = nblocks - 1; a2.tOff = 16; a2.skip_rhs = ((nfar - 1) % 32 == 0);
        const int g2 = (nblocks - 1 + F - 1) / F;
        CaqrArgs an = a; an.win = 1;
        auto timeit = [&](const char* name, auto&& go, double bytes_) {
            for (int i = 0; i < 2; ++i) go();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) go();
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            printf("%-34s %.3f ms   %.0f GB/s algorithmic (SURVEY 8d bytes)\n", name, ms, bytes_ / ms * 1e-6);
            return ms;
        };
        const double rk = (double)nblocks * 32;
        const double b1 = (double)batch * 8.0 * (2.0 * rk * ntrail + rk * 32), b2 = (double)batch * 8.0 * (2.0 * (rk - 32) * nfar + (rk - 32) * 32);
        const double bn = (double)batch * 8.0 * (2.0 * rk * 32 + rk * 32), bf = (double)batch * 8.0 * (2.0 * rk * nfar + rk * 32) + b2;
        if (pair_only) {
            // UB_EXACT=1: the grid covers the far columns only (no column block that exits at once: the library's grid spans ntrail)
            const int gcols = getenv("UB_EXACT") ? nfar - ap.skip_rhs : ntrail - ap.skip_rhs;
            const float t4 = timeit("pair, far columns", [&] { launch_update_v4(RPL, ap, groups, gcols, batch, 0); }, bf);
            const int ncb = (nfar - 1) / 32;
            printf("PAIRONLY m %d panel %d blocks %d nfar %d : %.3f ms  per column block %.4f ms  per (block x unit) %.3f ns\n", m, panel, nblocks, nfar, t4, t4 / ncb,
                   t4 / ncb / nblocks * 1e6);
            return 0;
        }
        for (int rep = 0; rep < 2; ++rep) {
            const float t1 = timeit("plain pass, first panel", [&] { launch_update_v4(RPL, a1, groups, ntrail - a1.skip_rhs, batch, 0); }, b1);
            const float t2 = timeit("plain pass, second panel", [&] { launch_update_v4(RPL, a2, g2, nfar - a2.skip_rhs, batch, 0); }, b2);
            const float t3 = timeit("narrow (second panel's columns)", [&] { launch_update_v4(RPL, an, groups, 32, batch, 0); }, bn);
            const float t4 = timeit("pair, far columns", [&] { launch_update_v4(RPL, ap, groups, ntrail - ap.skip_rhs, batch, 0); }, bf);
            printf("  two plain passes %.3f ms   narrow + pair %.3f ms   ratio %.3f\n", t1 + t2, t3 + t4, (t3 + t4) / (t1 + t2));
        }
#ifdef ENLSIP_V4_STAMPS
        {
        // the LAST kernel that ran was the pair's far update: entry 6 -> 0 start -> 1 product 1 (a) -> 4 reduction (a) -> 2 product 2 (a)
        // -> 3 product 1 (b) -> 7 reduction (b) -> 5 product 2 (b) + stores
        long long sp[64];
        CK(hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_v4_stamps), sizeof(sp)));
        for (int g = 0; g < 8; ++g)
            printf("PAIR wg (3, 5, %3d): setup %5.2f us | product 1 a %5.2f | reduce a %5.2f | product 2 a %5.2f | product 1 b %5.2f | reduce b %5.2f | product 2 b + stores %5.2f | total %5.2f\n",
                   32 * g + 7, (sp[g*8+0]-sp[g*8+6])*0.01, (sp[g*8+1]-sp[g*8+0])*0.01, (sp[g*8+4]-sp[g*8+1])*0.01, (sp[g*8+2]-sp[g*8+4])*0.01,
                   (sp[g*8+3]-sp[g*8+2])*0.01, (sp[g*8+7]-sp[g*8+3])*0.01, (sp[g*8+5]-sp[g*8+7])*0.01, (sp[g*8+5]-sp[g*8+6])*0.01);
        }
#endif
    }
#ifdef ENLSIP_V4_STAMPS
    if (groups <= 3) { const int zero = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_v4_stamp_x), &zero, sizeof(int))); }
#endif
    const int reps = 5;
    const double rows_k = (double)nblocks * 32;
    const double bytes = (double)batch * 8.0 * (2.0 * rows_k * ntrail + rows_k * 32);
    for (int which = 0; which < 2; ++which) {
        auto go = [&]() { which ? launch_update_v4(RPL, a, groups, ntrail, batch, 0) : launch_refl(a, groups, ntrail, batch); };
        for (int i = 0; i < 2; ++i) go();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) go();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%s: %.3f ms / launch   %.0f GB/s algorithmic   (%.2f us per workgroup-slot)\n", which ? "v4  " : "refl", ms, bytes / ms * 1e-6,
               ms * 1e3 / ((double)groups * ((ntrail + 31) / 32) * batch / 256.0));
    }
#ifdef ENLSIP_V4_STAMPS
    long long st[64];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_v4_stamps), sizeof(st)));
    for (int g = 0; g < 8; ++g)
        printf("wg (3, 5, %3d): setup %5.2f us | product1 (C, V loads + MFMA) %5.2f | partial->LDS %5.2f | barrier %5.2f | T-MFMA + barrier %5.2f | product2 + stores %5.2f | total %5.2f\n",
               32 * g + 7, (st[g*8+0]-st[g*8+6])*0.01, (st[g*8+1]-st[g*8+0])*0.01, (st[g*8+2]-st[g*8+1])*0.01, (st[g*8+3]-st[g*8+2])*0.01,
               (st[g*8+4]-st[g*8+3])*0.01, (st[g*8+5]-st[g*8+4])*0.01, (st[g*8+5]-st[g*8+6])*0.01);
#endif
    return 0;
}
