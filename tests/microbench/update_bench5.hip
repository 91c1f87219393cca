// Timing + check harness for the fifth form of the level-0 trailing update (gn_kernels_update_v5.hpp) beside the fourth
// (gn_kernels_update_v4.hpp) on synthetic data with the C2 geometry.  Vop (the tile's reflectors in operand order) is built on
// the host here; in the library the panel factorisation writes it.  The check compares the two kernels' results on all of problem 0.
// Build: hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -I enlsip.jl_amd/csrc [-DENLSIP_V5_NCT=1|2 -DENLSIP_V5_MT=0|1 -DENLSIP_V5_OCC=2|3|4]
//              -I tests/microbench -o <exe> tests/microbench/update_bench5.hip
// Run  : update_bench5 [batch=384] [panel=0]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "update_v5_experiment.hpp"

using namespace gn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 384;
    const int panel = argc > 2 ? atoi(argv[2]) : 0;
    const int m = 4096, n = 512, t = 64, RPL = 8, F = 16;
    const int ldw = 4128;
    const long long sW = (long long)ldw * (n + 1 + 32);
    const int n2 = n - t, kp = n2;
    const int nblocks = m / 32 - panel;
    const int groups = (nblocks + F - 1) / F;
    const int ntrail = n2 + 1 - (panel * 32 + 32);
    const long long sT = 64 * 32 * 32;
    const long long sVop = (long long)groups * F * V5_UNIT;
    printf("v5: NCT %d MT %d OCC %d | batch %d panel %d: nblocks %d groups %d ntrail %d\n", V5_NCT, (int)V5_MT, ENLSIP_V5_OCC, batch, panel, nblocks, groups, ntrail);

    std::vector<double> hW((size_t)sW * 2), hT((size_t)sT), hVop((size_t)sVop * 2, 0.0);
    uint64_t seed = 12345;
    for (auto& x : hW) x = (double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    for (int b = 0; b < 64; ++b)
        for (int i = 0; i < 32; ++i)
            for (int l = 0; l < 32; ++l) hT[(size_t)b * 1024 + l + i * 32] = (l <= i) ? 0.05 * ((double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5) : 0.0;
    const int r0 = 32 * panel, col0 = t + r0;
    for (int v = 0; v < 2; ++v)
        for (int g = 0; g < groups; ++g) {
            const int rows = std::min(F, nblocks - g * F) * 32;
            for (int s = 0; s < rows; ++s)
                for (int j = 0; j < 32; ++j) {
                    const long long row = (long long)r0 + (long long)g * F * 32 + s;
                    const double val = s > j ? hW[(size_t)v * sW + row + (size_t)(col0 + j) * ldw] : (s == j ? 1.0 : 0.0);
                    hVop[(size_t)v * sVop + ((size_t)g * F + (s >> 5)) * V5_UNIT + v5_vop_index(s & 31, j)] = val;
                }
        }
    double *dW, *dT, *dVop, *dW4;
    ProbState* dS;
    CK(hipMalloc(&dW, (size_t)sW * batch * 8));
    CK(hipMalloc(&dW4, (size_t)sW * 8));
    CK(hipMalloc(&dT, (size_t)sT * batch * 8));
    CK(hipMalloc(&dVop, (size_t)sVop * batch * 8));
    CK(hipMalloc(&dS, sizeof(ProbState) * batch));
    std::vector<ProbState> hs(batch);
    for (auto& s : hs) { s = ProbState{}; s.rankA = t; s.n2 = n2; s.kp = kp; }
    CK(hipMemcpy(dS, hs.data(), sizeof(ProbState) * batch, hipMemcpyHostToDevice));
    for (int b = 0; b < batch; ++b) {
        CK(hipMemcpy(dW + (size_t)b * sW, hW.data() + (size_t)(b & 1) * sW, (size_t)sW * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dT + (size_t)b * sT, hT.data(), (size_t)sT * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dVop + (size_t)b * sVop, hVop.data() + (size_t)(b & 1) * sVop, (size_t)sVop * 8, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(dW4, hW.data(), (size_t)sW * 8, hipMemcpyHostToDevice));

    CaqrArgs a{};
    a.m = m; a.n = n; a.ldw = ldw; a.panel = panel; a.level = 0; a.F = F; a.nblocks = nblocks; a.S = 32; a.tOff = 0;
    a.W = dW; a.sW = sW; a.Tbuf = dT; a.sT = sT; a.state = dS; a.prob0 = 0;
    V5Args a5{a, dVop, sVop};

    {   // problem 0: v5 in dW, v4 in dW4, compare everything
        launch_update_v5(RPL, a5, groups, ntrail, 1, 0);
        CaqrArgs a4 = a; a4.W = dW4;
        launch_update_v4(RPL, a4, groups, ntrail, 1, 0);
        CK(hipDeviceSynchronize());
        std::vector<double> r5((size_t)sW), r4((size_t)sW);
        CK(hipMemcpy(r5.data(), dW, r5.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r4.data(), dW4, r4.size() * 8, hipMemcpyDeviceToHost));
        double maxd = 0, maxc = 0;
        for (size_t i = 0; i < r5.size(); ++i) { maxd = fmax(maxd, fabs(r5[i] - r4[i])); maxc = fmax(maxc, fabs(r4[i] - hW[i])); }
        printf("check (problem 0, whole workspace): max |v5 - v4| = %.3e   (max change by the update %.3e)\n", maxd, maxc);
        CK(hipMemcpy(dW, hW.data(), (size_t)sW * 8, hipMemcpyHostToDevice));
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;
    const double rows_k = (double)nblocks * 32;
    const double bytes = (double)batch * 8.0 * (2.0 * rows_k * ntrail + rows_k * 32 + 1024.0 * groups);
    for (int which = 0; which < 4; ++which) {          // v4, v5, v4, v5: alternating
        auto go = [&]() { (which & 1) ? launch_update_v5(RPL, a5, groups, ntrail, batch, 0) : launch_update_v4(RPL, a, groups, ntrail, batch, 0); };
        for (int i = 0; i < 2; ++i) go();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) go();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%s: %.4f ms / launch   %.0f GB/s algorithmic = %.3f of 8 TB/s\n", (which & 1) ? "v5" : "v4", ms, bytes / ms * 1e-6, bytes / ms * 1e-6 / 8000.0);
    }
    return 0;
}
