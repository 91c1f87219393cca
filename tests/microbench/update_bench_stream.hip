// Timing + check harness for the streaming form of the level-0 trailing update (update_stream_experiment.hpp) beside the fourth
// form (gn_kernels_update_v4.hpp), C2 geometry, full 32-column blocks only.
// Build: hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -I enlsip.jl_amd/csrc -I tests/microbench [-DENLSIP_STREAM_OCC=2|3] -o <exe> tests/microbench/update_bench_stream.hip
// Run  : update_bench_stream [batch=384] [panel=0]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "update_stream_experiment.hpp"

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
    const int ntrail = (n2 - (panel * 32 + 32)) / 32 * 32;          // full blocks only
    const int ncb = ntrail / 32;
    const long long sT = 64 * 32 * 32;
    const long long sW2 = (long long)groups * ncb * 1024;
    printf("stream: OCC %d | batch %d panel %d: nblocks %d groups %d ntrail %d (full blocks)\n", ENLSIP_STREAM_OCC, batch, panel, nblocks, groups, ntrail);
    if (nblocks % F) { printf("partial last tile not supported by this harness\n"); return 0; }
    std::vector<double> hW((size_t)sW), hT((size_t)sT), hW2((size_t)sW2);
    uint64_t seed = 4711;
    for (auto& x : hW) x = (double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    for (auto& x : hW2) x = 0.05 * ((double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    for (int b = 0; b < 64; ++b)
        for (int i = 0; i < 32; ++i)
            for (int l = 0; l < 32; ++l) hT[(size_t)b * 1024 + l + i * 32] = (l <= i) ? 0.05 * ((double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5) : 0.0;
    double *dW, *dT, *dW2, *dW1;
    ProbState* dS;
    CK(hipMalloc(&dW, (size_t)sW * batch * 8));
    CK(hipMalloc(&dT, (size_t)sT * batch * 8));
    CK(hipMalloc(&dW2, (size_t)sW2 * batch * 8));
    CK(hipMalloc(&dW1, (size_t)sW2 * batch * 8));
    CK(hipMalloc(&dS, sizeof(ProbState) * batch));
    std::vector<ProbState> hs(batch);
    for (auto& s : hs) { s = ProbState{}; s.rankA = t; s.n2 = n2 - 1; s.kp = kp; }     // n2 - 1: the v4 launch then covers exactly the full blocks
    for (auto& s : hs) s.n2 = panel * 32 + 32 + ntrail - 1;
    CK(hipMemcpy(dS, hs.data(), sizeof(ProbState) * batch, hipMemcpyHostToDevice));
    for (int b = 0; b < batch; ++b) {
        CK(hipMemcpy(dW + (size_t)b * sW, hW.data(), (size_t)sW * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dT + (size_t)b * sT, hT.data(), (size_t)sT * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW2 + (size_t)b * sW2, hW2.data(), (size_t)sW2 * 8, hipMemcpyHostToDevice));
    }
    CaqrArgs a{};
    a.m = m; a.n = n; a.ldw = ldw; a.panel = panel; a.level = 0; a.F = F; a.nblocks = nblocks; a.S = 32; a.tOff = 0;
    a.W = dW; a.sW = sW; a.Tbuf = dT; a.sT = sT; a.state = dS; a.prob0 = 0;
    StreamArgs sa{a, dW2, dW1, sW2, ncb};
    {   // check problem 0 on sampled columns: C_new = C + V W2 ; W1 = V' C_new
        launch_update_stream(sa, groups, ntrail, 1, 0);
        CK(hipDeviceSynchronize());
        std::vector<double> r1((size_t)sW), w1((size_t)sW2);
        CK(hipMemcpy(r1.data(), dW, r1.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(w1.data(), dW1, w1.size() * 8, hipMemcpyDeviceToHost));
        const int r0 = 32 * panel, col0 = t + r0, first = r0 + 32;
        double maxd = 0, maxw = 0, maxc = 0;
        for (int g = 0; g < groups; ++g) {
            const int rows = F * 32;
            auto rowof = [&](int s) { return (long long)r0 + (long long)g * F * 32 + s; };
            std::vector<double> V((size_t)rows * 32);
            for (int s = 0; s < rows; ++s)
                for (int j = 0; j < 32; ++j) V[(size_t)s * 32 + j] = s > j ? hW[rowof(s) + (size_t)(col0 + j) * ldw] : (s == j ? 1.0 : 0.0);
            for (int cc = 0; cc < ntrail; cc += std::max(1, ntrail / 9) + 1) {
                const int cb = cc / 32, jc = cc % 32;
                const double* W2 = hW2.data() + ((size_t)g * ncb + cb) * 1024;
                const size_t co = (size_t)(t + first + cc) * ldw;
                std::vector<double> cn(rows);
                for (int s = 0; s < rows; ++s) {
                    double x = hW[rowof(s) + co];
                    for (int k = 0; k < 32; ++k) x += V[(size_t)s * 32 + k] * W2[k * 32 + jc];
                    cn[s] = x;
                    maxd = fmax(maxd, fabs(x - r1[rowof(s) + co]));
                    maxc = fmax(maxc, fabs(x - hW[rowof(s) + co]));
                }
                for (int vc = 0; vc < 32; ++vc) {
                    double sacc = 0;
                    for (int s = 0; s < rows; ++s) sacc += V[(size_t)s * 32 + vc] * cn[s];
                    maxw = fmax(maxw, fabs(sacc - w1[((size_t)g * ncb + cb) * 1024 + vc * 32 + jc]));
                }
            }
        }
        printf("check (problem 0, sampled columns): max |C - loops| = %.3e (change %.3e)   max |W1 - loops| = %.3e\n", maxd, maxc, maxw);
        CK(hipMemcpy(dW, hW.data(), (size_t)sW * 8, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;
    const double rows_k = (double)nblocks * 32;
    const double bytes = (double)batch * 8.0 * (2.0 * rows_k * ntrail + rows_k * 32 + 1024.0 * groups);
    for (int which = 0; which < 4; ++which) {
        auto go = [&]() { (which & 1) ? launch_update_stream(sa, groups, ntrail, batch, 0) : launch_update_v4(RPL, a, groups, ntrail, batch, 0); };
        for (int i = 0; i < 2; ++i) go();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) go();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%s: %.4f ms / launch   %.0f GB/s algorithmic = %.3f of 8 TB/s\n", (which & 1) ? "stream" : "v4    ", ms, bytes / ms * 1e-6, bytes / ms * 1e-6 / 8000.0);
    }
    return 0;
}
