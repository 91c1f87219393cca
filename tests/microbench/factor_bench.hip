// Timing harness for the panel factorisation of the CAQR (k_caqr_factor<8,8>, gn_kernels_caqr.hpp) on the C2 geometry: `batch`
// problems x 8 tiles of 512 x 32, timed with HIP events; the timing-only ablations (-DENLSIP_FACTOR_ABLATE=1..6, see the kernel
// header) say where a step's time goes, -DENLSIP_FACTOR_STAMPS adds wall-clock stamps of one workgroup's phases.
// Build: hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -DENLSIP_GN_LAB -I enlsip.jl_amd/csrc -o tests/microbench/factor_bench tests/microbench/factor_bench.hip
// Run  : factor_bench [batch=384] [tiles=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gn_kernels_caqr.hpp"
using namespace gn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 384;
    const int tiles = argc > 2 ? atoi(argv[2]) : 8;
    const int m = 512 * tiles, n = 64, F = 16;
    const int ldw = m + 32;
    const long long sW = (long long)ldw * (n + 1 + 32), sT = (long long)(tiles > 64 ? tiles : 64) * 32 * 32;   // one T block per tile
    std::vector<double> hW((size_t)sW);
    uint64_t seed = 777;
    for (auto& x : hW) x = (double)(sm(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    double *dW, *dW0, *dT;
    ProbState* dS;
    CK(hipMalloc(&dW, (size_t)sW * batch * 8));
    CK(hipMalloc(&dW0, (size_t)sW * batch * 8));
    CK(hipMalloc(&dT, (size_t)sT * batch * 8));
    CK(hipMalloc(&dS, sizeof(ProbState) * batch));
    std::vector<ProbState> hs(batch);
    for (auto& s : hs) { s = ProbState{}; s.rankA = 0; s.n2 = n; s.kp = n; }
    CK(hipMemcpy(dS, hs.data(), sizeof(ProbState) * batch, hipMemcpyHostToDevice));
    for (int b = 0; b < batch; ++b) CK(hipMemcpy(dW0 + (size_t)b * sW, hW.data(), (size_t)sW * 8, hipMemcpyHostToDevice));
    CaqrArgs a{};
    a.m = m; a.n = n; a.ldw = ldw; a.panel = 0; a.level = 0; a.F = F; a.nblocks = m / 32; a.S = 32; a.tOff = 0;
    a.W = dW; a.sW = sW; a.Tbuf = dT; a.sT = sT; a.state = dS; a.prob0 = 0; a.mode = 0; a.base = 0; a.skip = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0.f;
    const int reps = 5;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipMemcpy(dW, dW0, (size_t)sW * batch * 8, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_caqr_factor<8, 8>), dim3(tiles, batch), dim3(512), 0, 0, a);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) tot += ms;
    }
    printf("ablate %d: k_caqr_factor<8,8> %d tiles x %d problems: %.1f us per launch (%.2f us per tile-slot round of 512)\n", ENLSIP_FACTOR_ABLATE,
           tiles, batch, 1e3 * tot / reps, 1e3 * tot / reps / ((double)tiles * batch / 512.0));
#ifdef ENLSIP_FACTOR_STAMPS
    long long st[16];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_factor_stamps), sizeof(st)));
    printf("stamps of workgroup (3, 7) [us]: load %.2f  loop %.2f  store %.2f  T %.2f\n", (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01);
#endif
    return 0;
}
