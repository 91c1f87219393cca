// Stand-alone timing harness for k_jq1_v2 (gn_kernels_q1_v2.hpp) on synthetic data with the C2 geometry.
// Build: hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -I enlsip.jl_amd/csrc -o tests/microbench/jq1_bench tests/microbench/jq1_bench.hip
//        (-DENLSIP_JQ1_ABLATE=1|2|3: timing-only ablations, see the kernel)
// Run  : jq1_bench [batch=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gn_kernels_q1_v2.hpp"
using namespace gn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 256;
    const int m = 4096, n = 512, t = 64, ldw = 4128;
    const long long sW = (long long)ldw * (n + 1 + 32), sJ = (long long)m * n;
    double *J, *W, *FA, *TA, *p1, *rx, *VT; ProbState* S;
    CK(hipMalloc(&VT, (size_t)n * 64 * batch * 8)); CK(hipMemset(VT, 0, (size_t)n * 64 * batch * 8));
    CK(hipMalloc(&J, sJ * batch * 8)); CK(hipMalloc(&W, sW * batch * 8));
    CK(hipMalloc(&FA, (size_t)n * t * batch * 8)); CK(hipMalloc(&TA, 4096 * batch * 8));
    CK(hipMalloc(&p1, 64 * batch * 8)); CK(hipMalloc(&rx, (size_t)m * batch * 8)); CK(hipMalloc(&S, sizeof(ProbState) * batch));
    CK(hipMemset(J, 0, sJ * batch * 8)); CK(hipMemset(W, 0, sW * batch * 8)); CK(hipMemset(FA, 0, (size_t)n * t * batch * 8));
    CK(hipMemset(TA, 0, 4096 * batch * 8)); CK(hipMemset(p1, 0, 64 * batch * 8)); CK(hipMemset(rx, 0, (size_t)m * batch * 8));
    std::vector<ProbState> hs(batch);
    for (auto& s : hs) { s = ProbState{}; s.rankA = t; s.n2 = n - t; s.kp = n - t; }
    CK(hipMemcpy(S, hs.data(), sizeof(ProbState) * batch, hipMemcpyHostToDevice));
    JQ1Args a{};
    a.m = m; a.n = n; a.kA = t; a.ldw = ldw; a.J = J; a.ldj = m; a.strideJ = sJ; a.rx = rx; a.stride_rx = m;
    a.FA = FA; a.sFA = (long long)n * t; a.TA = TA; a.sTA = 4096; a.p1 = p1; a.sP1 = 64; a.W = W; a.sW = sW; a.state = S; a.prob0 = 0; a.VT = VT; a.sVT = (long long)n * 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (!launch_jq1_v2(a, batch, 0)) { printf("shape rejected\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch_jq1_v2(a, batch, 0);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = 4.0 * m * n * t * batch, bytes = 16.0 * m * n * batch;
    printf("k_jq1_v2 batch %d: %.3f ms  %.1f TFLOP/s  %.2f TB/s  (%s)\n", batch, ms, flop / ms * 1e-9, bytes / ms * 1e-9, hipGetErrorString(hipGetLastError()));
#ifdef ENLSIP_JQ1_STAMPS
    long long st[64];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_jq1_stamps), sizeof(st)));
    for (int g = 0; g < 8; ++g)
        printf("wg (37, %3d): load-issue %5.2f us | phase1 %5.2f | wait %5.2f | reduce+T %5.2f | phase3 %5.2f | store+d %5.2f | total %5.2f\n", 32 * g + 3,
               (st[g*8+1]-st[g*8])*0.01, (st[g*8+2]-st[g*8+1])*0.01, (st[g*8+3]-st[g*8+2])*0.01, (st[g*8+4]-st[g*8+3])*0.01,
               (st[g*8+5]-st[g*8+4])*0.01, (st[g*8+6]-st[g*8+5])*0.01, (st[g*8+6]-st[g*8])*0.01);
#endif
    return 0;
}
