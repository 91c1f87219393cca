// Latency of a dependency edge between two HIP streams on MI355X: kernel A on stream 1 -> kernel B on stream 2, where B may
// only start after A.  Three ways to express the edge:
//   same    : A and B on the SAME stream (in-order: the baseline)
//   event   : hipEventRecord(e, s1) + hipStreamWaitEvent(s2, e)
//   value   : hipStreamWriteValue32(s1, flag, k) + hipStreamWaitValue32(s2, flag, k, GEQ) on signal memory
// Each kernel stamps wall_clock64 (100 MHz) at its start and end; reported: (start of B) - (end of A), median over the chain
// links, for an idle chip and beside a long low-priority bulk kernel.  The whole chain is enqueued up front (as run_caqr does).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tests/microbench/edge_latency tests/microbench/edge_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_link(long long* stamps, int idx, int spin_ticks) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * idx] = t0;
    while (wall_clock64() - t0 < spin_ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * idx + 1] = wall_clock64();
}
__global__ void k_bulk(double* x, long long n, int reps) {       // bounded busy work over the whole chip
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double a = (double)i;
    for (int r = 0; r < reps; ++r) a = a * 1.0000001 + 0.5;
    if (i < n) x[i] = a;
}

int main(int argc, char** argv) {
    const int links = 64;
    hipStream_t s1, s2, sb;
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    uint32_t mask[8]; for (int i = 0; i < 8; ++i) mask[i] = 0xFFFFFFFFu;
    CK(hipExtStreamCreateWithCUMask(&s2, 8, mask));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, least));
    long long* stamps; CK(hipMalloc(&stamps, sizeof(long long) * 2 * (links + 1)));
    std::vector<long long> h(2 * (links + 1));
    double* bulk; const long long nb = 256LL * 2048 * 256; CK(hipMalloc(&bulk, nb * 8));
    uint32_t* flag = nullptr;
    const bool have_sig = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) == hipSuccess;
    if (!have_sig) { (void)hipGetLastError(); printf("no signal memory: value edges skipped\n"); }
    std::vector<hipEvent_t> ev(links + 1);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int load = 0; load < 2; ++load) {
        for (int mode = 0; mode < (have_sig ? 3 : 2); ++mode) {
            if (have_sig) CK(hipMemset(flag, 0, 64));
            CK(hipDeviceSynchronize());
            if (load) hipLaunchKernelGGL(k_bulk, dim3(256 * 2048), dim3(256), 0, sb, bulk, nb, 6000);     // ~ tens of ms of low-priority work
            // chain: link i on stream (i & 1 ? s2 : s1), edge between consecutive links
            for (int i = 0; i < links; ++i) {
                hipStream_t cur = (mode == 0 || !(i & 1)) ? s1 : s2;
                hipStream_t nxt = (mode == 0 || (i & 1)) ? s1 : s2;
                hipLaunchKernelGGL(k_link, dim3(4), dim3(256), 0, cur, stamps, i, 1000);                 // 10 us links
                if (mode == 1) { CK(hipEventRecord(ev[i], cur)); CK(hipStreamWaitEvent(nxt, ev[i], 0)); }
                if (mode == 2) { CK(hipStreamWriteValue32(cur, flag, (uint32_t)(i + 1), 0)); CK(hipStreamWaitValue32(nxt, flag, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu)); }
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * links, hipMemcpyDeviceToHost));
            std::vector<double> gap;
            for (int i = 1; i < links; ++i) gap.push_back((h[2 * i] - h[2 * i - 1]) * 0.01);
            std::sort(gap.begin(), gap.end());
            printf("%s  %-6s edge: median %.1f us  p10 %.1f  p90 %.1f  (link = 10 us, whole chain %.0f us)\n", load ? "beside bulk" : "idle chip  ",
                   mode == 0 ? "same" : (mode == 1 ? "event" : "value"), gap[gap.size() / 2], gap[gap.size() / 10], gap[gap.size() * 9 / 10],
                   (h[2 * links - 1] - h[0]) * 0.01);
        }
    }
    return 0;
}
