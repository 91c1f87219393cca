// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, alone and mixed.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_f64_rate mfma_f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int NFMA>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4 acc[NACC > 0 ? NACC : 1];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < (NACC > 0 ? 8 / NACC : 0); ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NFMA; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_fma(f[i], b, a);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NFMA>
void run(int wgs_per_cu) {
    int iters = 10000;
    int grid = 256 * wgs_per_cu;
    double* out;
    hipMalloc(&out, grid * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, NFMA><<<grid, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, NFMA><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)grid * 4;
    double mf = NACC > 0 ? waves * iters * 8.0 * 2048 : 0;
    double vf = waves * iters * NFMA * 8.0 * 64 * 2;
    double ns_per_mfma = NACC > 0 ? ms * 1e6 / (iters * 8.0) / wgs_per_cu : 0;   // per SIMD issue interval
    printf("nacc %d nfma %2d wgs/cu %d  %8.3f ms  mfma %5.1f TF  valu %5.1f TF  simd ns/mfma %.1f\n", NACC, NFMA, wgs_per_cu, ms,
           mf / ms * 1e-9, vf / ms * 1e-9, ns_per_mfma);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; ++w) {
        run<1, 0>(w); run<2, 0>(w); run<4, 0>(w); run<8, 0>(w);
        run<0, 8>(w);
        run<4, 4>(w);
    }
    return 0;
}
