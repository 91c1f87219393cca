// Micro-benchmark: what an in-place read-modify-write stream can reach on MI355X, by store flavour and occupancy.
// Same footprint as the level-0 trailing update of panel 0 (256 problems x 4096 rows x 448 columns, ld 4128).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o copy_variants copy_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int LD = 4128, NC = 449;

// MODE 0: in place, plain stores.  1: in place, nontemporal stores.  2: out of place (dst = second buffer).
// 3: in place, nontemporal loads + stores.   ROWS rows x 32 columns per workgroup.
template <int MODE, int ROWS>
__global__ __launch_bounds__(256) void k(double* W, double* D, long long sW, double add) {
    constexpr int NR = ROWS / 32;      // d2 per thread per column-group
    double* C = W + blockIdx.z * sW + (size_t)(blockIdx.y * 32) * LD + blockIdx.x * ROWS;
    double* O = (MODE == 2 ? D : W) + blockIdx.z * sW + (size_t)(blockIdx.y * 32) * LD + blockIdx.x * ROWS;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lq = l >> 4;
    d2 r[NR][2];
#pragma unroll
    for (int g = 0; g < NR; ++g)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const d2* p = (const d2*)&C[(size_t)(8 * w + 4 * ct + lq) * LD + 32 * g + 2 * lr];
            r[g][ct] = (MODE == 3) ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
    for (int g = 0; g < NR; ++g)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            d2 v = r[g][ct]; v[0] += add; v[1] += add;
            d2* p = (d2*)&O[(size_t)(8 * w + 4 * ct + lq) * LD + 32 * g + 2 * lr];
            if (MODE == 1 || MODE == 3) __builtin_nontemporal_store(v, p); else *p = v;
        }
}

// MODE 4: same footprint and non-temporal accesses, but every wave-instruction moves ONE KILOBYTE of one column (lane l <->
// rows 2l, 2l + 1 of a 128-row piece) instead of 4 columns x 256 B: is the 256-B segment shape of the MFMA fragment map what
// keeps the stream below the ~6.3 TB/s of a plain copy?
template <int ROWS>
__global__ __launch_bounds__(256) void k1k(double* W, long long sW, double add) {
    constexpr int NP = ROWS / 128;     // 128-row pieces per column
    double* C = W + blockIdx.z * sW + (size_t)(blockIdx.y * 32) * LD + blockIdx.x * ROWS;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    d2 r[8][NP];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int p = 0; p < NP; ++p) r[c][p] = __builtin_nontemporal_load((const d2*)&C[(size_t)(8 * w + c) * LD + 128 * p + 2 * l]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            d2 v = r[c][p]; v[0] += add; v[1] += add;
            __builtin_nontemporal_store(v, (d2*)&C[(size_t)(8 * w + c) * LD + 128 * p + 2 * l]);
        }
}

template <int MODE, int ROWS>
void run(double* W, double* D, long long sW, int batch) {
    dim3 grid(4096 / ROWS, 14, batch);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, ROWS><<<grid, 256>>>(W, D, sW, 0.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE, ROWS><<<grid, 256>>>(W, D, sW, 0.0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double bytes = 2.0 * 8 * 4096.0 * 448 * batch;
    printf("mode %d rows/wg %4d: %.3f ms  %.0f GB/s (read+write)  %s\n", MODE, ROWS, ms, bytes / ms * 1e-6, hipGetErrorString(hipGetLastError()));
}
int main() {
    const int batch = 256;
    long long sW = (long long)LD * NC;
    double *W, *D; hipMalloc(&W, sW * batch * 8); hipMalloc(&D, sW * batch * 8);
    hipMemset(W, 0, sW * batch * 8); hipMemset(D, 0, sW * batch * 8);
    run<0, 512>(W, D, sW, batch); run<1, 512>(W, D, sW, batch); run<2, 512>(W, D, sW, batch); run<3, 512>(W, D, sW, batch);
    run<0, 128>(W, D, sW, batch); run<1, 128>(W, D, sW, batch); run<2, 128>(W, D, sW, batch);
    run<0, 1024>(W, D, sW, batch); run<1, 1024>(W, D, sW, batch);
    {
        dim3 grid(4096 / 512, 14, batch);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k1k<512><<<grid, 256>>>(W, sW, 0.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) k1k<512><<<grid, 256>>>(W, sW, 0.0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("mode 4 (1 KB per wave-instruction, nt) rows/wg 512: %.3f ms  %.0f GB/s  %s\n", ms, 2.0 * 8 * 4096.0 * 448 * batch / ms * 1e-6,
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
