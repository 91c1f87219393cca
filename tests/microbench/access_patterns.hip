// Micro-benchmark: HBM read+write bandwidth of a (512 rows x 32 cols) block per workgroup for several
// lane->element maps (column-major matrix, ld = 4128 doubles).  Mirrors the grid of the level-0
// trailing update: grid (8 row tiles, 14 column blocks, 256 problems), 256 threads.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o access_patterns access_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int LD = 4128, NC = 449, ROWS = 512, CB = 32;

template <int PAT>
__global__ __launch_bounds__(256, 2) void k(double* W, long long sW, double add) {
    double* C = W + blockIdx.z * sW + (size_t)(blockIdx.y * CB) * LD + blockIdx.x * ROWS;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lq = l >> 4;
    if (PAT == 0) {   // lanes lr along rows (128 B), lq over 4 columns; 8 B per lane (current kernel)
        double r[8][2][4];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) r[ch][ct][q] = C[(size_t)(16 * ct + lq + 4 * q) * LD + 64 * ch + 16 * w + lr];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) C[(size_t)(16 * ct + lq + 4 * q) * LD + 64 * ch + 16 * w + lr] = r[ch][ct][q] + add;
    } else if (PAT == 1) {   // lanes lr along 16 columns, lane holds 4 consecutive rows (2 x 16 B); wave rows 128w..
        d2 r[8][2][2];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    r[jt][ct][h] = *(const d2*)&C[(size_t)(16 * ct + lr) * LD + 128 * w + 16 * jt + 4 * lq + 2 * h];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    d2 v = r[jt][ct][h]; v[0] += add; v[1] += add;
                    *(d2*)&C[(size_t)(16 * ct + lr) * LD + 128 * w + 16 * jt + 4 * lq + 2 * h] = v;
                }
    } else if (PAT == 2) {   // 64 lanes x 8 B along rows
        double r[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) { const int c = (i >> 1) , rr = (i & 1) * 256 + threadIdx.x; r[i] = C[(size_t)c * LD + rr]; }
#pragma unroll
        for (int i = 0; i < 64; ++i) { const int c = (i >> 1) , rr = (i & 1) * 256 + threadIdx.x; C[(size_t)c * LD + rr] = r[i] + add; }
    } else if (PAT == 3) {   // 64 lanes x 16 B along rows
        d2 r[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) r[i] = *(const d2*)&C[(size_t)i * LD + 2 * threadIdx.x];
#pragma unroll
        for (int i = 0; i < 32; ++i) { d2 v = r[i]; v[0] += add; v[1] += add; *(d2*)&C[(size_t)i * LD + 2 * threadIdx.x] = v; }
    } else if (PAT == 4) {   // like 1 but lane holds 2 consecutive rows per load and lq spans 8 rows: 16 cols x 64 B
        d2 r[8][2][2];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    r[jt][ct][h] = *(const d2*)&C[(size_t)(16 * ct + lr) * LD + 128 * w + 16 * jt + 8 * h + 2 * lq];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    d2 v = r[jt][ct][h]; v[0] += add; v[1] += add;
                    *(d2*)&C[(size_t)(16 * ct + lr) * LD + 128 * w + 16 * jt + 8 * h + 2 * lq] = v;
                }
    }
}

template <int PAT>
void run(double* W, long long sW, int batch) {
    dim3 grid(8, 14, batch);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<PAT><<<grid, 256>>>(W, sW, 0.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<PAT><<<grid, 256>>>(W, sW, 0.0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double bytes = 2.0 * 8 * 4096.0 * 448 * batch;
    printf("pattern %d: %.3f ms  %.0f GB/s (read+write)  err=%s\n", PAT, ms, bytes / ms * 1e-6, hipGetErrorString(hipGetLastError()));
}
int main() {
    const int batch = 256;
    long long sW = (long long)LD * NC;
    double* W; hipMalloc(&W, sW * batch * 8); hipMemset(W, 0, sW * batch * 8);
    run<0>(W, sW, batch); run<1>(W, sW, batch); run<2>(W, sW, batch); run<3>(W, sW, batch); run<4>(W, sW, batch);
    run<0>(W, sW, batch);
    return 0;
}
