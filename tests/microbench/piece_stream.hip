// Micro-benchmark: in-place read-modify-write of 32-row (256-byte) pieces of the columns of a batch of column-major matrices,
// the memory pattern of a CAQR tree-level update (the first 32 rows of every 512-row tile), against the same volume moved in
// contiguous column segments.  No arithmetic beyond one add: what rate does HBM deliver for this pattern?
// A workgroup owns 32 columns x NU pieces of one problem; wave w owns pieces w, w + 4, ...; a wave-instruction touches
// 4 columns x 256 contiguous bytes (lane = row pair lr, column lq + 4 r), like the update kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o piece_stream piece_stream.hip     Run: ./piece_stream [batch]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NGW>
__global__ __launch_bounds__(256, 2) void k(double* W, long long sW, int ldw, long long unit_stride_rows, int nloop, int ystep) {
    double* P = W + blockIdx.z * sW;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lq = l >> 4;
    for (int ib = 0; ib < nloop; ++ib) {
        const int cb0 = 32 * (blockIdx.y + ib * ystep);
        d2 x[NGW][8];
#pragma unroll
        for (int g = 0; g < NGW; ++g)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                x[g][c] = __builtin_nontemporal_load((const d2*)&P[(size_t)(cb0 + lq + 4 * c) * ldw + (w + 4 * g) * unit_stride_rows + 2 * lr]);
#pragma unroll
        for (int g = 0; g < NGW; ++g)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                d2 y = x[g][c];
                y[0] += 1.0; y[1] += 1.0;
                __builtin_nontemporal_store(y, (d2*)&P[(size_t)(cb0 + lq + 4 * c) * ldw + (w + 4 * g) * unit_stride_rows + 2 * lr]);
            }
    }
}

int main(int argc, char** argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 384;
    const int ldw = 4128, ncols = 384;
    const long long sW = (long long)ldw * (512 + 33);
    double* W;
    hipMalloc(&W, sizeof(double) * sW * batch);
    hipMemset(W, 0, sizeof(double) * sW * batch);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* what, int ngw, long long stride, int gy) {
        dim3 grid(1, gy, batch);
        const int nloop = (ncols / 32) / gy;
        auto go = [&]() {
            if (ngw == 2) k<2><<<grid, 256>>>(W, sW, ldw, stride, nloop, gy);
            else k<4><<<grid, 256>>>(W, sW, ldw, stride, nloop, gy);
        };
        go(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) go();
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double bytes = 2.0 * 8.0 * 32.0 * 4 * ngw * ncols * batch;
        printf("%-58s %2d units, grid y %2d: %.3f ms  %.0f GB/s\n", what, 4 * ngw, gy, ms, bytes / ms * 1e-6);
    };
    for (int gy : {12, 4}) {
        run("256-B pieces, 4 KB apart (tree of a pair's first panel)", 2, 512, gy);
        run("256-B pieces, 2 KB apart", 2, 256, gy);
        run("256-B pieces, 512 B apart", 2, 64, gy);
        run("contiguous (pieces adjacent)", 2, 32, gy);
        run("16 pieces, 4 KB apart", 4, 512, gy);
        run("16 pieces contiguous", 4, 32, gy);
    }
    // pieces of 512 B: units 2q, 2q + 1 adjacent, pairs 4 KB apart: emulate with stride 32 inside pairs -> use unit index mapping
    return 0;
}
