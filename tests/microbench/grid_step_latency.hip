// What does one step of a PERSISTENT distributed pivoted QR cost in synchronisation alone?  G workgroups (256 threads)
// run `steps` rounds of: publish an 8 KB column + a 16-byte record carrying the round number (device-coherent stores),
// one wave polls the G records until all carry the round number (device-coherent loads), everybody then reads the
// column of a "winner" that depends on the records (a dependent device-coherent load).  No arithmetic: the floor of a
// step.  Variant 1 replaces the tagged records by a fence + one atomic counter.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tests/microbench/grid_step_latency tests/microbench/grid_step_latency.hip
// Run  : grid_step_latency [G=129] [steps=1000] [rows=1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct alignas(16) Rec { double val; int pos; int tag; };
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i4 load_coherent16(const void* p) {
    i4 x;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    return x;
}
__device__ __forceinline__ void store_coherent16(void* p, i4 x) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(x) : "memory");
}
// variant 2: agent-scope (sc1) accesses, participants confined to ONE XCD (sc0 alone is workgroup scope: may hit a stale L1 — it hung)
__device__ __forceinline__ i4 load_l2_16(const void* p) {
    i4 x;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    return x;
}
__device__ __forceinline__ void store_l2_16(void* p, i4 x) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory"); }
// eight rows ln + 64 i (i = 0..7) of a column in one block: the results are not touched before the wait
__device__ __forceinline__ void ld8x8_l2(const double* p, double (&t)[8]) {
    asm volatile("global_load_dwordx2 %0, %8, off sc1\n\tglobal_load_dwordx2 %1, %8, off offset:512 sc1\n\t"
                 "global_load_dwordx2 %2, %8, off offset:1024 sc1\n\tglobal_load_dwordx2 %3, %8, off offset:1536 sc1\n\t"
                 "global_load_dwordx2 %4, %8, off offset:2048 sc1\n\tglobal_load_dwordx2 %5, %8, off offset:2560 sc1\n\t"
                 "global_load_dwordx2 %6, %8, off offset:3072 sc1\n\tglobal_load_dwordx2 %7, %8, off offset:3584 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                 : "v"(p) : "memory");
}
__device__ __forceinline__ void st8_l2(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ double ld8(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st8(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int POLL_LIMIT = 1 << 22;          // ~ a second of polling: then the round is abandoned and the host prints the failure
__device__ int g_gave_up;
template <int VARIANT>
__global__ __launch_bounds__(256) void k_rounds(Rec* recs, double* cols, unsigned* counter, int G, int steps, int rows, double* sink, int* xcc) {
    int* gave_up = &g_gave_up;
    __shared__ int win_s;
    if (VARIANT == 2 && (blockIdx.x & 7)) return;          // one XCD: workgroups are dealt round-robin to the 8 XCDs
    const int g = VARIANT == 2 ? blockIdx.x >> 3 : blockIdx.x, tid = threadIdx.x, ln = tid & 63, w = tid >> 6;
    if (tid == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[g] = (int)(id & 15); }
    double acc = 0.0;
    double mine[16];
    for (int i = 0; i < 16; ++i) mine[i] = g + 0.001 * (ln + 64 * i);
    for (int j = 0; j < steps; ++j) {
        // publish: wave 0 writes the column (rows/64 values per lane), then the record
        if (w == 0) {
            double* slot = cols + ((size_t)(j & 1) * G + g) * rows;
            if (VARIANT == 2) { for (int i = 0; i < rows / 64; ++i) st8_l2(slot + ln + 64 * i, mine[i] + j); }
            else for (int i = 0; i < rows / 64; ++i) st8(slot + ln + 64 * i, mine[i] + j);
            if (VARIANT == 0 || VARIANT == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the column is performed before the record is
                if (ln == 0) {
                    Rec r; r.val = (double)((g * 7919 + j * 104729) % 1000003); r.pos = g; r.tag = j + 1;
                    if (VARIANT == 2) store_l2_16(&recs[(size_t)(j & 1) * G + g], *(i4*)&r);
                    else store_coherent16(&recs[(size_t)(j & 1) * G + g], *(i4*)&r);
                }
            } else {
                if (ln == 0) {
                    Rec r; r.val = (double)((g * 7919 + j * 104729) % 1000003); r.pos = g; r.tag = j + 1;
                    *(Rec*)&recs[(size_t)(j & 1) * G + g] = r;
                }
                __threadfence();
                if (ln == 0) atomicAdd(counter, 1u);
            }
            // poll
            int win = -1;
            if (VARIANT == 0 || VARIANT == 2) {
                bool all;
                double bv; int bp;
                int spins = 0;           // every poll loop is bounded: a record that never arrives ends the kernel, not the box
                do {
                    bv = -1.0; bp = -1;
                    bool ok = true;
                    for (int e = ln; e < G; e += 64) {
                        const i4 x = VARIANT == 2 ? load_l2_16(&recs[(size_t)(j & 1) * G + e]) : load_coherent16(&recs[(size_t)(j & 1) * G + e]);
                        const Rec r = *(const Rec*)&x;
                        ok = ok && (r.tag == j + 1);
                        if (r.val > bv) { bv = r.val; bp = r.pos; }
                    }
                    all = __all(ok);
                    if (++spins > POLL_LIMIT) { if (ln == 0) atomicExch(gave_up, 1); all = true; }
                } while (!all);
                for (int o = 32; o > 0; o >>= 1) {
                    const double ov = __shfl_xor(bv, o); const int op = __shfl_xor(bp, o);
                    if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
                }
                win = bp;
            } else {
                if (ln == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G * (j + 1)) {
                        if (++spins > POLL_LIMIT) { atomicExch(gave_up, 1); break; }
                    }
                }
                __threadfence();
                double bv = -1.0; int bp = -1;
                for (int e = ln; e < G; e += 64) {
                    const Rec r = recs[(size_t)(j & 1) * G + e];
                    if (r.val > bv) { bv = r.val; bp = r.pos; }
                }
                for (int o = 32; o > 0; o >>= 1) {
                    const double ov = __shfl_xor(bv, o); const int op = __shfl_xor(bp, o);
                    if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
                }
                win = bp;
            }
            if (ln == 0) win_s = win;
        }
        __syncthreads();
        int win = win_s;
        win = win < 0 ? 0 : (win >= G ? G - 1 : win);
        // everybody reads the winner's column (dependent load)
        const double* src = cols + ((size_t)(j & 1) * G + win) * rows;
        double s = 0.0;
        if (VARIANT == 2) {
            for (int i0 = 0; i0 < rows / 64; i0 += 8) {
                double t[8];
                ld8x8_l2(src + ln + 64 * i0, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += t[i];
            }
        } else for (int i = 0; i < rows / 64; ++i) s += (VARIANT == 0) ? ld8(src + ln + 64 * i) : src[ln + 64 * i];
        acc += s;
        __syncthreads();
    }
    if (acc == 1.2345) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 129, steps = argc > 2 ? atoi(argv[2]) : 1000, rows = argc > 3 ? atoi(argv[3]) : 1024;
    {   // every variant needs all of its workgroups resident at once: refuse a G the device cannot hold
        hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
        const int cap = prop.multiProcessorCount * 2;           // 256-thread workgroups at the kernels' register budget: two per compute unit
        if (G > cap) { printf("G = %d exceeds the resident-workgroup capacity (%d)\n", G, cap); return 2; }
    }
    Rec* recs; double *cols, *sink; unsigned* ctr; int* xcc;
    CK(hipMalloc(&recs, sizeof(Rec) * 2 * G)); CK(hipMalloc(&cols, sizeof(double) * 2 * G * rows)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&ctr, 4));
    CK(hipMalloc(&xcc, 4 * G));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int only = argc > 4 ? atoi(argv[4]) : -1;
    for (int variant = 0; variant < 3; ++variant) {
        if (only >= 0 && variant != only) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(recs, 0, sizeof(Rec) * 2 * G)); CK(hipMemset(ctr, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(G), dim3(256), 0, 0, recs, cols, ctr, G, steps, rows, sink, xcc);
            else if (variant == 1) hipLaunchKernelGGL(k_rounds<1>, dim3(G), dim3(256), 0, 0, recs, cols, ctr, G, steps, rows, sink, xcc);
            else hipLaunchKernelGGL(k_rounds<2>, dim3(8 * G), dim3(256), 0, 0, recs, cols, ctr, G, steps, rows, sink, xcc);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int gave_up = 0, zero = 0;
            CK(hipMemcpyFromSymbol(&gave_up, HIP_SYMBOL(g_gave_up), 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gave_up), &zero, 4));
            if (gave_up) { printf("variant %d: a poll loop ran into its limit (records never arrived): no figure\n", variant); continue; }
            printf("variant %d (%s) G %d rows %d: %.3f us per round\n", variant, variant == 2 ? "ONE XCD, tagged records, L2-coherent ld/st" : (variant ? "fence + atomic counter" : "tagged records, coherent ld/st"), G, rows, ms * 1e3 / steps);
            fflush(stdout);
        }
    }
    int hx[1024]; CK(hipMemcpy(hx, xcc, 4 * G, hipMemcpyDeviceToHost));
    printf("xcc of workgroups 0..15:"); for (int i = 0; i < 16 && i < G; ++i) printf(" %d", hx[i]); printf("\n");
    return 0;
}
