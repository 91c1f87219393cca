// Level-0 trailing update of the CAQR, the dominant kernel of the solve:
//     C <- (I - V T V')' C = C - V (T' (V' C))
// for one tile (TR = 64*RPL contiguous rows) x one block of CB = 32 trailing columns per
// workgroup, with the two GEMM-shaped products on v_mfma_f64_16x16x4_f64.
//
// Data movement per workgroup: the C block is read from HBM once into registers (in the MFMA
// C/D fragment layout of the second product), staged chunk-wise through LDS as the B operand
// of the first product, updated in registers and written back once.  V (shared by all column
// blocks of the tile, L2-resident) is streamed twice through a 64-row LDS chunk.
//
// MFMA f64 16x16x4 fragment maps (cdna_hip_programming.md §3): A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], D[i = (l>>4) + 4 r][j = l&15], r = 0..3.
#pragma once
#include "gn_kernels_caqr.hpp"

namespace gn {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

constexpr int UM_CB = 32;        // trailing columns per workgroup
constexpr int UM_LD = 66;        // LDS leading dimension of a 64-row chunk (conflict-free)
constexpr int UM_WLD = 33;

// TRI = false: level 0 (a tile of 64*RPL contiguous rows).  TRI = true: a tree node — the rows are
// the top 32 rows of F lower-level groups (block q at row offset q*S), and V is the TSQR-stored
// reflector block: identity in block 0, upper triangles in the blocks q >= 1.
// FULL = true: interior workgroups only (full tile, full column block, full panel) — no predicates in
// the address / load code; FULL = false: the remaining edge workgroups.  Both are launched over the
// same grid and a workgroup of the other kind exits at once.
template <int RPL, bool TRI, bool FULL, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void k_caqr_update_mfma(CaqrArgs a) {
    __shared__ __attribute__((aligned(16))) double Vl[PB * UM_LD];        // V chunk  [col][row]
    __shared__ __attribute__((aligned(16))) double Cl[UM_CB * UM_LD];     // C chunk  [col][row]
    __shared__ __attribute__((aligned(16))) double Wp[4][PB * UM_WLD];    // per-wave partial V'C, later W1 / W2
    __shared__ __attribute__((aligned(16))) double Tl[PB * UM_WLD];       // T [l][i]

    const int prob = blockIdx.z + a.prob0;
    const ProbState st = a.state[prob];
    const int r0 = a.panel * PB;
    if (r0 >= st.kp) return;
    const int bw = (st.kp - r0) < PB ? (st.kp - r0) : PB;
    const int col0 = st.rankA + r0;
    const int g = blockIdx.x;
    const int first = r0 + bw;
    const int ncols = st.n2 + 1 - first;
    const int cb0 = blockIdx.y * UM_CB;
    if (cb0 >= ncols) return;
    const int ln = lane_id(), w = wave_id(), tid = threadIdx.x;
    const double* Wm = a.W + prob * a.sW;
    double* C = a.W + prob * a.sW + (size_t)(st.rankA + first) * a.ldw;
    const double* T = a.Tbuf + prob * a.sT + (a.tOff + g) * (long long)(PB * PB);
    // row of tile slot s (block q = s >> 5 of the group, row-in-block s & 31)
    const long long gblk0 = (long long)g * a.F;
    const long long tile_row0 = (long long)r0 + gblk0 * 32;          // level 0: contiguous rows
    auto rowof = [&](int s) -> long long {
        if (!TRI) return tile_row0 + s;
        return (long long)r0 + (gblk0 + (s >> 5)) * a.S + (s & 31);
    };
    // rows of this tile that exist: blocks g*F .. < nblocks, 32 rows each
    const long long blocks_here = (long long)a.nblocks - (long long)g * a.F;
    const int rows_valid = (int)((blocks_here < a.F ? blocks_here : a.F) * 32);
    const bool is_full = (rows_valid == 64 * RPL) && (ncols - cb0 >= UM_CB) && (bw == PB);
    if (SPLIT && is_full != FULL) return;   // SPLIT: interior and edge workgroups in separate launches

    // T into LDS: Tl[l * 33 + i] = T[l][i]
    for (int e = tid; e < PB * PB; e += 256) {
        const int l = e & 31, i = e >> 5;
        Tl[l * UM_WLD + i] = T[l + i * PB];
    }

    const int lr = ln & 15, lq = ln >> 4;
    // ---- load C block into registers, D-fragment layout of the second product -----------------
    // creg[ch][ct][r] = C[tile_row0 + 64 ch + 16 w + lr][cb0 + 16 ct + lq + 4 r]
    mfma_d4 creg[RPL][2];
#pragma unroll
    for (int ch = 0; ch < RPL; ++ch) {
        const int rloc = 64 * ch + 16 * w + lr;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cb0 + 16 * ct + lq + 4 * r;
                creg[ch][ct][r] = (FULL || (c < ncols && rloc < rows_valid)) ? C[rowof(rloc) + (size_t)c * a.ldw] : 0.0;
            }
    }

    // ---- first product: W1 = V' C, K = tile rows, split over chunks (and waves inside a chunk) --
    mfma_d4 acc[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[it][jt] = (mfma_d4){0.0, 0.0, 0.0, 0.0};

    const int vrow = tid & 63, vcg = tid >> 6;
    // V chunk loader: 8 values per thread (row vrow of the chunk, columns vcg + 4u), unit lower
    // trapezoid masked in.  The loads of chunk ch+1 are issued before the MFMAs of chunk ch so the
    // L2 latency overlaps with the matrix pipe (the data are only waited for at the next LDS write).
    auto load_v = [&](int ch, double* vr) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = vcg + 4 * u;
            const int s = 64 * ch + vrow;
            double v = 0.0;
            if (FULL || (j < bw && s < rows_valid)) {
                if (!TRI) {
                    if (s > j) v = Wm[rowof(s) + (size_t)(col0 + j) * a.ldw];
                    else if (s == j) v = 1.0;
                } else {
                    const int rb = s & 31;
                    if (s < 32) v = (rb == j) ? 1.0 : 0.0;
                    else if (rb <= j) v = Wm[rowof(s) + (size_t)(col0 + j) * a.ldw];
                }
            }
            vr[u] = v;
        }
    };
    double vreg[8];
    load_v(0, vreg);
#pragma unroll
    for (int ch = 0; ch < RPL; ++ch) {
        __syncthreads();  // previous chunk's operands fully consumed
#pragma unroll
        for (int u = 0; u < 8; ++u) Vl[(vcg + 4 * u) * UM_LD + vrow] = vreg[u];
        if (ch + 1 < RPL) load_v(ch + 1, vreg);
        else load_v(0, vreg);   // first chunk of the second product
        // stage C chunk from registers
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cl[(16 * ct + lq + 4 * r) * UM_LD + 16 * w + lr] = creg[ch][ct][r];
        __syncthreads();
        // wave w takes rows 16w .. 16w+15 of the chunk: 4 k-steps
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kr = 16 * w + 4 * ks + lq;
            double av[2], bv[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) av[it] = Vl[(16 * it + lr) * UM_LD + kr];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) bv[jt] = Cl[(16 * jt + lr) * UM_LD + kr];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[it], bv[jt], acc[it][jt], 0, 0, 0);
        }
    }
    // partial W1 of this wave -> LDS: Wp[w][i * 33 + j]
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Wp[w][(16 * it + lq + 4 * r) * UM_WLD + 16 * jt + lr] = acc[it][jt][r];
    __syncthreads();
    // reduce over waves into Wp[0]
    for (int e = tid; e < PB * UM_CB; e += 256) {
        const int i = e >> 5, j = e & 31;
        const int o = i * UM_WLD + j;
        Wp[0][o] = (Wp[0][o] + Wp[1][o]) + (Wp[2][o] + Wp[3][o]);
    }
    __syncthreads();
    // ---- W2 = -T' W1 (32 x 32, plain FMAs): Wp[1][k * 33 + j] ----------------------------------
    {
        const int i = tid & 31, jq = tid >> 5;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int l = 0; l <= i; ++l) {
            const double tl = Tl[l * UM_WLD + i];
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] += tl * Wp[0][l * UM_WLD + jq + 8 * q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) Wp[1][i * UM_WLD + jq + 8 * q] = -s4[q];
    }
    __syncthreads();
    // A operands of the second product: a2[ct][ks] = W2n[k = 4 ks + lq][col = 16 ct + lr]
    double a2[2][8];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a2[ct][ks] = Wp[1][(4 * ks + lq) * UM_WLD + 16 * ct + lr];

    // ---- second product: C += V * W2n, chunk by chunk ------------------------------------------
#pragma unroll
    for (int ch = 0; ch < RPL; ++ch) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) Vl[(vcg + 4 * u) * UM_LD + vrow] = vreg[u];
        if (ch + 1 < RPL) load_v(ch + 1, vreg);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const double bv = Vl[(4 * ks + lq) * UM_LD + 16 * w + lr];   // B[k][row]
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                creg[ch][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[ct][ks], bv, creg[ch][ct], 0, 0, 0);
        }
    }
    // ---- write back -----------------------------------------------------------------------------
#pragma unroll
    for (int ch = 0; ch < RPL; ++ch) {
        const int rloc = 64 * ch + 16 * w + lr;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cb0 + 16 * ct + lq + 4 * r;
                if (FULL || (c < ncols && rloc < rows_valid)) C[rowof(rloc) + (size_t)c * a.ldw] = creg[ch][ct][r];
            }
    }
}

inline void launch_update_mfma(int RPL, const CaqrArgs& a, int groups, int ncols, int batch, hipStream_t s) {
    dim3 grid(groups, (ncols + UM_CB - 1) / UM_CB, batch);
    if (a.level == 0) {
        // (a split into a predicate-free interior launch + an edge launch was measured SLOWER on MI355X:
        // the interior kernel gained nothing and all edge tiles share blockIdx.x, i.e. one XCD)
        if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_mfma<8, false, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_caqr_update_mfma<4, false, false>), grid, dim3(256), 0, s, a);
    } else {
        if (RPL == 8) hipLaunchKernelGGL((k_caqr_update_mfma<8, true, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_caqr_update_mfma<4, true, false>), grid, dim3(256), 0, s, a);
    }
}

}  // namespace gn
