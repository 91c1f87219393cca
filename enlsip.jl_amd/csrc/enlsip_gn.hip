// libenlsip_gn.so — C ABI (include/enlsip_gn.h) over the gfx950 kernels.
// One handle = one stream + one lazily grown device workspace.  No exceptions cross the ABI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>

#include "gn_context.hpp"
#include "gn_kernels_caqr.hpp"
#include "gn_kernels_constraint.hpp"
#include "gn_kernels_geqp3_reg.hpp"
#include "gn_kernels_final.hpp"
#include "gn_kernels_q1.hpp"
#include "gn_kernels_q1_mfma.hpp"
#include "gn_kernels_q1_v2.hpp"
#include "gn_kernels_q1_rows.hpp"
#include "gn_kernels_final_small.hpp"
#include "gn_kernels_small_fused.hpp"
#include "gn_kernels_constraint_small.hpp"
#include "gn_kernels_constraint_dist.hpp"
#include "gn_kernels_update_v4.hpp"
#include "gn_kernels_misc.hpp"
#include "gn_kernels_lagrange.hpp"
#include "gn_kernels_newton.hpp"
#include "gn_kernels_qrcp_dist.hpp"
#include "gn_kernels_qrcp_block.hpp"
#include "gn_kernels_qrcp_block_reg.hpp"
#include "gn_rescale.hpp"

using namespace gn;

#define GN_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            char buf__[512];                                                           \
            snprintf(buf__, sizeof buf__, "%s:%d %s -> %s", __FILE__, __LINE__, #call, \
                     hipGetErrorString(e__));                                          \
            h->err = buf__;                                                            \
            return (int)e__ > 0 ? (int)e__ : 999;                                      \
        }                                                                              \
    } while (0)

// No exception crosses the ABI: entry points that allocate on the host (std::vector, std::string, std::thread) run inside
// GN_TRY ... GN_CATCH(h), which turns an exception into an error code and a message.
#define GN_TRY try {
#define GN_CATCH(h)                                                                    \
    }                                                                                  \
    catch (const std::bad_alloc&) {                                                    \
        try { (h)->err = "out of host memory"; } catch (...) {}                        \
        return 998;                                                                    \
    }                                                                                  \
    catch (const std::exception& e__) {                                                \
        try { (h)->err = std::string("host exception: ") + e__.what(); } catch (...) {}\
        return 997;                                                                    \
    }                                                                                  \
    catch (...) {                                                                      \
        return 997;                                                                    \
    }

static inline long long rup(long long x, long long a) { return (x + a - 1) / a * a; }

#define GN_TRACE(h, ...)                                              \
    do {                                                              \
        if ((h)->trace) {                                             \
            (void)hipStreamSynchronize((h)->stream);                  \
            fprintf(stderr, "[enlsip_gn] " __VA_ARGS__);              \
            fputc('\n', stderr);                                      \
            fflush(stderr);                                           \
        }                                                             \
    } while (0)

static int grow(enlsip_gn_handle h, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return 0;
    if (b.p) GN_HIP(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    GN_HIP(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// plan: geometry + workspace carve for (batch, m, n, t)
// ---------------------------------------------------------------------------------------------
static int make_plan(enlsip_gn_handle h, long long batch, long long m, long long n, long long t) {
    Plan& P = h->plan;
    if (h->have_plan && P.batch == batch && P.m == m && P.n == n && P.t == t) return 0;
    P = Plan();
    P.batch = batch; P.m = m; P.n = n; P.t = t;
    P.kA = (int)std::min(n, t);
    P.RPL = (m <= 256 ? 256 : h->tile_rows) / 64;      // a problem of at most 256 rows is one 256-row tile
    P.F = 2 * P.RPL;
    P.ldw = (int)rup(std::max<long long>(m, 1), 32);
    // a leading dimension that is a multiple of 4 KB puts the same row range of every column on the
    // same few HBM channels (measured: 4x slower edge tiles at ldw = 4096): skew it by one 256-B block
    if (P.ldw % 512 == 0) P.ldw += 32;
    const long long kpmax = std::min(m, n);
    P.ldr = (int)rup(std::max<long long>(kpmax, 1), 8);
    P.npan_max = (int)((kpmax + PB - 1) / PB);
    long long running = 0;
    P.panels.resize(P.npan_max);
    // panel pairs (gn_kernels_caqr.hpp): from three panels on; the reflector-by-reflector A/B path keeps the plain sweep
    // ... and only where the far update is the bulk of the sweep: the pair costs two extra small launches per two panels
    // (the first panel's level-0 and tree reflectors on the second panel's 32 columns), which a latency-bound sweep does not
    // earn back.  Measured (MI355X): 384 x C2 +1.4 % solves/s and C4's 262144 rows 29.7 -> 27.4 ms with pairs, but a single
    // C2 problem 5.17 -> 5.37 ms, 64 of them 10.35 -> 10.47 ms, a 32768-row C4 shard 16.8 -> 17.2 ms.  Rule: at least ~8192
    // far-update workgroups in the first pair (tiles x 32-column blocks x problems); ENLSIP_GN_PAIR=1 forces pairs.
    const long long far_wgs = batch * ((std::max<long long>(m, 1) + 64 * P.RPL - 1) / (64 * P.RPL)) * ((std::max<long long>(n - P.kA, 1) + 31) / 32);
    P.pair = h->pair_enabled && !(h->flags & ENLSIP_GN_UPDATE_REFLECTORS) && P.npan_max >= 3 && (far_wgs >= 8192 || h->pair_forced);
    const long long mpad = rup(std::max<long long>(m, 1), 32);    // NOT ldw: the skew rows are never touched
    for (int k = 0; k < P.npan_max; ++k) {
        const bool second = P.pair && (k & 1);                    // second panel of the pair (k - 1, k): keeps the first one's tiles
        const long long anchor = 32LL * (k - (second ? 1 : 0));   // first row of tile 0
        const int nb0 = (int)((mpad - anchor) / 32);              // 32-row blocks from the anchor to the padded m
        const int ntiles = (nb0 + P.F - 1) / P.F;
        const int last_units = nb0 - (ntiles - 1) * P.F;          // blocks of the last tile
        auto push = [&](int level, int mode, long long base, int skip, int nblocks, int groups, long long S) {
            LevelPlan L;
            L.level = level; L.mode = mode; L.base = base; L.skip = skip;
            L.nblocks = nblocks; L.groups = groups; L.S = S;
            L.tOff = running;
            running += groups;
            P.panels[k].levels.push_back(L);
        };
        // level 0: the tiles (a tile that has no row of the second panel still gets its — zero — T block: the pair update indexes
        // T by tile)
        push(0, 0, anchor, second ? 1 : 0, nb0, ntiles, 32);
        if (ntiles <= 1) continue;
        int level = 1, nb;
        long long S = 32LL * P.F, base = 32LL * k;
        if (!second) nb = ntiles;
        else {
            // first tree level of the second panel: per tile the new R factor (rows 32..63) and, from tile 1 on, the rows 0..31
            // the first panel's tree left behind (dense in these columns): mode 2
            const int nblocks1 = (ntiles - (last_units == 1 ? 1 : 0)) + (ntiles - 1);
            const int groups1 = (nblocks1 + P.F - 1) / P.F;
            push(1, 2, anchor, 0, nblocks1, groups1, S);
            if (groups1 <= 1) continue;
            // group leaders: blocks F q of level 1 = (tile (F / 2) q, rows 32..63)
            nb = groups1;
            S = S * (P.F / 2);
            level = 2;
        }
        while (true) {
            const int groups = (nb + P.F - 1) / P.F;
            push(level, 1, base, 0, nb, groups, S);
            if (groups <= 1) break;
            nb = groups;
            S *= P.F;
            ++level;
        }
    }
    P.nTblocks = std::max<long long>(running, 1);
    const long long nblkA = std::max<long long>((P.kA + KBLK - 1) / KBLK, 1);
    auto pad = [](long long x) { return rup(std::max<long long>(x, 1), 32); };  // 256 B granules
    P.sFA = pad(n * t); P.sTauA = pad(P.kA); P.sJA = pad(t);
    P.sFL = pad(t * P.kA); P.sTauL = pad(P.kA); P.sJL = pad(P.kA);
    P.sTA = pad(nblkA * KBLK * KBLK); P.sP1 = pad(t); P.sB = pad(t);
    // 32 spare columns per problem: the trailing-update kernel reads (and discards) whole 32-column blocks
    P.sW = pad((long long)P.ldw * (n + 1 + 32));
    P.sT = pad(P.nTblocks * PB * PB);
    P.sRt = pad((long long)P.ldr * (n + 1));
    P.sTauJ = pad(kpmax); P.sJJ = pad(n); P.sZ = pad(kpmax);
    P.sVec = pad((long long)P.ldw * 2);
    // + 33 columns: k_sb_update_blk reads whole 32-column / 32-row blocks past the last valid element
    P.sM = pad((long long)P.ldr * (n + 1 + 33)); P.sVb = pad((long long)P.ldr * (std::max<long long>(kpmax, 1) + 33));
    P.sDiag = pad(kpmax); P.sVn = pad(n); P.sQI = pad(n);
    P.qdGmax = (int)((n + 1 + QD_CPW - 1) / QD_CPW);
    P.sCand = 2 * (long long)P.qdGmax;
    const long long per_dbl = P.sFA + P.sTauA + P.sFL + P.sTauL + P.sTA + P.sP1 + P.sB + P.sW + P.sT + P.sRt +
                              P.sTauJ + P.sZ + P.sVec + P.sM + P.sVb + P.sDiag + 2 * P.sVn + PB * PB;
    const long long per_i64 = P.sJA + P.sJL + P.sJJ;
    const long long per_i32 = 7 * P.sQI + 32;   // chosen + 2 x pos + 2 x colat + inblk + active list (n + 1 entries)
    const size_t bytes = (size_t)batch * (per_dbl * 8 + per_i64 * 8 + per_i32 * 4 + P.sCand * sizeof(QdCand)) +
                         (size_t)batch * (sizeof(ProbState) + sizeof(SbInfo)) + 8192;
    int rc = grow(h, h->ws, bytes);
    if (rc) return rc;
    char* p = (char*)h->ws.p;
    auto carve = [&](long long stride_elems) {
        char* r = p;
        p += (size_t)batch * stride_elems * 8;
        return r;
    };
    h->W = (double*)carve(P.sW);
    h->FA = (double*)carve(P.sFA); h->tauA = (double*)carve(P.sTauA);
    h->FL = (double*)carve(P.sFL); h->tauL = (double*)carve(P.sTauL);
    h->TA = (double*)carve(P.sTA); h->p1 = (double*)carve(P.sP1); h->bvec = (double*)carve(P.sB);
    h->Tbuf = (double*)carve(P.sT); h->Rt = (double*)carve(P.sRt); h->tauJ = (double*)carve(P.sTauJ);
    h->zsave = (double*)carve(P.sZ); h->vec = (double*)carve(P.sVec);
    h->qdM = (double*)carve(P.sM); h->qdVb = (double*)carve(P.sVb); h->qdDiag = (double*)carve(P.sDiag);
    h->qdVn1 = (double*)carve(P.sVn); h->qdVn2 = (double*)carve(P.sVn);
    h->sbT = (double*)carve(PB * PB);
    h->jpvtA = (long long*)carve(P.sJA); h->jpvtL = (long long*)carve(P.sJL); h->jpvtJ = (long long*)carve(P.sJJ);
    h->qdChosen = (int*)p; p += (size_t)batch * P.sQI * 4;
    h->qdPos = (int*)p; p += (size_t)batch * 2 * P.sQI * 4;
    h->qdColat = (int*)p; p += (size_t)batch * 2 * P.sQI * 4;
    h->qdCand = (void*)p; p += (size_t)batch * P.sCand * sizeof(QdCand);
    h->state = (ProbState*)p;
    p += (((size_t)batch * sizeof(ProbState) + 255) / 256) * 256;
    h->small = (unsigned*)p;        // 256 bytes of scalars (tail sums of the TSQR stages)
    p += 256;
    h->sbInfo = (void*)p;
    p += (((size_t)batch * sizeof(SbInfo) + 255) / 256) * 256;
    h->sbInblk = (int*)p;
    p += (size_t)batch * P.sQI * 4;
    h->sbAct = (int*)p;
    if (h->h_state_cap < (size_t)batch) {
        if (h->h_state) GN_HIP(hipHostFree(h->h_state));
        h->h_state = nullptr;
        GN_HIP(hipHostMalloc((void**)&h->h_state, (size_t)batch * sizeof(ProbState), hipHostMallocDefault));
        if (h->h_sbinfo) GN_HIP(hipHostFree(h->h_sbinfo));
        h->h_sbinfo = nullptr;
        GN_HIP(hipHostMalloc(&h->h_sbinfo, (size_t)batch * sizeof(SbInfo), hipHostMallocDefault));
        h->h_state_cap = (size_t)batch;
    }
    h->have_plan = true;
    h->factors_valid = false;
    return 0;
}

// Problems per launch: the batch index is a grid y / z dimension (limit 65535).  Larger batches are cut into
// consecutive chunks by solve_chunked; the limit is kept a power of two so that C5's 65536 problems are two even chunks.
constexpr long long GN_MAX_LAUNCH_BATCH = 32768;

static int check_limits(enlsip_gn_handle h, long long batch, long long m, long long n, long long t) {
    if (batch < 1) { h->err = "batch must be >= 1"; return -2; }
    if (batch > (1LL << 31) - 1) { h->err = "batch must be < 2^31"; return -2; }
    if (m < 1 || m > (1LL << 27) - 4096) { h->err = "m out of range (1 .. 2^27 - 4096: 32-bit lane offsets in the update kernels)"; return -3; }
    if (n < 1 || n > 1024) { h->err = "n must be in 1..1024 in this build"; return -4; }
    if (t < 0 || t > 1024) { h->err = "t must be in 0..1024 in this build"; return -5; }
    return 0;
}

// kernels that declare more than 64 KB of dynamic LDS need the opt-in attribute
template <class KernelT>
static void big_lds(KernelT k, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
#define GN_LAUNCH_BIG(kern, grid, block, lds, stream, args) \
    do {                                                     \
        big_lds(kern, lds);                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, args); \
    } while (0)

// dispatch helpers over the rows-per-lane instantiations of the single-workgroup kernels
// Problems with at most 64 rows run with 256 or 512 threads and an LDS carve sized to the problem, so that batches of
// small problems (C3, C5) keep several workgroups resident per CU; larger ones use 1024 threads.
static void launch_constraint(int rows, int batch, hipStream_t s, ConstraintArgs a) {
    if (launch_constraint_small(batch, s, a)) return;
    constraint_carve(a.n, a.t, a.fa_done, a.nv, a.blkd, a.gld, a.matd, a.need_T, a.fl_done);
    const size_t lds = constraint_lds_bytes(a.nv, a.blkd, a.gld, a.matd);
    if (!a.fa_done && (size_t)a.n * a.t > (size_t)CMAT_DOUBLES) GN_ROUTE(ENLSIP_GN_ROUTE_CONSTRAINT_GLOBAL);
    GN_ROUTE(rows <= 32 ? ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R1_256 : rows <= 64 ? ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R1_512 :
             rows <= 128 ? ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R2 : rows <= 256 ? ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R4 :
             rows <= 512 ? ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R8 : ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R16);
    if (rows <= 32) GN_LAUNCH_BIG((k_constraint<1, 8, 256>), dim3(batch), dim3(256), lds, s, a);
    else if (rows <= 64) GN_LAUNCH_BIG((k_constraint<1, 8, 512>), dim3(batch), dim3(512), lds, s, a);
    else if (rows <= 128) GN_LAUNCH_BIG((k_constraint<2, 8, 1024>), dim3(batch), dim3(1024), lds, s, a);
    else if (rows <= 256) GN_LAUNCH_BIG((k_constraint<4, 8, 1024>), dim3(batch), dim3(1024), lds, s, a);
    else if (rows <= 512) GN_LAUNCH_BIG((k_constraint<8, 4, 1024>), dim3(batch), dim3(1024), lds, s, a);
    else GN_LAUNCH_BIG((k_constraint<16, 2, 1024>), dim3(batch), dim3(1024), lds, s, a);
}
static void launch_pivot(int rows, int batch, hipStream_t s, FinalArgs a) {
    final_carve(a.m, a.n, a.t, a.nv, a.matd);
    const size_t lds = final_lds_bytes(a.nv, a.matd);
    GN_ROUTE(rows <= 32 ? ENLSIP_GN_ROUTE_PIVOT_LDS_R1_256 : rows <= 64 ? ENLSIP_GN_ROUTE_PIVOT_LDS_R1_512 :
             rows <= 128 ? ENLSIP_GN_ROUTE_PIVOT_LDS_R2 : rows <= 256 ? ENLSIP_GN_ROUTE_PIVOT_LDS_R4 :
             rows <= 512 ? ENLSIP_GN_ROUTE_PIVOT_LDS_R8 : ENLSIP_GN_ROUTE_PIVOT_LDS_R16);
    if (rows <= 32) GN_LAUNCH_BIG((k_pivot_solve<1, 8, 256>), dim3(batch), dim3(256), lds, s, a);
    else if (rows <= 64) GN_LAUNCH_BIG((k_pivot_solve<1, 8, 512>), dim3(batch), dim3(512), lds, s, a);
    else if (rows <= 128) GN_LAUNCH_BIG((k_pivot_solve<2, 8, 1024>), dim3(batch), dim3(1024), lds, s, a);
    else if (rows <= 256) GN_LAUNCH_BIG((k_pivot_solve<4, 8, 1024>), dim3(batch), dim3(1024), lds, s, a);
    else if (rows <= 512) GN_LAUNCH_BIG((k_pivot_solve<8, 4, 1024>), dim3(batch), dim3(1024), lds, s, a);
    else GN_LAUNCH_BIG((k_pivot_solve<16, 2, 1024>), dim3(batch), dim3(1024), lds, s, a);
}

static CaqrArgs caqr_args(enlsip_gn_handle h, int k, const LevelPlan& L) {
    const Plan& P = h->plan;
    CaqrArgs a{};
    a.m = (int)P.m; a.n = (int)P.n; a.ldw = P.ldw;
    a.panel = k; a.level = L.level; a.F = P.F; a.nblocks = L.nblocks; a.S = L.S; a.tOff = L.tOff;
    a.W = h->W; a.sW = P.sW; a.Tbuf = h->Tbuf; a.sT = P.sT; a.state = h->state;
    a.ext_cols = 0; a.C = nullptr; a.sC = 0; a.reverse = 0;
    a.mode = L.mode; a.base = L.base; a.skip = L.skip; a.win = 0; a.pair = 0; a.tOff2 = 0;
    return a;
}

static void launch_factor(enlsip_gn_handle h, const CaqrArgs& a, int groups, hipStream_t st = nullptr) {
    if (!st) st = h->stream;
    dim3 grid(groups, (unsigned)h->plan.batch);
    // one-tile problems of at most 256 rows: 4 waves x 8 columns issue ~20 % fewer instructions per step than 8 x 4
    // (measured on C5: panel stage 0.63 -> 0.57 ms); everywhere else the 8-wave form wins (a 16-wave form: 8.1 -> 11.2 ms)
    if (h->plan.m > 256) {
        // a tree level with ONE node of few blocks (the top of a tree; C2's first-panel trees: 8 blocks) needs only
        // ceil(blocks / 2) registers per lane and column: the smaller forms issue fewer multiply-adds per step on rows that hold
        // nothing (the geometry of a node is the same in every form: slot ln + 64 i = block (ln >> 5) + 2 i)
        int rpl = h->plan.RPL;
        if (rpl == 8 && ((h->factor_nw4 >= 1 && a.level == 0) || h->factor_nw4 >= 2)) {      // A/B: 4 waves x 8 columns
            hipLaunchKernelGGL((k_caqr_factor<8, 4>), grid, dim3(256), 0, st, a);
            return;
        }
        if (a.level > 0 && groups == 1) {
            const int need = (a.nblocks + 1) / 2;
            const int fit = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : 8));
            if (fit < rpl) rpl = fit;
        }
        if (rpl == 8) hipLaunchKernelGGL((k_caqr_factor<8, 8>), grid, dim3(512), 0, st, a);
        else if (rpl == 4) hipLaunchKernelGGL((k_caqr_factor<4, 8>), grid, dim3(512), 0, st, a);
        else if (rpl == 2) hipLaunchKernelGGL((k_caqr_factor<2, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_caqr_factor<1, 8>), grid, dim3(512), 0, st, a);
    } else {
        if (h->plan.RPL == 8) hipLaunchKernelGGL((k_caqr_factor<8, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_caqr_factor<4, 4>), grid, dim3(256), 0, st, a);
    }
}
static void launch_update_refl(enlsip_gn_handle h, const CaqrArgs& a, int groups, int ncols) {
    dim3 grid(groups, (ncols + 31) / 32, (unsigned)h->plan.batch);
    if (h->plan.RPL == 8) hipLaunchKernelGGL(k_caqr_update_refl<8>, grid, dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL(k_caqr_update_refl<4>, grid, dim3(256), 0, h->stream, a);
}

// Apply Q0' (reverse = 0) or Q0 (reverse = 1) of the resident CAQR factors to the external
// matrix C (ldw x ncols per problem).
static void caqr_apply_ext(enlsip_gn_handle h, double* C, long long sC, int ncols, int npan, bool reverse) {
    const Plan& P = h->plan;
    for (int kk = 0; kk < npan; ++kk) {
        const int k = reverse ? npan - 1 - kk : kk;
        const auto& lv = P.panels[k].levels;
        for (size_t li = 0; li < lv.size(); ++li) {
            const LevelPlan& L = reverse ? lv[lv.size() - 1 - li] : lv[li];
            CaqrArgs a = caqr_args(h, k, L);
            a.ext_cols = ncols; a.C = C; a.sC = sC; a.reverse = reverse ? 1 : 0;
            launch_update_refl(h, a, L.groups, ncols);
        }
    }
}

// the CAQR sweep over [J2 | d]
static int run_caqr(enlsip_gn_handle h, int n2_launch) {
    const Plan& P = h->plan;
    const int kp_launch = (int)std::min<long long>(P.m, n2_launch);
    const int npan = (kp_launch + PB - 1) / PB;
    const bool use_mfma = !(h->flags & ENLSIP_GN_UPDATE_REFLECTORS);
    GN_ROUTE(P.RPL == 8 ? ENLSIP_GN_ROUTE_SWEEP_TILE512 : ENLSIP_GN_ROUTE_SWEEP_TILE256);
    if (!use_mfma) GN_ROUTE(ENLSIP_GN_ROUTE_SWEEP_REFLECTORS);
    const double mpad = (double)rup(std::max<long long>(P.m, 1), 32);
    // the launch shape is the widest J2 of the batch; narrower ones exist only when some constraint matrix was rank deficient
    // (second attempt of solve_dev) or when the caller's problems differ
    const bool mixed = n2_launch != (int)(P.n - P.kA);
    // HIP events around the level-0 far updates (the dominant kernel) when profiling is on; `bytes` = SURVEY 8d's
    // B_trail = 8 (2 m_k n_k + m_k b + b^2) of every panel the launch applies, on the columns it applies them to
    auto timed = [&](double bytes, hipStream_t st, auto&& launch) -> int {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->profiling) {
            if (h->upd_used + 2 > h->upd_ev.size()) {
                for (int q = 0; q < 2; ++q) {
                    hipEvent_t e;
                    GN_HIP(hipEventCreate(&e));
                    h->upd_ev.push_back(e);
                }
            }
            e0 = h->upd_ev[h->upd_used++];
            e1 = h->upd_ev[h->upd_used++];
            GN_HIP(hipEventRecord(e0, st));
        }
        launch();
        if (e1) {
            GN_HIP(hipEventRecord(e1, st));
            h->upd_bytes += (double)P.batch * bytes;
            h->upd_launch_bytes.push_back((double)P.batch * bytes);
        }
        return 0;
    };
    // every other trailing-update launch (tree levels, the second panel's own columns): timed too, booked under
    // ENLSIP_GN_STAGE_UPDATE and reported by enlsip_gn_get_update_totals
    auto other = [&](hipStream_t st, auto&& launch) -> int {
        hipEvent_t e1 = nullptr;
        if (h->profiling && h->profile_all_updates) {
            if (h->oth_used + 2 > h->oth_ev.size()) {
                for (int q = 0; q < 2; ++q) {
                    hipEvent_t e;
                    GN_HIP(hipEventCreate(&e));
                    h->oth_ev.push_back(e);
                }
            }
            hipEvent_t e0 = h->oth_ev[h->oth_used++];
            e1 = h->oth_ev[h->oth_used++];
            GN_HIP(hipEventRecord(e0, st));
        }
        launch();
        if (e1) GN_HIP(hipEventRecord(e1, st));
        return 0;
    };
    // few problems with many tiles each: the far update of a pair reads its grid XCD-locally, so that a tile's V stays in one L2
    // between the tile's column blocks (k_caqr_update_v4_pair); batches keep the native order (a problem's tiles are neighbours)
    auto xmap_tiles = [&](int groups) -> int { return (h->xcd_map && P.batch <= 8 && groups >= 16 && !mixed) ? groups : 0; };
    auto btrail = [&](int k, double ncols) { const double mk = mpad - (double)k * PB; return 8.0 * (2.0 * mk * ncols + mk * PB + PB * PB); };
    // the MFMA update of every trailing column of the window; with the J2 columns filling whole 32-column blocks the carried
    // right-hand side (the 32 j + 1-th column: every panel of C2) would take a block of its own: it gets its own routine as the
    // last block index instead.  With a partial last block (C3: 24 columns) it simply rides in that block.
    auto update_l0 = [&](CaqrArgs a, const LevelPlan& L, int ncols_window, int ncols_grid, hipStream_t st) {
        if ((ncols_window - 1) % 32 == 0) {
            a.skip_rhs = 1;
            launch_update_v4(h->plan.RPL, a, L.groups, ncols_grid - 1, (int)P.batch, st);
        } else launch_update_v4(h->plan.RPL, a, L.groups, ncols_grid, (int)P.batch, st);
    };
    // Look-ahead (chain-bound sweeps: one or a few problems with many tiles — C4): a pair's far update is split into the NEXT
    // pair's 64 columns (this stream) and the rest (second stream), so that the next pair's chain of small dependent launches
    // (two level-0 factorisations, their trees, the second panel's own columns) runs beside the bulk of the previous far update
    // instead of behind it.  Order: chain(K) -> E1 ; [wait E2(K-1)] near(K) ; second stream: wait E1, rest(K) -> E2.
    const long long far_wgs0 = P.batch * (((long long)mpad + 64 * P.RPL - 1) / (64 * P.RPL)) * ((n2_launch + 31) / 32);
    const bool la = P.pair && use_mfma && h->lookahead && !mixed && !h->pair_debug && h->debug_stage < 0 &&
                    ((P.batch <= 8 && far_wgs0 >= 8192) || h->lookahead_forced);
    hipStream_t sA = h->stream, sB = nullptr;
    size_t la_ev = 0;
    hipEvent_t la_prev = nullptr;                 // E2 of the previous pair's rest (second stream), not yet waited for
    auto la_event = [&](hipEvent_t& e) -> int {
        if (la_ev >= h->la_events.size()) {
            hipEvent_t ne;
            GN_HIP(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
            h->la_events.push_back(ne);
        }
        e = h->la_events[la_ev++];
        return 0;
    };
    // The same idea for the PLAIN sweep of a chain-bound problem (a single C2 problem; a 32768-row shard of C4 where pairs do
    // not pay): the factorisations of a panel (tile level and tree levels: they only read the panel's own columns) run first,
    // its trailing updates are split into the next panel's 32 columns (this stream) and the rest (second stream), and the next
    // panel's factorisations run beside that rest.  Order per panel k: F(k) -> E1 ; [wait E2(k-1)] near(k) ; second stream:
    // wait E1, rest(k) -> E2.
    // Measured (MI355X, round 4): it does NOT pay at the sizes it was meant for — a single C2 problem 4.18 -> 4.38 ms, four of them
    // 4.59 -> 4.80 ms, a 32768 x 1024 shard 15.66 -> 15.63 ms: two event hand-overs per panel cost what the overlap of a ~25 us
    // update with a ~75 us factor chain brings.  Kept behind ENLSIP_GN_LOOKAHEAD=1 (parity-tested in both sweeps), off by default.
    const bool lap = !P.pair && use_mfma && h->lookahead && !mixed && h->debug_stage < 0 && npan >= 3 && h->lookahead_forced;
    if (la || lap) {
        if (!h->stream2) {
            // lowest priority: the chain's small kernels on the main stream must not queue behind the thousands of workgroups
            // of the bulk update for a free CU slot
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            GN_HIP(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, least));
        }
        sB = h->stream2;
        GN_ROUTE(ENLSIP_GN_ROUTE_SWEEP_LOOKAHEAD);
    }
    auto la_join = [&]() -> int {                 // this stream goes on only after the second stream's last far update
        if (la_prev) {
            GN_HIP(hipStreamWaitEvent(sA, la_prev, 0));
            la_prev = nullptr;
        }
        return 0;
    };
    for (int k = 0; k < npan;) {
        if (h->debug_maxpan >= 0 && k >= h->debug_maxpan) break;   // ENLSIP_GN_DEBUG_MAXPAN: stop the sweep (debugging aid)
        const int bwk = std::min(PB, kp_launch - k * PB);
        const int ntrail = n2_launch + 1 - (k * PB + bwk);  // trailing columns incl. the augmented one
        if (h->profiling && ntrail > 0) h->upd_all_bytes += (double)P.batch * btrail(k, ntrail);   // every column right of the panel
        if (P.pair && use_mfma && !(k & 1) && k + 1 < npan) {
            // ---- panel pair (k, k + 1): tiles shared, ONE pass over the far trailing columns for both (gn_kernels_caqr.hpp) ----
            GN_ROUTE(ENLSIP_GN_ROUTE_SWEEP_PAIRS);
            if (P.panels[k].levels.size() > 1) GN_ROUTE(ENLSIP_GN_ROUTE_SWEEP_TREE);
            const int kb = k + 1;
            const auto& LA = P.panels[k].levels;
            const auto& LB = P.panels[kb].levels;
            const int bwb = std::min(PB, kp_launch - kb * PB);
            const int nfar = ntrail - bwb;                   // columns beyond the pair, incl. the augmented one (>= 1)
            if (h->profiling && nfar > 0) h->upd_all_bytes += (double)P.batch * btrail(kb, nfar);
            // Grid of the launches over the far window (win = 2).  A problem whose J2 ends inside the pair has bwb fewer pair
            // columns and as many more far columns than the launch shape says, so with mixed widths the grid spans ntrail and the
            // column block past a problem's last column exits at once.  In a uniform batch that block is ALWAYS empty — and not
            // free: with 8 tiles in x (= the 8 XCDs) a grid whose y extent is a multiple of 4 hands the empty and the light
            // (right-hand side) block of every problem to the same two of an XCD's four dispatch queues, measured 5-20 % on the
            // far launches of pairs 1, 3, 5 of a C2 step (profiles/r4_notes.md).  The exact grid has an odd y extent.
            const int far_grid = mixed ? ntrail : nfar;
            auto live = [&](int st) { return h->debug_stage < 0 || st <= h->debug_stage; };   // ENLSIP_GN_DEBUG_STAGE (debugging aid)
            if (live(0)) {   // level 0 of the first panel, applied to the second panel's columns only
                CaqrArgs a = caqr_args(h, k, LA[0]);
                launch_factor(h, a, LA[0].groups);
                a.win = 1;
                if (int rc = other(h->stream, [&] { launch_update_v4(h->plan.RPL, a, LA[0].groups, bwb, (int)P.batch, h->stream); })) return rc;
            }
            if (live(1)) {   // level 0 of the second panel: the same tiles without their first 32 rows
                CaqrArgs a = caqr_args(h, kb, LB[0]);
                launch_factor(h, a, LB[0].groups);
            }
            for (size_t li = 1; li < LA.size() && live(2); ++li) {      // tree of the first panel, applied to the second panel's columns
                CaqrArgs a = caqr_args(h, k, LA[li]);
                launch_factor(h, a, LA[li].groups);
                a.win = 1;
                if (int rc = other(h->stream, [&] { launch_update_v4(h->plan.RPL, a, LA[li].groups, bwb, (int)P.batch, h->stream); })) return rc;
            }
            for (size_t li = 1; li < LB.size() && live(3); ++li) {      // tree of the second panel
                CaqrArgs a = caqr_args(h, kb, LB[li]);
                launch_factor(h, a, LB[li].groups);
            }
            // far columns: both level-0 reflectors in one pass (a problem whose J2 ends inside the pair has more far columns
            // than the launch shape says: the grid covers ntrail, workgroups past a problem's last column exit at once)
            if (!live(4)) {
            } else if (h->pair_debug) {    // A/B: the same pair geometry, far columns in two plain passes (first panel, then second)
                CaqrArgs a = caqr_args(h, k, LA[0]);
                a.win = 2;
                update_l0(a, LA[0], nfar, ntrail, h->stream);
                CaqrArgs b0 = caqr_args(h, kb, LB[0]);
                if (live(5)) update_l0(b0, LB[0], nfar, nfar, h->stream);
            } else if (la) {
                // far columns of the pair in two parts (see above); `far` launches level 0 and the trees on columns
                // [sub0, sub0 + subn) of the far window
                auto far = [&](int sub0, int subn, hipStream_t st) -> int {
                    const int ncw = (subn > 0 ? std::min(subn, nfar - sub0) : nfar - sub0);     // launch shape of the sub-window
                    CaqrArgs a = caqr_args(h, k, LA[0]);
                    a.win = 2; a.pair = 1; a.tOff2 = LB[0].tOff; a.sub0 = sub0; a.subn = subn;
                    a.xmap = xmap_tiles(LA[0].groups);
                    // the carried right-hand side is the window's last column: a sub-window that ends before it has J2 columns only
                    const bool has_rhs = (sub0 + ncw == nfar);
                    const double by = btrail(k, ncw) + btrail(kb, ncw);
                    int rc = timed(by, st, [&] {
                        if (has_rhs) update_l0(a, LA[0], ncw, ncw, st);          // look-ahead sweeps are uniform (la implies !mixed)
                        else launch_update_v4(h->plan.RPL, a, LA[0].groups, ncw, (int)P.batch, st);
                    });
                    if (rc) return rc;
                    for (size_t li = 1; li < LA.size(); ++li) {
                        CaqrArgs t = caqr_args(h, k, LA[li]);
                        t.win = 2; t.sub0 = sub0; t.subn = subn;
                        if (int rc2 = other(st, [&] { launch_update_v4(h->plan.RPL, t, LA[li].groups, ncw, (int)P.batch, st); })) return rc2;
                    }
                    for (size_t li = 1; li < LB.size(); ++li) {
                        CaqrArgs t = caqr_args(h, kb, LB[li]);
                        t.sub0 = sub0; t.subn = subn;
                        if (int rc2 = other(st, [&] { launch_update_v4(h->plan.RPL, t, LB[li].groups, ncw, (int)P.batch, st); })) return rc2;
                    }
                    return 0;
                };
                const int near = 2 * PB;
                if (nfar <= near) {               // nothing beyond the next pair's columns: one part, this stream
                    if (int rcj = la_join()) return rcj;
                    if (int rc = far(0, 0, sA)) return rc;
                } else {
                    hipEvent_t e1, e2;
                    if (int rc = la_event(e1)) return rc;
                    if (int rc = la_event(e2)) return rc;
                    GN_HIP(hipEventRecord(e1, sA));                     // the pair's reflectors and T factors are complete
                    if (int rcj = la_join()) return rcj;                // the previous pair's rest covers the columns `near` touches
                    if (int rc = far(0, near, sA)) return rc;
                    GN_HIP(hipStreamWaitEvent(sB, e1, 0));
                    if (int rc = far(near, 0, sB)) return rc;
                    GN_HIP(hipEventRecord(e2, sB));
                    la_prev = e2;
                }
                k += 2;
                continue;
            } else {
                CaqrArgs a = caqr_args(h, k, LA[0]);
                a.win = 2; a.pair = 1; a.tOff2 = LB[0].tOff;
                a.xmap = xmap_tiles(LA[0].groups);
                int rc = timed(btrail(k, nfar) + btrail(kb, nfar), h->stream, [&] { update_l0(a, LA[0], nfar, far_grid, h->stream); });
                if (rc) return rc;
                if (mixed) {        // problems whose J2 ends before the second panel: the first panel alone, every trailing column
                    a.pair = 2;
                    update_l0(a, LA[0], nfar, ntrail, h->stream);
                }
            }
            for (size_t li = 1; li < LA.size() && live(6); ++li) {
                CaqrArgs a = caqr_args(h, k, LA[li]);
                a.win = 2;
                if (int rc = other(h->stream, [&] { launch_update_v4(h->plan.RPL, a, LA[li].groups, far_grid, (int)P.batch, h->stream); })) return rc;
            }
            for (size_t li = 1; li < LB.size() && live(7); ++li) {
                CaqrArgs a = caqr_args(h, kb, LB[li]);
                if (int rc = other(h->stream, [&] { launch_update_v4(h->plan.RPL, a, LB[li].groups, nfar, (int)P.batch, h->stream); })) return rc;
            }
            k += 2;
            continue;
        }
        // ---- one panel ----
        // last panel narrower than 32 with d as the only trailing column: d rides through the factor kernels
        const bool passenger = (ntrail == 1 && bwk < PB && kp_launch == n2_launch);
        GN_ROUTE(passenger ? ENLSIP_GN_ROUTE_SWEEP_PASSENGER : ENLSIP_GN_ROUTE_SWEEP_PLAIN);
        if (P.panels[k].levels.size() > 1) GN_ROUTE(ENLSIP_GN_ROUTE_SWEEP_TREE);
        if (lap && !passenger && ntrail > 0) {
            const auto& LV = P.panels[k].levels;
            for (const LevelPlan& L : LV) {                      // every factorisation of the panel first
                CaqrArgs a = caqr_args(h, k, L);
                launch_factor(h, a, L.groups);
            }
            // updates of the levels, in order, on columns [sub0, sub0 + subn) of the trailing window (subn = 0: to its end)
            auto upd = [&](int sub0, int subn, hipStream_t st) -> int {
                const int ncw = (subn > 0 ? std::min(subn, ntrail - sub0) : ntrail - sub0);
                const bool has_rhs = (sub0 + ncw == ntrail);       // the carried right-hand side is the window's last column
                for (const LevelPlan& L : LV) {
                    CaqrArgs a = caqr_args(h, k, L);
                    a.sub0 = sub0; a.subn = subn;
                    if (L.level == 0) {
                        int rc = timed(btrail(k, ncw), st, [&] {
                            if (has_rhs) update_l0(a, L, ncw, ncw, st);
                            else launch_update_v4(h->plan.RPL, a, L.groups, ncw, (int)P.batch, st);
                        });
                        if (rc) return rc;
                    } else {
                        if (int rc = other(st, [&] { launch_update_v4(h->plan.RPL, a, L.groups, ncw, (int)P.batch, st); })) return rc;
                    }
                }
                return 0;
            };
            if (ntrail <= PB) {                                    // nothing beyond the next panel's columns
                if (int rcj = la_join()) return rcj;
                if (int rc = upd(0, 0, sA)) return rc;
            } else {
                hipEvent_t e1, e2;
                if (int rc = la_event(e1)) return rc;
                if (int rc = la_event(e2)) return rc;
                GN_HIP(hipEventRecord(e1, sA));                     // the panel's reflectors and T factors are complete
                if (int rcj = la_join()) return rcj;                // the previous panel's rest covers the columns `near` touches
                if (int rc = upd(0, PB, sA)) return rc;
                GN_HIP(hipStreamWaitEvent(sB, e1, 0));
                if (int rc = upd(PB, 0, sB)) return rc;
                GN_HIP(hipEventRecord(e2, sB));
                la_prev = e2;
            }
            ++k;
            continue;
        }
        if (int rcj = la_join()) return rcj;
        for (const LevelPlan& L : P.panels[k].levels) {
            CaqrArgs a = caqr_args(h, k, L);
            a.npass = passenger ? 1 : 0;
            launch_factor(h, a, L.groups);
            if (ntrail > 0 && !passenger) {
                if (use_mfma && L.level == 0) {
                    int rc = timed(btrail(k, ntrail), h->stream, [&] { update_l0(a, L, ntrail, ntrail, h->stream); });
                    if (rc) return rc;
                } else if (use_mfma) {
                    if (int rc = other(h->stream, [&] { launch_update_v4(h->plan.RPL, a, L.groups, ntrail, (int)P.batch, h->stream); })) return rc;
                }
                else launch_update_refl(h, a, L.groups, ntrail);
            }
        }
        ++k;
    }
    if (int rcj = la_join()) return rcj;
    GN_HIP(hipGetLastError());
    return 0;
}

// distributed column-pivoted QR of R0 (one launch per pivot step over all problems)
static int run_qrcp_dist(enlsip_gn_handle h, int n2_launch) {
    const Plan& P = h->plan;
    const int kp_launch = (int)std::min<long long>(P.m, n2_launch);
    QdArgs a{};
    a.n = (int)P.n; a.ldw = P.ldw; a.ldr = P.ldr; a.step = 0; a.prob0 = 0;
    a.W = h->W; a.sW = P.sW; a.M = h->qdM; a.sM = P.sM; a.Vb = h->qdVb; a.sVb = P.sVb; a.Rt = h->Rt; a.sRt = P.sRt;
    a.tau = h->tauJ; a.sTau = P.sTauJ; a.diag = h->qdDiag; a.sDiag = P.sDiag;
    a.vn1 = h->qdVn1; a.vn2 = h->qdVn2; a.sVn = P.sVn;
    a.chosen = h->qdChosen; a.pos = h->qdPos; a.colat = h->qdColat; a.sI = P.sQI;
    a.cand = (QdCand*)h->qdCand; a.sCand = P.sCand; a.Gmax = P.qdGmax;
    a.jpvt = h->jpvtJ; a.sJ = P.sJJ; a.state = h->state;
    a.n2cap = n2_launch;
    const int G = (n2_launch + 1 + QD_CPW - 1) / QD_CPW;
    dim3 grid(G, (unsigned)P.batch);
    GN_ROUTE(ENLSIP_GN_ROUTE_PIVOT_STEPS);
    const bool big = kp_launch > 512;
    if (big) hipLaunchKernelGGL(k_qd_init<16>, grid, dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL(k_qd_init<8>, grid, dim3(256), 0, h->stream, a);
    for (int j = 0; j < kp_launch; ++j) {
        a.step = j;
        if (big) hipLaunchKernelGGL(k_qd_step<16>, grid, dim3(256), 0, h->stream, a);
        else hipLaunchKernelGGL(k_qd_step<8>, grid, dim3(256), 0, h->stream, a);
    }
    hipLaunchKernelGGL(k_qd_assemble, grid, dim3(256), 0, h->stream, a);
    GN_HIP(hipGetLastError());
    return 0;
}

// blocked pivoted QR of R0 with verified pivots (gn_kernels_qrcp_block.hpp)
// More than 512 rows do not fit the register forms: the stage then opens with a launch-per-step HEAD (k_qd_step, 8.4 us per step)
// until 512 rows are left and hands the rest to the blocks (~3 us per step + ~50 us per block).  The two forms share their state
// (M, norms, maps, reflectors by position); the head ends on an even step so that the maps sit in parity 0, where the blocks keep
// them.  C4's 1024 x 1024 combine: 1024 head steps = 8.6 ms -> 512 head steps + 16 blocks.
static int run_qrcp_block(enlsip_gn_handle h, int n2_launch) {
    const Plan& P = h->plan;
    const int kp_launch = (int)std::min<long long>(P.m, n2_launch);
    int jhead = kp_launch > 512 ? kp_launch - 512 : 0;
    jhead += jhead & 1;
    GN_ROUTE(jhead > 0 ? ENLSIP_GN_ROUTE_PIVOT_HYBRID : ENLSIP_GN_ROUTE_PIVOT_BLOCKS);
    SbArgs a{};
    QdArgs& q = a.q;
    q.n = (int)P.n; q.ldw = P.ldw; q.ldr = P.ldr; q.step = -1; q.prob0 = 0;
    q.W = h->W; q.sW = P.sW; q.M = h->qdM; q.sM = P.sM; q.Vb = h->qdVb; q.sVb = P.sVb; q.Rt = h->Rt; q.sRt = P.sRt;
    q.tau = h->tauJ; q.sTau = P.sTauJ; q.diag = h->qdDiag; q.sDiag = P.sDiag;
    q.vn1 = h->qdVn1; q.vn2 = h->qdVn2; q.sVn = P.sVn;
    q.chosen = h->qdChosen; q.pos = h->qdPos; q.colat = h->qdColat; q.sI = P.sQI;
    q.cand = (QdCand*)h->qdCand; q.sCand = P.sCand; q.Gmax = P.qdGmax;
    q.jpvt = h->jpvtJ; q.sJ = P.sJJ; q.state = h->state;
    q.n2cap = n2_launch;
    a.info = (SbInfo*)h->sbInfo; a.inblk = h->sbInblk; a.sIn = P.sQI; a.blkid = 0;
    a.Tsb = h->sbT; a.sTsb = PB * PB; a.act = h->sbAct; a.sAct = P.sQI + 32;
    a.dbg = nullptr;
    const int G = (n2_launch + 1 + QD_CPW - 1) / QD_CPW;
    dim3 grid(G, (unsigned)P.batch);
    hipStream_t s = h->stream;
    // Row-count statistics per block id (see SbArgs::rows_stat).  Every block is launched in up to three forms of the select /
    // factor kernel and a problem runs in the one that fits its current row count; in a batch of similar problems two of the
    // three launches of every block id find nothing to do — 24-40 us each at batch 384 (a launch of 384 x 512 threads that only
    // reads its state), 1.3 ms of a C2 step.  The previous solve of the same shape on this handle tells which forms a block id
    // needs; a problem that a skipped form would have served simply makes no step in that block id, the end-of-stage check
    // sees it and the stage goes on WITHOUT hints, so the result never depends on them.
    {
        int rc = grow(h, h->sb_stat, 2 * SB_STAT_BLKS * sizeof(int));
        if (rc) return rc;
        if (!h->h_sb_stat) GN_HIP(hipHostMalloc((void**)&h->h_sb_stat, 2 * SB_STAT_BLKS * sizeof(int), hipHostMallocDefault));
        GN_HIP(hipMemsetAsync(h->sb_stat.p, 0, 2 * SB_STAT_BLKS * sizeof(int), s));
        a.rows_stat = (int*)h->sb_stat.p;
    }
    bool hints = h->sb_form_hints && h->sb_rows_kp == kp_launch && h->sb_rows_batch == P.batch && !h->sb_rows_max.empty();
    const bool hinted = hints;
    bool fell_back = false;
    if (jhead > 0) {
        hipLaunchKernelGGL(k_qd_init<16>, grid, dim3(256), 0, s, q);
        for (int j = 0; j < jhead; ++j) {
            q.step = j;
            hipLaunchKernelGGL(k_qd_step<16>, grid, dim3(256), 0, s, q);
        }
        q.step = -1;
        q.hyb = jhead;
    } else hipLaunchKernelGGL(k_qd_init<8>, grid, dim3(256), 0, s, q);
    hipLaunchKernelGGL(k_sb_reset, dim3(((unsigned)P.n + 255) / 256, (unsigned)P.batch), dim3(256), 0, s, a, (int)P.n, jhead);
    const int kp_blk = kp_launch - jhead;           // steps (= rows) left to the blocks
    dim3 ugrid((n2_launch + 1 + SB_UCW - 1) / SB_UCW, (unsigned)P.batch);
    int it = 0;
    // blocks of <= 32 steps; the first chunk is sized from the previous solve on this handle (one host check per
    // solve in steady state), later chunks are small
    int chunk = std::min(kp_blk, h->sb_hint > 0 ? h->sb_hint : kp_blk / 16 + 4);
    SbInfo* hinfo = (SbInfo*)h->h_sbinfo;
    // rows_bound: an upper bound of kp - j0 over the problems.  A block in which EVERY form the bound asks for was launched
    // serves every unfinished problem, and a served problem makes at least one step: the bound then falls by one.  A block
    // in which the hints suppressed one of those forms may have left a problem unserved: the bound stays, and the read-back
    // at the end of the chunk replaces it by the exact maximum.  (Until round 5 the forms were gated on kp_blk - it, which
    // assumes a step per block for everybody: a problem that a hinted chunk had left behind above 256 rows was never served
    // once kp_blk - it had fallen to 256, and the loop ran out with wrong factors.)
    int rows_bound = kp_blk;
    const int max_blocks = 2 * kp_blk + 64;         // without hints kp_blk blocks always suffice; hinted chunks may idle
    bool done = kp_blk <= 0;
    while (!done) {
        for (int i = 0; i < chunk && rows_bound > 0; ++i, ++it) {
            a.blkid = it;
            GN_TRACE(h, "  qrcp block %d", it);
            // candidates in the registers of one workgroup (kp <= 512), block reflector applied to the still-active columns.
            // Up to three forms per block, each problem runs in the one that fits its current row count kp - j0
            // (gn_kernels_qrcp_block_reg.hpp): the large forms are no longer launched once the bound fits a smaller one.
            const dim3 fg((unsigned)P.batch);
            const bool big0 = rows_bound > 256, med0 = rows_bound > 128;
            bool big = big0, med = med0, small = true;
            if (hints && it < (int)h->sb_rows_max.size() && h->sb_rows_max[it] > 0) {
                // what the previous solve saw at this block id, widened by a block's worth of steps either way
                const int lo = h->sb_rows_min[it] - 32, hi = h->sb_rows_max[it] + 32;
                big = big && hi > 256;
                med = med && hi > 128 && lo <= 256;
                small = lo <= 128;
            }
            if (big) {
                GN_ROUTE(kp_blk <= 448 ? ENLSIP_GN_ROUTE_PIVOT_BLOCKS_448 : ENLSIP_GN_ROUTE_PIVOT_BLOCKS_512);
                if (kp_blk <= 448) hipLaunchKernelGGL((k_sb_factor_reg<7, 8, 4>), fg, dim3(512), 0, s, a);
                else hipLaunchKernelGGL((k_sb_factor_reg<8, 8, 4>), fg, dim3(512), 0, s, a);
            }
            if (med) { GN_ROUTE(ENLSIP_GN_ROUTE_PIVOT_BLOCKS_256); hipLaunchKernelGGL((k_sb_factor_reg<4, 8, 2>), fg, dim3(512), 0, s, a); }
            if (small) { GN_ROUTE(ENLSIP_GN_ROUTE_PIVOT_BLOCKS_128); hipLaunchKernelGGL((k_sb_factor_reg<2, 8, 0>), fg, dim3(512), 0, s, a); }
            hipLaunchKernelGGL(k_sb_update_blk, ugrid, dim3(256), 0, s, a);
            if (big == big0 && med == med0 && small) --rows_bound;
        }
        GN_HIP(hipGetLastError());
        GN_HIP(hipMemcpyAsync(hinfo, h->sbInfo, (size_t)P.batch * sizeof(SbInfo), hipMemcpyDeviceToHost, s));
        GN_HIP(hipMemcpyAsync(h->h_state, h->state, (size_t)P.batch * sizeof(ProbState), hipMemcpyDeviceToHost, s));
        GN_HIP(hipMemcpyAsync(h->h_sb_stat, h->sb_stat.p, 2 * SB_STAT_BLKS * sizeof(int), hipMemcpyDeviceToHost, s));
        GN_HIP(hipStreamSynchronize(s));
        done = true;
        int rows_left = 0;                       // exact maximum of kp - j0 over the unfinished problems
        for (long long k = 0; k < P.batch; ++k) {
            if (h->h_state[k].n2 > n2_launch) continue;            // wider problems are skipped here
            const int left = h->h_state[k].kp - hinfo[k].j0;
            if (left > 0) {
                done = false;
                rows_left = std::max(rows_left, left);
            }
        }
        if (done) {
            int used = 0;
            for (long long k = 0; k < P.batch; ++k) used = std::max(used, hinfo[k].blk + 1);
            h->sb_hint = used + 1;
            // statistics of this solve = hints of the next one (only of a stage that ran on hints it could trust or on none:
            // a block id in which some problem found no form to run in shows smaller counts than it should)
            if (hinted && fell_back) {      // the hints misled this stage: its statistics are incomplete; the next solve runs without
                h->sb_rows_max.clear();
                h->sb_rows_min.clear();
                h->sb_rows_kp = -1;
                break;
            }
            const int nb = std::min(it, SB_STAT_BLKS);
            h->sb_rows_max.assign(nb, 0);
            h->sb_rows_min.assign(nb, 0);
            for (int b = 0; b < nb; ++b) {
                h->sb_rows_max[b] = h->h_sb_stat[b];
                h->sb_rows_min[b] = h->h_sb_stat[b] > 0 ? SB_STAT_OFF - h->h_sb_stat[SB_STAT_BLKS + b] : 0;
            }
            h->sb_rows_kp = kp_launch;
            h->sb_rows_batch = P.batch;
            break;
        }
        if (it >= max_blocks) {
            // cannot happen with every form launched (a served problem makes a step per block); never assemble half-done factors
            h->sb_rows_max.clear();
            h->sb_rows_min.clear();
            h->sb_rows_kp = -1;
            h->err = "blocked pivoted QR of R0 did not finish within its block budget";
            return 996;
        }
        hints = false;          // somebody is not done: the rest of the stage launches every form
        fell_back = true;
        rows_bound = std::min(rows_left, 512);
        chunk = 4;
    }
    hipLaunchKernelGGL(k_qd_assemble, grid, dim3(256), 0, s, q);
    GN_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// rescale path (gn_rescale.hpp): scale the resident factors of ONE problem back to the caller's data
// ---------------------------------------------------------------------------------------------
static void scale_region(hipStream_t s, double* X, long long ld, long long rows, long long cols, int shift, int upper) {
    if (rows <= 0 || cols <= 0 || shift == 0) return;
    hipLaunchKernelGGL(k_scale_region, dim3((unsigned)std::min<long long>((rows + 255) / 256, 1024), (unsigned)cols), dim3(256), 0, s,
                       X, ld, (int)rows, (int)cols, shift, upper);
}
// F_A.R, F_L11.R, b (and the pieces the distributed constraint stage keeps for the re-solve): x 2^-sc_eA.  p1, every tau, every
// reflector vector and the T blocks of Q1 do not depend on the scale.
static int unscale_constraint_side(enlsip_gn_handle h) {
    const Plan& P = h->plan;
    hipStream_t s = h->stream;
    const int sh = -h->sc_eA;
    scale_region(s, h->FA, P.n, P.kA, P.t, sh, 1);
    scale_region(s, h->FL, P.t, std::min<long long>(P.t, P.kA), P.kA, sh, 1);
    scale_region(s, h->bvec, P.t, P.t, 1, sh, 0);
    if (h->cdist.valid) {
        scale_region(s, const_cast<double*>(h->cdist.L), h->cdist.ldL, P.t, P.kA, sh, 0);
        scale_region(s, const_cast<double*>(h->cdist.qb), P.t, P.t, 1, sh, 0);
    }
    GN_HIP(hipGetLastError());
    return 0;
}
// J1 = (J*Q1)[:, 1:rankA], the carried right-hand side d, the pivoted factor R of J2 with its carried column, the saved leading
// entries of Q0'd, and the outputs d (device): x 2^-sc_eJ.  (R0 inside W is consumed by the pivoted QR only; V and T of the CAQR
// and of the pivoted QR do not depend on the scale.)
static int unscale_jacobian_side(enlsip_gn_handle h, double* dd) {
    const Plan& P = h->plan;
    hipStream_t s = h->stream;
    const int sh = -h->sc_eJ;
    const ProbState& st = h->h_state[0];
    scale_region(s, h->W, P.ldw, P.ldw, st.rankA, sh, 0);
    scale_region(s, h->W + (size_t)P.n * P.ldw, P.ldw, P.ldw, 1, sh, 0);
    scale_region(s, h->Rt, P.ldr, st.kp, st.n2, sh, 1);
    scale_region(s, h->Rt + (size_t)st.n2 * P.ldr, P.ldr, st.kp, 1, sh, 0);
    scale_region(s, h->zsave, P.ldr, st.kp, 1, sh, 0);
    scale_region(s, h->qdDiag, P.ldr, st.kp, 1, sh, 0);
    if (dd) scale_region(s, dd, P.m, P.m, 1, sh, 0);
    GN_HIP(hipGetLastError());
    return 0;
}
// largest |entry| of the four inputs of problem 0 of a one-problem solve -> power-of-two shifts that bring J, rx / A', cx to
// magnitude ~1 (0: inside the band, zero, or not finite — nothing to rescale)
static int extreme_shifts(enlsip_gn_handle h, long long m, long long n, long long t, const double* dJ, long long ldj, const double* drx,
                          const double* dAt, long long ldat, const double* dcx, int* shiftJ, int* shiftA) {
    hipStream_t s = h->stream;
    unsigned long long* dmx = (unsigned long long*)(h->small + 32);      // 2 words of the handle's 256-byte scalar area
    GN_HIP(hipMemsetAsync(dmx, 0, 16, s));
    if (dJ && drx) {
        hipLaunchKernelGGL(k_amax_bits, dim3((unsigned)n), dim3(256), 0, s, dJ, ldj, (int)m, (int)n, dmx);
        hipLaunchKernelGGL(k_amax_bits, dim3(1), dim3(256), 0, s, drx, m, (int)m, 1, dmx);
    }
    if (t > 0 && dAt && dcx) {
        hipLaunchKernelGGL(k_amax_bits, dim3((unsigned)t), dim3(256), 0, s, dAt, ldat, (int)n, (int)t, dmx + 1);
        hipLaunchKernelGGL(k_amax_bits, dim3(1), dim3(256), 0, s, dcx, t, (int)t, 1, dmx + 1);
    }
    GN_HIP(hipGetLastError());
    unsigned long long hb[2] = {0, 0};
    GN_HIP(hipMemcpyAsync(hb, dmx, 16, hipMemcpyDeviceToHost, s));
    GN_HIP(hipStreamSynchronize(s));
    auto shift_of = [](unsigned long long bits) -> int {
        double a;
        memcpy(&a, &bits, 8);
        if (!(a > 0.0) || !std::isfinite(a)) return 0;
        int e;
        (void)std::frexp(a, &e);                    // a = f 2^e, 0.5 <= f < 1
        return (e > GN_RESCALE_BAND || e < -GN_RESCALE_BAND) ? -(e - 1) : 0;
    };
    *shiftJ = shift_of(hb[0]);
    *shiftA = shift_of(hb[1]);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// core: device-pointer batched solve
// ---------------------------------------------------------------------------------------------
// Constraint stage with many constraints: both pivoted factorisations through the distributed QR (one launch per pivot step),
// then k_constraint only does the rank decisions, the triangular solves and the T blocks.
static int run_constraint_dist(enlsip_gn_handle h, ConstraintArgs ca, long long batch, long long n, long long t) {
    const Plan& P = h->plan;
    hipStream_t s = h->stream;
    const int kA = P.kA;
    auto pad = [](long long x) { return rup(std::max<long long>(x, 1), 32); };
    const long long ldc = rup(std::max(n, t), 8);
    const long long sMc = pad(ldc * (t + 2)), sVbc = pad(ldc * (kA + 1)), sRtc = pad(ldc * (t + 2)), sLc = pad(ldc * (kA + 1));
    const long long sVec = pad(std::max(n, t) + 1), sIc = pad(t + 1);
    const int Gc = (int)((t + 1 + QD_CPW - 1) / QD_CPW);
    const long long sCandc = 2LL * Gc;
    const size_t bytes = (size_t)batch * ((sMc + sVbc + sRtc + sLc + 5 * sVec) * 8 + 5 * sIc * 4 + sCandc * sizeof(QdCand) +
                                          2 * sizeof(ProbState)) + 4096;
    int rc = grow(h, h->cws, bytes);
    if (rc) return rc;
    char* p = (char*)h->cws.p;
    auto carve = [&](size_t b) { char* r = p; p += (b + 255) / 256 * 256; return r; };
    double* cM = (double*)carve((size_t)batch * sMc * 8);
    double* cVb = (double*)carve((size_t)batch * sVbc * 8);
    double* cRt = (double*)carve((size_t)batch * sRtc * 8);
    double* cL = (double*)carve((size_t)batch * sLc * 8);
    double* cDiag = (double*)carve((size_t)batch * sVec * 8);
    double* cVn1 = (double*)carve((size_t)batch * sVec * 8);
    double* cVn2 = (double*)carve((size_t)batch * sVec * 8);
    double* cBq = (double*)carve((size_t)batch * sVec * 8);       // b_buff, later F_L11.Q' b_buff
    double* cQb = (double*)carve((size_t)batch * sVec * 8);
    int* cChosen = (int*)carve((size_t)batch * sIc * 4);
    int* cPos = (int*)carve((size_t)batch * 2 * sIc * 4);
    int* cColat = (int*)carve((size_t)batch * 2 * sIc * 4);
    QdCand* cCand = (QdCand*)carve((size_t)batch * sCandc * sizeof(QdCand));
    ProbState* stA = (ProbState*)carve((size_t)batch * sizeof(ProbState));
    ProbState* stL = (ProbState*)carve((size_t)batch * sizeof(ProbState));
    const unsigned gb = (unsigned)((batch + 255) / 256);
    hipLaunchKernelGGL(k_fake_state, dim3(gb), dim3(256), 0, s, stA, (int)batch, kA, (int)t);
    hipLaunchKernelGGL(k_fake_state, dim3(gb), dim3(256), 0, s, stL, (int)batch, kA, kA);

    QdArgs q{};
    q.n = (int)n; q.ldw = 0; q.ldr = (int)ldc; q.prob0 = 0;
    q.M = cM; q.sM = sMc; q.Vb = cVb; q.sVb = sVbc; q.Rt = cRt; q.sRt = sRtc;
    q.diag = cDiag; q.sDiag = sVec; q.vn1 = cVn1; q.vn2 = cVn2; q.sVn = sVec;
    q.chosen = cChosen; q.pos = cPos; q.colat = cColat; q.sI = sIc;
    q.cand = cCand; q.sCand = sCandc; q.Gmax = Gc;
    auto factor = [&](int rows, int cols, int steps) {
        const dim3 grid((cols + 1 + QD_CPW - 1) / QD_CPW, (unsigned)batch);
        const bool big = rows > 512;
        q.step = 0;
        if (big) hipLaunchKernelGGL(k_qd_init<16>, grid, dim3(256), 0, s, q);
        else hipLaunchKernelGGL(k_qd_init<8>, grid, dim3(256), 0, s, q);
        for (int j = 0; j < steps; ++j) {
            q.step = j;
            if (big) hipLaunchKernelGGL(k_qd_step<16>, grid, dim3(256), 0, s, q);
            else hipLaunchKernelGGL(k_qd_step<8>, grid, dim3(256), 0, s, q);
        }
        q.step = 0;
        hipLaunchKernelGGL(k_qd_assemble, grid, dim3(256), 0, s, q);
    };
    // ---- F_A: the n x t matrix C.A' -------------------------------------------------------------------------------
    q.rows = (int)n; q.in_mode = 1; q.Ain = ca.At; q.ldain = ca.ldat; q.sAin = ca.strideAt;
    q.tau = h->tauA; q.sTau = P.sTauA; q.jpvt = h->jpvtA; q.sJ = P.sJA; q.state = stA;
    factor((int)n, (int)t, kA);
    hipLaunchKernelGGL(k_copy_cols, dim3((unsigned)t, (unsigned)batch), dim3(256), 0, s, h->FA, n, P.sFA, cRt, ldc, sRtc, (int)n, (int)t);
    // ---- F_L11: the t x kA lower trapezoid R_A', carrying b_buff = -cx[F_A.p] ----------------------------------------
    hipLaunchKernelGGL(k_bbuff, dim3((unsigned)((t + 255) / 256), (unsigned)batch), dim3(256), 0, s, cBq, sVec, ca.cx, ca.stride_cx,
                       h->jpvtA, P.sJA, (int)t);
    q.rows = (int)t; q.in_mode = 2; q.Ain = h->FA; q.ldain = n; q.sAin = P.sFA; q.rin = cBq; q.sRin = sVec; q.Lout = cL; q.sLout = sLc;
    q.tau = h->tauL; q.sTau = P.sTauL; q.jpvt = h->jpvtL; q.sJ = P.sJL; q.state = stL;
    factor((int)t, kA, (int)std::min<long long>(t, kA));
    hipLaunchKernelGGL(k_copy_cols, dim3((unsigned)kA, (unsigned)batch), dim3(256), 0, s, h->FL, t, P.sFL, cRt, ldc, sRtc, (int)t, kA);
    hipLaunchKernelGGL(k_copy_cols, dim3(1, (unsigned)batch), dim3(256), 0, s, cQb, sVec, sVec, cRt + (size_t)kA * ldc, ldc, sRtc, (int)t, 1);
    // ---- ranks, triangular solves, T blocks ---------------------------------------------------------------------------
    ca.fa_done = 1; ca.fl_done = 1; ca.need_T = 1;
    ca.Lmat = cL; ca.ldL = ldc; ca.sL = sLc; ca.qb = cQb; ca.sQb = sVec;
    h->cdist.L = cL; h->cdist.ldL = ldc; h->cdist.sL = sLc; h->cdist.qb = cQb; h->cdist.sQb = sVec; h->cdist.valid = true;
    launch_constraint((int)std::max(n, t), (int)batch, s, ca);
    GN_HIP(hipGetLastError());
    return 0;
}

// F_A, rankA, F_L11, b, p1, block T of Q1 for every problem of the batch (plan already made)
// prob0 / code_ov: the re-solve of ONE resident problem (enlsip_gn_resolve) takes the same route as the solve that produced its
// factors, so that they are rewritten bit for bit by the same kernels.
static int run_constraint_stage(enlsip_gn_handle h, long long batch, long long m, long long n, long long t, const double* dAt,
                                long long ldat, long long strideAt, const double* dcx, double eps_rank, long long dimA_ov,
                                int prob0 = 0, int code_ov = 0) {
    const Plan& P = h->plan;
    hipStream_t s = h->stream;
    if (prob0 == 0 && code_ov == 0) h->cdist.valid = false;
    // the resident problem was rescaled (sc_eA != 0: one problem): the stage runs on the scaled copies of A', cx with the absolute
    // rank threshold scaled alike, and what it leaves resident is scaled back below
    const bool scaledA = h->sc_eA != 0 && batch == 1 && prob0 == 0 && t > 0;
    if (scaledA) { dAt = h->rs_At; ldat = n; strideAt = n * t; dcx = h->rs_cx; }
    ConstraintArgs ca{};
    ca.n = (int)n; ca.t = (int)t; ca.kA = P.kA; ca.m = (int)m; ca.eps_rank = eps_rank;
    ca.abs_shift = scaledA ? h->sc_eA : 0;
    ca.dimA_override = (int)dimA_ov; ca.code_override = code_ov; ca.prob0 = prob0;
    ca.At = dAt; ca.ldat = ldat; ca.strideAt = strideAt; ca.cx = dcx; ca.stride_cx = t;
    ca.FA = h->FA; ca.sFA = P.sFA; ca.tauA = h->tauA; ca.sTauA = P.sTauA; ca.jpvtA = h->jpvtA; ca.sJA = P.sJA;
    ca.FL = h->FL; ca.sFL = P.sFL; ca.tauL = h->tauL; ca.sTauL = P.sTauL; ca.jpvtL = h->jpvtL; ca.sJL = P.sJL;
    ca.TA = h->TA; ca.sTA = P.sTA; ca.p1 = h->p1; ca.sP1 = P.sP1; ca.bvec = h->bvec; ca.sB = P.sB;
    ca.state = h->state;
    // many constraints: both factorisations through the distributed pivoted QR
    if (t > 64 && (size_t)n * t > (size_t)CMAT_DOUBLES) {
        GN_ROUTE(ENLSIP_GN_ROUTE_CONSTRAINT_DIST);
        int rcd = run_constraint_dist(h, ca, batch, n, t);
        if (rcd) return rcd;
        return scaledA ? unscale_constraint_side(h) : 0;
    }
    // F_A of a matrix that does not fit the LDS area of k_constraint: whole matrix in registers (gn_kernels_geqp3_reg.hpp)
    if (t >= 1 && t <= 64 && n <= 512 && (size_t)n * t > (size_t)CMAT_DOUBLES) {
        Geqp3RegArgs ga{};
        ga.rows = (int)n; ga.cols = (int)t; ga.A = dAt; ga.lda = ldat; ga.strideA = strideAt;
        ga.F = h->FA; ga.sF = P.sFA; ga.tau = h->tauA; ga.sTau = P.sTauA; ga.jpvt = h->jpvtA; ga.sJ = P.sJA;
        ga.T = h->TA; ga.sT = P.sTA; ga.prob0 = prob0;
        GN_ROUTE(n <= 256 ? ENLSIP_GN_ROUTE_CONSTRAINT_REG4 : ENLSIP_GN_ROUTE_CONSTRAINT_REG8);
        if (n <= 256) hipLaunchKernelGGL(k_geqp3_reg<4>, dim3((unsigned)batch), dim3(512), 0, s, ga);
        else hipLaunchKernelGGL(k_geqp3_reg<8>, dim3((unsigned)batch), dim3(512), 0, s, ga);
        ca.fa_done = 1;
    }
    // with F_A done the kernel only factors the t x kA matrix R_A': size its rows-per-lane instantiation (and LDS) for that
    launch_constraint(ca.fa_done ? (int)std::max<long long>(t, 1) : (int)std::max(n, t), (int)batch, s, ca);
    GN_HIP(hipGetLastError());
    return scaledA ? unscale_constraint_side(h) : 0;
}

static int solve_dev(enlsip_gn_handle h, long long batch, long long m, long long n, long long t,
                     const double* dJ, long long ldj, long long strideJ, const double* drx,
                     const double* dAt, long long ldat, long long strideAt, const double* dcx,
                     double eps_rank, long long dimA_ov, long long dimJ2_ov,
                     double* dp, double* db, double* dd, enlsip_gn_info* dinfo,
                     long long* djA, long long* djL, long long* djJ, enlsip_gn_info* hinfo = nullptr) {
    h->split = 0;   // routing of accessors to the pipeline child is (re)established by the batched entry point
    h->chunk0 = 0;  // ... and to the resident chunk by solve_chunked
    gn_route_acc = 0;
    const bool reuse = h->reuse_once;
    h->reuse_once = false;
    const bool upper_in = h->upper_once && t == 0 && m <= n;      // J is upper triangular (and unconstrained): it IS its own R0, Q0 = I
    h->upper_once = false;
    int rc = check_limits(h, batch, m, n, t);
    if (rc) return rc;
    if (ldj < m) { h->err = "ldj < m"; return -7; }
    if (t > 0 && ldat < n) { h->err = "ldat < n"; return -11; }
    GN_HIP(hipSetDevice(h->device));
    rc = make_plan(h, batch, m, n, t);
    if (rc) return rc;
    const Plan& P = h->plan;
    h->eps_rank = eps_rank;
    h->factors_valid = false;
    h->last_J = dJ; h->last_ldj = ldj; h->last_strideJ = strideJ;
    h->last_rx = drx; h->last_stride_rx = m;
    h->last_cx = dcx; h->last_stride_cx = t;
    h->last_At = dAt; h->last_ldat = ldat; h->last_strideAt = strideAt;
    h->sc_eJ = 0;
    if (!reuse) h->sc_eA = 0;       // (a resident constraint stage keeps the scale enlsip_gn_factor_constraints gave it)
    h->rescue_prob.clear();
    hipStream_t s = h->stream;
    if (h->profiling) {
        if (!h->ev_ready) {
            for (int i = 0; i < 8; ++i) GN_HIP(hipEventCreate(&h->ev[i]));
            h->ev_ready = true;
        }
        h->upd_used = 0;
        h->upd_bytes = 0.0;
        h->upd_launch_bytes.clear();
        h->oth_used = 0;
        h->upd_all_bytes = 0.0;
    }
    auto mark = [&](int i) { if (h->profiling) (void)hipEventRecord(h->ev[i], s); };

    mark(0);
    h->constraints_only = false;
    // 1. constraint stage
    GN_TRACE(h, "solve m=%lld n=%lld t=%lld batch=%lld: constraint stage%s", m, n, t, batch, reuse ? " (resident)" : "");
    if (!reuse) {       // enlsip_gn_solve_factored: F_A, F_L11, b, p1, T and the state record are those of enlsip_gn_factor_constraints
        rc = run_constraint_stage(h, batch, m, n, t, dAt, ldat, strideAt, dcx, eps_rank, dimA_ov);
        if (rc) return rc;
    }
    mark(1);
    GN_TRACE(h, "constraint stage done");

    // steps 2-4 for the Jacobian side given (the caller's J, rx — or their scaled copies, abs_shift = their power of two)
    auto attempts = [&](const double* dJ, long long ldj, long long strideJ, const double* drx, int abs_shift) -> int {
    int n2_launch = (int)(n - P.kA);  // speculate rankA = min(n, t); verified after the solve
    for (int attempt = 0; attempt < 2; ++attempt) {
        // 2. JQ1 = J*Q1, d_temp
        JQ1Args qa{};
        qa.m = (int)m; qa.n = (int)n; qa.kA = P.kA; qa.ldw = P.ldw;
        qa.J = dJ; qa.ldj = ldj; qa.strideJ = strideJ; qa.rx = drx; qa.stride_rx = m;
        qa.FA = h->FA; qa.sFA = P.sFA; qa.TA = h->TA; qa.sTA = P.sTA; qa.p1 = h->p1; qa.sP1 = P.sP1;
        qa.W = h->W; qa.sW = P.sW; qa.state = h->state;
        qa.prob0 = 0;
        // V T' of the fast path lives in the (still unused) working matrix of the pivoted QR
        qa.VT = (P.sM >= (long long)n * KBLK) ? h->qdM : nullptr; qa.sVT = P.sM;
        // one 256-row tile, one narrow panel (C5): J*Q1 and the panel factorisation in ONE launch, the tile handed over in LDS
        const bool fused = h->fuse_small && !upper_in && !(h->flags & ENLSIP_GN_UPDATE_REFLECTORS) &&
                           small_fused_applies(m, n, P.kA, n2_launch);
        if (fused) {
            CaqrArgs ca = caqr_args(h, 0, P.panels[0].levels[0]);
            ca.npass = 1;
            launch_jq1_factor_small(qa, ca, (int)batch, s);
            GN_HIP(hipGetLastError());
        } else if (h->flags & ENLSIP_GN_UPDATE_REFLECTORS) launch_jq1(qa, (int)batch, s);   // plain-FMA A/B partner
        else if (launch_jq1_rows(qa, (int)batch, s)) {}                             // small n, few reflectors
        else if (!launch_jq1_v2(qa, (int)batch, s)) launch_jq1_mfma(qa, (int)batch, s);      // regular shapes / general shapes
        mark(2);
        GN_TRACE(h, "attempt %d n2_launch=%d: J*Q1 done%s", attempt, n2_launch, fused ? " (fused with the panel factorisation)" : "");
        // 3. CAQR of [J2 | d]
        if (!upper_in && !fused) {
            rc = run_caqr(h, n2_launch);
            if (rc) return rc;
        }
        if (upper_in) GN_ROUTE(ENLSIP_GN_ROUTE_SWEEP_UPPER_INPUT);
        if (attempt > 0) GN_ROUTE(ENLSIP_GN_ROUTE_SECOND_ATTEMPT);
        mark(3);
        GN_TRACE(h, "CAQR done");
        // 4. pivoted QR of R0 + solves + outputs
        FinalArgs fa{};
        fa.m = (int)m; fa.n = (int)n; fa.t = (int)t; fa.kA = P.kA; fa.ldw = P.ldw; fa.ldr = P.ldr;
        fa.eps_rank = eps_rank; fa.abs_shift = abs_shift; fa.dimJ2_override = (int)dimJ2_ov; fa.refactor = 1;
        fa.W = h->W; fa.sW = P.sW; fa.Rt = h->Rt; fa.sRt = P.sRt; fa.tauJ = h->tauJ; fa.sTauJ = P.sTauJ;
        fa.jpvtJ = h->jpvtJ; fa.sJJ = P.sJJ; fa.FA = h->FA; fa.sFA = P.sFA; fa.tauA = h->tauA; fa.sTauA = P.sTauA;
        fa.p1 = h->p1; fa.sP1 = P.sP1; fa.bvec = h->bvec; fa.sB = P.sB; fa.zsave = h->zsave; fa.sZ = P.sZ;
        fa.p_out = dp; fa.sPo = n; fa.b_out = db; fa.sBo = t; fa.d_out = dd; fa.sDo = m;
        fa.jA_out = djA; fa.sJAo = t; fa.jpvtA = h->jpvtA; fa.sJA = P.sJA;
        fa.jL_out = djL; fa.sJLo = P.kA; fa.jpvtL = h->jpvtL; fa.sJL = P.sJL;
        fa.jJ_out = djJ; fa.sJJo = n;
        fa.state = h->state;
        fa.n2cap = n2_launch;
        {
            const int kp_launch = (int)std::min<long long>(m, n2_launch);
            if ((size_t)kp_launch * (n2_launch + 1) > (size_t)CMAT_DOUBLES) {
                // more than 512 rows do not fit the register form of the blocked factorisation: one launch per pivot step
                // (6.5 us per step; an LDS-slab blocked form was measured at 14 us per step and is gone)
                rc = (kp_launch > 512 && !h->qrcp_hybrid) ? run_qrcp_dist(h, n2_launch) : run_qrcp_block(h, n2_launch);
                if (rc) return rc;
                fa.refactor = 2;
            }
        }
        GN_TRACE(h, "pivoted QR of R0 done (refactor %d)", fa.refactor);
        if (!launch_pivot_small((int)std::min<long long>(m, n2_launch), n2_launch, (int)batch, s, fa))
            launch_pivot((int)std::min<long long>(m, n), (int)batch, s, fa);
        mark(4);
        GN_TRACE(h, "final kernel done");
        // nominate problems whose largest column norm overflowed or sits at the bottom of the exponent range (gn_rescale.hpp)
        if (h->rescale_enabled)
            hipLaunchKernelGGL(k_extreme_flags, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, s, h->state, (const double*)h->Rt, P.sRt,
                               (const double*)h->FA, P.sFA, P.kA, n2_launch, (int)batch);
        GN_HIP(hipGetLastError());
        GN_HIP(hipMemcpyAsync(h->h_state, h->state, (size_t)batch * sizeof(ProbState), hipMemcpyDeviceToHost, s));
        GN_HIP(hipStreamSynchronize(s));
        int n2max = 0;
        for (long long k = 0; k < batch; ++k) n2max = std::max(n2max, h->h_state[k].n2);
        if (n2max <= n2_launch) break;
        n2_launch = n2max;  // some A was rank deficient: J2 is wider than speculated, redo from J*Q1
    }
    return 0;
    };
    rc = attempts(dJ, ldj, strideJ, drx, 0);
    if (rc) return rc;
    // ---- magnitudes beyond the range of plain sums of squares: LAPACK's result through a power-of-two scaling (gn_rescale.hpp) ----
    {
        const int fl = GN_FLAG_NONFINITE | GN_FLAG_TINY;
        bool flagged = false;
        for (long long k = 0; k < batch; ++k) flagged = flagged || (h->h_state[k].status & fl);
        if (flagged && batch == 1) {
            int sJ = 0, sA = 0;
            rc = extreme_shifts(h, m, n, t, dJ, ldj, drx, reuse ? nullptr : dAt, ldat, reuse ? nullptr : dcx, &sJ, &sA);
            if (rc) return rc;
            if (sJ || sA) {
                GN_TRACE(h, "rescale: J, rx by 2^%d, A', cx by 2^%d", sJ, sA);
                const size_t nJ = (size_t)m * n, nA = (size_t)n * t;
                rc = grow(h, h->rs_buf, (nJ + (size_t)m + nA + (size_t)t + 8) * 8);
                if (rc) return rc;
                h->rs_J = (double*)h->rs_buf.p; h->rs_rx = h->rs_J + nJ; h->rs_At = h->rs_rx + m; h->rs_cx = h->rs_At + nA;
                auto copy_scaled = [&](double* dst, long long ldd, const double* src, long long lds, long long rows, long long cols, int sh) {
                    hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)std::min<long long>((rows + 255) / 256, 1024), (unsigned)cols), dim3(256), 0, s,
                                       dst, ldd, src, lds, (int)rows, (int)cols, sh);
                };
                if (sJ) {
                    copy_scaled(h->rs_J, m, dJ, ldj, m, n, sJ);
                    copy_scaled(h->rs_rx, m, drx, m, m, 1, sJ);
                }
                if (sA) {
                    copy_scaled(h->rs_At, n, dAt, ldat, n, t, sA);
                    copy_scaled(h->rs_cx, t, dcx, t, t, 1, sA);
                    h->sc_eA = sA;
                    rc = run_constraint_stage(h, 1, m, n, t, dAt, ldat, strideAt, dcx, eps_rank, dimA_ov);     // on the scaled copies; scaled back
                    if (rc) return rc;
                }
                h->sc_eJ = sJ;
                rc = attempts(sJ ? h->rs_J : dJ, sJ ? m : ldj, sJ ? (long long)nJ : strideJ, sJ ? h->rs_rx : drx, sJ);
                if (rc) return rc;
                if (sJ) {
                    rc = unscale_jacobian_side(h, dd);
                    if (rc) return rc;
                }
                GN_ROUTE(ENLSIP_GN_ROUTE_RESCALED);
            }
        } else if (flagged) {
            // a batch: every nominated problem whose inputs are beyond the band goes to a one-problem handle of its own (which
            // rescales in place as above); its outputs land in the caller's slots, the accessors are routed to it
            for (long long k = 0; k < batch; ++k) {
                if (!(h->h_state[k].status & fl)) continue;
                int sJ = 0, sA = 0;
                rc = extreme_shifts(h, m, n, t, dJ + k * strideJ, ldj, drx + k * m, dAt ? dAt + k * strideAt : nullptr, ldat,
                                    dcx ? dcx + k * t : nullptr, &sJ, &sA);
                if (rc) return rc;
                if (!sJ && !sA) continue;
                const size_t j = h->rescue_prob.size();
                if (j >= 64) { h->err = "more than 64 problems of the batch need rescaling (magnitudes beyond 2^+-400): solve them separately"; return -18; }
                if (j >= h->rescue.size()) {
                    enlsip_gn_opts o{};
                    o.device = h->device; o.flags = h->flags; o.panel_width = 0; o.tile_rows = h->tile_rows; o.stream = nullptr;
                    enlsip_gn_handle r = nullptr;
                    rc = enlsip_gn_create(&r, &o);
                    if (rc) { h->err = "could not create a handle for a rescaled problem"; return rc; }
                    r->is_rescue = true;
                    r->pipeline = false;
                    h->rescue.push_back(r);
                }
                enlsip_gn_handle r = h->rescue[j];
                const unsigned long long route_here = gn_route_acc;
                rc = solve_dev(r, 1, m, n, t, dJ + k * strideJ, ldj, strideJ, drx + k * m, dAt ? dAt + k * strideAt : nullptr, ldat, strideAt,
                               dcx ? dcx + k * t : nullptr, eps_rank, dimA_ov, dimJ2_ov, dp ? dp + k * n : nullptr, db ? db + k * t : nullptr,
                               dd ? dd + k * m : nullptr, nullptr, djA ? djA + k * t : nullptr, djL ? djL + k * P.kA : nullptr,
                               djJ ? djJ + k * n : nullptr, nullptr);
                gn_route_acc = route_here | r->route;
                if (rc) { h->err = r->err; return rc; }
                GN_HIP(hipSetDevice(h->device));
                h->h_state[k] = r->h_state[0];
                h->rescue_prob.push_back(k);
            }
        }
        if (flagged) {       // the nomination bits are host-internal
            for (long long k = 0; k < batch; ++k) h->h_state[k].status &= ~fl;
            hipLaunchKernelGGL(k_clear_status_bits, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, s, h->state, fl, (int)batch);
            GN_HIP(hipGetLastError());
        }
    }
    if (hinfo)
        for (long long k = 0; k < batch; ++k) {
            const ProbState& st = h->h_state[k];
            hinfo[k] = {st.rankA, st.rankJ2, st.code, st.dimA, st.dimJ2, st.status};
        }
    if (dinfo) {
        // info records are produced on the host from the state mirror and copied to the device buffer
        std::vector<enlsip_gn_info> tmp((size_t)batch);
        for (long long k = 0; k < batch; ++k) {
            const ProbState& st = h->h_state[k];
            tmp[k] = {st.rankA, st.rankJ2, st.code, st.dimA, st.dimJ2, st.status};
        }
        GN_HIP(hipMemcpyAsync(dinfo, tmp.data(), tmp.size() * sizeof(enlsip_gn_info), hipMemcpyHostToDevice, s));
        GN_HIP(hipStreamSynchronize(s));
    }
    if (h->profiling) {
        float ms;
        const int map[5][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {0, 4}};
        GN_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); h->stage_ms[ENLSIP_GN_STAGE_CONSTRAINT] = ms;
        GN_HIP(hipEventElapsedTime(&ms, h->ev[1], h->ev[2])); h->stage_ms[ENLSIP_GN_STAGE_JQ1] = ms;
        GN_HIP(hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
        float upd = 0.f;
        h->upd_launch_ms.clear();
        for (size_t i = 0; i + 1 < h->upd_used; i += 2) {
            float u;
            GN_HIP(hipEventElapsedTime(&u, h->upd_ev[i], h->upd_ev[i + 1]));
            upd += u;
            h->upd_launch_ms.push_back(u);
        }
        h->upd_launches = (long long)(h->upd_used / 2);
        h->upd_avg_ms = h->upd_launches ? upd / (float)h->upd_launches : 0.f;
        // every trailing-update launch of the sweep is "update" (far level-0 passes AND tree levels / second-panel columns);
        // "panel" = the factorisations (and whatever else the sweep launches)
        float oth = 0.f;
        for (size_t i = 0; i + 1 < h->oth_used; i += 2) {
            float u;
            GN_HIP(hipEventElapsedTime(&u, h->oth_ev[i], h->oth_ev[i + 1]));
            oth += u;
        }
        h->oth_ms = oth;
        h->oth_launches = (long long)(h->oth_used / 2);
        h->stage_ms[ENLSIP_GN_STAGE_UPDATE] = upd + oth;
        h->stage_ms[ENLSIP_GN_STAGE_PANEL] = ms - upd - oth;
        GN_HIP(hipEventElapsedTime(&ms, h->ev[3], h->ev[4])); h->stage_ms[ENLSIP_GN_STAGE_PIVOT] = ms;
        GN_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[4])); h->stage_ms[ENLSIP_GN_STAGE_TOTAL] = ms;
        (void)map;
    }
    h->factors_valid = true;
    h->route = gn_route_acc;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static void tsqr_drop_comm(enlsip_gn_handle h);      // gn_tsqr.inc

extern "C" {

int enlsip_gn_version(void) { return 200; }

// why the last enlsip_gn_create of this thread failed (no handle exists to carry the message): enlsip_gn_last_error(NULL)
static thread_local std::string g_create_err;

int enlsip_gn_create(enlsip_gn_handle* out, const enlsip_gn_opts* opts) {
    if (!out) return -1;
    *out = nullptr;
    g_create_err.clear();
    enlsip_gn_context* h = new (std::nothrow) enlsip_gn_context();
    if (!h) { g_create_err = "out of host memory"; return 998; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        delete h;
        g_create_err = std::string("no usable HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                       "): libenlsip_gn has no CPU code path";
        return e != hipSuccess ? (int)e : 100;  // hipErrorNoDevice
    }
    int dev = (opts && opts->device >= 0) ? opts->device : -1;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) {
        delete h;
        g_create_err = "opts->device is not a device ordinal of this process";
        return -2;
    }
    h->device = dev;
    h->flags = opts ? opts->flags : 0;
    h->tile_rows = (opts && opts->tile_rows == 256) ? 256 : 512;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            h->cu_count = prop.multiProcessorCount;
        h->trace = getenv("ENLSIP_GN_TRACE") != nullptr;
        const char* pp = getenv("ENLSIP_GN_PAIR");            // 0: plain sweep, one panel per pass over the trailing matrix (A/B)
        if (pp && pp[0] == '0') h->pair_enabled = false;
        if (pp && pp[0] == '2') h->pair_debug = true;         // 2: pair geometry, but the far columns in two plain passes (A/B)
        if (pp && (pp[0] == '1' || pp[0] == '2')) h->pair_forced = true;   // 1 / 2: pairs for every shape with three panels or more
        const char* lk = getenv("ENLSIP_GN_LOOKAHEAD");      // 0: the pair sweep on one stream (A/B)
        if (lk && lk[0] == '0') h->lookahead = false;
        if (lk && lk[0] == '1') h->lookahead_forced = true;   // 1: for every paired sweep (tests)
        const char* xm = getenv("ENLSIP_GN_XMAP");           // 0: native grid order for the far update of few problems with many tiles (A/B)
        if (xm && xm[0] == '0') h->xcd_map = false;
        const char* fs = getenv("ENLSIP_GN_FUSE_SMALL");     // 0: J*Q1 and the one-tile panel factorisation as two launches (A/B)
        if (fs && fs[0] == '0') h->fuse_small = false;
        const char* fh = getenv("ENLSIP_GN_SB_FORM_HINTS");   // 0: every block of the blocked pivoted QR in all of its forms (A/B)
        if (fh && fh[0] == '0') h->sb_form_hints = false;
        const char* hy = getenv("ENLSIP_GN_QRCP_HYBRID");     // 0: more than 512 rows = one launch per pivot step to the end (A/B)
        if (hy && hy[0] == '0') h->qrcp_hybrid = false;
#ifdef ENLSIP_GN_LAB       // laboratory build only (gn_device_utils.hpp): A/B and fault-location switches that no test of the suite uses
        const char* f4 = getenv("ENLSIP_GN_FACTOR_NW4");
        if (f4) h->factor_nw4 = atoi(f4);
        const char* dm = getenv("ENLSIP_GN_DEBUG_MAXPAN");
        if (dm) h->debug_maxpan = atoi(dm);
        const char* ds = getenv("ENLSIP_GN_DEBUG_STAGE");
        if (ds) h->debug_stage = atoi(ds);
#endif
        const char* rs = getenv("ENLSIP_GN_RESCALE");         // 0: no detection / rescaling of magnitudes beyond plain sums of squares (A/B, tests)
        if (rs && rs[0] == '0') h->rescale_enabled = false;
        const char* pl = getenv("ENLSIP_GN_PIPELINE");       // 0: never split a batch over two streams
        if (pl && pl[0] == '0') h->pipeline = false;
        if (pl && pl[0] == '1') h->pipeline_forced = true;    // 1: split even the small uniform shapes (A/B)
    }
    if (opts && opts->panel_width != 0 && opts->panel_width != PB) {
        delete h;
        return -2;
    }
    e = hipSetDevice(dev);
    if (e != hipSuccess) {
        delete h;
        return (int)e;
    }
    if (opts && opts->stream) {
        h->stream = (hipStream_t)opts->stream;
        h->own_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete h;
            return (int)e;
        }
        h->own_stream = true;
    }
    // single-workgroup kernels use > 64 KB of dynamic LDS
    *out = h;
    return 0;
}

int enlsip_gn_destroy(enlsip_gn_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);      // look-ahead sweep: an error return may have left work there
    if (h->ws.p) (void)hipFree(h->ws.p);
    if (h->in_stage.p) (void)hipFree(h->in_stage.p);
    if (h->out_stage.p) (void)hipFree(h->out_stage.p);
    if (h->lag.p) (void)hipFree(h->lag.p);
    if (h->newton.p) (void)hipFree(h->newton.p);
    if (h->cws.p) (void)hipFree(h->cws.p);
    if (h->scratch.p) (void)hipFree(h->scratch.p);
    if (h->xbuf.p) (void)hipFree(h->xbuf.p);
    tsqr_drop_comm(h);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->h_sbinfo) (void)hipHostFree(h->h_sbinfo);
    if (h->h_sb_stat) (void)hipHostFree(h->h_sb_stat);
    if (h->sb_stat.p) (void)hipFree(h->sb_stat.p);
    if (h->ev_ready)
        for (int i = 0; i < 8; ++i) (void)hipEventDestroy(h->ev[i]);
    for (hipEvent_t e : h->upd_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->oth_ev) (void)hipEventDestroy(e);
    if (h->rs_buf.p) (void)hipFree(h->rs_buf.p);
    for (enlsip_gn_handle r : h->rescue) (void)enlsip_gn_destroy(r);
    if (h->sub) (void)enlsip_gn_destroy(h->sub);
    if (h->child) (void)enlsip_gn_destroy(h->child);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->la_events) (void)hipEventDestroy(e);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

const char* enlsip_gn_last_error(enlsip_gn_handle h) {
    if (!h) return g_create_err.empty() ? "null handle" : g_create_err.c_str();
    if (h->err.empty() && h->child && !h->child->err.empty()) return h->child->err.c_str();
    return h->err.c_str();
}

int enlsip_gn_synchronize(enlsip_gn_handle h) {
    if (!h) return -1;
    GN_HIP(hipStreamSynchronize(h->stream));
    if (h->child) GN_HIP(hipStreamSynchronize(h->child->stream));
    return 0;
}

int enlsip_gn_set_profiling(enlsip_gn_handle h, int enable) {
    if (!h) return -1;
    h->profiling = enable != 0;
    h->profile_all_updates = enable >= 2;
    return 0;
}

int enlsip_gn_get_stage_ms(enlsip_gn_handle h, float* ms) {
    if (!h) return -1;
    if (!ms) return -2;
    for (int i = 0; i < ENLSIP_GN_STAGE_COUNT; ++i) ms[i] = h->stage_ms[i];
    return 0;
}

int enlsip_gn_get_update_table(enlsip_gn_handle h, int64_t cap, double* algorithmic_bytes, float* ms, int64_t* count) {
    if (!h) return -1;
    const size_t nl = std::min(h->upd_launch_ms.size(), h->upd_launch_bytes.size());
    if (count) *count = (int64_t)nl;
    for (size_t i = 0; i < nl && (int64_t)i < cap; ++i) {
        if (algorithmic_bytes) algorithmic_bytes[i] = h->upd_launch_bytes[i];
        if (ms) ms[i] = h->upd_launch_ms[i];
    }
    return 0;
}

// In-place read-modify-write stream over `bytes` of scratch memory with the trailing update's access shape (256-thread
// workgroups, 16-byte non-temporal loads and stores, 256 contiguous bytes per 16 lanes): the ceiling an in-place update of
// streamed data can reach on THIS device, measured with HIP events on the handle's stream.
__global__ __launch_bounds__(256) void k_stream_inplace(double* buf, long long n2pairs) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2* p = (d2*)buf;
    // one workgroup = 32 KB chunks of 256 x 16 B, eight of them (the update's 128 KB block of C per workgroup would be sixteen)
    const long long chunk = 256;
    for (long long c = (long long)blockIdx.x * 8; c < (long long)blockIdx.x * 8 + 8; ++c) {
        const long long i = c * chunk + threadIdx.x;
        if (i < n2pairs) {
            d2 x = __builtin_nontemporal_load(p + i);
            x[0] += 1.0; x[1] -= 1.0;
            __builtin_nontemporal_store(x, p + i);
        }
    }
}
int enlsip_gn_measure_stream(enlsip_gn_handle h, int64_t bytes, int reps, double* gbytes_per_s) {
    if (!h) return -1;
    GN_TRY
    if (bytes < (1 << 20) || reps < 1 || !gbytes_per_s) { h->err = "measure_stream: bytes >= 1 MiB, reps >= 1, non-NULL result"; return -2; }
    GN_HIP(hipSetDevice(h->device));
    int rc = grow(h, h->scratch, (size_t)bytes);
    if (rc) return rc;
    const long long pairs = bytes / 16;
    const unsigned grid = (unsigned)((pairs + 256 * 8 - 1) / (256 * 8));
    hipEvent_t e0, e1;
    GN_HIP(hipEventCreate(&e0));
    GN_HIP(hipEventCreate(&e1));
    GN_HIP(hipMemsetAsync(h->scratch.p, 0, (size_t)bytes, h->stream));
    hipLaunchKernelGGL(k_stream_inplace, dim3(grid), dim3(256), 0, h->stream, (double*)h->scratch.p, pairs);
    GN_HIP(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_stream_inplace, dim3(grid), dim3(256), 0, h->stream, (double*)h->scratch.p, pairs);
    GN_HIP(hipEventRecord(e1, h->stream));
    GN_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GN_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gbytes_per_s = 2.0 * (double)(pairs * 16) * reps / ((double)ms * 1e-3) / 1e9;
    return 0;
    GN_CATCH(h)
}

// debugging aid: the working matrix W (ldw x (n + 1)) of problem `prob` as it stands
int enlsip_gn_debug_copy_W(enlsip_gn_handle h, int64_t prob, double* out, int64_t* ldw_out, int64_t cap_doubles) {
    if (!h || !h->have_plan) return -1;
    const Plan& P = h->plan;
    if (ldw_out) *ldw_out = P.ldw;
    const size_t need = (size_t)P.ldw * (P.n + 1);
    if (!out || (size_t)cap_doubles < need) return -3;
    GN_HIP(hipSetDevice(h->device));
    GN_HIP(hipStreamSynchronize(h->stream));
    GN_HIP(hipMemcpy(out, h->W + prob * P.sW, need * 8, hipMemcpyDeviceToHost));
    return 0;
}

#ifdef ENLSIP_SB_STEP_STAMPS
// diagnostic build only: phase sums of the blocked pivoted QR's step (100 MHz ticks; [8] = steps; [16..23] block-level phases), reset on read; out: 24 words
extern "C" int enlsip_gn_debug_sb_phase(long long* out) {
    long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gn::g_sb_phase), sizeof(z)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(gn::g_sb_phase), z, sizeof(z)) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(gn::g_sb_blk), 8 * sizeof(long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(gn::g_sb_blk), z, 8 * sizeof(long long)) != hipSuccess) return 1;
    return 0;
}
#endif

int enlsip_gn_get_update_stats(enlsip_gn_handle h, float* avg_ms, int64_t* launches, double* bytes) {
    if (!h) return -1;
    if (avg_ms) *avg_ms = h->upd_avg_ms;
    if (launches) *launches = h->upd_launches;
    if (bytes) *bytes = h->upd_bytes;
    return 0;
}

int enlsip_gn_get_route(enlsip_gn_handle h, uint64_t* mask) {
    if (!h) return -1;
    if (!mask) return -2;
    *mask = (uint64_t)h->route;
    return 0;
}

const char* enlsip_gn_route_name(int bit) {
    static_assert(ENLSIP_GN_ROUTE_COUNT == 51, "one name per route bit");
    static const char* const names[ENLSIP_GN_ROUTE_COUNT] = {
        "constraint_wave32", "constraint_wave64", "constraint_lds_r1_256", "constraint_lds_r1_512", "constraint_lds_r2",
        "constraint_lds_r4", "constraint_lds_r8", "constraint_lds_r16", "constraint_global", "constraint_reg4", "constraint_reg8", "constraint_dist",
        "jq1_fused_small", "jq1_rows32", "jq1_rows2", "jq1_rows64", "jq1_v2_n128", "jq1_v2_n256", "jq1_v2_n384", "jq1_v2_n512",
        "jq1_mfma", "jq1_plain", "sweep_plain", "sweep_pairs", "sweep_lookahead", "sweep_passenger", "sweep_tree",
        "sweep_tile256", "sweep_tile512", "sweep_reflectors", "sweep_upper_input", "pivot_wave32", "pivot_wave64", "pivot_wave2",
        "pivot_lds_r1_256", "pivot_lds_r1_512", "pivot_lds_r2", "pivot_lds_r4", "pivot_lds_r8", "pivot_lds_r16", "pivot_blocks",
        "pivot_blocks_448", "pivot_blocks_512", "pivot_blocks_256", "pivot_blocks_128", "pivot_hybrid", "pivot_steps",
        "pipeline_split", "chunked", "second_attempt", "rescaled"};
    return (bit >= 0 && bit < ENLSIP_GN_ROUTE_COUNT) ? names[bit] : nullptr;
}

int enlsip_gn_get_launch_plan(enlsip_gn_handle h, int64_t* pipeline_split, int* panel_pairs, int64_t* tile_rows) {
    if (!h) return -1;
    if (!h->have_plan) { h->err = "no solve on this handle yet"; return -1; }
    if (pipeline_split) *pipeline_split = h->split;
    if (panel_pairs) *panel_pairs = h->plan.pair ? 1 : 0;
    if (tile_rows) *tile_rows = 64LL * h->plan.RPL;
    return 0;
}

int enlsip_gn_get_update_totals(enlsip_gn_handle h, float* far_ms, float* other_ms, int64_t* other_launches, double* all_panels_bytes) {
    if (!h) return -1;
    if (far_ms) *far_ms = h->upd_avg_ms * (float)h->upd_launches;
    if (other_ms) *other_ms = h->oth_ms;
    if (other_launches) *other_launches = h->oth_launches;
    if (all_panels_bytes) *all_panels_bytes = h->upd_all_bytes;
    return 0;
}

// One launch set over at most GN_MAX_LAUNCH_BATCH problems: either two pipelined halves on two streams or one solve_dev.
static int solve_launchable(enlsip_gn_handle h, int64_t batch, int64_t m, int64_t n, int64_t t,
                            const double* dJ, int64_t ldj, int64_t strideJ, const double* drx,
                            const double* dAt, int64_t ldat, int64_t strideAt, const double* dcx,
                            double eps_rank, int64_t dimA_ov, int64_t dimJ2_ov, double* dp, double* db, double* dd,
                            enlsip_gn_info* dinfo, long long* djA, long long* djL, long long* djJ, enlsip_gn_info* hinfo) {
    h->split = 0;
    // one-tile problems of the wave-per-problem pipeline (n <= 64, m <= 512: C3, C5) are a handful of short, uniform launches with
    // nothing latency-bound to hide behind them: the split costs C3 4 % (1.276 -> 1.325 M solves/s without it), C5 nothing
    const bool small_uniform = (n <= 64 && m <= 512) && !h->pipeline_forced;
    const bool plain = (dimA_ov < 0 && dimJ2_ov < 0 && !h->reuse_once);
    if (plain && h->pipeline && !h->profiling && batch >= h->pipeline_min && !small_uniform) {
        // two halves on two streams (see gn_context.hpp); the child's stream is ordered after everything the caller
        // has enqueued on this handle's stream, and both halves are complete when this call returns
        GN_HIP(hipSetDevice(h->device));
        if (!h->child) {
            enlsip_gn_opts o{};
            o.device = h->device; o.flags = h->flags; o.panel_width = 0; o.tile_rows = h->tile_rows; o.stream = nullptr;
            int rc = enlsip_gn_create(&h->child, &o);
            if (rc) { h->err = "could not create the second pipeline handle"; return rc; }
            h->child->pipeline = false;
        }
        if (!h->ev_fork) GN_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        GN_HIP(hipEventRecord(h->ev_fork, h->stream));
        GN_HIP(hipStreamWaitEvent(h->child->stream, h->ev_fork, 0));
        const int64_t b0 = (batch + 1) / 2, b1 = batch - b0;
        const int64_t kA = std::min(n, t);
        enlsip_gn_handle c = h->child;
        int rc1 = 0;
        std::thread worker([&] {
            (void)hipSetDevice(c->device);
            try {
                rc1 = solve_dev(c, b1, m, n, t, dJ + b0 * strideJ, ldj, strideJ, drx + b0 * m, dAt ? dAt + b0 * strideAt : nullptr,
                                ldat, strideAt, dcx ? dcx + b0 * t : nullptr, eps_rank, -1, -1, dp ? dp + b0 * n : nullptr,
                                db ? db + b0 * t : nullptr, dd ? dd + b0 * m : nullptr, dinfo ? dinfo + b0 : nullptr,
                                djA ? djA + b0 * t : nullptr, djL ? djL + b0 * kA : nullptr, djJ ? djJ + b0 * n : nullptr,
                                hinfo ? hinfo + b0 : nullptr);
            } catch (...) {
                c->err = "exception in the second pipeline half (out of host memory?)";
                rc1 = 997;
            }
        });
        int rc0;
        try {
            rc0 = solve_dev(h, b0, m, n, t, dJ, ldj, strideJ, drx, dAt, ldat, strideAt, dcx, eps_rank, -1, -1, dp, db,
                            dd, dinfo, djA, djL, djJ, hinfo);
        } catch (...) {
            worker.join();
            throw;
        }
        worker.join();
        if (rc0) return rc0;
        if (rc1) { h->err = c->err; return rc1; }
        h->split = b0;
        h->route |= c->route | (1ull << ENLSIP_GN_ROUTE_PIPELINE_SPLIT);
        return 0;
    }
    return solve_dev(h, batch, m, n, t, dJ, ldj, strideJ, drx, dAt, ldat, strideAt, dcx, eps_rank, dimA_ov, dimJ2_ov, dp, db,
                     dd, dinfo, djA, djL, djJ, hinfo);
}

// Any batch: consecutive chunks of at most GN_MAX_LAUNCH_BATCH problems (the problem index is a grid y / z dimension).  The
// factors that stay resident are those of the LAST chunk; accessors address problems by their index in the whole batch and
// report an error for the earlier chunks (gn_accessors.inc: need_factors).
static int solve_chunked(enlsip_gn_handle h, int64_t batch, int64_t m, int64_t n, int64_t t,
                         const double* dJ, int64_t ldj, int64_t strideJ, const double* drx,
                         const double* dAt, int64_t ldat, int64_t strideAt, const double* dcx,
                         double eps_rank, int64_t dimA_ov, int64_t dimJ2_ov, double* dp, double* db, double* dd,
                         enlsip_gn_info* dinfo, long long* djA, long long* djL, long long* djJ, enlsip_gn_info* hinfo) {
    h->chunk0 = 0;
    const int64_t kA = std::min(n, t);
    const int64_t nchunks = (batch + GN_MAX_LAUNCH_BATCH - 1) / GN_MAX_LAUNCH_BATCH;
    const int64_t per = (batch + nchunks - 1) / nchunks;
    unsigned long long route_all = nchunks > 1 ? (1ull << ENLSIP_GN_ROUTE_CHUNKED) : 0ull;
    for (int64_t c0 = 0; c0 < batch; c0 += per) {
        const int64_t nb = std::min(per, batch - c0);
        int rc = solve_launchable(h, nb, m, n, t, dJ + c0 * strideJ, ldj, strideJ, drx + c0 * m,
                                  dAt ? dAt + c0 * strideAt : nullptr, ldat, strideAt, dcx ? dcx + c0 * t : nullptr, eps_rank,
                                  dimA_ov, dimJ2_ov, dp ? dp + c0 * n : nullptr, db ? db + c0 * t : nullptr,
                                  dd ? dd + c0 * m : nullptr, dinfo ? dinfo + c0 : nullptr, djA ? djA + c0 * t : nullptr,
                                  djL ? djL + c0 * kA : nullptr, djJ ? djJ + c0 * n : nullptr, hinfo ? hinfo + c0 : nullptr);
        if (rc) return rc;
        h->chunk0 = c0;
        route_all |= h->route;
        h->route = route_all;
    }
    return 0;
}

int enlsip_gn_solve_batched_dev(enlsip_gn_handle h, int64_t batch, int64_t m, int64_t n, int64_t t,
                                const double* dJ, int64_t ldj, int64_t strideJ, const double* drx,
                                const double* dAt, int64_t ldat, int64_t strideAt, const double* dcx,
                                double eps_rank, double* dp, double* db, double* dd, enlsip_gn_info* dinfo,
                                int64_t* djpvtA, int64_t* djpvtL, int64_t* djpvtJ2) {
    if (!h) return -1;
    GN_TRY
    int rc = check_limits(h, batch, m, n, t);
    if (rc) return rc;
    if (!dJ) { h->err = "dJ is NULL"; return -6; }
    if (!drx) { h->err = "drx is NULL"; return -9; }
    if (t > 0 && (!dAt || !dcx)) { h->err = "dAt / dcx is NULL with t > 0"; return -10; }
    return solve_chunked(h, batch, m, n, t, dJ, ldj, strideJ, drx, dAt, ldat, strideAt, dcx, eps_rank, -1, -1, dp, db, dd,
                         dinfo, (long long*)djpvtA, (long long*)djpvtL, (long long*)djpvtJ2, nullptr);
    GN_CATCH(h)
}

static int solve_host(enlsip_gn_handle h, int64_t batch, int64_t m, int64_t n, int64_t t, const double* J,
                      int64_t ldj, int64_t strideJ, const double* rx, const double* At, int64_t ldat,
                      int64_t strideAt, const double* cx, double eps_rank, int64_t dimA_ov, int64_t dimJ2_ov,
                      double* p, double* b, double* d, enlsip_gn_info* info, int64_t* jA, int64_t* jL, int64_t* jJ,
                      bool factored = false) {
    if (!h) return -1;
    GN_TRY
    int rc = check_limits(h, batch, m, n, t);
    if (rc) return rc;
    if (!J) { h->err = "J is NULL"; return -6; }
    if (ldj < m) { h->err = "ldj < m"; return -7; }
    if (!rx) { h->err = "rx is NULL"; return -9; }
    if (factored) {
        const Plan& P = h->plan;
        if (!(h->factors_valid && h->constraints_only && h->have_plan && P.batch == 1 && P.m == m && P.n == n && P.t == t)) {
            h->err = "enlsip_gn_solve_factored needs enlsip_gn_factor_constraints with the same m, n, t right before";
            return -1;
        }
    } else {
        if (t > 0 && (!At || !cx)) { h->err = "At / cx is NULL with t > 0"; return -10; }
        if (t > 0 && ldat < n) { h->err = "ldat < n"; return -11; }
    }
    // truncation dimensions index the triangular factors: dimA <= min(n, t) = rows of F_L11.R, dimJ2 <= min(m, n) >= kp
    // (the kernels clamp dimJ2 to kp = min(m, n - rankA), which is only known on the device)
    if (dimA_ov > std::min(n, t)) { h->err = "dimA_override > min(n, t)"; return -16; }
    if (dimJ2_ov > std::min(m, n)) { h->err = "dimJ2_override > min(m, n)"; return -17; }
    GN_HIP(hipSetDevice(h->device));
    const int kA = (int)std::min(n, t);
    // staging: inputs packed (ld = m / n), outputs packed
    const size_t inJ = (size_t)batch * m * n, inAt = (size_t)batch * n * t;
    const size_t in_bytes = (inJ + (size_t)batch * m + inAt + (size_t)batch * t) * 8 + 1024;
    rc = grow(h, h->in_stage, in_bytes);
    if (rc) return rc;
    const size_t out_dbl = (size_t)batch * (n + t + m);
    const size_t out_i64 = (size_t)batch * (t + kA + n);
    rc = grow(h, h->out_stage, (out_dbl + out_i64) * 8 + 1024);
    if (rc) return rc;
    double* dJ = (double*)h->in_stage.p;
    double* drx = dJ + inJ;
    double* dAt = drx + (size_t)batch * m;
    double* dcx = dAt + inAt;
    double* dp = (double*)h->out_stage.p;
    double* db = dp + (size_t)batch * n;
    double* dd = db + (size_t)batch * t;
    long long* djA = (long long*)(dd + (size_t)batch * m);
    long long* djL = djA + (size_t)batch * t;
    long long* djJ = djL + (size_t)batch * kA;
    hipStream_t s = h->stream;
    for (int64_t k = 0; k < batch; ++k) {
        GN_HIP(hipMemcpy2DAsync(dJ + (size_t)k * m * n, (size_t)m * 8, J + (size_t)k * strideJ, (size_t)ldj * 8,
                                (size_t)m * 8, (size_t)n, hipMemcpyHostToDevice, s));
        if (t > 0 && !factored)
            GN_HIP(hipMemcpy2DAsync(dAt + (size_t)k * n * t, (size_t)n * 8, At + (size_t)k * strideAt,
                                    (size_t)ldat * 8, (size_t)n * 8, (size_t)t, hipMemcpyHostToDevice, s));
    }
    GN_HIP(hipMemcpyAsync(drx, rx, (size_t)batch * m * 8, hipMemcpyHostToDevice, s));
    if (t > 0 && !factored) GN_HIP(hipMemcpyAsync(dcx, cx, (size_t)batch * t * 8, hipMemcpyHostToDevice, s));
    h->reuse_once = factored;      // A', cx (same staging slots) and the constraint factors are resident
    rc = solve_chunked(h, batch, m, n, t, dJ, m, m * n, drx, dAt, n, n * t, dcx, eps_rank, dimA_ov, dimJ2_ov, dp, db, dd,
                       nullptr, djA, djL, djJ, info);
    if (rc) return rc;
    if (p) GN_HIP(hipMemcpyAsync(p, dp, (size_t)batch * n * 8, hipMemcpyDeviceToHost, s));
    if (b && t > 0) GN_HIP(hipMemcpyAsync(b, db, (size_t)batch * t * 8, hipMemcpyDeviceToHost, s));
    if (d) GN_HIP(hipMemcpyAsync(d, dd, (size_t)batch * m * 8, hipMemcpyDeviceToHost, s));
    if (jA && t > 0) GN_HIP(hipMemcpyAsync(jA, djA, (size_t)batch * t * 8, hipMemcpyDeviceToHost, s));
    if (jL && kA > 0) GN_HIP(hipMemcpyAsync(jL, djL, (size_t)batch * kA * 8, hipMemcpyDeviceToHost, s));
    if (jJ) GN_HIP(hipMemcpyAsync(jJ, djJ, (size_t)batch * n * 8, hipMemcpyDeviceToHost, s));
    GN_HIP(hipStreamSynchronize(s));
    return 0;
    GN_CATCH(h)
}

int enlsip_gn_factor_constraints(enlsip_gn_handle h, int64_t m, int64_t n, int64_t t, const double* At, int64_t ldat,
                                 const double* cx, double eps_rank, enlsip_gn_info* info) {
    if (!h) return -1;
    int rc = check_limits(h, 1, m, n, t);
    if (rc) return rc;
    if (t > 0 && (!At || !cx)) return -5;
    if (t > 0 && ldat < n) return -6;
    GN_HIP(hipSetDevice(h->device));
    h->split = 0;
    h->chunk0 = 0;
    rc = make_plan(h, 1, m, n, t);
    if (rc) return rc;
    // same staging layout as solve_host, so that a following solve of the same shape reuses the buffers
    const size_t inJ = (size_t)m * n, inAt = (size_t)n * t;
    rc = grow(h, h->in_stage, (inJ + (size_t)m + inAt + (size_t)t) * 8 + 1024);
    if (rc) return rc;
    double* dAt = (double*)h->in_stage.p + inJ + (size_t)m;
    double* dcx = dAt + inAt;
    hipStream_t s = h->stream;
    if (t > 0) {
        GN_HIP(hipMemcpy2DAsync(dAt, (size_t)n * 8, At, (size_t)ldat * 8, (size_t)n * 8, (size_t)t, hipMemcpyHostToDevice, s));
        GN_HIP(hipMemcpyAsync(dcx, cx, (size_t)t * 8, hipMemcpyHostToDevice, s));
    }
    h->eps_rank = eps_rank;
    h->factors_valid = false;
    h->last_J = nullptr; h->last_rx = nullptr;
    h->last_cx = dcx; h->last_stride_cx = t;
    h->last_At = dAt; h->last_ldat = n; h->last_strideAt = (long long)n * t;
    h->sc_eJ = 0; h->sc_eA = 0;
    h->rescue_prob.clear();
    rc = run_constraint_stage(h, 1, m, n, t, dAt, n, (long long)n * t, dcx, eps_rank, -1);
    if (rc) return rc;
    if (h->rescale_enabled && t > 0)
        hipLaunchKernelGGL(k_extreme_flags, dim3(1), dim3(256), 0, s, h->state, (const double*)nullptr, 0LL, (const double*)h->FA, h->plan.sFA,
                           h->plan.kA, 0, 1);
    GN_HIP(hipMemcpyAsync(h->h_state, h->state, sizeof(ProbState), hipMemcpyDeviceToHost, s));
    GN_HIP(hipStreamSynchronize(s));
    if (h->h_state[0].status & (GN_FLAG_NONFINITE | GN_FLAG_TINY)) {
        // A', cx beyond the range of plain sums of squares: the stage again on copies scaled by a power of two (gn_rescale.hpp)
        int sJ = 0, sA = 0;
        rc = extreme_shifts(h, m, n, t, nullptr, 0, nullptr, dAt, n, dcx, &sJ, &sA);
        if (rc) return rc;
        if (sA) {
            const size_t nJ = (size_t)m * n, nA = (size_t)n * t;
            rc = grow(h, h->rs_buf, (nJ + (size_t)m + nA + (size_t)t + 8) * 8);
            if (rc) return rc;
            h->rs_J = (double*)h->rs_buf.p; h->rs_rx = h->rs_J + nJ; h->rs_At = h->rs_rx + m; h->rs_cx = h->rs_At + nA;
            hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024), (unsigned)t), dim3(256), 0, s, h->rs_At,
                               (long long)n, (const double*)dAt, (long long)n, (int)n, (int)t, sA);
            hipLaunchKernelGGL(k_scale_copy, dim3(1, 1), dim3(256), 0, s, h->rs_cx, (long long)t, (const double*)dcx, (long long)t, (int)t, 1, sA);
            h->sc_eA = sA;
            rc = run_constraint_stage(h, 1, m, n, t, dAt, n, (long long)n * t, dcx, eps_rank, -1);
            if (rc) return rc;
            GN_HIP(hipMemcpyAsync(h->h_state, h->state, sizeof(ProbState), hipMemcpyDeviceToHost, s));
            GN_HIP(hipStreamSynchronize(s));
            h->route |= (1ull << ENLSIP_GN_ROUTE_RESCALED);
        }
        h->h_state[0].status &= ~(GN_FLAG_NONFINITE | GN_FLAG_TINY);
        hipLaunchKernelGGL(k_clear_status_bits, dim3(1), dim3(256), 0, s, h->state, GN_FLAG_NONFINITE | GN_FLAG_TINY, 1);
        GN_HIP(hipGetLastError());
    }
    h->factors_valid = true;
    h->constraints_only = true;
    if (info) {
        const ProbState& st = h->h_state[0];
        *info = {st.rankA, 0, st.code, st.dimA, 0, st.status};
    }
    return 0;
}

int enlsip_gn_solve_factored(enlsip_gn_handle h, int64_t m, int64_t n, int64_t t, const double* J, int64_t ldj,
                             const double* rx, double eps_rank, int64_t dimJ2_override, double* p, double* b, double* d,
                             enlsip_gn_info* info, int64_t* jpvtA, int64_t* jpvtL, int64_t* jpvtJ2) {
    return solve_host(h, 1, m, n, t, J, ldj, (int64_t)ldj * n, rx, nullptr, n, (int64_t)n * t, nullptr, eps_rank, -1,
                      dimJ2_override, p, b, d, info, jpvtA, jpvtL, jpvtJ2, true);
}

int enlsip_gn_solve_batched(enlsip_gn_handle h, int64_t batch, int64_t m, int64_t n, int64_t t, const double* J,
                            int64_t ldj, int64_t strideJ, const double* rx, const double* At, int64_t ldat,
                            int64_t strideAt, const double* cx, double eps_rank, double* p, double* b, double* d,
                            enlsip_gn_info* info, int64_t* jpvtA, int64_t* jpvtL, int64_t* jpvtJ2) {
    return solve_host(h, batch, m, n, t, J, ldj, strideJ, rx, At, ldat, strideAt, cx, eps_rank, -1, -1, p, b, d,
                      info, jpvtA, jpvtL, jpvtJ2);
}

int enlsip_gn_solve(enlsip_gn_handle h, int64_t m, int64_t n, int64_t t, const double* J, int64_t ldj,
                    const double* rx, const double* At, int64_t ldat, const double* cx, double eps_rank,
                    int64_t dimA_override, int64_t dimJ2_override, double* p, double* b, double* d,
                    enlsip_gn_info* info, int64_t* jpvtA, int64_t* jpvtL, int64_t* jpvtJ2) {
    return solve_host(h, 1, m, n, t, J, ldj, (int64_t)ldj * n, rx, At, ldat, (int64_t)ldat * t, cx, eps_rank,
                      dimA_override, dimJ2_override, p, b, d, info, jpvtA, jpvtL, jpvtJ2);
}

}  // extern "C"

#include "gn_accessors.inc"
#include "gn_tsqr.inc"
#include "gn_lagrange.inc"
#include "gn_newton.inc"
