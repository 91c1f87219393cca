// Host-side state behind an enlsip_gn_handle: shape plan, device workspace carve, stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/enlsip_gn.h"
#include "gn_device_utils.hpp"

namespace gn {

struct LevelPlan {
    int level;
    int nblocks;      // 32-row blocks entering this level
    int groups;       // workgroups (= blocks of the next level)
    long long S;      // block stride (rows)
    long long tOff;   // first T block index
    int mode = 0;     // row geometry (CaqrArgs::mode): 0 level-0 tiles, 1 plain tree level, 2 first tree level of a pair's second panel
    long long base = 0;   // row of block 0
    int skip = 0;     // level 0: leading 32-row units of every tile that belong to the pair's first panel
};
struct PanelPlan {
    std::vector<LevelPlan> levels;
};

struct Plan {
    long long batch = 0, m = 0, n = 0, t = 0;
    int kA = 0;
    int RPL = 8;         // CAQR rows per lane (tile rows = 64 * RPL)
    int F = 16;          // blocks per group
    int ldw = 0, ldr = 0;
    int npan_max = 0;    // panels if n2 = n
    bool pair = false;   // panels (2K, 2K+1) share their tiles and one pass over the far trailing columns (gn_kernels_caqr.hpp, "Panel pairs")
    long long nTblocks = 0;
    std::vector<PanelPlan> panels;
    // per-problem strides (elements)
    long long sFA, sTauA, sJA, sFL, sTauL, sJL, sTA, sP1, sB, sW, sT, sRt, sTauJ, sJJ, sZ, sVec;
    // distributed pivoted QR (gn_kernels_qrcp_dist.hpp)
    long long sM, sVb, sDiag, sVn, sQI, sCand;
    int qdGmax = 0;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace gn

struct enlsip_gn_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int flags = 0;
    int tile_rows = 512;
    std::string err;
    gn::Plan plan;
    bool have_plan = false;
    bool factors_valid = false;
    gn::DevBuf cws;                      // workspace of the distributed constraint stage (many constraints)
    struct {                             // what the re-solve needs of it: unfactored L11 and F_L11.Q' b_buff of the last solve
        const double* L = nullptr; long long ldL = 0, sL = 0;
        const double* qb = nullptr; long long sQb = 0;
        bool valid = false;
    } cdist;
    bool upper_once = false;         // the next solve_dev gets an upper-triangular J (one gathered triangle of the TSQR combine stage): no CAQR needed
    bool reuse_once = false;         // the next solve_dev skips the constraint stage (enlsip_gn_solve_factored)
    bool trace = false;              // ENLSIP_GN_TRACE=1: stage names on stderr with a stream synchronisation after each (fault hunting)
    bool constraints_only = false;   // resident: F_A, F_L11 only (enlsip_gn_factor_constraints); everything about J is absent
    double eps_rank = 0.0;

    // workspace (one allocation, carved)
    gn::DevBuf ws;
    double *FA = nullptr, *tauA = nullptr, *FL = nullptr, *tauL = nullptr, *TA = nullptr, *p1 = nullptr,
           *bvec = nullptr, *W = nullptr, *Tbuf = nullptr, *Rt = nullptr, *tauJ = nullptr, *zsave = nullptr,
           *vec = nullptr, *qdM = nullptr, *qdVb = nullptr, *qdDiag = nullptr, *qdVn1 = nullptr, *qdVn2 = nullptr;
    int *qdChosen = nullptr, *qdPos = nullptr, *qdColat = nullptr;
    void* qdCand = nullptr;
    unsigned* small = nullptr;
    // row counts the blocks of the previous blocked QRCP started with (min / max over the problems, by block id), valid for
    // sb_rows_kp == kp_launch and the same batch: which forms of the select / factor kernel a block id needs
    std::vector<int> sb_rows_min, sb_rows_max;
    int sb_rows_kp = -1;
    long long sb_rows_batch = -1;
    gn::DevBuf sb_stat;          // device statistics (2 x SB_STAT_BLKS ints)
    int* h_sb_stat = nullptr;    // pinned mirror
    bool sb_form_hints = true;   // ENLSIP_GN_SB_FORM_HINTS=0: every block id in all three forms (A/B)
    bool qrcp_hybrid = true;     // pivoted QR of more than 512 rows: launch-per-step head, register blocks for the last 512 (ENLSIP_GN_QRCP_HYBRID=0: A/B)
    int sb_hint = 0;             // blocks the previous blocked QRCP needed (+1): size of the first launch chunk
    double* sbT = nullptr;       // per problem: T factor of the current QRCP block (32 x 32)
    void* sbInfo = nullptr;      // SbInfo per problem (device)
    int* sbInblk = nullptr;      // per column block id (device)
    int* sbAct = nullptr;        // per problem: columns the current block update touches (device)
    void* h_sbinfo = nullptr;    // pinned mirror of sbInfo
    int cu_count = 256;
    enlsip_gn_context* sub = nullptr;   // handle for the stacked problem of the TSQR combine stage
    // Two-stream pipelining of large batches (enlsip_gn_solve_batched_dev): the second half of the problems runs on
    // a child handle (own stream + workspace) driven by a host thread, so the latency-bound kernels of one half
    // overlap the bandwidth-bound kernels of the other.  Accessors route a problem index to the half that owns it.
    enlsip_gn_context* child = nullptr;
    bool pair_debug = false, pair_forced = false;
    bool lookahead_forced = false;
    bool lookahead = true;              // ENLSIP_GN_LOOKAHEAD=0: chain-bound pair sweeps on one stream
    hipStream_t stream2 = nullptr;      // second stream of the look-ahead sweep (the bulk of a pair's far update)
    std::vector<hipEvent_t> la_events;
    bool xcd_map = true;                // ENLSIP_GN_XMAP=0: native grid order (A/B)
    bool fuse_small = true;             // ENLSIP_GN_FUSE_SMALL=0: two launches for J*Q1 + panel factorisation of one-tile problems
    int factor_nw4 = 0;                 // ENLSIP_GN_FACTOR_NW4 (A/B): 1 = level-0 tiles factored by 4 waves x 8 columns, 2 = tree nodes too
    int debug_maxpan = -1, debug_stage = -1;
    bool pair_enabled = true;           // ENLSIP_GN_PAIR=0: one panel per pass over the trailing matrix
    bool pipeline = true;               // ENLSIP_GN_PIPELINE=0 disables
    bool pipeline_forced = false;   // ENLSIP_GN_PIPELINE=1
    long long pipeline_min = 128;       // smallest batch that is split
    long long split = 0;                // problems [split, batch) of the last solve live on `child` (0: not split)
    hipEvent_t ev_fork = nullptr;
    // Rescale path (gn_rescale.hpp): the resident problem of a one-problem solve whose inputs were beyond the range of plain sums
    // of squares was solved on copies scaled by 2^sc_eJ (J, rx) / 2^sc_eA (A', cx), kept in rs_buf for the entry points that run the
    // constraint stage again (re-solve, Newton); the resident factors are scaled back, i.e. they are those of the caller's data.
    // In a batch such a problem is handed to a one-problem rescue handle of its own, to which the accessors are routed.
    int sc_eJ = 0, sc_eA = 0;
    gn::DevBuf rs_buf;
    double *rs_J = nullptr, *rs_rx = nullptr, *rs_At = nullptr, *rs_cx = nullptr;
    bool rescale_enabled = true;        // ENLSIP_GN_RESCALE=0: detection and rescaling off (A/B; tests)
    bool is_rescue = false;
    std::vector<enlsip_gn_context*> rescue;     // one-problem handles of rescaled problems of the last batch
    std::vector<long long> rescue_prob;          // their problem indices (same length while the batch is resident)
    unsigned long long route = 0;       // ENLSIP_GN_ROUTE_* bits of the last solve (enlsip_gn_get_route)
    long long chunk0 = 0;               // first problem (index in the caller's batch) of the resident chunk: batches above the launch limit run in chunks
    long long tsqr_n2 = -1;             // n2 of the last tsqr_local on this handle
    // communicator of enlsip_gn_solve_tsqr: an RCCL communicator (created here or handed in) or the caller's all-gather
    void* tsqr_comm = nullptr;
    bool tsqr_comm_owned = false;
    enlsip_gn_allgather_fn tsqr_xfn = nullptr;
    void* tsqr_xctx = nullptr;
    int tsqr_ranks = 1, tsqr_rank = 0;
    bool tsqr_broken = false;           // enlsip_gn_tsqr_init_rccl failed: enlsip_gn_solve_tsqr refuses until a communicator is set again
    int tsqr_tags_seen = -1;            // gathered messages of the last enlsip_gn_solve_tsqr whose header carried the rank of their slot
    int tsqr_transport = 0;             // what moved the triangles in the last enlsip_gn_solve_tsqr (ENLSIP_GN_TRANSPORT_*)
    gn::DevBuf xbuf;                    // send message + G received messages
    float tsqr_ms[3] = {};              // local / exchange / combine of the last enlsip_gn_solve_tsqr (profiling on)
    long long *jpvtA = nullptr, *jpvtL = nullptr, *jpvtJ = nullptr;
    gn::ProbState* state = nullptr;
    // staging for the host-pointer API
    gn::DevBuf in_stage, out_stage, scratch, lag, newton;
    gn::ProbState* h_state = nullptr;   // pinned
    size_t h_state_cap = 0;
    // device-pointer inputs of the last solve (needed by resolve / get_JQ1 paths)
    const double* last_J = nullptr; long long last_ldj = 0, last_strideJ = 0;
    const double* last_rx = nullptr; long long last_stride_rx = 0;
    const double* last_cx = nullptr; long long last_stride_cx = 0;
    const double* last_At = nullptr; long long last_ldat = 0, last_strideAt = 0;

    // profiling
    bool profiling = false;
    bool profile_all_updates = false;   // enlsip_gn_set_profiling(h, 2): events around EVERY trailing-update launch (hundreds in a C4 sweep)
    hipEvent_t ev[8] = {};
    bool ev_ready = false;
    float stage_ms[ENLSIP_GN_STAGE_COUNT] = {};
    std::vector<hipEvent_t> upd_ev;   // pairs around level-0 update launches
    size_t upd_used = 0;
    double upd_bytes = 0.0;
    std::vector<double> upd_launch_bytes;   // per timed launch, same order as the event pairs
    std::vector<float> upd_launch_ms;
    float upd_avg_ms = 0.f;
    long long upd_launches = 0;
    // every OTHER trailing-update launch of the sweep (tree levels, a pair's second-panel columns): event pairs, summed time
    std::vector<hipEvent_t> oth_ev;
    size_t oth_used = 0;
    float oth_ms = 0.f;
    long long oth_launches = 0;
    double upd_all_bytes = 0.0;             // SURVEY 8d bytes of EVERY panel of the sweep on ALL its trailing columns
};
