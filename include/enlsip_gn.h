/*
 * enlsip_gn.h — C ABI of libenlsip_gn.so: the Gauss-Newton search-direction subproblem of
 * Enlsip.jl as MI355X-native (gfx950) HIP kernels.
 *
 * Drop-in boundary.  The reference has no FFI; the seam is two internal Julia
 * functions, and every entry point below names the reference lines it replaces
 * (paths relative to the Enlsip.jl repository):
 *
 *   enlsip_gn_solve*            gn_search_direction   src/enlsip_functions.jl:206-234
 *                               sub_search_direction  src/enlsip_functions.jl:116-153
 *                               + the QR / rank lines of update_working_set
 *                                                     src/enlsip_functions.jl:700, 768-769
 *                               pseudo_rank           src/enlsip_functions.jl:17-31
 *   enlsip_gn_resolve           sub_search_direction re-entry with truncated dimA/dimJ2
 *                                                     src/enlsip_functions.jl:1249-1253
 *   enlsip_gn_get_R / _diagR / _jpvt / _apply_qt / _apply_q / _get_JQ1
 *                               the QRPivoted accessors (.R, .p, .Q', .Q) and J*F_A.Q that
 *                               first/second_lagrange_mult_estimate!, search_direction_analys,
 *                               choose_subspace_dimensions, determine_solving_dim consume
 *                                                     src/enlsip_functions.jl:461-537, 1118-1291
 *   enlsip_gn_newton_direction  newton_search_direction after its Hessian sums    src/enlsip_functions.jl:348-423
 *   enlsip_gn_solve_tsqr        (new design, no reference counterpart) the same subproblem with the ROWS of one
 *                               tall J sharded over the GPUs of a node: `JQ1 = J * F_A.Q` ... `qr(J2, ColumnNorm())`
 *                               (src/enlsip_functions.jl:219-223) as a TSQR whose one exchange step is an RCCL
 *                               all-gather over xGMI inside the library; enlsip_gn_tsqr_local_dev / _combine_dev are
 *                               its two stages for callers that bring their own transport.  INTEGRATION.md section 5.
 *
 * Conventions
 *   - All matrices column-major (Julia / LAPACK layout), fp64, explicit leading dimensions.
 *   - Permutations are returned as 1-based LAPACK jpvt (Julia's F.p), int64.
 *   - Integers that mirror Julia Int / BlasInt are int64_t.
 *   - Return value: 0 = ok; < 0 = -(index of the offending argument), LAPACK style;
 *     > 0 = HIP runtime error code (text via enlsip_gn_last_error).  No exceptions cross the ABI.
 *   - Caller owns every buffer passed in; the library owns device workspaces (grown lazily,
 *     freed by enlsip_gn_destroy).  One handle = one HIP stream; a handle is not thread-safe,
 *     distinct handles are independent.
 *   - "_dev" entry points take DEVICE pointers (hipMalloc memory on the handle's device) and
 *     leave results in device memory; the others take HOST pointers and stage through PCIe.
 *   - Factors stay resident on the device behind the handle until the next solve on it.
 *   - Magnitudes.  qr(., ColumnNorm()) of the reference is dgeqp3, whose norms and reflectors scale internally: inputs of any
 *     magnitude are factored alike and pseudo_rank's absolute first test (src/enlsip_functions.jl:19) decides the rank.  The
 *     kernels square plainly; enlsip_gn_solve*, _solve_batched*, _factor_constraints and _solve_factored therefore detect inputs
 *     whose entries leave the range of plain sums of squares (beyond about 2^+-500) on their result and solve them again on
 *     copies scaled by a power of two, with the resident factors and the outputs scaled back: ranks, pivots,
 *     p, b, d and every accessor are those of the caller's data, as LAPACK would return them.  NOT covered: the row shards of the
 *     TSQR entry points (all shards of one matrix would have to agree on one scale before their local stages).
 */
#ifndef ENLSIP_GN_H
#define ENLSIP_GN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct enlsip_gn_context* enlsip_gn_handle;
typedef int (*enlsip_gn_allgather_fn)(void* ctx, const void* dsend, void* drecv, size_t bytes_per_rank, void* hip_stream);

/* which factorisation an accessor addresses */
enum {
    ENLSIP_GN_FACTOR_A = 0,   /* F_A   = qr(C.A', ColumnNorm())     n  x t   */
    ENLSIP_GN_FACTOR_L11 = 1, /* F_L11 = qr(F_A.R', ColumnNorm())   t  x kA  */
    ENLSIP_GN_FACTOR_J2 = 2   /* F_J2  = qr(J2,  ColumnNorm())      m  x n2  */
};

/* option flags */
enum {
    ENLSIP_GN_UPDATE_MFMA = 1,      /* trailing update through v_mfma_f64_16x16x4 (default on)   */
    ENLSIP_GN_UPDATE_REFLECTORS = 2 /* trailing update by sequential reflectors (debug / A-B)    */
};

typedef struct enlsip_gn_opts {
    int32_t device;        /* HIP device ordinal; -1 = current device                          */
    int32_t flags;         /* 0 = defaults; see enum above                                     */
    int32_t panel_width;   /* 0 = default (32)                                                 */
    int32_t tile_rows;     /* 0 = default (512); rows of one CAQR tile, 256 or 512             */
    void*   stream;        /* hipStream_t to run on, NULL = library creates its own            */
} enlsip_gn_opts;

/* per-problem scalar results, mirrors Iteration.{rankA,rankJ2,dimA,dimJ2} + code
 * (src/structures.jl:63-91, src/enlsip_functions.jl:217, 226-229) */
typedef struct enlsip_gn_info {
    int64_t rankA;
    int64_t rankJ2;
    int64_t code;   /* 1 = rankA == t, -1 = stabilised path */
    int64_t dimA;
    int64_t dimJ2;
    int64_t status; /* 0 ok; bit0: a triangular diagonal was exactly 0 (Julia would throw SingularException) */
} enlsip_gn_info;

int enlsip_gn_version(void);

int enlsip_gn_create(enlsip_gn_handle* h, const enlsip_gn_opts* opts);
int enlsip_gn_destroy(enlsip_gn_handle h);
/* message of the last failed call on h; h = NULL: why the last enlsip_gn_create of the calling thread failed */
const char* enlsip_gn_last_error(enlsip_gn_handle h);
int enlsip_gn_synchronize(enlsip_gn_handle h);

/*
 * One subproblem, host buffers.  Replaces, for given J (m x n), rx (m), At = C.A' (n x t,
 * column-major, i.e. the memory of Julia's t x n C.A read row-wise is NOT what is wanted: pass
 * the transpose explicitly), cx (t):
 *     F_A = qr(C.A', ColumnNorm()); rankA; F_L11 = qr(F_A.R', ColumnNorm());
 *     p_gn, F_J2 = gn_search_direction(J, rx, cx, F_A, F_L11, rankA, t, eps_rank, iter)
 * dimA_override / dimJ2_override: -1 = use rankA / rankJ2 (gn_search_direction);
 * Outputs: p (n), b (t), d (m), info, jpvtA (t), jpvtL (min(n,t)), jpvtJ2 (n - rankA; buffer of
 * n entries required).  Any output pointer may be NULL.
 */
int enlsip_gn_solve(enlsip_gn_handle h, int64_t m, int64_t n, int64_t t,
                    const double* J, int64_t ldj, const double* rx,
                    const double* At, int64_t ldat, const double* cx,
                    double eps_rank, int64_t dimA_override, int64_t dimJ2_override,
                    double* p, double* b, double* d, enlsip_gn_info* info,
                    int64_t* jpvtA, int64_t* jpvtL, int64_t* jpvtJ2);

/*
 * The constraint stage alone: F_A = qr(C.A', ColumnNorm()) (src/enlsip_functions.jl:700), rankA (:768, :17-31) and
 * F_L11 (:769) of one problem, left resident for enlsip_gn_first_lagrange (pass grad_fx), the F_A / F_L11 accessors and
 * apply_q / apply_qt — what update_working_set needs BEFORE it decides which constraint to drop (:700-704).  m is the row
 * count of the solve that follows (the workspace plan is shared).  J-related entries (F_J2, get_JQ1, resolve, gradient,
 * second estimate, jacobian_times) report an error until the next solve.  info: rankA, code, dimA (J2 fields zero).
 */
int enlsip_gn_factor_constraints(enlsip_gn_handle h, int64_t m, int64_t n, int64_t t, const double* At, int64_t ldat,
                                 const double* cx, double eps_rank, enlsip_gn_info* info);

/*
 * The solve that follows enlsip_gn_factor_constraints when the working set did not change: update_working_set factors C.A' once
 * (src/enlsip_functions.jl:700) and, in its s == 0 branch, goes on with that SAME factorisation (:768-771: rankA, F_L11,
 * gn_search_direction).  Same m, n, t as the enlsip_gn_factor_constraints call right before; F_A, F_L11, b, p1 stay as they are,
 * only J and rx are taken in.  Outputs as enlsip_gn_solve.
 */
int enlsip_gn_solve_factored(enlsip_gn_handle h, int64_t m, int64_t n, int64_t t, const double* J, int64_t ldj,
                             const double* rx, double eps_rank, int64_t dimJ2_override, double* p, double* b, double* d,
                             enlsip_gn_info* info, int64_t* jpvtA, int64_t* jpvtL, int64_t* jpvtJ2);

/*
 * Batch of independent subproblems of one shape, host buffers.  Problem k uses
 * J + k*strideJ, rx + k*m, At + k*strideAt, cx + k*t; outputs p + k*n, b + k*t, d + k*m,
 * info[k], jpvtA + k*t, jpvtL + k*min(n,t), jpvtJ2 + k*n.
 */
int enlsip_gn_solve_batched(enlsip_gn_handle h, int64_t batch, int64_t m, int64_t n, int64_t t,
                            const double* J, int64_t ldj, int64_t strideJ, const double* rx,
                            const double* At, int64_t ldat, int64_t strideAt, const double* cx,
                            double eps_rank,
                            double* p, double* b, double* d, enlsip_gn_info* info,
                            int64_t* jpvtA, int64_t* jpvtL, int64_t* jpvtJ2);

/*
 * Same, DEVICE buffers in, DEVICE buffers out (inputs are not modified).  Output pointers may be
 * NULL (results then stay only behind the handle).  The call returns after the work has been
 * enqueued and the per-problem info has been checked (one stream synchronisation).
 */
int enlsip_gn_solve_batched_dev(enlsip_gn_handle h, int64_t batch, int64_t m, int64_t n, int64_t t,
                                const double* dJ, int64_t ldj, int64_t strideJ, const double* drx,
                                const double* dAt, int64_t ldat, int64_t strideAt, const double* dcx,
                                double eps_rank,
                                double* dp, double* db, double* dd, enlsip_gn_info* dinfo,
                                int64_t* djpvtA, int64_t* djpvtL, int64_t* djpvtJ2);

/* ---- accessors on the resident factors of problem `prob` of the last solve (host buffers) ---- */

/* rows/cols of F.R for `which`: A: min(n,t) x t; L11: min(t,kA) x kA; J2: min(m,n2) x n2 */
int enlsip_gn_factor_shape(enlsip_gn_handle h, int which, int64_t prob, int64_t* rows, int64_t* cols);
/* F.R = triu(factors[1:min,:]) into R (ldr >= rows) */
int enlsip_gn_get_R(enlsip_gn_handle h, int which, int64_t prob, double* R, int64_t ldr);
int enlsip_gn_get_diagR(enlsip_gn_handle h, int which, int64_t prob, double* diag);
int enlsip_gn_get_jpvt(enlsip_gn_handle h, int which, int64_t prob, int64_t* jpvt);
/* v <- F.Q' * v  (length n for A, t for L11, m for J2) */
int enlsip_gn_apply_qt(enlsip_gn_handle h, int which, int64_t prob, double* v);
/* v <- F.Q * v */
int enlsip_gn_apply_q(enlsip_gn_handle h, int which, int64_t prob, double* v);
/* J * F_A.Q (m x n) into out (ld >= m) — src/enlsip_functions.jl:219, :526, :1249 */
int enlsip_gn_get_JQ1(enlsip_gn_handle h, int64_t prob, double* out, int64_t ld);
/*
 * sub_search_direction(J1, rx, cx, F_A, F_L11, F_J2, n, t, rankA, dimA, dimJ2, code) on the
 * resident factors (src/enlsip_functions.jl:1253 calls it with code = -1).  code must be 1 or -1.
 */
int enlsip_gn_resolve(enlsip_gn_handle h, int64_t prob, int64_t dimA, int64_t dimJ2, int64_t code,
                      double* p, double* b, double* d);

/* ---- multiplier estimates on the resident data of the last solve -------------------------
 * The two consumers the reference runs right after the subproblem, computed where F_A, J and J*F_A.Q already
 * are instead of on copies: host vectors in / out, `prob` = problem index of the last (batched) solve,
 * diag_scale = Constraint.diag_scale when row scaling is on (NULL: no back-transform), eps_rank as in the solve.
 *
 * enlsip_gn_gradient          grad = J' * rx                           src/enlsip_functions.jl:2690, :2734, :2830
 * enlsip_gn_first_lagrange    first_lagrange_mult_estimate!            src/enlsip_functions.jl:461-508
 *                             lambda (t) and iter.grad_res; grad_fx = NULL uses J' * rx of the resident J, rx
 * enlsip_gn_second_lagrange   second_lagrange_mult_estimate!           src/enlsip_functions.jl:514-537
 *                             b = J1' (rx + J p_gn) with the resident J1 = (J*F_A.Q)[:, 1:t] (the reference
 *                             recomputes J*F_A.Q here, :526).  Returns -7 if the pseudo-rank under eps_rank
 *                             exceeds the rank the solve used (those J1 columns were factored in place).
 * enlsip_gn_jacobian_times    Jp = J * p (m) and Ap = C.A * p (t, active rows), the products the line search sets up
 *                             with (src/enlsip_functions.jl:2226-2229: `Jp = J * p`, `active_Ap = (active_constraint.A) * p`),
 *                             on the J and A' of the last solve; either output may be NULL.  (The product with the FULL
 *                             constraint Jacobian, :2227: enlsip_gn_full_constraints_times.)
 * enlsip_gn_full_constraints_times   Ap = A * p with the FULL constraint Jacobian A (l x n, column-major, host; inactive rows
 *                             included): the other product of src/enlsip_functions.jl:2227 (`Ap = A * p`).  A is staged through
 *                             PCIe for one gemv — offered so that the whole line-search set-up can stay behind the ABI; a host
 *                             BLAS call is the faster choice for small l.  Needs no resident factors.
 * enlsip_gn_matrix_times_QA   out = M * F_A.Q for a HOST matrix M with the row count m of the last solve (rows x n, ldm >= rows):
 *                             what the reference writes as `J * F_A.Q` (src/enlsip_functions.jl:384, :526, :1249) for ANY such
 *                             matrix — one launch of the J*Q1 kernel on the resident reflectors; -3 when rows != m.  The Julia
 *                             glue's `Base.:*(::AbstractMatrix, ::DeviceQ)` rests on it.
 */
int enlsip_gn_gradient(enlsip_gn_handle h, int64_t prob, double* grad);
int enlsip_gn_full_constraints_times(enlsip_gn_handle h, int64_t l, int64_t n, const double* A, int64_t lda, const double* p,
                                     double* Ap);
int enlsip_gn_matrix_times_QA(enlsip_gn_handle h, int64_t prob, int64_t rows, const double* M, int64_t ldm, double* out,
                              int64_t ldo);
int enlsip_gn_jacobian_times(enlsip_gn_handle h, int64_t prob, const double* p, double* Jp, double* Ap);
int enlsip_gn_first_lagrange(enlsip_gn_handle h, int64_t prob, const double* grad_fx, const double* diag_scale,
                             double eps_rank, double* lambda, double* grad_res);
int enlsip_gn_second_lagrange(enlsip_gn_handle h, int64_t prob, const double* p_gn, const double* diag_scale,
                              double eps_rank, double* lambda);

/* ---- Newton direction on the resident data of the last solve -------------------------------------------------
 * newton_search_direction (src/enlsip_functions.jl:348-423) after its two Hessian sums (:391-396), which are callback-bound and
 * stay with the caller: Gamma = r_mat - c_mat (n x n, host, column-major, ldg >= n).  Computes E = F_A.Q' Gamma F_A.Q (:398),
 * W22 = E22 + J2'J2, W21 = E21 + J2'J1 (:405-409), d = -W21 p1 - J2' rx (:411), cholesky((W22 + W22')/2) and the two triangular
 * solves (:414-420), p = F_A.Q [p1; p2] (:421) on the resident F_A, p1, J (J * F_A.Q is recomputed, as at :384).
 * *not_posdef = 1 (and p = 0) when the symmetrised W22 is not positive definite (:417-420: `error = true`).  With rankA == n the
 * reference returns p1 as it is (:374-376); so does this.  A rank-deficient working set (t > rankA) takes the reference's other
 * branch: p1 = F_L11.P[1:rankA, 1:rankA] * dp1 (:371-373) and E = E[F_L11.p, F_L11.p] (:396-399) — defined for t >= n only,
 * because F_L11.p has min(n, t) entries and :402-403 read rows up to n; with n > t > rankA the reference runs out of bounds and
 * this entry point returns -7. */
int enlsip_gn_newton_direction(enlsip_gn_handle h, int64_t prob, const double* Gamma, int64_t ldg, double* p, int64_t* not_posdef);

/* ---- row-sharded TSQR building blocks (multi-GPU config C4) ----------
 * One tall residual Jacobian whose ROWS are sharded over G GPUs; the (small) constraint data
 * At, cx are replicated.  Same mathematics as enlsip_gn_solve:  F_A, rankA, F_L11, p1 are
 * computed redundantly on every rank (bitwise identical), J*Q1 and d_temp = -J1 p1 - rx are
 * row-local, the unpivoted QR of [J2 | d_temp] is done on the local rows, and only the
 * n2 x n2 triangles travel.
 *
 * Local stage (device buffers in, device buffers out):
 *   dRloc  n2 x n2 upper-triangular factor of the local rows, column-major, PACKED (ld = n2), so the
 *          first n2*n2 doubles can be all-gathered as they are; the caller provides n*n doubles
 *          (n2 = n - rankA <= n is only known afterwards and is returned in *n2_out)
 *   dzloc  (Q_loc' d_loc)[1:n2]             (n doubles provided)
 *   tail_sq  ||(Q_loc' d_loc)[n2+1:]||^2    (host)
 */
int enlsip_gn_tsqr_local_dev(enlsip_gn_handle h, int64_t m_loc, int64_t n, int64_t t,
                             const double* dJ, int64_t ldj, const double* drx,
                             const double* dAt, int64_t ldat, const double* dcx, double eps_rank,
                             double* dRloc, double* dzloc, double* tail_sq, int64_t* n2_out);
/*
 * Combine stage, run redundantly on every rank after the all-gather (RCCL) of the G local results:
 *   dRstack  G blocks of n2 x n2 (each column-major, ld = n2, i.e. the dRloc buffers packed to n2*n2)
 *   dzstack  G blocks of n2
 * Factors the stacked (G*n2) x n2 matrix (unpivoted CAQR, then the pivoted QR of its R — the same
 * plan as a single-GPU solve), solves, and applies Q1 of the handle's resident F_A (from the
 * local stage on the SAME handle).  HOST outputs: p (n), dlead (n2) = leading entries of
 * F_J2.Q' d, comb_tail_sq = squared norm of the remaining entries of the stacked rhs (add the
 * ranks' tail_sq for ||d||^2), info, jpvtJ2 (n2).
 */
int enlsip_gn_tsqr_combine_dev(enlsip_gn_handle h, int64_t G, int64_t n2,
                               const double* dRstack, const double* dzstack, double eps_rank,
                               double* p, double* dlead, double* comb_tail_sq,
                               enlsip_gn_info* info, int64_t* jpvtJ2);

/* ---- the same, as ONE collective call (every rank of the communicator calls it with its row block) -------------------------
 *
 * Communicator of a handle (default: one rank, no exchange).  Exactly one of:
 *   enlsip_gn_tsqr_init_rccl     the library creates an RCCL communicator: rank 0 obtains 128 bytes from
 *                                enlsip_gn_tsqr_unique_id, the caller hands them to every rank by whatever means it has (Julia:
 *                                Distributed / MPI.jl; Python: torch.distributed broadcast), every rank calls init_rccl (collective,
 *                                ncclCommInitRank).  The handle's device must be the rank's GPU.  RCCL is loaded at run time
 *                                (librccl.so.1; ENLSIP_GN_RCCL_LIB overrides): the library has no link-time dependency on it.
 *   enlsip_gn_tsqr_set_comm      an existing ncclComm_t of the caller (not destroyed by the library); NULL = back to one rank.
 *   enlsip_gn_tsqr_set_exchange  any other transport: fn(ctx, dsend, drecv, bytes_per_rank, hip_stream) must all-gather
 *                                bytes_per_rank bytes of DEVICE memory from every rank into drecv (rank-major) and return 0 once
 *                                drecv is complete or the transfer is enqueued on hip_stream; dsend is complete when fn is called.
 *
 * enlsip_gn_solve_tsqr: rank g passes its m_loc rows of J and rx (device pointers; the row blocks may have different heights)
 * and the replicated At, cx.  One message per rank travels: the packed upper triangle of the local R (8 n2 (n2 + 1) / 2 bytes:
 * 4.2 MB at n2 = 1024), z = (Q_loc' d_loc)[1:n2], the squared norm of the local tail and the sender's n2; its length depends on n
 * alone, so the ranks' counts agree even if their n2 do not — that case (the ranks see different constraint ranks) returns -13.
 * A rank that fails BEFORE the exchange (bad arguments, out of memory, a HIP error in its local stage) leaves its peers waiting
 * in the all-gather: such a failure is fatal for the communicator.  After a failed enlsip_gn_tsqr_init_rccl the handle has no
 * communicator and enlsip_gn_solve_tsqr returns an error until one is set again.  Every rank returns the same HOST
 * outputs: p (n), dlead (n2 <= n entries: leading entries of F_J2.Q' d), d_norm = ||d||_2 over all ranks, info, jpvtJ2 (n2 <= n).
 */
int enlsip_gn_tsqr_unique_id(void* id128);
int enlsip_gn_tsqr_init_rccl(enlsip_gn_handle h, const void* id128, int nranks, int rank);
int enlsip_gn_tsqr_set_comm(enlsip_gn_handle h, void* nccl_comm, int nranks, int rank);
int enlsip_gn_tsqr_set_exchange(enlsip_gn_handle h, enlsip_gn_allgather_fn fn, void* ctx, int nranks, int rank);
int enlsip_gn_solve_tsqr(enlsip_gn_handle h, int64_t m_loc, int64_t n, int64_t t,
                         const double* dJ, int64_t ldj, const double* drx,
                         const double* dAt, int64_t ldat, const double* dcx, double eps_rank,
                         double* p, double* dlead, double* d_norm, enlsip_gn_info* info, int64_t* jpvtJ2);
/* local / exchange / combine time (ms, HIP events) of the last enlsip_gn_solve_tsqr with profiling enabled */
int enlsip_gn_tsqr_get_stage_ms(enlsip_gn_handle h, float* ms3);
/* what moved the messages in the last enlsip_gn_solve_tsqr of this handle: an attached RCCL communicator is used for the
 * exchange even when it has ONE rank (a self-gather), so that the RCCL leg runs on a one-GPU box too */
enum {
    ENLSIP_GN_TRANSPORT_NONE = 0,     /* one rank, no communicator: a device copy */
    ENLSIP_GN_TRANSPORT_RCCL = 1,     /* ncclAllGather on the handle's stream */
    ENLSIP_GN_TRANSPORT_CALLBACK = 2  /* the caller's all-gather (enlsip_gn_tsqr_set_exchange) */
};
int enlsip_gn_tsqr_get_transport(enlsip_gn_handle h, int* transport);
/* The same with the evidence that the collective really spanned the communicator: *ranks = the handle's rank count, and
 * *rank_tags_seen = how many of the gathered messages carried, in their header, the rank their slot belongs to AND the same rank
 * count (every message is tagged by its sender; -1: nothing was gathered, n2 = 0).  transport == RCCL and rank_tags_seen == ranks
 * on every rank is "RCCL moved one message from each of N distinct ranks" (bench.py --config C4 prints and asserts it). */
int enlsip_gn_tsqr_get_exchange(enlsip_gn_handle h, int* transport, int* ranks, int* rank_tags_seen);

/* ---- instrumentation: HIP-event time (ms) of the stages of the last solve ------------------ */
enum {
    ENLSIP_GN_STAGE_CONSTRAINT = 0, /* F_A, F_L11, p1, T factor            */
    ENLSIP_GN_STAGE_JQ1 = 1,        /* J*Q1 and d_temp                      */
    ENLSIP_GN_STAGE_PANEL = 2,      /* all CAQR panel factorisations (tiles and tree nodes)                     */
    ENLSIP_GN_STAGE_UPDATE = 3,     /* EVERY trailing-update launch of the sweep: the level-0 far passes, the tree
                                     * levels and a pair's second-panel columns (sum of their HIP-event times)  */
    ENLSIP_GN_STAGE_PIVOT = 4,      /* pivoted QR of R0 + triangular solves */
    ENLSIP_GN_STAGE_TOTAL = 5,
    ENLSIP_GN_STAGE_COUNT = 6
};
/* enable = 1 records events per stage and around the level-0 far updates (adds stream bubbles; off by default); enable = 2 also
 * around every other trailing-update launch of the sweep (tree levels, second-panel columns): ENLSIP_GN_STAGE_UPDATE then is the
 * sum over ALL update launches and enlsip_gn_get_update_totals reports them; with enable = 1 it is the far passes alone and the
 * other update launches stay inside ENLSIP_GN_STAGE_PANEL (hundreds of event pairs would stretch a C4 sweep by ~3 %) */
int enlsip_gn_set_profiling(enlsip_gn_handle h, int enable);
int enlsip_gn_get_stage_ms(enlsip_gn_handle h, float* ms /* ENLSIP_GN_STAGE_COUNT */);
/* average duration (ms) and count of the level-0 trailing-update launches of the last solve,
 * measured with HIP events on the handle's stream (bench.py roofline leg) */
int enlsip_gn_get_update_stats(enlsip_gn_handle h, float* avg_ms, int64_t* launches,
                               double* algorithmic_bytes);
/* the same launch by launch (sweep order: one entry per panel, or per panel pair where two panels share a pass): SURVEY 8d
 * bytes 8 (2 m_k n_k + m_k b + b^2) of the panels the launch applies, and its HIP-event time.  *count = launches recorded;
 * at most cap entries are written. */
int enlsip_gn_get_update_table(enlsip_gn_handle h, int64_t cap, double* algorithmic_bytes, float* ms, int64_t* count);
/* How the last solve on this handle was launched (what the library chose on its own): *pipeline_split = number of problems the
 * first of the two pipelined halves owned (0: one stream), *panel_pairs = 1 when the CAQR sweep applied two panels per pass over
 * the far trailing columns, *tile_rows = rows of a level-0 CAQR tile.  Tests use it to assert that a configuration really took
 * the path a benchmark times. */
int enlsip_gn_get_launch_plan(enlsip_gn_handle h, int64_t* pipeline_split, int* panel_pairs, int64_t* tile_rows);
/* All trailing-update launches of the last profiled solve: HIP-event time of the level-0 far passes (the launches of
 * enlsip_gn_get_update_table) and of every OTHER update launch (tree levels, second-panel columns), and SURVEY 8d's
 * 8 (2 m_k n_k + m_k b + b^2) summed over EVERY panel of the sweep with n_k = all columns right of the panel (times the batch):
 * bytes / (far_ms + other_ms) is the trailing update's rate counted over all of its kernels (bench.py: roofline.all_update_kernels) */
int enlsip_gn_get_update_totals(enlsip_gn_handle h, float* far_ms, float* other_ms, int64_t* other_launches,
                                double* all_panels_bytes);
/* GB/s (read + write) of an in-place non-temporal read-modify-write stream over `bytes` of the handle's scratch memory with the
 * trailing update's access shape, HIP events around `reps` passes: the same-box ceiling of an in-place update (bench.py) */
int enlsip_gn_measure_stream(enlsip_gn_handle h, int64_t bytes, int reps, double* gbytes_per_s);
/* Which kernel-selection branches of the host code the last solve on this handle took: one bit per branch (and per template
 * instantiation a branch chooses between), ORed over both pipeline halves, every chunk and both attempts of a solve.  Every
 * place in the library that picks a kernel from the shape sets a bit; tests/test_dispatch_grid.py derives a stratified shape
 * list from the same conditions and asserts that every bit is hit by a case that is compared against the oracle — a new fast
 * path gets a bit here and cannot ship without such a case.  enlsip_gn_route_name(bit) = its name, NULL past the last bit. */
enum {
    ENLSIP_GN_ROUTE_CONSTRAINT_WAVE32 = 0,   /* k_constraint_small<32>: n, t <= 32                                     */
    ENLSIP_GN_ROUTE_CONSTRAINT_WAVE64,       /* k_constraint_small<64>: n <= 64, t <= 63                               */
    ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R1_256,   /* k_constraint, matrices in LDS, rows <= 32                              */
    ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R1_512,   /* rows <= 64                                                             */
    ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R2,       /* rows <= 128                                                            */
    ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R4,       /* rows <= 256                                                            */
    ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R8,       /* rows <= 512                                                            */
    ENLSIP_GN_ROUTE_CONSTRAINT_LDS_R16,      /* rows <= 1024                                                           */
    ENLSIP_GN_ROUTE_CONSTRAINT_GLOBAL,       /* k_constraint factoring in global memory (beyond the LDS area, n > 512) */
    ENLSIP_GN_ROUTE_CONSTRAINT_REG4,         /* F_A by k_geqp3_reg<4> (n <= 256, beyond the LDS area)                  */
    ENLSIP_GN_ROUTE_CONSTRAINT_REG8,         /* F_A by k_geqp3_reg<8> (n <= 512)                                       */
    ENLSIP_GN_ROUTE_CONSTRAINT_DIST,         /* more than 64 constraints beyond the LDS area: launch-per-step QR       */
    ENLSIP_GN_ROUTE_JQ1_FUSED_SMALL,         /* k_jq1_factor_small: J*Q1 + the one narrow panel in one launch          */
    ENLSIP_GN_ROUTE_JQ1_ROWS32,              /* k_jq1_rows<32>: a lane per row, n <= 32                                */
    ENLSIP_GN_ROUTE_JQ1_ROWS2,               /* k_jq1_rows2: two lanes per row, 32 < n <= 64                           */
    ENLSIP_GN_ROUTE_JQ1_ROWS64,              /* k_jq1_rows<64> (leading dimensions beyond 2^23)                        */
    ENLSIP_GN_ROUTE_JQ1_V2_N128,             /* k_jq1_v2<2,4,2>: kA = 64, n = 128, m multiple of 32                    */
    ENLSIP_GN_ROUTE_JQ1_V2_N256,
    ENLSIP_GN_ROUTE_JQ1_V2_N384,
    ENLSIP_GN_ROUTE_JQ1_V2_N512,
    ENLSIP_GN_ROUTE_JQ1_MFMA,                /* k_jq1_mfma: every other shape                                          */
    ENLSIP_GN_ROUTE_JQ1_PLAIN,               /* k_jq1 (ENLSIP_GN_UPDATE_REFLECTORS; also behind get_JQ1 / Newton)      */
    ENLSIP_GN_ROUTE_SWEEP_PLAIN,             /* CAQR: one panel per pass over the trailing columns                     */
    ENLSIP_GN_ROUTE_SWEEP_PAIRS,             /* two panels per pass over the far columns                               */
    ENLSIP_GN_ROUTE_SWEEP_LOOKAHEAD,         /* a pair's far update split over two streams                             */
    ENLSIP_GN_ROUTE_SWEEP_PASSENGER,         /* last narrow panel: d rides through the factor kernel                   */
    ENLSIP_GN_ROUTE_SWEEP_TREE,              /* more than one tile: tree levels                                        */
    ENLSIP_GN_ROUTE_SWEEP_TILE256,           /* 256-row tiles (m <= 256 or opts.tile_rows = 256)                       */
    ENLSIP_GN_ROUTE_SWEEP_TILE512,
    ENLSIP_GN_ROUTE_SWEEP_REFLECTORS,        /* trailing update reflector by reflector (ENLSIP_GN_UPDATE_REFLECTORS)   */
    ENLSIP_GN_ROUTE_SWEEP_UPPER_INPUT,       /* J already upper triangular (TSQR combine of one shard): no sweep       */
    ENLSIP_GN_ROUTE_PIVOT_WAVE32,            /* k_pivot_small<32>: kp <= 32, n2 + 1 <= 64, one problem                 */
    ENLSIP_GN_ROUTE_PIVOT_WAVE64,            /* k_pivot_small<64>: kp <= 64, n2 + 1 <= 64                              */
    ENLSIP_GN_ROUTE_PIVOT_WAVE2,             /* k_pivot_small2: two problems per wave, kp <= 32, n2 + 1 <= 32, batch   */
    ENLSIP_GN_ROUTE_PIVOT_LDS_R1_256,        /* k_pivot_solve (pivoted QR of R0 in LDS, or only the solves behind the  */
    ENLSIP_GN_ROUTE_PIVOT_LDS_R1_512,        /*                blocked form), by rows min(m, n): <= 32, 64, 128, ...   */
    ENLSIP_GN_ROUTE_PIVOT_LDS_R2,
    ENLSIP_GN_ROUTE_PIVOT_LDS_R4,
    ENLSIP_GN_ROUTE_PIVOT_LDS_R8,
    ENLSIP_GN_ROUTE_PIVOT_LDS_R16,
    ENLSIP_GN_ROUTE_PIVOT_BLOCKS,            /* run_qrcp_block: register blocks (R0 beyond the LDS area, kp <= 512)    */
    ENLSIP_GN_ROUTE_PIVOT_BLOCKS_448,        /*   forms of the block kernel that were launched: 448 rows               */
    ENLSIP_GN_ROUTE_PIVOT_BLOCKS_512,
    ENLSIP_GN_ROUTE_PIVOT_BLOCKS_256,
    ENLSIP_GN_ROUTE_PIVOT_BLOCKS_128,
    ENLSIP_GN_ROUTE_PIVOT_HYBRID,            /* kp > 512: launch-per-step head, then the register blocks               */
    ENLSIP_GN_ROUTE_PIVOT_STEPS,             /* kp > 512 with ENLSIP_GN_QRCP_HYBRID=0: one launch per step to the end  */
    ENLSIP_GN_ROUTE_PIPELINE_SPLIT,          /* batch >= 128: two halves on two streams                                */
    ENLSIP_GN_ROUTE_CHUNKED,                 /* batch > 32768: consecutive chunks                                      */
    ENLSIP_GN_ROUTE_SECOND_ATTEMPT,          /* some A was rank deficient: J2 wider than speculated, redone            */
    ENLSIP_GN_ROUTE_RESCALED,                /* inputs beyond the range of plain sums of squares: solved on a copy     */
                                             /* scaled by a power of two (LAPACK's dnrm2 / dlarfg behaviour)           */
    ENLSIP_GN_ROUTE_COUNT
};
int enlsip_gn_get_route(enlsip_gn_handle h, uint64_t* mask);
const char* enlsip_gn_route_name(int bit);
/* Debugging aid (tests/probes/pair_probe_w*.py): copies the working matrix W of problem `prob` (ldw x (n + 1): J*Q1 with the CAQR factors of
 * [J2 | d] in place) as it stands to host memory; *ldw_out = its leading dimension; -3 when cap_doubles is too small.  Together with
 * ENLSIP_GN_DEBUG_MAXPAN / ENLSIP_GN_DEBUG_STAGE (stop the CAQR sweep after so many panels / inside the first pair) this is how an
 * orthogonality defect is located stage by stage. */
int enlsip_gn_debug_copy_W(enlsip_gn_handle h, int64_t prob, double* out, int64_t* ldw_out, int64_t cap_doubles);

#ifdef __cplusplus
}
#endif
#endif /* ENLSIP_GN_H */
