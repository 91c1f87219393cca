/* Plain C99 caller of include/enlsip_gn.h (no C++, no HIP headers): what a cgo / ccall / JNI binding sees.
 * Builds a small deterministic problem (m = 12, n = 5, t = 2), solves it through enlsip_gn_solve and checks, in C,
 * the two optimality conditions of the subproblem: A p + c = 0 and Z'J'(J p + r) = 0 via the resident F_A (Q' applied to
 * the gradient: its last n - t entries must vanish).  Exit code 0 = ok, 3 = no usable GPU (library refused to create a
 * handle, with a message), anything else = failure.  Test infrastructure (tests/test_abi.py). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "enlsip_gn.h"

#define M 12
#define N 5
#define T 2

int main(void) {
    enlsip_gn_handle h = NULL;
    enlsip_gn_opts opts = {-1, 0, 0, 0, NULL};
    int rc = enlsip_gn_create(&h, &opts);
    if (rc != 0) {
        const char* msg = enlsip_gn_last_error(NULL);
        printf("create refused (rc %d): %s\n", rc, msg ? msg : "(no message)");
        /* every entry point must turn a NULL handle into an error code, never into an access */
        {
            double x[4] = {0, 0, 0, 0};
            int64_t i4[4] = {0, 0, 0, 0}, r = 0, c = 0;
            enlsip_gn_info inf;
            int bad = 0;
            bad += enlsip_gn_solve(NULL, 2, 2, 0, x, 2, x, NULL, 2, NULL, 1e-8, -1, -1, x, NULL, x, &inf, NULL, NULL, i4) == 0;
            bad += enlsip_gn_solve_batched(NULL, 1, 2, 2, 0, x, 2, 4, x, NULL, 2, 0, NULL, 1e-8, x, NULL, x, &inf, NULL, NULL, i4) == 0;
            bad += enlsip_gn_solve_batched_dev(NULL, 1, 2, 2, 0, x, 2, 4, x, NULL, 2, 0, NULL, 1e-8, x, NULL, x, NULL, NULL, NULL, i4) == 0;
            bad += enlsip_gn_factor_constraints(NULL, 2, 2, 0, NULL, 2, NULL, 1e-8, &inf) == 0;
            bad += enlsip_gn_factor_shape(NULL, 0, 0, &r, &c) == 0;
            bad += enlsip_gn_get_R(NULL, 0, 0, x, 2) == 0;
            bad += enlsip_gn_apply_qt(NULL, 0, 0, x) == 0;
            bad += enlsip_gn_resolve(NULL, 0, 0, 0, 1, x, x, x) == 0;
            bad += enlsip_gn_gradient(NULL, 0, x) == 0;
            bad += enlsip_gn_newton_direction(NULL, 0, x, 2, x, &r) == 0;
            bad += enlsip_gn_tsqr_set_exchange(NULL, NULL, NULL, 1, 0) == 0;
            bad += enlsip_gn_solve_tsqr(NULL, 2, 2, 0, x, 2, x, NULL, 2, NULL, 1e-8, x, x, x, &inf, i4) == 0;
            bad += enlsip_gn_synchronize(NULL) == 0;
            bad += enlsip_gn_set_profiling(NULL, 1) == 0;
            bad += enlsip_gn_destroy(NULL) != 0;            /* destroying nothing is fine */
            if (bad) { printf("%d entry points accepted a NULL handle\n", bad); return 8; }
        }
        return (msg && msg[0]) ? 3 : 4;
    }
    double J[M * N], rx[M], At[N * T], cx[T];
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) J[i + j * M] = sin(1.0 + 0.7 * i + 1.3 * j) + (i == j ? 2.0 : 0.0);
    for (int i = 0; i < M; ++i) rx[i] = cos(0.3 * i) - 0.5;
    for (int k = 0; k < T; ++k)
        for (int j = 0; j < N; ++j) At[j + k * N] = cos(0.9 * j + 2.1 * k) + (j == k ? 1.5 : 0.0);
    cx[0] = 0.25; cx[1] = -0.75;

    double p[N], b[T], d[M];
    int64_t jA[T], jL[T], jJ[N];
    enlsip_gn_info info;
    rc = enlsip_gn_solve(h, M, N, T, J, M, rx, At, N, cx, 1.4901161193847656e-08, -1, -1, p, b, d, &info, jA, jL, jJ);
    if (rc != 0) { printf("solve failed (rc %d): %s\n", rc, enlsip_gn_last_error(h)); return 5; }
    if (info.rankA != T || info.rankJ2 != N - T || info.code != 1) { printf("unexpected ranks\n"); return 6; }

    double worst = 0.0;
    for (int k = 0; k < T; ++k) {                       /* A p + c */
        double s = cx[k];
        for (int j = 0; j < N; ++j) s += At[j + k * N] * p[j];
        if (fabs(s) > worst) worst = fabs(s);
    }
    double res[M], g[N];
    for (int i = 0; i < M; ++i) {                       /* J p + r */
        double s = rx[i];
        for (int j = 0; j < N; ++j) s += J[i + j * M] * p[j];
        res[i] = s;
    }
    for (int j = 0; j < N; ++j) {                       /* g = J'(J p + r) */
        double s = 0.0;
        for (int i = 0; i < M; ++i) s += J[i + j * M] * res[i];
        g[j] = s;
    }
    rc = enlsip_gn_apply_qt(h, ENLSIP_GN_FACTOR_A, 0, g);   /* Q1' g: entries t..n-1 = Z' g */
    if (rc != 0) { printf("apply_qt failed (rc %d)\n", rc); return 7; }
    for (int j = T; j < N; ++j)
        if (fabs(g[j]) > worst) worst = fabs(g[j]);
    printf("C client: ranks (%lld, %lld), worst optimality residual %.3e\n", (long long)info.rankA, (long long)info.rankJ2, worst);
    enlsip_gn_destroy(h);
    return worst < 1e-10 ? 0 : 8;
}
