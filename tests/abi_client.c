/* A plain-C client of libslamhip.so -- what a cgo / ccall / JNI binding sees: include/slamhip.h compiled AS C, the
 * library linked with no Python, torch or C++ runtime of the caller's.  Drives the hand-derived known-answer tests
 * of SURVEY.md 8c (KAT-1 .. KAT-4; values worked out by hand there, not taken from any implementation) through the
 * EKF entry points in both dtypes, checks that the sweep and the grid form of the gating decide alike on a small
 * map, and takes a FastSLAM filter through one enqueued step.  The CPU suite compiles this file (the header must be
 * valid C, every symbol must resolve); the GPU suite runs it.  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slamhip.h"
#include "slamhip_diag.h"

static int failures = 0;

#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            ++failures;                                    \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                  \
            fprintf(stderr, "\n");                         \
        }                                                  \
    } while (0)

#define OK(call)                                                                         \
    do {                                                                                 \
        const int rc_ = (call);                                                          \
        if (rc_ != SLAM_OK) {                                                            \
            ++failures;                                                                  \
            fprintf(stderr, "FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, slam_last_error()); \
        }                                                                                \
    } while (0)

static int close_to(double a, double b, double tol) { return fabs(a - b) <= tol * (1.0 + fabs(b)); }

/* upload a state given in doubles to a handle of either dtype */
static int upload(slam_ekf_t h, int dtype, const double* x, const double* P, int n) {
    if (dtype == SLAM_F64) return slam_ekf_set_state(h, x, P, n, n);
    float* xf = (float*)malloc(sizeof(float) * (size_t)n);
    float* Pf = (float*)malloc(sizeof(float) * (size_t)n * (size_t)n);
    for (int i = 0; i < n; ++i) xf[i] = (float)x[i];
    for (int i = 0; i < n * n; ++i) Pf[i] = (float)P[i];
    const int rc = slam_ekf_set_state(h, xf, Pf, n, n);
    free(xf);
    free(Pf);
    return rc;
}

static int download(slam_ekf_t h, int dtype, double* x, double* P, int n) {
    if (dtype == SLAM_F64) return slam_ekf_get_state(h, x, P, n, n);
    float* xf = (float*)malloc(sizeof(float) * (size_t)n);
    float* Pf = (float*)malloc(sizeof(float) * (size_t)n * (size_t)n);
    const int rc = slam_ekf_get_state(h, xf, Pf, n, n);
    for (int i = 0; i < n; ++i) x[i] = xf[i];
    for (int i = 0; i < n * n; ++i) P[i] = Pf[i];
    free(xf);
    free(Pf);
    return rc;
}

static void ekf_kats(int dtype) {
    const double tol = dtype == SLAM_F64 ? 1e-12 : 2e-6;
    const double PI = 3.14159265358979323846;
    const double R[4] = {0.1 * 0.1, 0.0, 0.0, (PI / 180.0) * (PI / 180.0)};      /* column-major 2 x 2 */
    slam_ekf_t h = NULL;
    OK(slam_ekf_create(&h, dtype, 8, 0));
    if (!h) return;
    int dt = -1;
    OK(slam_ekf_dtype(h, &dt));
    CHECK(dt == dtype, "dtype %d", dt);

    /* KAT-3: predict from x = 0, P = 0; v = 8, g = 0, w = 4, dt = 0.025, Q = diag(0.5^2, (3 deg)^2) */
    {
        const double x0[3] = {0, 0, 0}, P0[9] = {0};
        const double Q[4] = {0.25, 0.0, 0.0, (3.0 * PI / 180.0) * (3.0 * PI / 180.0)};
        OK(upload(h, dtype, x0, P0, 3));
        OK(slam_ekf_predict(h, 8.0, 0.0, 4.0, Q, 0.025));
        double x[3], P[9];
        OK(download(h, dtype, x, P, 3));
        CHECK(close_to(x[0], 0.2, tol) && close_to(x[1], 0.0, tol) && close_to(x[2], 0.0, tol), "KAT-3 x = %g %g %g", x[0], x[1], x[2]);
        CHECK(close_to(P[0], 1.5625e-4, 1e-6) && close_to(P[4], 1.09662271e-4, 1e-6) && close_to(P[5], 2.74155678e-5, 1e-6) &&
                  close_to(P[8], 6.85389195e-6, 1e-6) && close_to(P[7], P[5], tol),
              "KAT-3 P = %g %g %g %g", P[0], P[4], P[5], P[8]);
    }
    /* KAT-4: add_features from x = 0, P = 0 with z = (10, 0): landmark at (10, 0), P_ff = diag(R11, 100 R22), cross blocks 0 */
    {
        const double x0[3] = {0, 0, 0}, P0[9] = {0}, zn[2] = {10.0, 0.0};
        OK(upload(h, dtype, x0, P0, 3));
        OK(slam_ekf_augment(h, zn, 1, R));
        int N = -1;
        OK(slam_ekf_num_landmarks(h, &N));
        CHECK(N == 1, "KAT-4 N = %d", N);
        double x[5], P[25];
        OK(download(h, dtype, x, P, 5));
        CHECK(close_to(x[3], 10.0, tol) && close_to(x[4], 0.0, tol), "KAT-4 landmark %g %g", x[3], x[4]);
        CHECK(close_to(P[3 * 5 + 3], R[0], 1e-6) && close_to(P[4 * 5 + 4], 100.0 * R[3], 1e-6) && fabs(P[3 * 5 + 4]) < 1e-9 &&
                  fabs(P[3]) < 1e-12 && fabs(P[4]) < 1e-12,
              "KAT-4 P_ff = %g %g", P[3 * 5 + 3], P[4 * 5 + 4]);
    }
    /* KAT-1 / KAT-2: x = (0, 0, 0, 10, 0), P = I: predict_observation, nis and nd of z = (10.5, 0.02) */
    {
        const double x0[5] = {0, 0, 0, 10, 0};
        double P0[25] = {0};
        for (int i = 0; i < 5; ++i) P0[i * 5 + i] = 1.0;
        OK(upload(h, dtype, x0, P0, 5));
        double zp[2], Hv[6], Hf[4];
        OK(slam_ekf_predict_observation(h, 1, zp, Hv, Hf));
        /* H = [[-1, 0, 0, 1, 0], [0, -0.1, -1, 0, 0.1]]; Hv, Hf column-major */
        CHECK(close_to(zp[0], 10.0, tol) && close_to(zp[1], 0.0, tol), "KAT-1 zp = %g %g", zp[0], zp[1]);
        CHECK(close_to(Hv[0], -1.0, tol) && close_to(Hv[1], 0.0, tol) && close_to(Hv[3], -0.1, tol) && close_to(Hv[5], -1.0, tol) &&
                  close_to(Hf[0], 1.0, tol) && close_to(Hf[3], 0.1, tol) && close_to(Hf[1], 0.0, tol) && close_to(Hf[2], 0.0, tol),
              "KAT-1 H blocks");
        const double z[2] = {10.5, 0.02};
        double out[2];
        OK(slam_ekf_nis(h, z, 1, R, out));
        CHECK(close_to(out[0], 0.12477014923494524, 10 * tol) && close_to(out[1], 0.843006098545911, 10 * tol), "KAT-2 nis %.15g nd %.15g",
              out[0], out[1]);
        int32_t a = 99;
        OK(slam_ekf_associate(h, z, 1, R, 4.0, 25.0, &a));
        CHECK(a == 1, "KAT-2 association %d", a);
        /* errors do not cross the ABI as exceptions: a status code and a message */
        const int32_t bad = 7;
        CHECK(slam_ekf_update(h, z, &bad, 1, R, SLAM_FORM_CHOLESKY) == SLAM_E_BADARG, "out-of-range idf accepted");
        CHECK(strlen(slam_last_error()) > 0, "no error text");
    }
    /* the two forms of the gating on a small map: identical decisions (slam_ekf_set_gate_mode) */
    {
        enum { N = 6, n = 3 + 2 * N, NZ = 4 };
        double x[n], P[n * n];
        memset(P, 0, sizeof P);
        x[0] = 1.0; x[1] = 2.0; x[2] = 0.4;
        for (int j = 0; j < N; ++j) {
            x[3 + 2 * j] = 1.0 + 12.0 * cos(0.9 * j);
            x[4 + 2 * j] = 2.0 + 9.0 * sin(0.7 * j + 0.3);
        }
        for (int i = 0; i < n; ++i) P[i * n + i] = 0.02;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) P[i * n + j] += 1e-3 * sin(0.37 * (i + 1)) * sin(0.37 * (j + 1));     /* rank-1, symmetric */
        double z[2 * NZ];
        for (int q = 0; q < NZ; ++q) {
            const int j = (q * 2) % N;
            const double dx = x[3 + 2 * j] - x[0], dy = x[4 + 2 * j] - x[1];
            z[2 * q] = sqrt(dx * dx + dy * dy) + (q == 3 ? 40.0 : 0.03);                 /* the last one matches nothing */
            z[2 * q + 1] = atan2(dy, dx) - x[2] + 0.002;
        }
        int32_t a_sweep[NZ], a_grid[NZ];
        int64_t info[8];
        OK(upload(h, dtype, x, P, n));
        OK(slam_ekf_set_gate_mode(h, SLAM_GATE_SWEEP));
        OK(slam_ekf_associate(h, z, NZ, R, 4.0, 25.0, a_sweep));
        OK(slam_ekf_set_gate_mode(h, SLAM_GATE_GRID));
        OK(slam_ekf_associate(h, z, NZ, R, 4.0, 25.0, a_grid));
        OK(slam_ekf_gate_info(h, info));
        CHECK(info[0] == SLAM_GATE_GRID && info[2] == N, "gate_info form %lld in_grid %lld", (long long)info[0], (long long)info[2]);
        for (int q = 0; q < NZ; ++q) CHECK(a_sweep[q] == a_grid[q], "gating forms differ at %d: %d vs %d", q, a_sweep[q], a_grid[q]);
        CHECK(a_sweep[0] == 1 && a_sweep[1] == 3 && a_sweep[2] == 5 && a_sweep[3] == -1, "associations %d %d %d %d", a_sweep[0], a_sweep[1],
              a_sweep[2], a_sweep[3]);
        CHECK(slam_ekf_set_gate_mode(h, 9) == SLAM_E_BADARG, "unknown gate mode accepted");
        OK(slam_ekf_set_gate_mode(h, SLAM_GATE_AUTO));
        /* the fused step: update with the three matches, the fourth observation becomes landmark 7 */
        int32_t a_obs[NZ];
        OK(slam_ekf_observe(h, z, NZ, R, 4.0, 25.0, SLAM_FORM_CHOLESKY, a_obs));
        OK(slam_ekf_sync(h));
        int Nn = -1;
        OK(slam_ekf_num_landmarks(h, &Nn));
        CHECK(Nn == N + 1, "after observe: N = %d", Nn);
        double pose[3];
        OK(slam_ekf_get_pose(h, pose));
        CHECK(fabs(pose[0] - 1.0) < 0.2 && fabs(pose[1] - 2.0) < 0.2, "pose after observe %g %g", pose[0], pose[1]);
    }
    OK(slam_ekf_destroy(h));
}

static void pf_step(void) {
    const double PI = 3.14159265358979323846;
    const double R[4] = {0.01, 0.0, 0.0, (PI / 180.0) * (PI / 180.0)};
    const double Q[4] = {0.25, 0.0, 0.0, (3.0 * PI / 180.0) * (3.0 * PI / 180.0)};
    slam_pf_t pf = NULL;
    OK(slam_pf_create(&pf, SLAM_F32, 4096, 4096, 0, 8, 0, 1234u));
    if (!pf) return;
    const double pose[3] = {0.0, 0.0, 0.3};
    double lm[16];
    for (int l = 0; l < 8; ++l) {
        lm[2 * l] = 20.0 * cos(0.8 * l);
        lm[2 * l + 1] = 20.0 * sin(0.8 * l);
    }
    OK(slam_pf_set_pose(pf, pose));
    OK(slam_pf_init_landmarks(pf, lm, 8, 0.01, 0.1));
    double z[4];
    const int32_t ids[2] = {1, 4};
    for (int q = 0; q < 2; ++q) {
        const double dx = lm[2 * (ids[q] - 1)], dy = lm[2 * (ids[q] - 1) + 1];
        z[2 * q] = sqrt(dx * dx + dy * dy);
        z[2 * q + 1] = atan2(dy, dx) - 0.3;
    }
    for (int s = 0; s < 3; ++s) OK(slam_pf_step_auto(pf, 1.0, 0.0, 4.0, Q, 0.025, z, ids, 2, R, 0.75, 0, 0));
    double out[4] = {0, 0, 0, 0};
    OK(slam_pf_flush(pf, out));
    CHECK(out[3] == 3.0 && out[0] > 1.0 && out[0] <= 4096.0, "flush: Neff %g, steps %g", out[0], out[3]);
    double sums[4];
    OK(slam_pf_mean_pose_sums(pf, sums));
    CHECK(sums[0] == sums[0], "mean pose sums are NaN");
    OK(slam_pf_destroy(pf));
}

int main(void) {
    if (slam_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", slam_last_error());
        return 2;
    }
    ekf_kats(SLAM_F64);
    ekf_kats(SLAM_F32);
    pf_step();
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("abi_client: all checks passed\n");
    return 0;
}
