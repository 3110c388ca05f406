/* slamhip.h -- C ABI of libslamhip.so: the MI355X (gfx950) EKF-SLAM / FastSLAM
 * filter core that replaces the hot path of andrewadare/SLAM.jl.
 *
 * The reference has no FFI/plugin interface: its hot path is four plain Julia
 * functions.  Each entry point below cites the reference
 * function (path:line relative to the SLAM.jl tree) whose work it takes over;
 * the Julia-side binding a maintainer would add is shown in INTEGRATION.md and
 * shipped as slam.jl_amd/SLAMHip.jl.
 *
 * Conventions
 *  - every function returns an int status (SLAM_OK = 0, negative = error) and
 *    never throws; slam_last_error() gives a thread-local message;
 *  - on error the filter state is unchanged;
 *  - the state (x, P) lives on the device for the life of the handle
 *    (P at N = 10k is 1.6 GB as a matrix: moving it per call would dwarf the update);
 *  - host arrays passed in are borrowed for the duration of the call only;
 *  - matrices are column-major (Julia order).  Small matrices Q, R are
 *    double[4] = {m11, m21, m12, m22}.  Observations are double pairs
 *    (range, bearing) = the memory order of Julia's 2 x nz matrix z;
 *  - landmark indices are 1-based like the reference's idf;
 *  - dtype selects the storage / bulk-arithmetic type of x and P (and of the
 *    host buffers of set_state/get_state); the scalar geometry (Jacobians,
 *    innovations, the k x k factorisation) is always evaluated in double;
 *  - one HIP stream per handle; calls on one handle must be serialised by the
 *    caller, distinct handles may be used from distinct threads.
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_OK            0
#define SLAM_E_BADARG     -1   /* null pointer, negative size, index out of range   */
#define SLAM_E_CAPACITY   -2   /* augment beyond max_landmarks (Julia: would grow)   */
#define SLAM_E_NOTPD      -3   /* innovation covariance not positive definite
                                  (Julia: chol throws PosDefException, ekf.jl:70)    */
#define SLAM_E_HIP        -4   /* HIP runtime error / no usable device               */
#define SLAM_E_NOMEM      -5   /* device or host allocation failed                   */

#define SLAM_PF_HALTED      1   /* slam_pf_step_auto / slam_pf_flush on a SHARDED filter: a queued step decided to
                                  resample; the caller exchanges weights and records, then slam_pf_resume           */

#define SLAM_F32 0
#define SLAM_F64 1

#define SLAM_FORM_CHOLESKY 0   /* P -= W1*W1'  (reference form, src/ekf.jl:67-75)    */
#define SLAM_FORM_JOSEPH   1   /* P -= K*T' + T*K', T = P*H' - K*S/2 (not in the
                                  reference; BASELINE.json config 5)                 */

/* kernel ids for slam_ekf_timing_read */
#define SLAM_K_GATE      0     /* gating sweep                                      */
#define SLAM_K_GATE_FIN  1     /* per-observation decision                          */
#define SLAM_K_PREDICT   2
#define SLAM_K_AUGMENT   3
#define SLAM_K_PHT       4     /* P*H' panel                                        */
#define SLAM_K_FACTOR    5     /* S, factorisation, C = inv(chol(S))                */
#define SLAM_K_W1        6     /* W1 = PHt*C, x += W*v                              */
#define SLAM_K_SYRK      7     /* P -= W1*W1' (or Joseph rank-2k)                   */
#define SLAM_K_COUNT     8

typedef struct slam_ekf* slam_ekf_t;

const char* slam_last_error(void);
/* Number of usable HIP devices (0 if none); never fails. */
int slam_device_count(void);

/* ---- EKFSlamState (src/common.jl:22-28): construction, I/O ------------------ */

/* Replaces `EKFSlamState(x, cov)` (src/common.jl:25-28; built at
 * sim/ekfslam-sim.jl:42).  State starts as x = 0 (3), P = 0 (3x3), 0 landmarks;
 * capacity for max_landmarks is allocated once (the reference re-allocates P per
 * new feature, src/ekf.jl:108-109). */
int slam_ekf_create(slam_ekf_t* h, int dtype, int max_landmarks, int device);
int slam_ekf_destroy(slam_ekf_t h);

/* Upload x (n) and P (n x n, leading dimension ldP >= n, column-major) from host
 * buffers of the handle's dtype; n = 3 + 2*N.  Replaces assigning state.x /
 * state.cov (sim/ekfslam-sim.jl:100,117,120; sim/browser/wsserver.jl:161-174). */
int slam_ekf_set_state(slam_ekf_t h, const void* x, const void* P, int n, int ldP);
/* Same, from DEVICE buffers on the handle's device (stream-ordered D2D copy). */
int slam_ekf_set_state_device(slam_ekf_t h, const void* d_x, const void* d_P, int n, int ldP);
/* Download; either pointer may be NULL.  Reads of state.x / state.cov.
 * Peak device memory of set_state (host source) / get_state: the state itself plus a staging buffer of at most
 * max(256 MiB, 128 columns) -- the matrix is repacked band by band, never through a second n x n copy. */
int slam_ekf_get_state(slam_ekf_t h, void* x, void* P, int n, int ldP);
/* state.cov[r0+1 : r0+nr, c0+1 : c0+nc] (0-based r0, c0 here) into a column-major host buffer of the handle's
 * dtype with leading dimension ld_out >= nr, and diag(state.cov) (n values): the way to look at parts of a
 * covariance that is too large to download (80 GB at N = 50k fp64; the reference ships the whole matrix per step,
 * sim/browser/wsserver.jl:36).  Elements come from the symmetric view, whichever triangle holds them. */
int slam_ekf_get_block(slam_ekf_t h, int r0, int c0, int nr, int nc, void* out, int ld_out);
/* The landmarks' 2 x 2 covariance blocks (what compute_association, src/data-association.jl:59, and the ellipses of
 * sim/browser/wsserver.jl:72-85 need of state.cov), packed: out[0][j] = P[f, f], out[1][j] = P[f+1, f], out[2][j] =
 * P[f+1, f+1], f = 3 + 2 j; three rows of N values in the handle's dtype.  Read from the side array the gating sweep
 * streams instead of gathering the matrix's diagonal (kept by every writer of these entries). */
int slam_ekf_get_landmark_blocks(slam_ekf_t h, void* out);
int slam_ekf_get_pose(slam_ekf_t h, double pose[3]);       /* state.x[1:3]          */
int slam_ekf_num_landmarks(slam_ekf_t h, int* N);          /* (length(x)-3)/2       */
/* Raw device views (for zero-copy interop, e.g. a torch tensor over x): d_x has 3+2*max_landmarks elements.
 * d_P is NOT a column-major matrix: the covariance is stored TILE-MAJOR, BLOCK LOWER -- only the square tiles (edge E =
 * 128 for fp32, 64 for fp64) on and below the diagonal exist, each one contiguous E x E column-major block, the tiles of
 * column band J one after the other (I = J, J+1, ..., T-1), band after band: tile (I, J) is block number
 * J*T - J*(J-1)/2 + (I - J) with T = ld / E (ld is returned for that purpose); element (r, c), r >= c tile-wise, sits at
 * block * E*E + (c % E) * E + (r % E).  Diagonal tiles are complete and symmetric: a caller who writes an off-diagonal
 * entry of a diagonal tile must write BOTH mirrored entries (r, c) and (c, r) -- the kernels read whichever is cheaper, and
 * slam_ekf_state_written takes a landmark's 2 x 2 block from the entries (f, f), (f + 1, f), (f + 1, f + 1), i.e. the LOWER one.
 * slam_ekf_get_state / slam_ekf_get_block return ordinary column-major data. */
int slam_ekf_device_ptrs(slam_ekf_t h, void** d_x, void** d_P, int* ld, void** stream);
/* The raw views are READ-ONLY as far as the landmarks go -- unless the caller says so afterwards.  The gating
 * (associate / observe, src/data-association.jl:21-63) does not read the landmarks' 2 x 2 covariance blocks from d_P but
 * from a packed side array that every writer INSIDE the library keeps, bounds the landmarks' variances with a value kept
 * on the device, and from 16384 landmarks on visits a grid of the landmark means built earlier: a caller who writes
 * landmark entries of x or P through the raw views (the reference's callers assign state.x / state.cov freely,
 * sim/ekfslam-sim.jl:100,117,120) must call slam_ekf_state_written before the next associate / observe, or the
 * decisions are made against the old values.  It rebuilds the side array from d_P, invalidates the variance bound and
 * forces a grid rebuild (all on the device, enqueued).  Writes of the POSE entries x[0:3] / P[0:3, 0:3] and of the
 * landmark-pose cross blocks need no call: the gating reads those from x and P. */
int slam_ekf_state_written(slam_ekf_t h);

/* ---- the hot path ----------------------------------------------------------- */

/* predict(state, vehicle, Q, dt)  src/ekf.jl:8-43.
 * (v, g, wheelbase) = vehicle.measured_speed, .measured_gamma, .wheelbase (:14-16).
 * In place, enqueued on the handle's stream (no host sync). */
int slam_ekf_predict(slam_ekf_t h, double v, double g, double wheelbase,
                     const double Q[4], double dt);

/* associate(state, z, R, gate1, gate2)  src/data-association.jl:1-51.
 * z: nz (range, bearing) pairs.  assoc[i] = j >= 1: observation i matched landmark
 * j (goes to zf/idf); 0: dropped; -1: new feature (goes to zn).  The reference's
 * (zf, idf, zn) are rebuilt from assoc in observation order by the host wrapper.
 * Synchronises (host output). */
int slam_ekf_associate(slam_ekf_t h, const double* z, int nz, const double R[4],
                       double gate1, double gate2, int32_t* assoc);

/* How associate / observe search the map -- the reference's TODO, src/data-association.jl:18-20 ("a quick bounding-box
 * threshold to remove distant features; or, better yet, a balanced k-d tree lookup").  Every form returns IDENTICAL
 * decisions; the choice is cost only.
 *   SLAM_GATE_SWEEP  every (observation, landmark) pair, O(N) per call (with the threshold pre-gate from 32768
 *                    landmarks on).
 *   SLAM_GATE_GRID   a uniform grid over the landmark means, kept on the device: an observation visits only the
 *                    landmarks of the cells its gate can reach -- O(candidates).  The grid follows the filter by itself
 *                    (updates record how far they moved a mean, add_features appends to a tail, a rebuild happens on
 *                    the device when either grows too large).  Needs R positive definite and gate1 <= gate2 < inf;
 *                    otherwise the call falls back to the sweep.  "Identical" presumes what the bound's proof does: every
 *                    landmark's S = H P H' + R positive definite AS COMPUTED (a landmark whose 2 x 2 block has lost its
 *                    definiteness to rounding -- condition above 1/eps of the dtype -- yields a negative nis in the
 *                    sweep, which the grid, like the threshold pre-gate, may not visit).
 *   SLAM_GATE_AUTO   (default) the grid from 16384 landmarks on (below that the two cost the same: 11-14 us).
 * slam_ekf_gate_info: out = {form of the last gating, cells per axis, landmarks in the grid, landmarks in the tail,
 * rebuilds, grid queries, landmarks visited, landmarks fully evaluated (the last four: totals since create)}.
 * Synchronises. */
#define SLAM_GATE_AUTO   0
#define SLAM_GATE_SWEEP  1
#define SLAM_GATE_GRID   2

/* compute_association(x, P, z, R, idf)  src/data-association.jl:53-63.
 * out = {nis, nd}.  Synchronises. */
int slam_ekf_nis(slam_ekf_t h, const double z1[2], int j, const double R[4], double out[2]);

/* predict_observation(x, idf)  src/common.jl:139-165.  zp = {range, bearing};
 * Hv = the 2x3 pose block H[:,1:3], Hf = the 2x2 block H[:,fpos:fpos+1], both
 * column-major; all other columns of the reference's dense H are zero.
 * Synchronises. */
int slam_ekf_predict_observation(slam_ekf_t h, int j, double zp[2], double Hv[6], double Hf[4]);

/* update(state, z, R, idf)  src/ekf.jl:46-77.  zf: m (range, bearing) pairs, idf: m
 * 1-based landmark indices (duplicates allowed, rows are stacked like the
 * reference).  m = 0 is a no-op.  In place on the device.  Returns SLAM_E_NOTPD
 * (state unchanged) if S is not positive definite; to report that, the call
 * synchronises unless slam_ekf_set_async(h, 1) was set, in which case the status
 * is deferred to slam_ekf_sync(). */
int slam_ekf_update(slam_ekf_t h, const double* zf, const int32_t* idf, int m,
                    const double R[4], int form);

/* add_features(state, z, R)  src/ekf.jl:84-122.  zn: nn (range, bearing) pairs.
 * Returns SLAM_E_CAPACITY (state unchanged) if N + nn > max_landmarks.  Enqueued. */
int slam_ekf_augment(slam_ekf_t h, const double* zn, int nn, const double R[4]);

/* One observation step of sim!  (sim/ekfslam-sim.jl:114-120):  associate -> update -> add_features
 * in ONE call, same results as the three calls above in sequence.  The association
 * vector (meaning as in slam_ekf_associate) is compacted into the update's inputs on
 * the device and the update kernels read the matched count from device memory, so
 * they are queued behind the gating without a host round trip; the host only waits
 * for assoc[] (to learn how many features to append).  S not positive definite:
 * as slam_ekf_update (the features are still appended).  SLAM_E_CAPACITY: the
 * update was applied, no feature was added. */
int slam_ekf_observe(slam_ekf_t h, const double* z, int nz, const double R[4],
                     double gate1, double gate2, int form, int32_t* assoc);

/* Telemetry without downloading P: feature_ellipses(x, cov) (sim/browser/wsserver.jl:72-85) into
 * `features` (5 x N column-major: cx, cy, rx, ry, phi; may be NULL) and the vehicle-position ellipse of
 * monitor() (:60-65) into `vehicle` = {cx, cy, vehicle_phi, rx, ry, phi} (may be NULL).  rx <= ry are the
 * square roots of the ascending eigenvalues of the 2 x 2 block, phi the direction of the FIRST eigenvector
 * with its sign fixed so that phi lies in [-pi/2, pi/2] (LAPACK's sign in the reference is arbitrary).
 * Synchronises. */
int slam_ekf_ellipses(slam_ekf_t h, double* features, double vehicle[6]);

/* ---- stream / timing -------------------------------------------------------- */

int slam_ekf_set_async(slam_ekf_t h, int async_updates);
/* Wait for the handle's stream; returns the first deferred error (and clears it). */
int slam_ekf_sync(slam_ekf_t h);

/* ---- FastSLAM-1.0 particle path (known correspondences) ----------------------------
 *
 * The reference implements no particle filter: only the types Particle (src/common.jl:14-20)
 * and PFSlamState (src/common.jl:31-34) exist and README.md:6 says "FastSLAM is ongoing".
 * These entry points take over what a PFSlamState-based predict/update would do, as specified
 * in SURVEY.md 8a rows F1-F4 from the reference's EKF building blocks (each function cites them).
 *
 * One handle owns the GLOBAL particle ids [first_id, first_id + n_local) of a filter with
 * n_global particles (one process per GPU).  Random numbers are Philox4x32-10 keyed by
 * (seed, step, global id), so results do not depend on the split.  Collectives (three scalars
 * per step; all log-weights and the migrating particle records on a resampling step) are
 * issued by the host between these calls (torch.distributed over RCCL): see slam.jl_amd/pf.py.
 * Limits: n_global < 2^31; one landmark's five rows of a shard (5 * n_local values) must fit a 4 GiB buffer descriptor,
 * i.e. n_local < 2^32 / (5 * sizeof(T)) -- 214 M particles in fp32, 107 M in fp64 (SLAM_E_BADARG otherwise). */
typedef struct slam_pf* slam_pf_t;

int slam_pf_create(slam_pf_t* h, int dtype, int64_t n_local, int64_t n_global, int64_t first_id,
                   int max_landmarks, int device, uint64_t seed);
int slam_pf_destroy(slam_pf_t h);
/* Every particle at `pose`, weights uniform (1 / n_global).  Particle.pose, src/common.jl:15. */
int slam_pf_set_pose(slam_pf_t h, const double pose[3]);
/* Landmarks 1..nl known to every particle at lm_xy + N(0, jitter_sigma^2) with covariance
 * diag(var, var) (the synthetic start of BASELINE.json config 4).  Particle.features / .fcov. */
int slam_pf_init_landmarks(slam_pf_t h, const double* lm_xy, int nl, double var, double jitter_sigma);
/* F1: V, G perturbed per particle like add_control_noise! (sim/sim-utils.jl:35-38), then the
 * pose update of predict (src/ekf.jl:39-41).  Enqueued. */
int slam_pf_predict(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt);
/* F2/F3: m (range, bearing) pairs with KNOWN 1-based landmark ids.  Seen landmark: the 2x2
 * EKF update (src/common.jl:162 + src/ekf.jl:67-75 on the feature block) and
 * logw += log N(v; 0, S).  First sighting: initialisation like add_features
 * (src/ekf.jl:94-103,112) without the vehicle-covariance term.  Enqueued. */
int slam_pf_update_known(slam_pf_t h, const double* z, const int32_t* ids, int m, const double R[4]);
/* SURVEY 8f N4 (no reference code): FastSLAM-1.0 with UNKNOWN correspondences.  slam_pf_clear_landmarks marks
 * every landmark slot of every particle unused (Pxx = -1).  slam_pf_update_unknown: m <= 16 (range, bearing) pairs;
 * every particle associates them with its OWN landmarks by the gated nearest-neighbour rule of associate()
 * (src/data-association.jl:1-51) with compute_association (:53-63) on the landmark's 2 x 2 block, all against the
 * map before this step's updates; then, in observation order, matched landmarks get the update of
 * slam_pf_update_known and new ones start in the particle's lowest unused slot (none left: dropped).  d_assoc:
 * DEVICE pointer to m * n int32 ([m][n]) or NULL; slot >= 0 matched, -1 new, -2 dropped.  Enqueued. */
int slam_pf_clear_landmarks(slam_pf_t h);
int slam_pf_update_unknown(slam_pf_t h, const double* z, int m, const double R[4], double gate1, double gate2,
                           int32_t* d_assoc);
/* F1 + F2/F3 + the local part of F4 as ONE sweep over the particles: slam_pf_predict, slam_pf_update_known
 * and slam_pf_weight_stats in one kernel (same particles bit for bit; out as slam_pf_weight_stats).  Synchronises. */
int slam_pf_step(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt,
                 const double* z, const int32_t* ids, int m, const double R[4], double out[3]);
/* SURVEY 8f N4 (no reference code): the FastSLAM-2.0 step -- slam_pf_step with the pose drawn from the proposal that
 * already knows this step's observations (Montemerlo et al. 2003).  The proposal lives in control space: pose =
 * f(pose, V + u0, G + u1) (src/ekf.jl:39-41) with u = chol(Q) w, w ~ N(0, I) a priori; every observation of a
 * landmark the particle already holds is assimilated as a linear 2 x 2 measurement of w (Jacobians
 * src/common.jl:161-162 and src/ekf.jl:27-29, Cholesky form of src/ekf.jl:67-75) and its predictive density goes
 * into the weight; w is then sampled with the same two normals slam_pf_step uses, and the landmarks are updated
 * from the sampled pose with the weight left alone.  Q: any symmetric positive definite 2 x 2 matrix.  With
 * m == 0 it is slam_pf_step bit for bit.  Same statistics in out.  Synchronises. */
int slam_pf_step_proposal(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt,
                          const double* z, const int32_t* ids, int m, const double R[4], double out[3]);
/* slam_pf_step + slam_pf_normalize with the shard's own statistics, for a filter on ONE GPU (n == n_global):
 * one library call per filter step.  out = {max logw, sum, sum2, Neff}. */
int slam_pf_step_normalized(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt,
                            const double* z, const int32_t* ids, int m, const double R[4], double out[4]);
/* F4, local part: out = {max logw, sum exp(logw - max), sum exp(2 (logw - max))}.  Synchronises. */
int slam_pf_weight_stats(slam_pf_t h, double out[3]);
/* logw -= gmax + log(gsum) with the GLOBAL max / sum (after the all-reduce). */
int slam_pf_normalize(slam_pf_t h, double gmax, double gsum);
/* Copy the local log-weights (handle dtype, n_local values) into a DEVICE buffer, e.g. this
 * rank's slice of the all-gather input.  Synchronises. */
int slam_pf_copy_logw(slam_pf_t h, void* d_dst);
/* Systematic resampling over the GLOBAL weights: d_logw_all holds all n_global log-weights
 * (device, handle dtype), gmax their maximum, u0 in [0,1) the shared offset.  d_anc (device,
 * n_local int32) receives the global ancestor id of every local slot.  Synchronises. */
int slam_pf_ancestors(slam_pf_t h, const void* d_logw_all, double gmax, double u0, int32_t* d_anc);
/* The same for EVERY slot of the filter: d_anc_all (device, n_global int32).  All ranks compute the identical table,
 * so each rank knows which of its particles every other rank needs and the record exchange is ONE all-to-all with
 * no request round.  Synchronises. */
int slam_pf_ancestors_all(slam_pf_t h, const void* d_logw_all, double gmax, double u0, int32_t* d_anc_all);
/* Resampling of a filter that lives WHOLLY on this shard (n_local == n_global) as one call: cdf of the weights,
 * systematic-resampling ancestors (offset u0, gmax = the maximum log-weight), then the LAZY step: poses are permuted
 * and small ancestor tables composed, the particles' maps stay where they are and move landmark by landmark when
 * next updated (csrc/pf_legacy.hip, "lazy resampling"; SLAMHIP_PF_EAGER=1 or an exhausted table pool: the eager gather of
 * whole records).  Same particles, bit for bit, as slam_pf_copy_logw + slam_pf_ancestors + slam_pf_resample_apply.
 * Enqueued. */
int slam_pf_resample_local(slam_pf_t h, double gmax, double u0);
/* Rows of one particle record: 3 pose rows + 5 rows per landmark (x, y, Pxx, Pxy, Pyy). */
int slam_pf_record_rows(slam_pf_t h, int* rows);
/* records[row][c] = state row of local particle d_local_idx[c]  (device buffers, handle dtype):
 * what a rank sends to the ranks whose slots descend from its particles.  Synchronises. */
int slam_pf_pack(slam_pf_t h, const int32_t* d_local_idx, int cnt, void* d_records);
/* Replace every local particle by its ancestor: local ancestors are gathered from the
 * resident state, remote ones from d_remote_records ([rows][nremote], columns in the order
 * of the ascending global ids d_remote_ids).  Weights return to 1 / n_global.  Synchronises. */
int slam_pf_resample_apply(slam_pf_t h, const int32_t* d_anc, const int32_t* d_remote_ids, int nremote,
                           const void* d_remote_records);
/* out = {sum w x, sum w y, sum w sin(phi), sum w cos(phi)} over the local particles, w = exp(logw). */
int slam_pf_mean_pose_sums(slam_pf_t h, double out[4]);
/* Download (host buffers, handle dtype; any may be NULL): pose [3][n], logw [n], lm [nl][5][n]. */
int slam_pf_download(slam_pf_t h, void* pose, void* logw, void* lm);
int slam_pf_sync(slam_pf_t h);

/* ---- the filter step without the host in the loop ("auto mode") -------------------------------------------------
 * slam_pf_step_auto: one whole filter step -- slam_pf_step (proposal = 0) or slam_pf_step_proposal (1), the
 * normalisation, Neff, the decision to resample (force < 0: Neff < neff_frac * n_global; 0 / 1: never / always) and, for
 * a filter that lives wholly on this shard, the resampling itself (slam_pf_resample_local) -- ENQUEUED: the call
 * returns at once, steps queue back to back, the statistics, the decision and the bookkeeping of the lazy resampling
 * stay on the device.  m <= 64.  Same particles as the synchronous calls.  The other slam_pf_* entry points may be
 * mixed in freely (they wait for the queue first).
 * Sharded filter (slam_pf_attach_exchange): the ranks' GPUs exchange their three scalars per step through a shared
 * page of pinned host memory, so steps that do not resample need no host either.  A step that does resample needs the
 * all-gather of the log-weights and the record exchange, which the caller issues (RCCL): that step HALTS, the steps
 * queued behind it are skipped on the device, and the next slam_pf_step_auto / slam_pf_flush returns SLAM_PF_HALTED
 * (nothing enqueued by that call).  The caller then resamples with slam_pf_halt_info + the legacy entry points
 * (slam_pf_copy_logw, slam_pf_ancestors_all, slam_pf_pack, slam_pf_resample_apply), calls slam_pf_resume -- the
 * skipped steps are enqueued again from the library's log -- and repeats the call. */
int slam_pf_step_auto(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt, const double* z,
                      const int32_t* ids, int m, const double R[4], double neff_frac, int force, int proposal);
/* K consecutive slam_pf_step_auto calls in ONE call -- the same filter afterwards, bit for bit.  Step k: control
 * (VG[2k], VG[2k+1]) = (V, G); its m[k] observations are the (range, bearing) pairs at z + 2 zstride k, their landmark
 * ids at ids + zstride k (m[k] <= zstride); force[k] as slam_pf_step_auto's force (force == NULL: the Neff rule at every
 * step).  wheelbase, Q, dt, R, neff_frac, proposal are common to the K steps.
 * Runs of at least four consecutive steps that CANNOT resample (force[k] == 0) go -- where the caller allows it (flags bit 1)
 * and the filter does: fp32, the whole filter on this shard, proposal = 0, m[k] <= 32, at most 2048 landmarks and 1024
 * particles per compute unit of the device -- as persistent launches of up to 16 steps (csrc/pf_batch.hip): poses and
 * weights stay in registers between the steps, every workgroup reduces the step's statistics itself, the statistics tail of
 * a step runs under the next step's sweep.  The persistent grid takes every compute unit whole and its workgroups wait for
 * each other: the caller sets flags bit 1 only when nothing else (another filter's kernels, another process) keeps the
 * device's compute units busy meanwhile -- a grid that cannot become co-resident gives up after 2 s and the filter is dead.
 * Every other step is enqueued as slam_pf_step_auto enqueues it.
 * *enqueued (may be NULL): the steps taken -- less than K only together with SLAM_PF_HALTED (sharded halting flow: resolve
 * the halt, call again with the remaining steps). */
int slam_pf_step_auto_batch(slam_pf_t h, int K, const double* VG, double wheelbase, const double Q[4], double dt,
                            const double* z, const int32_t* ids, const int32_t* m, int zstride, const double R[4],
                            double neff_frac, const int32_t* force, int proposal, int flags, int* enqueued);
/* Wait for everything queued.  out (may be NULL) = {Neff of the last step, 1 if it resampled, resamplings so far,
 * steps so far}.  (A queued step reports to the host only every eighth step, when it halts or fails; this call asks the
 * device for the last step's outcome.) */
int slam_pf_flush(slam_pf_t h, double out[4]);
int slam_pf_halt_info(slam_pf_t h, double out[2]);      /* {largest normalised log-weight, resamplings so far}        */
int slam_pf_resume(slam_pf_t h, int64_t resamplings);   /* resamplings: the caller's count after its own resampling   */
/* The systematic-resampling offset of resampling k is Philox(counter (0, 0, k, 2), key seed): the count is part of the
 * filter state (slam.jl_amd/pf.py: FastSLAM.resamples). */
int slam_pf_resample_count(slam_pf_t h, int64_t* count);        /* (waits for the queue, like slam_pf_flush) */
int slam_pf_set_resample_count(slam_pf_t h, int64_t count);
/* The ranks' shared scalar page (host memory every rank has mapped, >= 2 * world * 64 bytes, zeroed): see above.
 * (The legacy form of the per-step exchange; with slam_pf_attach_peers the scalars travel GPU to GPU.) */
int slam_pf_attach_exchange(slam_pf_t h, int rank, int world, void* page, size_t bytes);

/* ---- sharding behind the C ABI: peers (SURVEY 8b `n_devices`, 8e) --------------------------------------------------
 * The particle types the reference declares (Particle / PFSlamState, src/common.jl:14-20,31-34) say nothing about
 * devices; SURVEY 8e: one process per GPU, rank r owns the global particle ids [r n, (r + 1) n).  The library links no
 * collective library: at set-up every rank exports ONE blob (SLAM_PF_PEER_BLOB_BYTES: hipIpc handles of its pose /
 * landmark / log-weight / ancestor-table buffers and of an inbox page; raw pointers for shards of the same process),
 * the CALLER moves the blobs between the ranks (MPI_Allgather, files, torch.distributed -- 4 KB per rank, once), and
 * every rank attaches all `world` blobs in rank order (<= 8 ranks, equal slices).  After that
 *   - the per-step weight statistics travel as one record {max, sum w, sum w^2} per 1024 particles, written by each rank's
 *     GPU into every peer's inbox over xGMI and polled in LOCAL device memory (no host page, no PCIe); every rank reduces
 *     the same records along the same fixed tree over the global particle index, so the normalised log-weights are
 *     bit-identical to a one-rank filter's whenever a rank's slice is a multiple of 1024 particles;
 *   - a step that resamples does so ON THE DEVICE, like a one-GPU filter: the all-gather of the log-weights is the
 *     scan kernel's loads from the peers' buffers, remote ancestors' poses and ancestor-table entries are read from
 *     their owners, and the particles' MAPS do not move at all -- a table entry is a global particle id and a remote
 *     ancestor's landmark record is read from its owner when that landmark is next updated.  slam_pf_step_auto never
 *     returns SLAM_PF_HALTED for this reason (only when the ancestor-table pool is exhausted, or SLAMHIP_PF_EAGER=1).
 * Entry points that need plain maps (slam_pf_download with landmarks, slam_pf_pack, slam_pf_resample_apply, the legacy
 * sweeps slam_pf_update_known / slam_pf_step / slam_pf_step_proposal, slam_pf_update_unknown, slam_pf_detach_peers) are
 * COLLECTIVE while peers are attached: every rank must call them in the same order (they first bring remote records
 * home, with barriers among the ranks' streams).  Detach (collectively) before any rank destroys its handle; a handle
 * destroyed while attached first tells its peers (their queued steps then fail with a "peer is gone" error instead of
 * reading freed memory -- best effort).
 * Limits of the IPC mappings on ROCm 7.2 (both checked BEFORE anything is opened; slam_pf_attach_peers returns
 * SLAM_E_CAPACITY and the caller keeps the halting flow -- slam.jl_amd/pf.py does that by itself): no exported buffer of a
 * peer in ANOTHER process may exceed 2047 MiB (a filter with a larger one hung in its attach flow, cause unknown; the landmark records
 * are therefore kept in chunks of at most 1 GiB, so 262144 particles x 512 landmarks per rank attach), and the inbox (the only
 * fine-grained export) must stay within 2 MiB, i.e. n_global <= 32 M particles.
 * slam_pf_comm_info: out = {ranks, 1 if peers are attached, SLAM_PF_HALTED returns so far, resamplings so far}. */
#define SLAM_PF_PEER_BLOB_BYTES 4096
int slam_pf_export_peer(slam_pf_t h, void* blob);
int slam_pf_attach_peers(slam_pf_t h, int rank, int world, const void* blobs);
int slam_pf_detach_peers(slam_pf_t h);

/* ---- SURVEY 8b's whole-filter calls (filter wholly on this shard) --------------------------------------------------
 * slam_pf_resample: F4 -- normalise, and resample (systematic, slam_pf_resample_local) if Neff < neff_frac * n;
 * *resampled (may be NULL) tells whether it did.  slam_pf_get_mean_pose: weighted mean [x, y, phi].
 * slam_pf_get_weights: w = exp(logw) of the local particles (double[n_local]).  Particle.weight, src/common.jl:19. */
int slam_pf_resample(slam_pf_t h, double neff_frac, int* resampled);
int slam_pf_get_mean_pose(slam_pf_t h, double pose[3]);
int slam_pf_get_weights(slam_pf_t h, double* w);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
