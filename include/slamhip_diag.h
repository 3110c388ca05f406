/* slamhip_diag.h -- the MEASUREMENT and introspection entry points of libslamhip.so: event timing of the kernels, phase
 * stamps, the copy floor of the down-date, which form of the gating ran, what the exchange between the ranks of a sharded
 * filter saw.  Nothing here is part of the drop-in boundary (include/slamhip.h: what the reference's module surface maps
 * onto); bench.py, the tests and the profiling tools use them.  Same conventions: extern "C", int status codes. */
#ifndef SLAMHIP_DIAG_H
#define SLAMHIP_DIAG_H

#include "slamhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The diagonal of P (n values, handle dtype) -- a read-out for checks; slam_ekf_get_block / slam_ekf_get_landmark_blocks are the API. */
int slam_ekf_get_diag(slam_ekf_t h, void* out);

/* SLAM_F32 or SLAM_F64, as given to slam_ekf_create. */
int slam_ekf_dtype(slam_ekf_t h, int* dtype);

/* A/B knob of the gating: SLAM_GATE_AUTO (the sweep below 16384 landmarks, the grid from there on), SLAM_GATE_SWEEP, SLAM_GATE_GRID.
 * The decisions are the same in every mode (tested). */
int slam_ekf_set_gate_mode(slam_ekf_t h, int mode);

/* out = {form of the last gating (SLAM_GATE_SWEEP / _GRID), grid cells per axis, landmarks in the grid, its tail, rebuilds, queries,
 * landmarks visited, landmarks evaluated} (counters since create). */
int slam_ekf_gate_info(slam_ekf_t h, int64_t out[8]);

/* enable = 1: every kernel launch is bracketed by HIP events on the handle's
 * stream; enable = a mask of (2 << SLAM_K_x): only those kernels (an event pair costs
 * ~10 us of stream time, so a benchmark brackets the dominant kernel only); 0: off.
 * timing_read synchronises, folds the pending events into per-kernel
 * totals and returns total milliseconds and launch count for kernel id `kid`. */
int slam_ekf_timing(slam_ekf_t h, int enable);

int slam_ekf_timing_read(slam_ekf_t h, int kid, double* total_ms, int64_t* launches);

/* The fastest bracketed launch of kernel `kid` since the last reset, in milliseconds (0: none).  Synchronises. */
int slam_ekf_timing_min(slam_ekf_t h, int kid, double* min_ms);

/* out = {bracketed launches, their mean, sample standard deviation and minimum in milliseconds} since the last reset.  Synchronises. */
int slam_ekf_timing_stats(slam_ekf_t h, int kid, double out[4]);

int slam_ekf_timing_reset(slam_ekf_t h);

/* Diagnostics: when enabled the factorisation kernel records 100 MHz wall-clock stamps at its
 * phase boundaries; out16 (may be NULL) receives the stamps of the last update: [0..7] the workgroup
 * that factors S, [8..15] the first of the workgroups that form W1 in the same launch (zero when
 * the update took the two-launch form). */
int slam_ekf_debug_stamps(slam_ekf_t h, int enable, uint64_t* out16);

/* Measurement hook (bench.py: roofline.copy_floor_ms): the bare memory side of the covariance down-date (src/ekf.jl:75) on
 * THIS handle's matrix -- every stored tile the down-date touches read once and written back unchanged (bit-exact), in
 * the down-date's own band-major order, no panels, no matrix-core work; `reps` individually timed passes of each of two launch
 * forms.  out = {milliseconds of the FASTEST pass, its form's index (0: one workgroup per tile, 1: persistent grid)}.  The
 * down-date's launch time over this figure compares across the boxes of a pool whose memory systems differ by a few
 * per cent.  Synchronises; the state is unchanged. */
int slam_ekf_copy_floor(slam_ekf_t h, int reps, double out[2]);

/* The filter's HIP stream (interop: event timing around its kernels). */
int slam_pf_stream(slam_pf_t h, void** stream);

/* Collective: a barrier among the attached ranks through their inboxes; SLAM_OK when every peer's word arrived within
 * timeout_ms.  The caller's check, right after attaching, that the GPUs see each other's writes. */
int slam_pf_peer_selftest(slam_pf_t h, int timeout_ms);

/* out = {ranks, 1 if peers are attached, SLAM_PF_HALTED returns so far, resamplings so far}. */
int slam_pf_comm_info(slam_pf_t h, int64_t out[4]);

/* Diagnostics: 100 MHz wall-clock stamps of the last auto step: kernel start, every workgroup's statistics collected,
 * statistics folded, decision taken, bookkeeping done, published; [6] the collecting workgroup finished its own share,
 * [7] = [0] + 100 x the number of polls it needed.  Waits for the queue. */
int slam_pf_debug_stamps(slam_pf_t h, uint64_t out[8]);

#ifdef __cplusplus
}
#endif

#endif /* SLAMHIP_DIAG_H */
