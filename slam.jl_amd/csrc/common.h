// common.h -- internal declarations shared by the HIP sources of libslamhip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../../include/slamhip.h"
#include "../../include/slamhip_diag.h"

#define SLAM_PI 3.14159265358979323846

// ---- error plumbing ---------------------------------------------------------
void slam_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            slam_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                           __FILE__, __LINE__);                                        \
            return (_e == hipErrorOutOfMemory) ? SLAM_E_NOMEM : SLAM_E_HIP;            \
        }                                                                              \
    } while (0)

#define ARG_CHECK(cond, msg)                                                           \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            slam_set_error("bad argument: %s", msg);                                   \
            return SLAM_E_BADARG;                                                      \
        }                                                                              \
    } while (0)

// ---- roctx ranges around the C-ABI calls (tracing) -------------------
// rocprofv3 --marker-trace shows one range per library call, named after the entry point.  The marker library
// (librocprofiler-sdk-roctx / libroctx64) is looked up with dlopen at the first call -- already loaded by the tool or
// the application, or loaded here when SLAMHIP_ROCTX=1 -- so libslamhip.so itself still depends on libamdhip64 only;
// without it the ranges cost one predictable branch.
void slam_roctx_push(const char* name);
void slam_roctx_pop(void);
struct SlamRange {
    explicit SlamRange(const char* name) { slam_roctx_push(name); }
    ~SlamRange() { slam_roctx_pop(); }
    SlamRange(const SlamRange&) = delete;
    SlamRange& operator=(const SlamRange&) = delete;
};
#define SLAM_RANGE() SlamRange slam_range_scope_(__func__)

// ---- geometry of the buffers -------------------------------------------------
// Covariance tile edge of the rank-k down-date; the panel buffers (PHt, W1, ...)
// are padded to whole tiles with zero rows so the down-date needs no row guards.
#define SLAM_TILE 128
// k = 2m is padded to a multiple of SLAM_KPAD with zero columns.
#define SLAM_KPAD 32

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// A/B knobs of the measurement tooling (tile order, workgroups per list, super-row height, the scalar factorisation ...):
// read from the environment ONLY in the experiments build (`make exp`, libslamhip_exp.so); the product library always
// takes the default, so no untested code path can be switched on from outside.  What the product library does read:
// SLAMHIP_X bits 8 / 32 / 64 / 512 (fp32 matrix cores instead of the split-bf16 down-date; pre-gate never / always; round 3's
// register-staged chunk pipeline instead of the LDS-DMA one) and SLAMHIP_PF_EAGER -- each has a test of its own.
#ifdef SLAMHIP_EXPERIMENTS
#include <stdlib.h>
static inline int slam_exp_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
constexpr int SLAM_XFLAGS_MASK = ~0;
#else
static inline int slam_exp_env(const char*, int dflt) { return dflt; }
constexpr int SLAM_XFLAGS_MASK = 8 | 32 | 64 | 128 | 512;
#endif

struct TimingPair {
    hipEvent_t a, b;
    int kid;
};

struct slam_ekf {
    int dtype;        // SLAM_F32 / SLAM_F64
    int device;
    int num_cus;      // compute units of the device
    int maxN;         // landmark capacity
    int N;            // landmarks in the map
    int ncap;         // 3 + 2*maxN
    int ld;           // = npad: carries the tile-row count of P's allocation (T = ld / E) and is the panels' row count
    int npad;         // rows of the panel buffers, multiple of SLAM_TILE
    size_t esz;       // element size
    void* x;          // [ncap]
    void* P;          // tile-major, block lower (device_math.h): T(T+1)/2 tiles of E x E, T = npad / E; rows/cols >= n are zero padding
    void* Pside;      // [3][npad / 2]: the landmarks' 2 x 2 diagonal blocks, packed (device_math.h: side_note)
    hipStream_t stream;
    double* PHtS;          // compact panel [3 + kcap][kcap]: the rows of P*H' the factorisation needs
    hipEvent_t stage_ev;   // marks the last H2D copy out of the pinned staging buffers
    int stage_pending;

    // update workspace, (re)allocated when k grows
    int kcap;         // padded k capacity (multiple of SLAM_KPAD)
    double* PHt;      // [npad][kcap]   row-major, double
    double* Kd;       // [npad][kcap]   row-major, double (Joseph: K = PHt*inv(S))
    void* W1;         // [npad][2*kcap] row-major, dtype (Joseph uses both halves: [K|T])
    void* W2;         // [npad][2*kcap] row-major, dtype (Joseph: [T|K])
    void* Wimg;       // fp32 handles: W1 split into bf16 (h, m, l), stored as the LDS image of the split-bf16 down-date:
                      // [npad/128 row blocks][kcap/16 chunks][3 splits][128 rows][16 bf16, 16-byte halves swizzled]
    double* Cmat;     // [kcap][kcap]   row-major, double (C = inv(chol(S)) upper, or inv(S) for Joseph)
    double* Smat;     // [kcap][kcap]   double (Joseph: S);  also global scratch for big k
    double* Mwork;    // [kcap][kcap+1] double, factor scratch when it does not fit LDS
    double* gvec;     // [kcap]  g = C*C'*v  (x += PHt*g)
    double* obsbuf;   // [ocap][2]  observations on device
    int32_t* idfbuf;  // [ocap]
    int ocap;
    double* h_obs;    // pinned staging
    int32_t* h_idf;   // pinned staging
    int32_t* h_assoc; // pinned staging [ocap]
    int32_t* d_assoc; // [ocap]
    double* h_obs_dev;     // device-side addresses of the pinned h_obs / h_assoc (observe(): zero-copy staging)
    int32_t* h_assoc_dev;
    double* znbuf;    // [2*ocap] observe(): the new-feature observations, compacted on the device
    int32_t* d_count; // [4]      observe(): {matched m, new nn}
    int32_t* h_flag;       // pinned: observe() polls it for obs_seq (written by the compaction after the decisions)
    int32_t* h_flag_dev;
    int32_t obs_seq;

    // down-date tile order (ekf_syrk.hip): workgroup b computes tile tiles[b]
    int2* tiles;         // two orders in one buffer: the super-row order (8 lists of tiles_len), then the band-major
    int tiles_T, tiles_len, tiles_cap;     // order of the split-bf16 path (8 lists of tilesB_len, at offset tilesB_off)
    int tilesB_off, tilesB_len;
    int tilesH_off, tilesH_len;            // (experiments build: the half-tile experiment's lists)
    unsigned* dd_claim;  // [8][16] per-XCD tile counters of the claiming down-date launch (zeroed before every launch)
    int tiles_xlen[8];   // valid entries of each XCD's list
    int diag_off, diag_len, diag_xlen[8];   // fp32: the diagonal tiles, listed after the main lists

    // N2 pre-gate: upper bound of the landmarks' variances (ekf_gate.hip)
    double* d_pmax;      // device double, >= max diag(P_ff)
    int pmax_valid;      // 0: recompute before the next sweep (upload, Joseph-form update)

    // N2, the O(candidates) form: a uniform grid over the landmark means (ekf_gate.hip)
    int gate_mode;       // SLAM_GATE_AUTO / SLAM_GATE_SWEEP / SLAM_GATE_GRID
    int gate_last;       // the form the last gating used
    void* grid_meta;     // device GridMeta (origin, cell size, drift bound, counters, the updates' displacement slots)
    int32_t* grid_cells; // [G*G + 1] first item of every cell
    void* grid_items;    // [maxN]    GridItem: landmark index + its mean at build time, sorted by cell
    int grid_force;      // rebuild at the next query (state upload, too many updates between two queries)
    int grid_upd;        // updates enqueued since the last query (their displacement bounds: GridMeta::slot[0 .. grid_upd))
    int grid_live;       // a grid exists: updates record their displacement bounds
    int grid_n_seen;     // N when the fold / rebuild check was last enqueued

    // gating partials
    double* gate_part;   // [gate_blocks][ocap][3]
    int gate_blocks_cap;

    // small device scratch + pinned mirror for scalar outputs
    double* d_small;     // 64 doubles: [0..11] small results, [40..55] debug stamps, [56] the ready word factor_w1_kernel's workgroups meet on
    double* h_small;     // pinned, 64 doubles
    int32_t* d_status;   // [4]  [0] = not-PD flag of the last update
    int32_t* h_status;   // pinned

    int xflags;          // SLAMHIP_X: experiment switches of the production down-date (1: no start stagger, 2: no s_setprio around the MFMAs, 4: no streaming path, 8: no split-bf16 path -- the fp32 matrix cores do the down-date)
    int debug_flags;     // SLAMHIP_DEBUG env bits: 1 = no P stores, 2 = no MFMAs, 4 = no P loads (timing experiments, WRONG results)
    void* dd_prof;       // SLAMHIP_DEBUG & 8: per-wave phase clocks of the fp32 down-date (printed at destroy)
    int factor_blocked;  // K4: blocked MFMA elimination (default) or the scalar one (SLAMHIP_FACTOR=scalar)
    int debug_stamps;    // factor kernel writes 100 MHz wall-clock stamps into d_small[40..55]
    int async_updates;
    int deferred;        // first deferred error
    int pending_status;  // an update's status word has not been read back yet

    // timing
    int timing;
    std::vector<TimingPair> pairs;
    std::vector<TimingPair> free_pairs;
    double t_ms[SLAM_K_COUNT];
    double t_sq[SLAM_K_COUNT];    // sum of the squared launch durations (slam_ekf_timing_stats)
    double t_min[SLAM_K_COUNT];   // fastest bracketed launch since the last reset (0: none)
    int64_t t_n[SLAM_K_COUNT];
};

// RAII-less helper: bracket a launch with events when timing is on.
struct KTimer {
    slam_ekf* h;
    TimingPair p;
    bool on;
    hipStream_t s;
    KTimer(slam_ekf* h_, int kid, hipStream_t stream = nullptr);    // default: the handle's main stream
    ~KTimer();
};

// ---- kernel launchers (one per .hip file) -----------------------------------
int launch_ellipses(slam_ekf* h, double* d_out);     // [N + 1][5]: vehicle, then the landmarks
int launch_block_gather(slam_ekf* h, int r0, int c0, int nr, int nc, int diag, void* d_out);   // dense nr x nc copy of P[r0.., c0..] (symmetric view)
int launch_pack(slam_ekf* h, const void* d_src, int lds, int n, int cf, int ncols);    // columns [cf, cf + ncols) of the column-major matrix (d_src: that band) -> the tile-major state
int launch_unpack(slam_ekf* h, void* d_dst, int ldd, int n, int cf, int ncols);       // columns [cf, cf + ncols) of the full symmetric matrix -> d_dst (that band, column-major)
int launch_side_rebuild(slam_ekf* h);    // Pside <- the entries of the matrix itself (slam_ekf_state_written)
int launch_copy_floor(slam_ekf* h, int reps, double out[2]);   // bare read + rewrite of the stored tiles, timed (ekf_syrk.hip)
int launch_predict(slam_ekf* h, double v, double g, double w, const double Q[4], double dt);
int launch_augment(slam_ekf* h, int nn, const double R[4], const double* zn_dev);   // zn already on the device (obsbuf or znbuf)
int launch_gate(slam_ekf* h, int nz, const double R[4], double gate1, double gate2, const double* z_host, bool compact);
// z_src: device-readable (obsbuf or pinned host); compact (observe()): d_assoc -> idfbuf/obsbuf (matched, in order),
// znbuf (new), d_count = {m, nn}, h_assoc (pinned) -- done by the last gate_final launch
constexpr int SLAM_GRID_SLOTS = 64;    // updates whose displacement bounds the grid keeps apart; an update's bound is the max over
                                       // 16 words SLAM_GRID_SLOTS apart (ekf_gate.hip: GridMeta::slot[sub][update])
int gate_kernels_init();
int gate_info(slam_ekf* h, int64_t out[8]);
unsigned long long* grid_drift_slot(slam_ekf* h);   // where the update about to be enqueued records its largest landmark displacement (null: no grid)
int ensure_pmax(slam_ekf* h);       // the pre-gate's variance bound is current
int launch_nis(slam_ekf* h, const double z1[2], int j, const double R[4]);
int launch_obs_model(slam_ekf* h, int j);
int launch_update(slam_ekf* h, int m, const double R[4], int form, bool device_count);   // device_count: m is an upper bound, kernels read d_count[0]
int ensure_update_workspace(slam_ekf* h, int m);
int ensure_obs_capacity(slam_ekf* h, int nobs);
