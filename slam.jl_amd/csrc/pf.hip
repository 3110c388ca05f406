// pf.hip -- K9..K12: the FastSLAM-1.0 particle path (known correspondences) and its C ABI.
//
// The reference has NO particle-filter code, only the types Particle / PFSlamState
// (src/common.jl:14-20,31-34; README.md:6 "FastSLAM is ongoing").  The algorithm is the one
// specified in SURVEY.md 8a rows F1-F4 from the reference's EKF building blocks:
//   F1  control noise per particle (sim/sim-utils.jl:35-38) + pose update (src/ekf.jl:39-41)
//   F2  per-landmark 2x2 EKF: feature block of predict_observation (src/common.jl:162) and the
//       Cholesky-form update (src/ekf.jl:67-75) restricted to that block; w *= N(v; 0, S)
//   F3  new landmark (src/ekf.jl:94-103,112 without the vehicle term)
//   F4  normalisation, Neff, systematic resampling
//
// Layout (HBM): structure of arrays, particle index fastest --
//   pose[3][n], logw[n], lm[max_landmarks][5][n]  with 5 = (x, y, Pxx, Pxy, Pyy)
// so a known-correspondence update streams five fully coalesced rows per observed landmark.
// The reference's Particle type is an array of heap objects (layout hint only).
//
// Sharding: one handle owns the global particle ids [first, first + n).  Random numbers are
// Philox4x32-10 keyed by (seed, step, global id): results do not depend on the number of GPUs.
// The only cross-particle steps are three scalars per step (max, sum w, sum w^2) and, on a
// resampling step, the log-weights of all particles; both collectives are issued by the host
// (torch.distributed over RCCL), this library provides the local pieces.
#include <stdlib.h>
#include <unistd.h>

#include <vector>

#include "common.h"

#define PF_PI 3.14159265358979323846

// No FMA contraction in this file: the particle arithmetic is specified operation by operation (the oracle is
// NumPy, which rounds every operation), and the fused step kernel must reproduce the separate kernels bit for
// bit whatever the compiler would otherwise fuse across the predict/update boundary.  The kernels are HBM-bound.
#pragma clang fp contract(off)

// ---- auto mode: device-resident control block, its pinned mirror, and the host's log of queued steps ----------
constexpr int PF_CTL_TABS = 64;             // = PF_TAB_MAX (asserted below)
constexpr int PF_CTL_MAXOBS = 64;           // = PF_AUTO_MAXOBS

// ---- sharded filter: every rank's buffers as THIS rank's GPU addresses them (slam_pf_attach_peers) ---------------------
// One process per GPU; at attach time the ranks exchange IPC handles of their state buffers and of an "inbox" page, so every
// rank's kernels can read every peer's log-weights, poses, ancestor tables and landmark records over xGMI and WRITE into
// every peer's inbox (per-step scalars, hand-shake words): posted writes to the peer, polls of local memory.
constexpr int PF_MAX_WORLD = 8;

// ---- the landmark records: [landmark][5][n] in CHUNKS of whole landmarks ----------------------------------------------------
// One allocation per chunk of 2^shift landmarks, every chunk below 2 GiB: hipIpcOpenMemHandle of a larger allocation never
// returns on this runtime (ROCm 7.2, dmabuf IPC), and BASELINE.json's weak-scaling shape (262144 particles x
// 512 landmarks per rank) is 2.5 GiB per buffer.  A filter whose buffer stays below 1 GiB has ONE chunk (every shape of the
// fixed-size filter from two ranks on).  The table lives in device memory, is written once at create and is read through the
// constant address space (wave-uniform index: scalar loads).
constexpr int PF_LM_MAXC = 16;
struct PfLmTab {
    void* c[2][PF_LM_MAXC];      // [buffer][chunk]: the landmarks [k << shift, (k + 1) << shift), 5 rows of n values each
    int32_t shift, nchunks;
};
typedef const __attribute__((address_space(4))) PfLmTab* PfLmTabK;
template <typename T>
struct LmView {
    const PfLmTab* tab;
    // the five rows of landmark l in buffer `buf` (row k at + k n)
    __device__ __forceinline__ T* rows(int buf, int l, int64_t n) const {
        const PfLmTabK k = (PfLmTabK)tab;
        const int sh = k->shift;
        return (T*)k->c[buf][l >> sh] + (size_t)(l & ((1 << sh) - 1)) * 5 * (size_t)n;
    }
    // row `row` of the flat [5 nl][n] view of a buffer (the eager gather / pack kernels)
    __device__ __forceinline__ T* flat_row(int buf, int64_t row, int64_t n) const {
        const int l = (int)(row / 5), kk = (int)(row - 5 * (int64_t)l);
        return rows(buf, l, n) + (size_t)kk * (size_t)n;
    }
};

struct PfInbox {                 // lives in its owner's device memory; slot [r] is written by rank r (its own too)
    unsigned long long ready[PF_MAX_WORLD][8];       // [r][0]: last resampling step whose step kernel rank r has COMPLETED
    unsigned long long bar[PF_MAX_WORLD][8];         // [r][0]: rank r's count of peer barriers (materialise)
    unsigned long long gone[PF_MAX_WORLD][8];        // [r][0] != 0: rank r is destroying its handle -- its buffers are about to
                                                     // be freed; every kernel that would touch peer memory stops with PF_ERR_PEER
};
// Behind the header: the ranks' 1024-particle weight records of a step (the canonical tree's nodes, see WRec), two parities:
// double rec[2][rec_cap][4] = {m, s1, s2, tag}; record (rank r, local block j) sits at index r * ceil(n_local / 1024) + j and is
// written by rank r's step kernel into EVERY rank's inbox.  rec_cap = ceil(n_global / 1024) + PF_MAX_WORLD.
__host__ __device__ inline double* pf_inbox_recs(PfInbox* ib) { return reinterpret_cast<double*>(ib + 1); }
inline size_t pf_inbox_bytes(int64_t n_global) {
    return sizeof(PfInbox) + (size_t)2 * (size_t)((n_global + 1023) / 1024 + PF_MAX_WORLD) * 4 * sizeof(double);
}
struct PfPeers {                 // device memory of each rank, filled at attach time
    void* pose[PF_MAX_WORLD][2];
    PfLmTab lm[PF_MAX_WORLD];     // each rank's chunk table, as THIS GPU addresses the chunks
    void* logw[PF_MAX_WORLD][2];
    int32_t* tab[PF_MAX_WORLD][2];
    PfInbox* inbox[PF_MAX_WORLD];
};
constexpr int PF_ERR_HANDOVER = 1;   // a workgroup's statistics line never came (2 s)
constexpr int PF_ERR_EXCHANGE = 2;   // a rank's per-step scalars never came (20 s): a rank is gone
constexpr int PF_ERR_PEER = 3;       // a peer hand-shake (resampling / materialise barrier) timed out

struct PfCtl {                   // device memory; written by the LAST workgroup of a step kernel, read by later kernels
    double shift_next;           // normalisation shift the next kernel that reads logw subtracts on the way
    double shift_scan;           // shift of the step that decided to resample (the cdf is formed through it)
    double gmax_norm;            // largest normalised log-weight of that step, as the storage type holds it
    double u0;                   // systematic-resampling offset of that step
    double stats[8];             // {max, sum w, sum w^2, 0, 0, 0, 0, Neff}, w = exp(logw - ceil(max / ln 2) ln 2) (see WRec)
    long long seq;               // last completed step
    long long resample_seq;      // the step whose (lazy) resampling the conditional kernels apply
    long long halt_seq;          // != 0: that step wants a resampling the device cannot do; later steps are skipped
    int32_t arrive;              // (unused: the first form of the hand-over counted arrivals here)
    int32_t error;               // PF_ERR_*: the filter is dead, every later kernel returns at once
    int32_t nresamples;          // resamplings so far
    int32_t pcur, tside;         // live pose buffer / ancestor-table side
    int32_t lwcur;               // live log-weight buffer (flips with every device-side resampling: the peers of a sharded
                                 // filter still read the old weights while this rank already writes the uniform ones)
    int32_t identity;            // landmarks without an ancestor table
    int32_t tl_count, tl_fresh;  // the pending lazy resampling: live tables to compose, index of the fresh one (-1: none)
    int32_t tl_idx[PF_CTL_TABS];
    int32_t tref[PF_CTL_TABS];   // landmarks referring to each table
    unsigned long long stamps[8];    // diagnostics: 100 MHz wall-clock stamps of the last step (kernel start, tail phases)
};

struct PfMirror {                // pinned host memory, written with system-scope stores: read by the host without a sync
    long long done_seq;          // last completed (not skipped) step
    long long halt_seq;
    long long resampled_seq;     // last step that resampled (on the device)
    long long nresamples;
    double neff;
    double stats[8];
    long long error;
};

struct PfStepRec {               // one queued slam_pf_step_auto call, kept until the device confirms it
    long long seq;
    uint32_t rng_step;
    int m, force, proposal;
    double V, G, wheelbase, Q[4], dt, R[4], neff_frac;
    double z[2 * PF_CTL_MAXOBS];
    int32_t ids[PF_CTL_MAXOBS];
};

struct slam_pf {
    int dtype, device;
    size_t esz;
    int64_t n, n_global, first;
    int nl;
    uint64_t seed;
    uint32_t step;
    hipStream_t stream;
    void* pose[2];       // [3][n]
    PfLmTab lmtab;       // the landmark records [nl][5][n], two buffers, in chunks (see PfLmTab)
    PfLmTab* d_lmtab;    // its device copy
    size_t lm_chunk_bytes;
    void* logw;          // [n]: the LIVE one of logw2 (what the legacy entry points work on)
    void* logw2[2];
    int lwcur;
    int cur;             // landmarks: the buffer legacy (non-lazy) kernels work on; valid when !lazy_dirty
    int pcur;            // poses: which of the two buffers is live
    // Lazy resampling (whole filter on this shard): a resampling step permutes POSES and composes ancestor tables; a
    // landmark's records move only when the landmark is next updated.  Landmark l's record of particle p sits in
    // buffer lbuf[l] at slot tab[ltab[l]][p] (ltab = -1: slot p).  See "lazy resampling" below.
    std::vector<int8_t> lbuf;
    std::vector<int16_t> ltab;
    std::vector<int> tref;       // landmarks referring to each table
    std::vector<int32_t> prior;  // staging scratch: a landmark's location before the current call (-1: not yet observed in it)
    int32_t* d_tab[2];           // [PF_TAB_MAX][n] ancestor tables, two sides (composition is out of place)
    int tside;
    int lazy_dirty;              // some landmark is not (buffer cur, identity table)
    int lazy_off;                // SLAMHIP_PF_EAGER=1: always the eager gather
    int32_t* d_lmeta;            // [nl] per-landmark work list of the materialise kernel
    std::vector<char> seen;
    int32_t* h_ids;      // pinned, [2][ocap]: observation landmark ids (0-based; bit 30 marks "new landmark"); two
    double* h_obs;       // pinned, [2][ocap][2]   staging slots used alternately, each guarded by an event, read by the kernels
    hipEvent_t stage_ev[2];
    int stage_used[2], stage_slot, stage_last;
    int32_t* h_ids_dev;  // device-side addresses of the pinned slots
    double* h_obs_dev;
    int ocap;
    double* d_part;      // [blocks][4] reduction partials
    double* d_out;       // [8]
    double* h_out;       // pinned [8]: seven statistics + the sequence word the host polls
    double* h_out_dev;   // its device-side address
    long long out_seq;
    double pending_shift;    // slam_pf_normalize defers its shift: the next kernel that touches logw applies it
    int has_pending;
    double* d_cdf;       // [n_global]
    double* d_bsum;      // [scan blocks]
    int32_t* d_src;      // [n] gather source: >= 0 local index, < 0: -(recv position + 1)
    int32_t* d_anc;      // [n] ancestors of slam_pf_resample_local
    int red_blocks;
    // ---- auto mode (slam_pf_step_auto): the per-step statistics, the Neff decision, the lazy-resampling bookkeeping
    // and the resampling itself stay on the device; the host only enqueues.  See "auto mode" below.
    PfCtl* d_ctl;                // device control block
    int32_t* d_lmstate;          // [nl] per-landmark state word (table + 1 | buffer << 8 | seen << 9)
    PfMirror* h_mir;             // pinned: what the host may look at without synchronising
    PfMirror* h_mir_dev;
    int auto_on;                 // the device copy of the bookkeeping is the live one
    long long auto_seq;          // last step enqueued
    long long pub_seq;           // last step whose publication to the mirror is enqueued (with it, or by the publish kernel)
    long long nresamples;        // resamplings so far (drives the systematic-resampling offset)
    std::vector<PfStepRec> log;       // queued steps not yet confirmed by the device (replayed after a halt)
    double* d_xchg;              // device address of the ranks' shared scalar page (sharded filter, legacy form), or null
    void* xchg_host;
    int xchg_rank, xchg_world;
    // peers (slam_pf_attach_peers): the sharded filter resamples on the device
    PfPeers* d_peers;            // device copy of the table below (null: no peers attached)
    PfPeers peers;
    PfInbox* inbox;              // this rank's inbox (device memory, exported)
    size_t inbox_bytes;
    void* peer_open[PF_MAX_WORLD][7 + 2 * PF_LM_MAXC];   // what hipIpcOpenMemHandle returned (closed at detach); null for in-process peers
    int64_t par_max_n;           // filters / shards of at most this many particles take the observation-parallel step kernel
    int64_t way4_max_n, way2_max_n;   // ... up to these: 4 / 2 observation ways on 256-particle workgroups
    long long bar_count;         // peer barriers enqueued so far (the same on every rank: the calls are collective)
    long long halts;             // SLAM_PF_HALTED returns so far
    double last_out[4];          // {Neff, resampled?, resamplings, step} of the last confirmed step
    int halted;                  // a sharded filter's step wants a resampling: the caller exchanges, then slam_pf_resume
    double halt_gmax;            // largest normalised log-weight of the halted step
    long long last_resampled_seq;
};

namespace {

// Timing experiment (make exp with -DPF_EXP_STAMPS): where does ONE workgroup of the auto step (the middle one) spend its
// time?  g_xs: [0] its first instruction, [1] control block read, [2] observations planned, [3] pose predicted,
// [4] map updates done, [5] statistics stored; [7] workgroup 0's first instruction.  g_wg: the same three points of
// EVERY workgroup (printed by slam_pf_debug_stamps), which is how the 3-of-4 residency at 135 registers was found.
#if defined(SLAMHIP_EXPERIMENTS) && defined(PF_EXP_STAMPS)
__device__ unsigned long long g_xs[8];
__device__ unsigned long long g_wg[3][4096];        // every workgroup: first instruction, map updates done, statistics stored
#define PF_WG(k)                                                                                                   \
    do {                                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_wg[k][blockIdx.x] = wall_clock64();                           \
    } while (0)
#define PF_XS(k)                                                                                                   \
    do {                                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) g_xs[k] = wall_clock64();                             \
    } while (0)
#else
#define PF_XS(k) do { } while (0)
#define PF_WG(k) do { } while (0)
#endif

// ---- Philox4x32-10 -------------------------------------------------------------------------------
__host__ __device__ inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c0;
        const uint64_t p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename T>
__device__ inline T u01(uint32_t x) {      // 24 random bits, offset by half a step: never 0 or 1
    return ((T)(x >> 8) + (T)0.5) * (T)(1.0 / 16777216.0);
}

// ---- fp32 arithmetic of the sweep ----------------------------------------------------------------------------
// The fp32 sweep is bound by instruction issue as much as by memory (about 4500 vector instructions per particle at C4
// on four waves per SIMD), and a third of those were the IEEE-exact library forms of log, atan2, sin and cos (range
// reduction for arguments up to 1e38, denormal and infinity handling).  The fp32 instantiation uses the hardware's
// transcendental unit instead: v_log_f32 / v_sin_f32 / v_cos_f32 / v_sqrt_f32 / v_rcp_f32 (absolute error about 1e-6
// on sin and cos of an angle of a few radians, 1 ulp on the others) and a degree-15 odd polynomial for atan (8e-8).
// That is inside the rounding of the fp32 state itself; fp64 keeps the exact forms.  PF_FAST_MATH=0 builds the exact
// forms for fp32 too.
#ifndef PF_FAST_MATH
#define PF_FAST_MATH 1
#endif
template <typename T>
constexpr bool kFast = PF_FAST_MATH && sizeof(T) == 4;

template <typename T>
__device__ __forceinline__ T m_log(T x) {
    if constexpr (kFast<T>) return 0.69314718f * __builtin_amdgcn_logf(x);
    else return log(x);
}
template <typename T>
__device__ __forceinline__ T m_sqrt(T x) {
    if constexpr (kFast<T>) return __builtin_amdgcn_sqrtf(x);
    else return sqrt(x);
}
template <typename T>
__device__ __forceinline__ void m_sincos(T a, T& sn, T& cs) {            // |a| up to a few hundred radians
    if constexpr (kFast<T>) {
        const float rev = a * 0.15915494f;                               // the unit takes revolutions
        sn = __builtin_amdgcn_sinf(rev);
        cs = __builtin_amdgcn_cosf(rev);
    } else {
        sn = sin(a);
        cs = cos(a);
    }
}
template <typename T>
__device__ __forceinline__ T m_atan2(T y, T x) {
    if constexpr (kFast<T>) {
        const float ax = fabsf(x), ay = fabsf(y);
        const float t = fminf(ax, ay) * __builtin_amdgcn_rcpf(fmaxf(ax, ay));          // [0, 1]
        const float q = t * t;
        // atan t = t + t^3 P(t^2) on [0, 1], near-minimax (Lawson-weighted least squares), |error| < 8.3e-8 in fp32
        float p = 0.002622197614982724f;
        p = fmaf(p, q, -0.015132341533899307f);
        p = fmaf(p, q, 0.041121527552604675f);
        p = fmaf(p, q, -0.0736667662858963f);
        p = fmaf(p, q, 0.10573917627334595f);
        p = fmaf(p, q, -0.14185971021652222f);
        p = fmaf(p, q, 0.1999039649963379f);
        p = fmaf(p, q, -0.33332985639572144f);
        float a = fmaf(t * q, p, t);
        a = ay > ax ? 1.57079633f - a : a;
        a = x < 0.0f ? 3.14159265f - a : a;
        return copysignf(a, y);
    } else {
        return atan2(y, x);
    }
}

template <typename T>
__device__ inline void normals2(uint64_t gid, uint32_t step, uint32_t stream, uint64_t seed, T& e1, T& e2) {
    uint32_t r[4];
    philox((uint32_t)gid, (uint32_t)(gid >> 32), step, stream, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const T u1 = u01<T>(r[0]), u2 = u01<T>(r[1]);
    const T rad = m_sqrt<T>((T)-2.0 * m_log<T>(u1));
    if constexpr (kFast<T>) {
        e1 = rad * __builtin_amdgcn_cosf(u2);          // cos(2 pi u2): the unit takes revolutions
        e2 = rad * __builtin_amdgcn_sinf(u2);
    } else {
        const T ang = (T)(2.0 * PF_PI) * u2;
        e1 = rad * cos(ang);
        e2 = rad * sin(ang);
    }
}

template <typename T>
__device__ inline T wrap_pi(T a) {         // mpi_to_pi, src/common.jl:102-110: single conditional wrap
    if (a > (T)PF_PI) return a - (T)(2.0 * PF_PI);
    if (a < (T)-PF_PI) return a + (T)(2.0 * PF_PI);
    return a;
}

constexpr uint32_t STREAM_PREDICT = 0, STREAM_INIT = 1;

// ---- F1 ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pf_predict_kernel(T* __restrict__ pose, int64_t n, int64_t first, uint32_t step,
                                                          uint64_t seed, T V, T G, T wheelbase, T sigV, T sigG, T dt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    T e1, e2;
    normals2<T>((uint64_t)(first + p), step, STREAM_PREDICT, seed, e1, e2);
    const T Vn = V + sigV * e1;                       // sim/sim-utils.jl:36
    const T Gn = G + sigG * e2;                       // :37
    const T x = pose[p], y = pose[n + p], phi = pose[2 * n + p];
    T sgp, cgp, sg, cg;
    m_sincos<T>(Gn + phi, sgp, cgp);
    m_sincos<T>(Gn, sg, cg);
    pose[p] = x + Vn * dt * cgp;                      // src/ekf.jl:39-41
    pose[n + p] = y + Vn * dt * sgp;
    pose[2 * n + p] = wrap_pi<T>(phi + Vn * dt * sg / wheelbase);
}

template <typename T>
__global__ __launch_bounds__(256) void pf_set_pose_kernel(T* __restrict__ pose, T* __restrict__ logw, int64_t n, T x, T y,
                                                           T phi, T lw) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    pose[p] = x; pose[n + p] = y; pose[2 * n + p] = phi;
    logw[p] = lw;
}

template <typename T>
__global__ __launch_bounds__(256) void pf_init_lm_kernel(LmView<T> lv, int buf, int64_t n, int64_t first, uint64_t seed,
                                                          const double* __restrict__ xy, int nl, T var, T jitter) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    for (int l = 0; l < nl; ++l) {
        T e1, e2;
        normals2<T>((uint64_t)(first + p), (uint32_t)l, STREAM_INIT, seed, e1, e2);
        T* row = lv.rows(buf, l, n) + p;
        row[0] = (T)xy[2 * l] + jitter * e1;
        row[n] = (T)xy[2 * l + 1] + jitter * e2;
        row[2 * n] = var;
        row[3 * n] = (T)0;
        row[4 * n] = var;
    }
}

// ---- F2 / F3 -------------------------------------------------------------------------------------
constexpr int32_t NEW_FLAG = 1 << 30;       // observation code: first sighting of this landmark
constexpr int32_t FRESH_FLAG = 1 << 29;     // a further observation of a landmark first seen in the SAME call
constexpr int32_t ID_MASK = FRESH_FLAG - 1;
// second staged word per observation: where this observation's landmark record is read and written
constexpr int32_t META_TAB = 0xff;          // table index + 1 (0: the particle's own slot)
constexpr int32_t META_RBUF = 1 << 8;       // buffer the record is read from
constexpr int32_t META_WBUF = 1 << 9;       // buffer the updated record goes to (slot p)
constexpr int META_PRIOR_SHIFT = 10;        // bits 10..18: table + 1 and buffer of the record as it was BEFORE this call
                                            // (the FastSLAM-2.0 proposal reads every observation against the prior map)
constexpr int PF_OCAP = 1024;               // observations per call; the meta words sit PF_OCAP ints behind the codes
constexpr int PF_TAB_MAX = 64;              // live ancestor tables before the maps are materialised
constexpr int PF_AUTO_MAXOBS = 64;          // observations per slam_pf_step_auto call (planned per workgroup in LDS)
constexpr int PF_LOG = 32;                  // steps the host may run ahead of the device
constexpr int PF_PUBLISH_EVERY = 8;         // a step publishes to the host's mirror when its number is a multiple of this
                                            // (or when it halts / fails); slam_pf_flush asks for the last one
constexpr int32_t LS_TAB = 0xff, LS_BUF = 1 << 8, LS_SEEN = 1 << 9;     // per-landmark state word of the auto mode
static_assert(PF_CTL_TABS == PF_TAB_MAX && PF_CTL_MAXOBS == PF_AUTO_MAXOBS, "control-block sizes");

// One landmark record of one particle (5 strided values).
template <typename T>
struct LmRow {
    T lx, ly, pxx, pxy, pyy;
};
template <typename T>
__device__ __forceinline__ LmRow<T> load_row(const T* __restrict__ row, int64_t n) {
    LmRow<T> r;
    r.lx = row[0]; r.ly = row[n]; r.pxx = row[2 * n]; r.pxy = row[3 * n]; r.pyy = row[4 * n];
    return r;
}

__device__ inline double block_reduce(double v, double* sh, bool is_max, int nw = 0);
__device__ __forceinline__ void fold_partials(const double* __restrict__ part, int nblocks, int relative,
                                              double* __restrict__ out, double* __restrict__ host_out, long long seq);

// Per-block weight statistics with the block's OWN maximum as the shift (one pass; pf_fold_kernel rescales):
// part[b] = {m_b, sum e, sum e^2, sum e x, sum e y, sum e sin(phi), sum e cos(phi)},  e = exp(logw - shift_b),
// shift_b = m_b if `relative` else 0.
// SC1 (auto mode): the partials are stored write-through at agent scope (global_store ... sc1) together with a tag, the
// form in which another workgroup of the SAME launch may read them without an L2 write-back (part_key, pf_auto_tail).
// six block sums at once: one LDS exchange and one barrier pair for all of them.  Same order of additions as six
// block_reduce calls (xor tree inside the wave, then wave 0 + wave 1 + ...), so the sums are the same bit for bit.
__device__ __forceinline__ void block_sum6(double (&v)[6], double (*sh6)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 6; ++i) sh6[wave][i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double r = sh6[0][i];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += sh6[w][i];
        v[i] = r;
    }
}

// the first two of them only (same order of additions for those two: the same sums bit for bit)
__device__ __forceinline__ void block_sum2(double (&v)[6], double (*sh6)[6]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sh6[wave][0] = v[0]; sh6[wave][1] = v[1]; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        double r = sh6[0][i];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += sh6[w][i];
        v[i] = r;
    }
}

// ---- hand-over of the per-workgroup statistics inside ONE launch (auto mode) --------------------------------------
// A workgroup's seven partials and a TAG fill one 64-byte line of `part`: tag = key(step) xor the (rotated) bit patterns
// of the seven values, all eight stored write-through at agent scope and NOT waited for.  The launch's last workgroup
// polls the lines (agent-scope loads) until every line's tag fits its values and this step's key: a line that is stale
// (an earlier step's key), half written or torn does not fit.  Nothing else is needed -- no drain of the storing wave's
// outstanding record stores (3-4 us behind 80 non-temporal stores), no arrival counter (1024 adds to one address:
// another 3-5 us), which the first form of this hand-over paid on every workgroup's way out.
__device__ __forceinline__ unsigned long long part_key(long long seq) {
    return (unsigned long long)seq * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
}
__device__ __forceinline__ unsigned long long part_hash(const double (&v)[7]) {
    unsigned long long x = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
        x ^= (b << (9 * i + 3)) | (b >> (64 - (9 * i + 3)));
    }
    return x;
}

// POSE = false (the filter-step kernels, round 3): the weighted pose sums are left out (zeros in the record) -- nothing reads
// them from a step (the mean pose is asked for through pf_stats_kernel), and they cost a double-precision sincos per
// particle and four of the six block sums in a sweep that is bound by instruction issue as much as by memory.
template <typename T, bool SC1 = false, bool POSE = true>
__device__ __forceinline__ void block_weight_stats(T lw, T x, T y, T phi, bool valid, int relative, double* __restrict__ part,
                                                   long long seq = 0) {
    __shared__ double sh[16];
    __shared__ double sh6[16][6];
    const double m = block_reduce(valid ? (double)lw : -__builtin_inf(), sh, true);
    const double shift = relative ? m : 0.0;
#if defined(SLAMHIP_EXPERIMENTS) && defined(PF_EXP_NOSTATS)          // timing experiment: WRONG statistics
    const double e = valid ? 1.0 + ((double)lw - shift) : 0.0;
    double sn = (double)phi, cs = 1.0;
#else
    const double e = valid ? exp((double)lw - shift) : 0.0;
    double sn = 0.0, cs = 0.0;
    if constexpr (POSE) sincos((double)phi, &sn, &cs);
#endif
    double v[6] = {e, e * e, 0.0, 0.0, 0.0, 0.0};
    if constexpr (POSE) {
        v[2] = e * (double)x; v[3] = e * (double)y; v[4] = e * sn; v[5] = e * cs;
        block_sum6(v, sh6);
    } else {
        block_sum2(v, sh6);
    }
    if (threadIdx.x == 0) {
        double* o = part + (size_t)blockIdx.x * 8;
        const double w[7] = {m, v[0], v[1], v[2], v[3], v[4], v[5]};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (SC1) __hip_atomic_store(o + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else o[i] = w[i];
        }
        if (SC1)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(o + 7), part_hash(w) ^ part_key(seq), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}


// ---- the CANONICAL weight statistics of the auto mode (round 4) ---------------------------------------------------------------
// SURVEY 8e asks for results that do not depend on the number of GPUs.  Particles and maps never did; the NORMALISATION did by
// ulps, because a rank folded its own workgroups' partial sums and the ranks' sums were folded in rank order.  Now the three
// statistics (max log-weight, sum w, sum w^2) are DEFINED as the root of one fixed reduction tree over the GLOBAL particle
// index, whatever computes its nodes:
//   leaf    a wave's 64 consecutive particles: m = their largest log-weight, k = ceil(m / ln 2) (an integer), e_i =
//           exp(logw_i - k ln 2) in double, s1 = sum e_i, s2 = sum e_i^2 by the xor butterfly (every lane ends with the same bits)
//   node    wrec_combine4 of its four children in index order: M = max m, K = ceil(M / ln 2), every child's sums rescaled by
//           2^(k_child - K) -- a power of two, EXACT -- and added left to right.  An absent child is {-inf, 0, 0}, and combining
//           with absent children returns the present one bit for bit, so ragged sizes and any padding of the depth change nothing.
// A step kernel's workgroup stores one tagged line per 64 particles (observation-parallel kernel) or per 256 (the tree's next
// level, formed in the workgroup); the launch's last workgroup climbs to the 1024-particle records, and -- sharded filter with
// peers -- every rank writes ITS records into every rank's inbox and all ranks reduce the same sequence of records with the same
// tree: log-weights bit-identical to the one-rank filter's whenever a rank's slice is a multiple of 1024 particles (every shape
// of BASELINE.json's filter).  (The legacy entry points keep block_weight_stats / fold_partials; they agree to a few ulp.)
struct WRec {
    double m, s1, s2;
};
constexpr double PF_LN2 = 0.693147180559945309417232121458;
constexpr double PF_INV_LN2 = 1.442695040888963407359924681002;
__device__ __forceinline__ WRec wrec_empty() { return WRec{-__builtin_inf(), 0.0, 0.0}; }
__device__ __forceinline__ double wrec_k(double m) { return ceil(m * PF_INV_LN2); }      // the record's binary exponent
__device__ __forceinline__ WRec wrec_combine4(const WRec& a, const WRec& b, const WRec& c, const WRec& d) {
    const double NEG = -__builtin_inf();
    WRec r;
    r.m = fmax(fmax(a.m, b.m), fmax(c.m, d.m));
    if (!(r.m > NEG)) { r.s1 = 0.0; r.s2 = 0.0; return r; }
    const double K = wrec_k(r.m);
    auto sc = [&](const WRec& x, double& f1, double& f2) {
        if (!(x.m > NEG)) { f1 = 0.0; f2 = 0.0; return; }
        const int dk = (int)fmax(wrec_k(x.m) - K, -4000.0);            // <= 0
        f1 = ldexp(x.s1, dk);
        f2 = ldexp(x.s2, 2 * dk);
    };
    double a1, a2, b1, b2, c1, c2, d1, d2;
    sc(a, a1, a2); sc(b, b1, b2); sc(c, c1, c2); sc(d, d1, d2);
    r.s1 = ((a1 + b1) + c1) + d1;
    r.s2 = ((a2 + b2) + c2) + d2;
    return r;
}
// leaf: the wave's 64 particles (every lane returns the same record)
template <typename T>
__device__ __forceinline__ WRec wrec_wave(T lw, bool valid) {
    const double NEG = -__builtin_inf();
    double m = valid ? (double)lw : NEG;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    WRec r;
    r.m = m;
    const bool live = valid && (double)lw > NEG && m > NEG;
#if defined(SLAMHIP_EXPERIMENTS) && defined(PF_EXP_NOSTATS)          // timing experiment: WRONG statistics
    const double e = live ? 1.0 + ((double)lw - m) : 0.0;
#else
    const double e = live ? exp((double)lw - wrec_k(m) * PF_LN2) : 0.0;
#endif
    double s1 = e, s2 = e * e;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    r.s1 = s1; r.s2 = s2;
    return r;
}
__device__ __forceinline__ unsigned long long wrec_hash(const WRec& r) {
    const unsigned long long b0 = (unsigned long long)__double_as_longlong(r.m), b1 = (unsigned long long)__double_as_longlong(r.s1),
                             b2 = (unsigned long long)__double_as_longlong(r.s2);
    return ((b0 << 7) | (b0 >> 57)) ^ ((b1 << 23) | (b1 >> 41)) ^ ((b2 << 41) | (b2 >> 23));
}
// one tagged line {m, s1, s2, tag} at part[8 line ..]: write-through at agent scope, NOT waited for (see part_key)
__device__ __forceinline__ void wrec_store_line(double* __restrict__ part, int line, const WRec& r, long long seq) {
    double* o = part + (size_t)line * 8;
    __hip_atomic_store(o + 0, r.m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, r.s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 2, r.s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(o + 3), wrec_hash(r) ^ part_key(seq), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// a workgroup whose first four waves hold 256 consecutive particles' weights (the sweep kernels): the tree's 256-particle
// node, stored as line `blockIdx.x`.  All threads of the workgroup must call it (one barrier).
template <typename T>
__device__ __forceinline__ void wrec_block_line(T lw, bool valid, double* __restrict__ part, long long seq) {
    __shared__ double sh_w[4][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        const WRec r = wrec_wave<T>(lw, valid);
        if (lane == 0) { sh_w[wave][0] = r.m; sh_w[wave][1] = r.s1; sh_w[wave][2] = r.s2; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const WRec q0{sh_w[0][0], sh_w[0][1], sh_w[0][2]}, q1{sh_w[1][0], sh_w[1][1], sh_w[1][2]},
                   q2{sh_w[2][0], sh_w[2][1], sh_w[2][2]}, q3{sh_w[3][0], sh_w[3][1], sh_w[3][2]};
        wrec_store_line(part, (int)blockIdx.x, wrec_combine4(q0, q1, q2, q3), seq);
    }
}
// 256 threads (four waves), records held by the threads with tid % stride == 0 (stride 1, 4 or 16, consecutive tree
// positions): the node above all of them, returned to every thread.  sh: [4][3] doubles.
__device__ __forceinline__ WRec wrec_tree256(WRec r, int stride, double (*sh)[3]) {
    for (int s = stride; s < 64; s *= 4) {
        WRec b, c, d;
        b.m = __shfl_down(r.m, s); b.s1 = __shfl_down(r.s1, s); b.s2 = __shfl_down(r.s2, s);
        c.m = __shfl_down(r.m, 2 * s); c.s1 = __shfl_down(r.s1, 2 * s); c.s2 = __shfl_down(r.s2, 2 * s);
        d.m = __shfl_down(r.m, 3 * s); d.s1 = __shfl_down(r.s1, 3 * s); d.s2 = __shfl_down(r.s2, 3 * s);
        r = wrec_combine4(r, b, c, d);                     // valid in the lanes with lane % (4 s) == 0
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sh[wave][0] = r.m; sh[wave][1] = r.s1; sh[wave][2] = r.s2; }
    __syncthreads();
    return wrec_combine4(WRec{sh[0][0], sh[0][1], sh[0][2]}, WRec{sh[1][0], sh[1][1], sh[1][2]}, WRec{sh[2][0], sh[2][1], sh[2][2]},
                         WRec{sh[3][0], sh[3][1], sh[3][2]});
}

// F3: first sighting of a landmark -- src/ekf.jl:94-103,112 without the pose term.
// Where an updated record goes: a plain pointer to the particle's first value (rows n apart), or -- the sweep -- a buffer
// descriptor of the landmark's rows plus the lane's byte offset (see lm_rsrc).
template <typename T, typename R>
struct BufRow {
    R rs;
    uint32_t voff, row;
};
template <typename T>
__device__ __forceinline__ void row_store(T* row, int64_t n, int k, T v) { row[k * n] = v; }
template <typename T, typename R>
__device__ __forceinline__ void row_store(const BufRow<T, R>& b, int64_t, int k, T v);

template <typename T, typename ROW>
__device__ __forceinline__ void lm_init(const ROW& row, int64_t n, T x, T y, T phi, T r, T b, T R00, T R10, T R01, T R11,
                                        bool valid) {
    T s, c;
    m_sincos<T>(phi + b, s, c);
    const T g00 = c, g01 = -r * s, g10 = s, g11 = r * c;
    const T a00 = g00 * R00 + g01 * R10, a01 = g00 * R01 + g01 * R11;
    const T a10 = g10 * R00 + g11 * R10, a11 = g10 * R01 + g11 * R11;
    if (valid) {
        row_store<T>(row, n, 0, x + r * c);
        row_store<T>(row, n, 1, y + r * s);
        row_store<T>(row, n, 2, a00 * g00 + a01 * g01);
        row_store<T>(row, n, 3, a00 * g10 + a01 * g11);
        row_store<T>(row, n, 4, a10 * g10 + a11 * g11);
    }
}

// F2: the 2 x 2 EKF update of one landmark record (`cur`, its 5 values) and the log-weight increment.
template <typename T, typename ROW>
__device__ __forceinline__ void lm_update(const ROW& row, int64_t n, const LmRow<T>& cur, T x, T y, T phi, T r, T b, T R00,
                                          T R10, T R01, T R11, bool valid, T& lw) {
    const T lx = cur.lx, ly = cur.ly, pxx = cur.pxx, pxy = cur.pxy, pyy = cur.pyy;
    const T dx = lx - x, dy = ly - y;
    const T d2 = dx * dx + dy * dy;
    // fp32: hardware reciprocal square roots (v_rsq_f32, 1 ulp) instead of IEEE sqrt + eight IEEE divisions -- the
    // kernel's time is one third arithmetic at four waves per SIMD; fp64 keeps the exact operations.
    T d, h00, h01, h10, h11;
    if constexpr (sizeof(T) == 4) {
        const T rd = __builtin_amdgcn_rsqf(d2);
        d = d2 * rd;
        const T rd2 = rd * rd;
        h00 = dx * rd; h01 = dy * rd; h10 = -dy * rd2; h11 = dx * rd2;           // src/common.jl:162
    } else {
        d = sqrt(d2);
        h00 = dx / d; h01 = dy / d; h10 = -dy / d2; h11 = dx / d2;
    }
    const T v0 = r - d;                                               // src/ekf.jl:58
    const T v1 = wrap_pi<T>(b - (m_atan2<T>(dy, dx) - phi));
    const T t00 = pxx * h00 + pxy * h01, t01 = pxx * h10 + pxy * h11;  // PHt
    const T t10 = pxy * h00 + pyy * h01, t11 = pxy * h10 + pyy * h11;
    const T s00 = h00 * t00 + h01 * t10 + R00;                         // S = Hf PHt + R (:68)
    const T s01a = h00 * t01 + h01 * t11 + R01;
    const T s10a = h10 * t00 + h11 * t10 + R10;
    const T s11 = h10 * t01 + h11 * t11 + R11;
    const T s01 = (T)0.5 * (s01a + s10a);                             // (:69)
    T u00, u01, u11, c00, c01, c11;                                   // chol(S) upper (:70), C = inv(U)
    if constexpr (sizeof(T) == 4) {
        c00 = __builtin_amdgcn_rsqf(s00);
        u00 = s00 * c00;
        u01 = s01 * c00;
        const T tt = s11 - u01 * u01;
        c11 = __builtin_amdgcn_rsqf(tt);
        u11 = tt * c11;
        c01 = -u01 * (c00 * c11);
    } else {
        u00 = sqrt(s00);
        u01 = s01 / u00;
        u11 = sqrt(s11 - u01 * u01);
        c00 = (T)1 / u00; c01 = -u01 / (u00 * u11); c11 = (T)1 / u11;
    }
    const T w00 = t00 * c00, w01 = t00 * c01 + t01 * c11;             // W1 = PHt C (:71)
    const T w10 = t10 * c00, w11 = t10 * c01 + t11 * c11;
    const T y0 = c00 * v0, y1 = c01 * v0 + c11 * v1;                  // C' v
    if (valid) {
        row_store<T>(row, n, 0, lx + w00 * y0 + w01 * y1);                            // x += W v (:72,:74)
        row_store<T>(row, n, 1, ly + w10 * y0 + w11 * y1);
        row_store<T>(row, n, 2, pxx - (w00 * w00 + w01 * w01));               // P -= W1 W1' (:75)
        row_store<T>(row, n, 3, pxy - (w00 * w10 + w01 * w11));
        row_store<T>(row, n, 4, pyy - (w10 * w10 + w11 * w11));
    }
    lw += (T)-0.5 * (y0 * y0 + y1 * y1) - m_log<T>(u00 * u11) - (T)1.8378770664093453;   // log(2 pi)
}

// The m known-id observations of one particle at pose (x, y, phi), in order: F2 on a landmark the filter has seen,
// F3 on a first sighting.  The sweep is bound by memory LATENCY (one particle per lane, four waves per SIMD at C4), so
// the records of the next PF_DEPTH observations are kept in flight per particle: a ring of PF_DEPTH records in
// registers, the record of observation i + PF_DEPTH requested before observation i is processed, the first PF_DEPTH
// before the motion model runs (KnownRing::start).  A landmark that one of the PF_DEPTH observations before it writes
// (a repeat inside the call) cannot be requested ahead: it is read after that store, at its turn.  Records are read
// where the lazy resampling left them (sweep_load: buffer + slot through the landmark's ancestor table) and written to
// the particle's own slot of the buffer the staging chose.  The arithmetic and its order do not depend on the depth.
#ifndef PF_DEPTH
#define PF_DEPTH 4
#endif
// The sweep reads and writes records through BUFFER instructions: a wave-uniform descriptor per landmark (base = the
// landmark's five rows in the buffer read or written, 5 n values), the field's row as the scalar offset k n sizeof(T),
// the lane's slot as a 32-bit byte offset -- no vector address arithmetic at all (it was 12 of the ~200 vector
// instructions per observation, in a kernel that is bound by instruction issue as much as by memory).  Non-temporal:
// a record is touched once per step, 5 GB of other records pass before it is touched again.
// (slam_pf_create bounds n so that 5 n sizeof(T) fits 32 bits.)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ auto lm_rsrc(const T* base, int64_t n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), (short)0, (int)(uint32_t)(5 * n * (int64_t)sizeof(T)), 0x00020000);
}
template <typename T, int AUX = 2, typename R>
__device__ __forceinline__ T rec_load(R rs, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(T) == 4) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX));
    else return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX));
}
template <typename T, typename R>
__device__ __forceinline__ void rec_store(T v, R rs, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rs, voff, soff, 2);
}

template <typename T, typename R>
__device__ __forceinline__ void row_store(const BufRow<T, R>& b, int64_t, int k, T v) {
    rec_store<T>(v, b.rs, b.voff, (uint32_t)k * b.row);
}

// How the sweep of a SHARDED filter resolves an ancestor-table entry: the entry is a GLOBAL particle id; its owner's
// buffers are addressed through the peer table (this rank's own slice through the local descriptors, as before).
struct PfShardCtx {
    const PfPeers* peers;
    uint32_t first, n;           // this rank's slice [first, first + n)
    int rank, world;
};
template <typename T>
__device__ __forceinline__ T ld_sys(const T* p) {          // a load that a peer GPU's store is visible to (sc0 sc1)
    if constexpr (sizeof(T) == 4)
        return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    else
        return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_SYSTEM));
}
// owner of global id g when the ranks own equal slices of n (world <= 8: seven compares instead of a division)
__device__ __forceinline__ uint32_t pf_owner(uint32_t g, uint32_t n, int world) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 1; k < PF_MAX_WORLD; ++k) r += (k < world && g >= (uint32_t)k * n) ? 1u : 0u;
    return r;
}

// The record of observation (code, meta) as particle p reads it: its own slot or, after a lazy resampling, its
// ancestor's through the landmark's table.  SH (sharded filter with peers): the ancestor may live on another rank --
// its record is then read from that rank's buffer over xGMI (system-scope loads; the owner wrote it in a kernel that
// had completed before the resampling that created the entry, see pf_peer_gate_kernel).
template <typename T, int AUX = 2, bool SH = false>      // AUX: cache policy of a record read from the particle's own slot (2 = non-temporal)
__device__ __forceinline__ LmRow<T> sweep_load(const LmView<T> lv, const int32_t* __restrict__ tabs, int64_t n, uint32_t p,
                                               int32_t code, int32_t meta, const PfShardCtx& sc) {
    const int t = meta & META_TAB;
    const auto rs = lm_rsrc<T>(lv.rows((meta & META_RBUF) ? 1 : 0, code & ID_MASK, n), n);
    const uint32_t row = (uint32_t)n * (uint32_t)sizeof(T);
    LmRow<T> r;
    if (t) {                                                           // uniform
        // through a table: several particles -- of other waves too -- read the same ancestor's record, so these loads
        // keep the default cache policy (one-box A/B against non-temporal: 74.0 against 80.3 us per resampling step)
        const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(tabs + (size_t)(t - 1) * n), (short)0,
                                                          (int)(uint32_t)(n * 4), 0x00020000);
        uint32_t slot = __builtin_amdgcn_raw_buffer_load_b32(rt, p * 4u, 0, 2);
        bool local = true;
        if constexpr (SH) {
            const uint32_t owner = pf_owner(slot, sc.n, sc.world);
            local = owner == (uint32_t)sc.rank;
            if (!local) {
                const LmView<T> pv{&sc.peers->lm[owner]};       // the owner's chunks as this GPU addresses them
                const T* base = pv.rows((meta & META_RBUF) ? 1 : 0, code & ID_MASK, n) + (slot - owner * sc.n);
                r.lx = ld_sys(base);
                r.ly = ld_sys(base + n);
                r.pxx = ld_sys(base + 2 * n);
                r.pxy = ld_sys(base + 3 * n);
                r.pyy = ld_sys(base + 4 * n);
            }
            slot -= sc.first;
        }
        if (local) {
            const uint32_t voff = slot * (uint32_t)sizeof(T);
            r.lx = rec_load<T, 0>(rs, voff, 0u);
            r.ly = rec_load<T, 0>(rs, voff, row);
            r.pxx = rec_load<T, 0>(rs, voff, 2u * row);
            r.pxy = rec_load<T, 0>(rs, voff, 3u * row);
            r.pyy = rec_load<T, 0>(rs, voff, 4u * row);
        }
    } else {
        const uint32_t voff = p * (uint32_t)sizeof(T);
        r.lx = rec_load<T, AUX>(rs, voff, 0u);
        r.ly = rec_load<T, AUX>(rs, voff, row);
        r.pxx = rec_load<T, AUX>(rs, voff, 2u * row);
        r.pxy = rec_load<T, AUX>(rs, voff, 3u * row);
        r.pyy = rec_load<T, AUX>(rs, voff, 4u * row);
    }
    return r;
}

template <typename T, bool SH = false>
struct KnownRing {
    LmRow<T> ring[PF_DEPTH];
    bool have[PF_DEPTH];

    // a value read from LDS at a wave-uniform address IS uniform: say so, and everything derived from it -- the branches
    // on the codes, the landmark's base address -- is scalar work
    static __device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

    // may observation j's record be requested PF_DEPTH observations ahead?  (uniform: the codes sit in LDS)
    static __device__ __forceinline__ bool ahead(const int32_t* s_ids, int j) {
        const int32_t c = uni(s_ids[j]);
        if (c & NEW_FLAG) return false;
        const int l = c & ID_MASK;
        bool ok = true;
#pragma unroll
        for (int k = 1; k <= PF_DEPTH; ++k)
            if (j - k >= 0 && (uni(s_ids[j - k]) & ID_MASK) == l) ok = false;
        return ok;
    }

    __device__ __forceinline__ void start(const LmView<T> lv, const int32_t* __restrict__ tabs, int64_t n, uint32_t p,
                                          const int32_t* s_ids, const int32_t* s_meta, int m, const PfShardCtx& sc) {
#pragma unroll
        for (int u = 0; u < PF_DEPTH; ++u) {
            have[u] = false;
            ring[u] = LmRow<T>{0, 0, 0, 0, 0};
            if (u < m && ahead(s_ids, u)) {
                ring[u] = sweep_load<T, 2, SH>(lv, tabs, n, p, uni(s_ids[u]), uni(s_meta[u]), sc);
                have[u] = true;
            }
        }
    }

    __device__ __forceinline__ void run(const LmView<T> lv, const int32_t* __restrict__ tabs, int64_t n, uint32_t p,
                                        const T* s_obs, const int32_t* s_ids, const int32_t* s_meta, int m, T x, T y, T phi,
                                        T R00, T R10, T R01, T R11, bool valid, T& lw, const PfShardCtx& sc) {
#if defined(SLAMHIP_EXPERIMENTS) && defined(PF_EXP_NOOBS)           // timing experiment: no map updates
        m = 0;
#endif
        for (int i0 = 0; i0 < m; i0 += PF_DEPTH) {
#pragma unroll
            for (int u = 0; u < PF_DEPTH; ++u) {
                const int i = i0 + u;
                if (i >= m) break;                                 // uniform
                const int32_t code = uni(s_ids[i]), meta = uni(s_meta[i]);
                const int l = code & ID_MASK;
                const T r = s_obs[2 * i], b = s_obs[2 * i + 1];
                const BufRow<T, decltype(lm_rsrc<T>((const T*)nullptr, n))> row{lm_rsrc<T>(lv.rows((meta & META_WBUF) ? 1 : 0, l, n), n),
                                                                  p * (uint32_t)sizeof(T), (uint32_t)n * (uint32_t)sizeof(T)};
                LmRow<T> cur = ring[u];
                const bool have_cur = have[u];
                have[u] = false;
                const int j = i + PF_DEPTH;
                if (j < m && ahead(s_ids, j)) {                    // uniform
                    ring[u] = sweep_load<T, 2, SH>(lv, tabs, n, p, uni(s_ids[j]), uni(s_meta[j]), sc);
                    have[u] = true;
                }
                if (code & NEW_FLAG) {                             // F3: src/ekf.jl:94-103,112 without the pose term
                    lm_init<T>(row, n, x, y, phi, r, b, R00, R10, R01, R11, valid);
                    continue;
                }
                if (!have_cur) cur = sweep_load<T, 2, SH>(lv, tabs, n, p, code, meta, sc);
                lm_update<T>(row, n, cur, x, y, phi, r, b, R00, R10, R01, R11, valid, lw);
            }
        }
    }
};

template <typename T, bool SH = false>
__device__ __forceinline__ void apply_known(const LmView<T> lv, const int32_t* __restrict__ tabs, int64_t n, int64_t p,
                                            const T* s_obs, const int32_t* s_ids, const int32_t* s_meta, int m, T x, T y,
                                            T phi, T R00, T R10, T R01, T R11, bool valid, T& lw, const PfShardCtx& sc) {
    KnownRing<T, SH> k;
    k.start(lv, tabs, n, (uint32_t)p, s_ids, s_meta, m, sc);
    k.run(lv, tabs, n, (uint32_t)p, s_obs, s_ids, s_meta, m, x, y, phi, R00, R10, R01, R11, valid, lw, sc);
}

// One particle's filter step: predict (PREDICT), the m known-id updates, the log-weight.  Shared by the legacy kernels
// (observation codes staged by the host) and the auto mode's kernel (codes planned on the device).
// PRELOADED (the auto mode's kernel): x, y, phi hold the particle's pose, lw its stored log-weight and e1, e2 its two
// normal deviates on entry -- requested / computed before the observation plan's barriers, off the critical path.
template <typename T, bool PREDICT, bool PRELOADED = false, bool SH = false>
__device__ __forceinline__ void step_core(T* __restrict__ pose, const LmView<T> lv, const int32_t* __restrict__ tabs,
                                          T* __restrict__ logw, int64_t n, int64_t first, uint32_t step, uint64_t seed, T V, T G,
                                          T wheelbase, T sigV, T sigG, T dt, const T* s_obs, const int32_t* s_ids,
                                          const int32_t* s_meta, int m, T R00, T R10, T R01, T R11, T pend, int64_t p, bool valid,
                                          T& x, T& y, T& phi, T& lw, T e1 = 0, T e2 = 0, const PfShardCtx& sc = PfShardCtx{}) {
    if (!PRELOADED) {
        x = pose[p]; y = pose[n + p]; phi = pose[2 * n + p];
        lw = logw[p];
    }
    lw -= pend;                   // `pend`: the normalisation shift deferred by slam_pf_normalize (0 if none)
    KnownRing<T, SH> known;
    known.start(lv, tabs, n, (uint32_t)p, s_ids, s_meta, m, sc);  // the first records are in flight during the motion model
    if (PREDICT) {
#if defined(SLAMHIP_EXPERIMENTS) && defined(PF_EXP_NOPREDICT)       // timing experiment: no noise
        e1 = (T)0.1; e2 = (T)-0.1;
#else
        if (!PRELOADED) normals2<T>((uint64_t)(first + p), step, STREAM_PREDICT, seed, e1, e2);
#endif
        const T Vn = V + sigV * e1;                       // sim/sim-utils.jl:36
        const T Gn = G + sigG * e2;                       // :37
        T sgp, cgp, sg, cg;
        m_sincos<T>(Gn + phi, sgp, cgp);
        m_sincos<T>(Gn, sg, cg);
        const T xn = x + Vn * dt * cgp;                   // src/ekf.jl:39-41
        const T yn = y + Vn * dt * sgp;
        const T pn = wrap_pi<T>(phi + Vn * dt * sg / wheelbase);
        x = xn; y = yn; phi = pn;
        if (valid) { pose[p] = x; pose[n + p] = y; pose[2 * n + p] = phi; }
    }
    PF_XS(3);
    known.run(lv, tabs, n, (uint32_t)p, s_obs, s_ids, s_meta, m, x, y, phi, R00, R10, R01, R11, valid, lw, sc);
    if (valid) logw[p] = lw;
}

// F1 + F2/F3 (+ F4 partials): one pass over the particle -- predict (PREDICT), the m known-id updates, and
// (STATS) the block's weight statistics, so that a filter step is ONE sweep of HBM instead of five launches.
template <typename T, bool PREDICT, bool STATS>
__global__ __launch_bounds__(256) void pf_step_kernel(T* __restrict__ pose, LmView<T> lv, const int32_t* __restrict__ tabs,
                                                       T* __restrict__ logw,
                                                       int64_t n, int64_t first, uint32_t step, uint64_t seed, T V, T G,
                                                       T wheelbase, T sigV, T sigG, T dt, const double* __restrict__ z,
                                                       const int32_t* __restrict__ ids, int m, T R00, T R10, T R01, T R11,
                                                       double* __restrict__ part, T pend) {
    // the observation list may live in pinned HOST memory (zero-copy staging): one read per workgroup into LDS
    extern __shared__ double s_raw[];              // room for [m][2] doubles, then [m] codes, then [m] meta words
    T* s_obs = reinterpret_cast<T*>(s_raw);        // the observations in the state dtype: converted once per workgroup
    int32_t* s_ids = reinterpret_cast<int32_t*>(s_raw + 2 * m);
    int32_t* s_meta = s_ids + m;
    for (int i = threadIdx.x; i < 2 * m; i += blockDim.x) s_obs[i] = (T)z[i];
    for (int i = threadIdx.x; i < m; i += blockDim.x) { s_ids[i] = ids[i]; s_meta[i] = ids[PF_OCAP + i]; }
    __syncthreads();
    const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pi < n;
    if (!STATS && !valid) return;
    const int64_t p = valid ? pi : n - 1;          // (STATS: idle lanes shadow the last particle, stores are masked)
    T x, y, phi, lw;
    step_core<T, PREDICT>(pose, lv, tabs, logw, n, first, step, seed, V, G, wheelbase, sigV, sigG, dt, s_obs, s_ids, s_meta,
                          m, R00, R10, R01, R11, pend, p, valid, x, y, phi, lw);
    // (folding the partials in the last workgroup to finish behind an agent-scope release/acquire was tried: that is an L2
    //  write-back + invalidate on this multi-XCD part and doubled the kernel's time; here a 1-workgroup fold kernel
    //  follows, the auto mode's kernel uses write-through partials instead: pf_auto_step_kernel)
    if (STATS) block_weight_stats<T, false, false>(lw, x, y, phi, valid, 1, part);
}

// One particle's FastSLAM-2.0 step (see pf_proposal_kernel).  Shared by the legacy kernel and the auto mode's kernel.
template <typename T, bool SH = false>
__device__ __forceinline__ void proposal_core(T* __restrict__ pose, const LmView<T> lv, const int32_t* __restrict__ tabs,
                                              T* __restrict__ logw, int64_t n, int64_t first, uint32_t step, uint64_t seed, T V,
                                              T G, T wheelbase, T lq00, T lq10, T lq11, T dt, const T* s_obs,
                                              const int32_t* s_ids, const int32_t* s_meta, int m, T R00, T R10, T R01, T R11,
                                              T pend, int64_t p, bool valid, T& xo, T& yo, T& po, T& lwo,
                                              const PfShardCtx& sc = PfShardCtx{}) {
    const T x = pose[p], y = pose[n + p], phi = pose[2 * n + p];
    // motion mean (w = 0) and GL = Gu Lq
    T s, c, sG, cG;
    m_sincos<T>(G + phi, s, c);
    m_sincos<T>(G, sG, cG);
    const T vts = V * dt * s, vtc = V * dt * c;
    const T xm = x + vtc, ym = y + vts;
    const T pm = wrap_pi<T>(phi + V * dt * sG / wheelbase);
    const T gu20 = dt * sG / wheelbase, gu21 = V * dt * cG / wheelbase;
    const T gl00 = dt * c * lq00 + (-vts) * lq10, gl01 = (-vts) * lq11;
    const T gl10 = dt * s * lq00 + vtc * lq10, gl11 = vtc * lq11;
    const T gl20 = gu20 * lq00 + gu21 * lq10, gl21 = gu21 * lq11;
    T mu0 = 0, mu1 = 0, g00 = 1, g01 = 0, g11 = 1;
    T lw = logw[p] - pend;
    // pass 1 reads the PRIOR map only (nothing is written): the records of the next PF_DEPTH observations are kept in
    // flight as in the sweep's second pass (KnownRing), through the same buffer descriptors
    auto uni = [](int32_t v) { return __builtin_amdgcn_readfirstlane(v); };
    auto prior_row = [&](int j) {
        // (default cache policy: the second pass reads the same records again -- one-box A/B against non-temporal:
        //  69.0 against 73.5 us per step)
        return sweep_load<T, 0, SH>(lv, tabs, n, (uint32_t)p, uni(s_ids[j]), uni(s_meta[j]) >> META_PRIOR_SHIFT, sc);
    };
    auto informative = [&](int j) { return j < m && !(uni(s_ids[j]) & (NEW_FLAG | FRESH_FLAG)); };
    LmRow<T> ring[PF_DEPTH];
#pragma unroll
    for (int u = 0; u < PF_DEPTH; ++u) {
        ring[u] = LmRow<T>{0, 0, 0, 0, 0};
        if (informative(u)) ring[u] = prior_row(u);
    }
    for (int i0 = 0; i0 < m; i0 += PF_DEPTH)
#pragma unroll
    for (int u = 0; u < PF_DEPTH; ++u) {
        const int i = i0 + u;
        if (i >= m) break;                             // uniform
        const int32_t code = uni(s_ids[i]);
        const LmRow<T> cur = ring[u];
        if (informative(i + PF_DEPTH)) ring[u] = prior_row(i + PF_DEPTH);
        if (code & (NEW_FLAG | FRESH_FLAG)) continue;  // a landmark first seen in this call says nothing about the pose
        const T r = (T)s_obs[2 * i], b = (T)s_obs[2 * i + 1];
        const T dx = cur.lx - xm, dy = cur.ly - ym;
        const T d2 = dx * dx + dy * dy;
        T d, h00, h01, h10, h11;
        if constexpr (sizeof(T) == 4) {
            const T rd = __builtin_amdgcn_rsqf(d2);
            d = d2 * rd;
            const T rd2 = rd * rd;
            h00 = dx * rd; h01 = dy * rd; h10 = -dy * rd2; h11 = dx * rd2;           // src/common.jl:162
        } else {
            d = sqrt(d2);
            h00 = dx / d; h01 = dy / d; h10 = -dy / d2; h11 = dx / d2;
        }
        // B = Hv GL with Hv = [-h00 -h01 0; -h10 -h11 -1]  (src/common.jl:161)
        const T b00 = -(h00 * gl00 + h01 * gl10), b01 = -(h00 * gl01 + h01 * gl11);
        const T b10 = -(h10 * gl00 + h11 * gl10) - gl20, b11 = -(h10 * gl01 + h11 * gl11) - gl21;
        const T v0 = (r - d) - (b00 * mu0 + b01 * mu1);
        const T v1 = wrap_pi<T>(b - (m_atan2<T>(dy, dx) - pm)) - (b10 * mu0 + b11 * mu1);
        const T t00 = cur.pxx * h00 + cur.pxy * h01, t01 = cur.pxx * h10 + cur.pxy * h11;      // Pf Hf'
        const T t10 = cur.pxy * h00 + cur.pyy * h01, t11 = cur.pxy * h10 + cur.pyy * h11;
        const T f00 = h00 * t00 + h01 * t10 + R00;                                            // Sf, symmetrised
        const T f01 = (T)0.5 * ((h00 * t01 + h01 * t11 + R01) + (h10 * t00 + h11 * t10 + R10));
        const T f11 = h10 * t01 + h11 * t11 + R11;
        const T q00 = g00 * b00 + g01 * b01, q01 = g00 * b10 + g01 * b11;                       // Sig B'
        const T q10 = g01 * b00 + g11 * b01, q11 = g01 * b10 + g11 * b11;
        const T s00 = b00 * q00 + b01 * q10 + f00;                                            // S = B Sig B' + Sf
        const T s01 = (T)0.5 * ((b00 * q01 + b01 * q11 + f01) + (b10 * q00 + b11 * q10 + f01));
        const T s11 = b10 * q01 + b11 * q11 + f11;
        T u00, u01, u11, c00, c01, c11;                                                       // chol(S) upper, C = inv(U)
        if constexpr (sizeof(T) == 4) {
            c00 = __builtin_amdgcn_rsqf(s00);
            u00 = s00 * c00;
            u01 = s01 * c00;
            const T tt = s11 - u01 * u01;
            c11 = __builtin_amdgcn_rsqf(tt);
            u11 = tt * c11;
            c01 = -u01 * (c00 * c11);
        } else {
            u00 = sqrt(s00);
            u01 = s01 / u00;
            u11 = sqrt(s11 - u01 * u01);
            c00 = (T)1 / u00; c01 = -u01 / (u00 * u11); c11 = (T)1 / u11;
        }
        const T w00 = q00 * c00, w01 = q00 * c01 + q01 * c11;
        const T w10 = q10 * c00, w11 = q10 * c01 + q11 * c11;
        const T y0 = c00 * v0, y1 = c01 * v0 + c11 * v1;
        mu0 = mu0 + (w00 * y0 + w01 * y1);
        mu1 = mu1 + (w10 * y0 + w11 * y1);
        g00 = g00 - (w00 * w00 + w01 * w01);
        g01 = g01 - (w00 * w10 + w01 * w11);
        g11 = g11 - (w10 * w10 + w11 * w11);
        lw += (T)-0.5 * (y0 * y0 + y1 * y1) - m_log<T>(u00 * u11) - (T)1.8378770664093453;
    }
    // w ~ N(mu, Sig), the control, the pose
    T e1, e2;
    normals2<T>((uint64_t)(first + p), step, STREAM_PREDICT, seed, e1, e2);
    const T l00 = sqrt(g00);
    const T l10 = g01 / l00;
    const T l11 = sqrt(g11 - l10 * l10);
    const T w0 = mu0 + l00 * e1;
    const T w1 = mu1 + l10 * e1 + l11 * e2;
    const T Vn = V + lq00 * w0;
    const T Gn = G + (lq10 * w0 + lq11 * w1);
    T sgp, cgp, sgn, cgn;
    m_sincos<T>(Gn + phi, sgp, cgp);
    m_sincos<T>(Gn, sgn, cgn);
    const T xn = x + Vn * dt * cgp;                   // src/ekf.jl:39-41
    const T yn = y + Vn * dt * sgp;
    const T pn = wrap_pi<T>(phi + Vn * dt * sgn / wheelbase);
    if (valid) { pose[p] = xn; pose[n + p] = yn; pose[2 * n + p] = pn; logw[p] = lw; }
    T unused = 0;
    apply_known<T, SH>(lv, tabs, n, p, s_obs, s_ids, s_meta, m, xn, yn, pn, R00, R10, R01, R11, valid, unused, sc);
    xo = xn; yo = yn; po = pn; lwo = lw;
}

// ---- N4: FastSLAM-2.0 proposal ------------------------------------------------------------------------
// One step in which the pose is drawn from the proposal that already knows this step's observations (Montemerlo
// et al. 2003; no reference code -- SURVEY 8f N4; specified in oracle/pf_ref.py::step_proposal).  The proposal
// lives in CONTROL space: pose = f(pose, V + u0, G + u1) (src/ekf.jl:39-41), u = Lq w, Lq = chol(Q), w ~ N(0, I)
// a priori.  Around w = 0 the pose moves by GL w, GL = Gu Lq (Gu: src/ekf.jl:27-29), so an observation of a landmark
// the particle holds is a linear 2 x 2 measurement of w with noise Sf = Hf Pf Hf' + R: pass 1 assimilates them in
// the Cholesky form of src/ekf.jl:67-75 and multiplies their predictive densities into the weight, the pose is
// sampled with the SAME two normals FastSLAM-1.0's predict uses (no observation: the same pose bit for bit), and
// pass 2 is apply_known from the sampled pose with the weight left alone.  One sweep, records read twice: 69 us
// against the 48 us of pf_step_kernel at 262144 particles x 16 observations (the 84 MB of pass 1 at HBM speed).
// (Keeping the 16 records in registers between the passes, all requested up front, was measured: 166 VGPRs, three
//  waves per SIMD instead of six, and the step went from 107 to 122 us on the same box.)
template <typename T>
__global__ __launch_bounds__(256) void pf_proposal_kernel(T* __restrict__ pose, LmView<T> lv, const int32_t* __restrict__ tabs,
                                                           T* __restrict__ logw,
                                                           int64_t n, int64_t first, uint32_t step, uint64_t seed, T V, T G,
                                                           T wheelbase, T lq00, T lq10, T lq11, T dt,
                                                           const double* __restrict__ z, const int32_t* __restrict__ ids,
                                                           int m, T R00, T R10, T R01, T R11, double* __restrict__ part, T pend) {
    extern __shared__ double s_raw[];              // room for [m][2] doubles, then [m] codes, then [m] meta words
    T* s_obs = reinterpret_cast<T*>(s_raw);        // the observations in the state dtype: converted once per workgroup
    int32_t* s_ids = reinterpret_cast<int32_t*>(s_raw + 2 * m);
    int32_t* s_meta = s_ids + m;
    for (int i = threadIdx.x; i < 2 * m; i += blockDim.x) s_obs[i] = (T)z[i];
    for (int i = threadIdx.x; i < m; i += blockDim.x) { s_ids[i] = ids[i]; s_meta[i] = ids[PF_OCAP + i]; }
    __syncthreads();
    const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pi < n;
    const int64_t p = valid ? pi : n - 1;          // idle lanes shadow the last particle, stores are masked
    T xn, yn, pn, lw;
    proposal_core<T>(pose, lv, tabs, logw, n, first, step, seed, V, G, wheelbase, lq00, lq10, lq11, dt, s_obs, s_ids, s_meta, m,
                     R00, R10, R01, R11, pend, p, valid, xn, yn, pn, lw);
    block_weight_stats<T, false, false>(lw, xn, yn, pn, valid, 1, part);
}

// ---- N4: unknown correspondences --------------------------------------------------------------------
// Per-particle gated nearest neighbour over the particle's OWN landmark slots (a slot with Pxx < 0 holds no
// landmark): the rule of associate() (src/data-association.jl:1-51 in the order-independent form of SURVEY 3.2)
// with compute_association() (:53-63) restricted to the landmark's 2 x 2 block.  One thread per particle.
// Pass 1 sweeps the slots ONCE (coalesced: the particle index is the fastest one) and keeps, for each of the
// m <= UNK_MAX observations, the best candidate in registers; all observations are associated against the map as
// it is BEFORE this step's updates.  Pass 2 applies them in observation order: matched -> lm_update on that slot,
// new -> lm_init in the particle's lowest unused slot (none left: dropped).
constexpr int UNK_MAX = 16;

template <typename T>
__global__ __launch_bounds__(256) void pf_update_unknown_kernel(const T* __restrict__ pose, LmView<T> lv, int buf,
                                                                 T* __restrict__ logw, int64_t n, int nl,
                                                                 const double* __restrict__ z, int m, T R00, T R10, T R01,
                                                                 T R11, T gate1, T gate2, T pend,
                                                                 int32_t* __restrict__ assoc_out) {
    __shared__ double s_obs[2 * UNK_MAX];
    for (int i = threadIdx.x; i < 2 * m; i += blockDim.x) s_obs[i] = z[i];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const T x = pose[p], y = pose[n + p], phi = pose[2 * n + p];
    T lw = logw[p] - pend;
    const T INF = (T)__builtin_inf();
    T best_nd[UNK_MAX];
    int best_l[UNK_MAX];
    unsigned near = 0u;
#pragma unroll
    for (int i = 0; i < UNK_MAX; ++i) { best_nd[i] = INF; best_l[i] = -1; }
    for (int l = 0; l < nl; ++l) {
        const T* row = lv.rows(buf, l, n) + p;
        const T pxx = row[2 * n];
        if (pxx < (T)0) continue;
        const T lx = row[0], ly = row[n], pxy = row[3 * n], pyy = row[4 * n];
        const T dx = lx - x, dy = ly - y;
        const T d2 = dx * dx + dy * dy;
        const T d = sqrt(d2);
        const T zp1 = atan2(dy, dx) - phi;
        const T h00 = dx / d, h01 = dy / d, h10 = -dy / d2, h11 = dx / d2;      // src/common.jl:162
        const T t00 = pxx * h00 + pxy * h01, t01 = pxx * h10 + pxy * h11;
        const T t10 = pxy * h00 + pyy * h01, t11 = pxy * h10 + pyy * h11;
        const T s00 = h00 * t00 + h01 * t10 + R00;                              // S = Hf Pf Hf' + R (:59), not symmetrised
        const T s01 = h00 * t01 + h01 * t11 + R01;
        const T s10 = h10 * t00 + h11 * t10 + R10;
        const T s11 = h10 * t01 + h11 * t11 + R11;
        const T det = s00 * s11 - s01 * s10;
        const T rdet = (T)1 / det;
        const T qa = s11 * rdet, qb = -(s01 + s10) * rdet, qc = s00 * rdet;
        const T logdet = log(det);
#pragma unroll
        for (int i = 0; i < UNK_MAX; ++i) {
            if (i < m) {
                const T v0 = (T)s_obs[2 * i] - d;
                const T v1 = wrap_pi<T>((T)s_obs[2 * i + 1] - zp1);              // :57
                const T nis = qa * v0 * v0 + qb * v0 * v1 + qc * v1 * v1;        // :60
                const T nd = nis + logdet;                                       // :61
                if (nis < gate1 && nd < best_nd[i]) { best_nd[i] = nd; best_l[i] = l; }     // strict: lowest slot wins a tie
                if (nis <= gate2) near |= 1u << i;
            }
        }
    }
    int next_free = 0;                                   // unused slots are handed out in ascending order
#pragma unroll
    for (int i = 0; i < UNK_MAX; ++i) {
        if (i < m) {
            const int a = best_l[i] >= 0 ? best_l[i] : (((near >> i) & 1u) ? -2 : -1);
            if (assoc_out) assoc_out[(size_t)i * n + p] = a;
            const T r = (T)s_obs[2 * i], b = (T)s_obs[2 * i + 1];
            if (a >= 0) {
                T* row = lv.rows(buf, a, n) + p;
                const LmRow<T> cur = load_row<T>(row, n);
                lm_update<T>(row, n, cur, x, y, phi, r, b, R00, R10, R01, R11, true, lw);
            } else if (a == -1) {
                int slot = next_free;
                while (slot < nl && !(lv.rows(buf, slot, n)[2 * n + p] < (T)0)) ++slot;
                if (slot < nl) {
                    lm_init<T>(lv.rows(buf, slot, n) + p, n, x, y, phi, r, b, R00, R10, R01, R11, true);
                    next_free = slot + 1;
                }
            }
        }
    }
    logw[p] = lw;
}

template <typename T>
__global__ __launch_bounds__(256) void pf_clear_lm_kernel(LmView<T> lv, int buf, int64_t n, int nl) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    for (int l = 0; l < nl; ++l) {
        T* row = lv.rows(buf, l, n) + p;
        row[0] = (T)0; row[n] = (T)0; row[2 * n] = (T)-1; row[3 * n] = (T)0; row[4 * n] = (T)0;
    }
}

// ---- F4: reductions ------------------------------------------------------------------------------
// nw: waves taking part (0: all of the workgroup; the tail of the observation-parallel step kernel runs on four of eight)
__device__ inline double block_reduce(double v, double* sh, bool is_max, int nw) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(v, off);
        v = is_max ? fmax(v, o) : v + o;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (nw == 0) nw = (int)(blockDim.x >> 6);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double r = sh[0];
    for (int w = 1; w < nw; ++w) r = is_max ? fmax(r, sh[w]) : r + sh[w];
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void pf_stats_kernel(const T* __restrict__ logw, const T* __restrict__ pose, int64_t n,
                                                        int relative, double* __restrict__ part) {
    const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pi < n;
    const int64_t p = valid ? pi : n - 1;
    block_weight_stats<T>(logw[p], pose[p], pose[n + p], pose[2 * n + p], valid, relative, part);
}

// out = {M, sum, sum2, sx, sy, ss, sc} over all blocks: M = max_b m_b, block sums rescaled by exp(m_b - M)
// (its square for the second moment).  One workgroup.
__device__ __forceinline__ void fold_partials(const double* __restrict__ part, int nblocks, int relative,
                                              double* __restrict__ out, double* __restrict__ host_out, long long seq) {
    __shared__ double sh[4];
    // one pass over the partials: up to four records per thread stay in registers between the max and the sums
    // (more than 1024 partials: the remainder goes through the plain two-pass loop below)
    double q[4][7];
    double m = -__builtin_inf();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int b = threadIdx.x + 256 * u;
#pragma unroll
        for (int i = 0; i < 7; ++i) q[u][i] = b < nblocks ? part[(size_t)b * 8 + i] : (i == 0 ? -__builtin_inf() : 0.0);
        m = fmax(m, q[u][0]);
    }
    for (int b = threadIdx.x + 1024; b < nblocks; b += 256) m = fmax(m, part[(size_t)b * 8]);
    const double M = block_reduce(m, sh, true);
    double acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const double f = relative ? (q[u][0] == -__builtin_inf() ? 0.0 : exp(q[u][0] - M)) : 1.0;
        acc[0] += q[u][1] * f;
        acc[1] += q[u][2] * f * f;
        acc[2] += q[u][3] * f; acc[3] += q[u][4] * f; acc[4] += q[u][5] * f; acc[5] += q[u][6] * f;
    }
    for (int b = threadIdx.x + 1024; b < nblocks; b += 256) {
        const double* qq = part + (size_t)b * 8;
        const double f = relative ? exp(qq[0] - M) : 1.0;
        acc[0] += qq[1] * f;
        acc[1] += qq[2] * f * f;
        acc[2] += qq[3] * f; acc[3] += qq[4] * f; acc[4] += qq[5] * f; acc[5] += qq[6] * f;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = block_reduce(acc[i], sh, false);
    if (threadIdx.x == 0) {
        out[0] = M;
        for (int i = 0; i < 6; ++i) out[1 + i] = acc[i];
        // the host polls pinned memory for `seq` (no copy kernel, no event, no interrupt-driven wake-up)
        // write-through system-scope stores, drained, then the sequence word (a system-scope fence here is a write-back
        // of the XCD's L2, full of the sweep's dirty landmark records: it cost most of this kernel's 7 us)
        __hip_atomic_store(host_out, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = 0; i < 6; ++i) __hip_atomic_store(host_out + 1 + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(reinterpret_cast<long long*>(host_out + 7), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void pf_fold_kernel(const double* __restrict__ part, int nblocks, int relative,
                                                      double* __restrict__ out, double* __restrict__ host_out,
                                                      long long seq) {
    fold_partials(part, nblocks, relative, out, host_out, seq);
}

template <typename T>
__global__ __launch_bounds__(256) void pf_fill_kernel(T* __restrict__ a, int64_t n, T v) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) a[p] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void pf_shift_kernel(T* __restrict__ logw, int64_t n, T shift) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) logw[p] -= shift;
}

// ---- F4: systematic resampling over the GLOBAL weights -----------------------------------------------
constexpr int SCAN_BLOCK = 1024;

// per-block inclusive scan of w = exp(logw - max) (double) + block totals
template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void pf_scan1_kernel(const T* __restrict__ logw_all, int64_t n, double gmax,
                                                               double* __restrict__ cdf, double* __restrict__ bsum, T pend) {
    __shared__ double sh[SCAN_BLOCK];
    const int64_t i = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    // `pend`: a normalisation shift not yet applied to the stored values (rounded as pf_shift_kernel would store it)
    sh[threadIdx.x] = i < n ? exp((double)(T)(logw_all[i] - pend) - gmax) : 0.0;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        const double v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if (i < n) cdf[i] = sh[threadIdx.x];
    if (threadIdx.x == SCAN_BLOCK - 1) bsum[blockIdx.x] = sh[threadIdx.x];
}

// exclusive scan of the block totals, in place.  The additions run in index order on ONE thread (the oracle's
// order: the ancestor table must be exact), but out of LDS: loads and stores are done by the whole workgroup, so
// the serial part is ~10 cycles per block total instead of one L2 round trip.
constexpr int SCAN2_CHUNK = 4096;
__global__ __launch_bounds__(256) void pf_scan2_kernel(double* __restrict__ bsum, int nb) {
    __shared__ double sh[SCAN2_CHUNK];
    __shared__ double carry;
    if (threadIdx.x == 0) carry = 0.0;
    for (int base = 0; base < nb; base += SCAN2_CHUNK) {
        const int cnt = nb - base < SCAN2_CHUNK ? nb - base : SCAN2_CHUNK;
        for (int i = threadIdx.x; i < cnt; i += 256) sh[i] = bsum[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            double run = carry;
            for (int i = 0; i < cnt; ++i) {
                const double v = sh[i];
                sh[i] = run;
                run += v;
            }
            carry = run;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) bsum[base + i] = sh[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = carry;                      // grand total
}

// ancestor of global slot g = first j with cdf[j] >= (g + u0)/N * total   (binary search)
__global__ __launch_bounds__(256) void pf_ancestor_kernel(const double* __restrict__ cdf, const double* __restrict__ bsum,
                                                           int nb, int64_t n_global, int64_t first, int64_t n, double u0,
                                                           int32_t* __restrict__ anc) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double total = bsum[nb];
    const double target = ((double)(first + p) + u0) / (double)n_global * total;
    int64_t lo = 0, hi = n_global - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const double c = cdf[mid] + bsum[mid / SCAN_BLOCK];
        if (c >= target) hi = mid; else lo = mid + 1;
    }
    anc[p] = (int32_t)lo;
}

// gather source per local slot: >= 0 local index, < 0: -(position in the sorted remote id list + 1)
__global__ __launch_bounds__(256) void pf_src_kernel(const int32_t* __restrict__ anc, int64_t n, int64_t first,
                                                      const int32_t* __restrict__ remote_ids, int nremote,
                                                      int32_t* __restrict__ src) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t a = anc[p];
    if (a >= first && a < first + n) { src[p] = (int32_t)(a - first); return; }
    int lo = 0, hi = nremote - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (remote_ids[mid] >= a) hi = mid; else lo = mid + 1;
    }
    src[p] = -(lo + 1);
}

// new[row][p] = old[row][src] or remote[row][pos].  A thread owns one particle and GATHER_ROWS consecutive rows
// (grid.y = row chunks): the source index is read once and the row loop keeps eight independent loads in flight.
constexpr int GATHER_ROWS = 64;
template <typename T>
__global__ __launch_bounds__(256) void pf_gather_kernel(const T* __restrict__ pose_old, LmView<T> lv, int bold,
                                                         T* __restrict__ pose_new, int bnew, int64_t n,
                                                         int nrows, const int32_t* __restrict__ src,
                                                         const T* __restrict__ remote, int nremote) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t s = src[p];
    const int row0 = blockIdx.y * GATHER_ROWS;
    const int row1 = row0 + GATHER_ROWS < nrows ? row0 + GATHER_ROWS : nrows;
    if (s >= 0) {
        int row = row0;
        for (; row < 3 && row < row1; ++row) pose_new[(size_t)row * n + p] = pose_old[(size_t)row * n + s];
        // (row - 3 of the flat [5 nl][n] view: a chunk boundary may fall inside the block, so every row finds its own chunk)
        for (; row + 8 <= row1; row += 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = lv.flat_row(bold, row - 3 + u, n)[s];
#pragma unroll
            for (int u = 0; u < 8; ++u) lv.flat_row(bnew, row - 3 + u, n)[p] = v[u];
        }
        for (; row < row1; ++row) lv.flat_row(bnew, row - 3, n)[p] = lv.flat_row(bold, row - 3, n)[s];
    } else {
        const T* rr = remote + (size_t)(-s - 1);
        for (int row = row0; row < row1; ++row) {
            T* new_row = row < 3 ? pose_new + (size_t)row * n : lv.flat_row(bnew, row - 3, n);
            new_row[p] = rr[(size_t)row * nremote];
        }
    }
}

// records[row][c] = state[row][idx[c]]
template <typename T>
__global__ __launch_bounds__(256) void pf_pack_kernel(const T* __restrict__ pose, LmView<T> lv, int buf, int64_t n,
                                                       const int32_t* __restrict__ idx, int cnt, T* __restrict__ rec) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cnt) return;
    const int row = blockIdx.y;
    const T* src_row = row < 3 ? pose + (size_t)row * n : lv.flat_row(buf, row - 3, n);
    rec[(size_t)row * cnt + c] = src_row[idx[c]];
}

// ---- lazy resampling ------------------------------------------------------------------------------------
// When the whole filter lives on this shard, resampling does not copy the particles' maps (2.7 GB per step at
// 262144 x 512).  It gathers the POSES and composes one small table per group of landmarks: landmark l's record of
// particle p is found in buffer lbuf[l] at slot tab[ltab[l]][p].  With known correspondences every particle updates
// the SAME landmarks in a call, so an update reads through the table, writes the particle's own slot of the OTHER
// buffer, and the landmark is "identity" again; landmarks that were identity at a resampling share the new table
// (= the ancestor vector), older tables are composed with it (tab'[p] = tab[anc[p]]) and die when their last
// landmark is updated.  At m observations per call there are about nl / m live tables: a resampling step moves
// megabytes instead of gigabytes.  Everything that wants plain maps (download, pack, the unknown-correspondence
// sweep, a sharded filter's record exchange) calls pf_materialise first: the eager gather, landmark by landmark.
template <typename T>
__global__ __launch_bounds__(256) void pf_pose_gather_kernel(const T* __restrict__ pose_old, T* __restrict__ pose_new, int64_t n,
                                                              const int32_t* __restrict__ anc) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t a = anc[p];
#pragma unroll
    for (int r = 0; r < 3; ++r) pose_new[(size_t)r * n + p] = pose_old[(size_t)r * n + a];
}

struct TabList {
    int32_t count;               // live tables to compose
    int32_t fresh;               // index of the new table (= anc), or -1
    int16_t idx[PF_TAB_MAX];
};

__global__ __launch_bounds__(256) void pf_compose_kernel(const int32_t* __restrict__ tin, int32_t* __restrict__ tout, int64_t n,
                                                          const int32_t* __restrict__ anc, TabList tl) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t a = anc[p];
    if (tl.fresh >= 0) tout[(size_t)tl.fresh * n + p] = a;
    for (int i = 0; i < tl.count; ++i) {
        const size_t t = (size_t)tl.idx[i];
        tout[t * n + p] = tin[t * n + a];
    }
}

// the three per-particle pieces of a lazy resampling step in one launch: poses, tables, uniform weights
template <typename T>
__global__ __launch_bounds__(256) void pf_lazy_apply_kernel(const T* __restrict__ pose_old, T* __restrict__ pose_new,
                                                             const int32_t* __restrict__ tin, int32_t* __restrict__ tout,
                                                             T* __restrict__ logw, int64_t n, const int32_t* __restrict__ anc,
                                                             TabList tl, T lw) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t a = anc[p];
#pragma unroll
    for (int r = 0; r < 3; ++r) pose_new[(size_t)r * n + p] = pose_old[(size_t)r * n + a];
    if (tl.fresh >= 0) tout[(size_t)tl.fresh * n + p] = a;
    for (int i = 0; i < tl.count; ++i) {
        const size_t t = (size_t)tl.idx[i];
        tout[t * n + p] = tin[t * n + a];
    }
    logw[p] = lw;
}

// work[l]: -1 nothing to do, else (table + 1) | source buffer << 8 | destination buffer << 9
constexpr int MAT_LMS = 12;      // landmarks per thread
// SH (sharded filter with peers): a table entry is a global particle id; a remote ancestor's record is read from its owner.
template <typename T, bool SH>
__global__ __launch_bounds__(256) void pf_materialise_kernel(LmView<T> lv, const int32_t* __restrict__ tabs, int64_t n, int nl,
                                                              const int32_t* __restrict__ work, PfShardCtx sc) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int l0 = blockIdx.y * MAT_LMS, l1 = l0 + MAT_LMS < nl ? l0 + MAT_LMS : nl;
    for (int l = l0; l < l1; ++l) {
        const int32_t w = work[l];
        if (w < 0) continue;                                   // (uniform)
        const int t = w & META_TAB;
        int64_t slot = t ? (int64_t)tabs[(size_t)(t - 1) * n + p] : p;
        const T* src = lv.rows((w & META_RBUF) ? 1 : 0, l, n);
        bool remote = false;
        if constexpr (SH) {
            if (t) {
                const uint32_t owner = pf_owner((uint32_t)slot, sc.n, sc.world);
                remote = owner != (uint32_t)sc.rank;
                slot -= (int64_t)owner * sc.n;
                if (remote) src = LmView<T>{&sc.peers->lm[owner]}.rows((w & META_RBUF) ? 1 : 0, l, n);
            }
        }
        src += slot;
        T* dst = lv.rows((w & META_WBUF) ? 1 : 0, l, n) + p;
        T v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if constexpr (SH) v[c] = remote ? ld_sys(src + (size_t)c * n) : src[(size_t)c * n];
            else v[c] = src[(size_t)c * n];
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) dst[(size_t)c * n] = v[c];
    }
}


// ---- auto mode -------------------------------------------------------------------------------------------------------
// slam_pf_step_auto: a filter step that needs NO answer from the host.  What the legacy entry points keep on the host
// -- the folded weight statistics, the normalisation shift, Neff and the decision to resample, the bookkeeping of the
// lazy resampling (which buffer and which ancestor table holds each landmark), the live pose buffer -- lives in a
// device-resident control block (PfCtl) and a per-landmark state word:
//   * every workgroup of the step kernel plans the observation codes itself, in LDS, from the state words (the host's
//     pf_stage, a few dozen integer operations);
//   * the per-block weight statistics are stored write-through (sc1) as one tagged 64-byte line per workgroup and the
//     workgroup that is dispatched LAST collects them (polling until every line's tag fits, see part_key), folds them,
//     forms shift / Neff / the decision, applies the state transitions of this step's observations and, if the filter
//     resamples and lives wholly on this shard, prepares the lazy resampling (table list, fresh table, buffer flips)
//     -- pf_auto_tail.  No release/acquire fence (an L2 write-back + invalidate on this multi-XCD part), no drain of
//     the storing waves and no arrival counter is involved;
//   * two conditional kernels follow every step (cdf; ancestors + lazy apply) and return at once unless the control
//     block says that THIS step resamples.
// A sharded filter (or an exhausted table pool) cannot resample on the device: the tail then records a HALT, the steps
// already queued behind it return without touching anything, and the host -- which notices at its next call -- does the
// resampling the legacy way and re-enqueues the skipped steps from its log.  The ranks of a sharded filter exchange
// their three scalars (max, sum w, sum w^2) through a page of pinned host memory that every rank's GPU can write and
// poll: no host in the loop, no collective launch per step.

template <typename T>
__device__ __forceinline__ T ld_sc1(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the host's philox_uniform(step, stream, seed) (pf.py): counter (0, 0, step, stream)
__host__ __device__ inline double resample_offset(uint32_t count, uint64_t seed) {
    uint32_t r[4];
    philox(0u, 0u, count, 2u /* STREAM_RESAMPLE */, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return ((double)(r[0] >> 8) + 0.5) * (1.0 / 16777216.0);
}

struct PfAutoArgs {
    void *pose0, *pose1, *logw0, *logw1;
    const PfLmTab* lmtab;        // the landmark records' chunk table (device memory)
    int32_t *tab0, *tab1;
    long long n, first, n_global, seq;
    unsigned long long seed;
    unsigned int step;
    int m, nl, force, lazy_ok, rank, world, publish, rec_cap;
    double V, G, wheelbase, a0, a1, a2, dt, R00, R10, R01, R11, neff_frac;
    double* part;
    PfCtl* ctl;
    int32_t* lmstate;
    PfMirror* mir;
    double* xchg;
    const PfPeers* peers;        // sharded filter with peers attached (else null)
    PfInbox* inbox;              // this rank's inbox
    // The step's observations travel IN the kernel arguments (1.3 KB of the 4 KB a launch may carry): every one of the
    // ~1000 workgroups reads them at its start, and from a pinned host page (the zero-copy staging of the legacy
    // calls) that is ~5000 64-byte reads across PCIe per step -- measured: 19 us of a 46 us kernel before the first
    // landmark record moves.  The argument segment is read through the scalar/L2 caches like any other constant.
    double z[2 * PF_AUTO_MAXOBS];
    int32_t ids[PF_AUTO_MAXOBS];
};
static_assert(sizeof(PfAutoArgs) <= 4096, "kernel argument segment");

// The planning of pf_stage on the device: observation i of landmark l = ids[i] - 1 gets its code (landmark, first
// sighting / repeat of a first sighting) and its meta word (where the record is read and written) from the landmark's
// state word; a repeat inside the call sees the state its first occurrence leaves behind.
// (l, st: thread i < m holds observation i's landmark and its state word, loaded by the caller ahead of time)
__device__ __forceinline__ void plan_obs(int l_mine, int32_t st_mine, int m, int32_t* s_l,
                                         int32_t* s_st, int32_t* s_ids, int32_t* s_meta, int32_t* s_first) {
    const int tid = threadIdx.x;
    if (tid < m) {
        s_l[tid] = l_mine;
        s_st[tid] = st_mine;
    }
    __syncthreads();
    if (tid < m) {
        const int l = s_l[tid];
        int j0 = tid;
        for (int j = 0; j < tid; ++j)
            if (s_l[j] == l) { j0 = j; break; }
        const int32_t st = s_st[tid];
        const int tab = st & LS_TAB, rb = (st & LS_BUF) ? 1 : 0;
        const int wb = tab ? (rb ^ 1) : rb;                  // behind a table the update goes to the OTHER buffer
        const int32_t prior = tab | (rb ? META_RBUF : 0);
        int32_t code, meta;
        if (j0 == tid) {
            code = l | ((st & LS_SEEN) ? 0 : NEW_FLAG);
            meta = tab | (rb ? META_RBUF : 0) | (wb ? META_WBUF : 0);
        } else {                                             // the first occurrence has made the landmark (buffer wb, identity)
            code = l | ((st & LS_SEEN) ? 0 : FRESH_FLAG);
            meta = (wb ? META_RBUF : 0) | (wb ? META_WBUF : 0);
        }
        s_ids[tid] = code;
        s_meta[tid] = meta | (prior << META_PRIOR_SHIFT);
        s_first[tid] = j0 == tid;
    }
    __syncthreads();
}

// the first NS of six sums at once: one LDS exchange and one barrier pair for all of them (256 threads); the result reaches every thread
template <int NS = 6>
__device__ __forceinline__ void block_reduce6(double (&v)[6], double (*sh6)[6]) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NS; ++i) sh6[wave][i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) v[i] = sh6[0][i] + sh6[1][i] + sh6[2][i] + sh6[3][i];
}

// has a peer announced that it is going away?  (uniform: the words sit in this rank's own inbox)
__device__ __forceinline__ bool pf_peer_gone(const PfInbox* inbox, int world) {
    unsigned long long g = 0;
    for (int r = 0; r < world; ++r) g |= __hip_atomic_load(&inbox->gone[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return g != 0;
}

// One lane writes a step's outcome to the host's mirror (pinned memory).
__device__ __forceinline__ void pf_publish(PfMirror* mir, double neff, long long nresamples, long long resampled_seq, int error,
                                           long long halt_seq, long long seq) {
    __hip_atomic_store(&mir->neff, neff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mir->nresamples, nresamples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mir->resampled_seq, resampled_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (error) __hip_atomic_store(&mir->error, (long long)error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (halt_seq) {
        __hip_atomic_store(&mir->halt_seq, halt_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_store(&mir->done_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// slam_pf_flush's request: the outcome of the LAST completed step, whatever its number.  (After a halt nothing is to be
// said: the halting step has published itself and the steps behind it were skipped.)
__global__ void pf_auto_publish_kernel(const PfCtl* __restrict__ ctl, PfMirror* mir) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || ctl->halt_seq != 0) return;
    pf_publish(mir, ctl->stats[7], (long long)ctl->nresamples, ctl->resample_seq, ctl->error, 0ll, ctl->seq);
}

// Runs in the launch's last workgroup, after its own share of the sweep (all 256 threads).  s_l / s_st / s_first: this step's plan (LDS).
template <typename T>
__device__ __forceinline__ void pf_auto_tail(const PfAutoArgs& a, const int32_t* s_l, const int32_t* s_st, const int32_t* s_first,
                                             int pcur, int tside, int lwcur, int line_level) {
    // line_level: what a statistics line of this launch is -- 0: the tree's leaf (64 particles, observation-parallel kernel),
    // 1: the 256-particle node (the sweep kernels)
    __shared__ double s_w[4][3];
    __shared__ double s_pass[16][3];              // the tree nodes above each pass of 1024 lines / records
    __shared__ double s_g[12];
    __shared__ double s_rv[PF_MAX_WORLD][3];      // the ranks' records of the legacy scalar exchange (thread 0)
    __shared__ int s_tref[PF_TAB_MAX];
    __shared__ int s_i[4];          // [0] identity landmarks, [1] outcome (0 none, 1 lazy resampling, 2 halt), [2] fresh table
    const int tid = threadIdx.x;
    PfCtl* ctl = a.ctl;
    const int nblocks = (int)gridDim.x;
    // Every global load of the tail -- the table reference counts, this thread's landmark state words, its share of the
    // partials -- is issued up front: the memory system is still draining the sweep's stores and a load takes microseconds
    // to come back, so the tail pays that latency once, not once per phase.
    const int my_tref = tid < PF_TAB_MAX ? ctl->tref[tid] : 0;
    const int identity0 = ctl->identity;                      // landmarks without a table before this step
    const int nres0 = ctl->nresamples;                        // (requested here, with the rest: used by the bookkeeping only)
    const long long res0 = ctl->resample_seq;
    const unsigned long long key = part_key(a.seq);
    const unsigned long long t_poll = wall_clock64();
    if (tid == 0) ctl->stamps[6] = t_poll;                    // the collecting workgroup has done its own share
    int rounds = 0;
    __shared__ int s_perr;
    if (tid == 0) s_perr = 0;
    if (tid < PF_TAB_MAX) s_tref[tid] = my_tref;
    if (tid == 0) s_i[0] = identity0;
    __syncthreads();
    // Four consecutive records {m, s1, s2, tag} starting at `first` (64 bytes apart in `src`), polled until their tags fit
    // (see part_key: a line that is stale, half written or torn does not fit; nothing else orders the stores).  Records from
    // `count` on read as absent.  SYS: written by peer GPUs into this rank's inbox (system-scope loads), else by this launch's
    // workgroups (agent scope).
    auto collect4 = [&](const double* src, int stride_d, int first, int count, unsigned long long kkey, bool sys, WRec (&q)[4]) {
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = first + u;
                q[u] = wrec_empty();
                if (b < count) {
                    const double* o = src + (size_t)b * stride_d;
                    unsigned long long tag;
                    if (sys) {
                        q[u].m = __hip_atomic_load(o + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        q[u].s1 = __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        q[u].s2 = __hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        tag = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(o + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    } else {
                        q[u].m = ld_sc1(o + 0); q[u].s1 = ld_sc1(o + 1); q[u].s2 = ld_sc1(o + 2);
                        tag = ld_sc1(reinterpret_cast<const unsigned long long*>(o + 3));
                    }
                    if ((wrec_hash(q[u]) ^ tag) != (kkey ^ (sys ? (unsigned long long)b * 0xD6E8FEB86659FD93ull : 0ull))) ok = false;
                }
            }
            ++rounds;
            if (ok) break;
            __builtin_amdgcn_s_sleep(8);
            // 2 s (lines of this launch) / 20 s (a rank's records: that rank is gone) at 100 MHz: give up, report
            if (wall_clock64() - t_poll > (sys ? 2000000000ull : 200000000ull)) { atomicOr(&s_perr, sys ? PF_ERR_EXCHANGE : PF_ERR_HANDOVER); break; }
        }
    };
    // ---- this rank's lines -> the tree's 1024-particle records (C) and above, 1024 lines per pass ----
    const bool xpeers = a.world > 1 && a.peers != nullptr;    // sharded with peers: the 1024-particle records go to every rank
    const int lines_per_c = line_level == 0 ? 16 : 4;
    const int nc_local = (int)((a.n + 1023) / 1024);
    const int par = (int)(a.seq & 1);
    const unsigned long long xkey = key ^ 0x5851F42D4C957F2Dull;
    const int npass_l = (nblocks + 1023) / 1024;
    for (int ps = 0; ps < npass_l; ++ps) {
        WRec q[4];
        collect4(a.part, 8, 1024 * ps + 4 * tid, nblocks, key, false, q);
        WRec r = wrec_combine4(q[0], q[1], q[2], q[3]);
        int stride = 1;
        if (line_level == 0) {                                // leaves: one more level to reach the 1024-particle record
            WRec b, c, d;
            b.m = __shfl_down(r.m, 1); b.s1 = __shfl_down(r.s1, 1); b.s2 = __shfl_down(r.s2, 1);
            c.m = __shfl_down(r.m, 2); c.s1 = __shfl_down(r.s1, 2); c.s2 = __shfl_down(r.s2, 2);
            d.m = __shfl_down(r.m, 3); d.s1 = __shfl_down(r.s1, 3); d.s2 = __shfl_down(r.s2, 3);
            r = wrec_combine4(r, b, c, d);
            stride = 4;
        }
        if (ps == 0 && tid == 0) ctl->stamps[1] = wall_clock64();      // (first pass: every workgroup's statistics are in)
        if (xpeers) {
            // r (threads with tid % stride == 0) is the record of local block cj: into every rank's inbox, tagged
            const int cj = (1024 * ps) / lines_per_c + tid / stride;
            if (tid % stride == 0 && cj < nc_local) {
                const int gi = a.rank * nc_local + cj;
                const unsigned long long tag = wrec_hash(r) ^ xkey ^ ((unsigned long long)gi * 0xD6E8FEB86659FD93ull);
                for (int rr = 0; rr < a.world; ++rr) {
                    double* o = pf_inbox_recs(a.peers->inbox[rr]) + ((size_t)par * a.rec_cap + gi) * 4;
                    __hip_atomic_store(o + 0, r.m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(o + 1, r.s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(o + 2, r.s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(o + 3), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        } else {
            const WRec pr = wrec_tree256(r, stride, s_w);
            if (tid == 0 && ps < 16) { s_pass[ps][0] = pr.m; s_pass[ps][1] = pr.s1; s_pass[ps][2] = pr.s2; }
        }
    }
    int npass = npass_l;
    // ---- this step's state transitions (what pf_stage does on the host): one thread per observation; the state word
    //      of every observed landmark is still in LDS from the plan.  Written only HERE, after the LAST local collect: every
    //      workgroup of the launch has then stored its statistics line, i.e. has long finished planning from the state
    //      words (a grid of more than 1024 workgroups is not resident at once: a workgroup beyond the first 1024 lines
    //      may not even have started when the first pass returns). ----
    if (tid < a.m && s_first[tid]) {
        const int32_t st = s_st[tid];
        const int tab = st & LS_TAB, rb = (st & LS_BUF) ? 1 : 0;
        if (tab) {
            atomicSub(&s_tref[tab - 1], 1);
            atomicAdd(&s_i[0], 1);                            // released its table: a landmark without one ("identity")
        }
        a.lmstate[s_l[tid]] = LS_SEEN | ((tab ? (rb ^ 1) : rb) ? LS_BUF : 0);
    }
    if (xpeers) {
        // ---- every rank's records, as they arrive in THIS rank's inbox (local memory, written by the peers over xGMI): the
        //      all-gather of the step's statistics.  Every rank reduces the same sequence with the same tree. ----
        const int nc_global = a.world * nc_local;
        const double* recs = pf_inbox_recs(a.inbox) + (size_t)par * a.rec_cap * 4;
        npass = (nc_global + 1023) / 1024;
        for (int ps = 0; ps < npass; ++ps) {
            WRec q[4];
            collect4(recs, 4, 1024 * ps + 4 * tid, nc_global, xkey, true, q);
            const WRec pr = wrec_tree256(wrec_combine4(q[0], q[1], q[2], q[3]), 1, s_w);
            if (tid == 0 && ps < 16) { s_pass[ps][0] = pr.m; s_pass[ps][1] = pr.s1; s_pass[ps][2] = pr.s2; }
        }
    }
    __syncthreads();
    if (tid == 0) ctl->stamps[7] = ctl->stamps[0] + 100ull * (unsigned long long)rounds;      // (diagnostic: polls of thread 0)
    if (tid == 0) {
        ctl->stamps[2] = wall_clock64();
        // the passes' nodes -> the root, still the radix-4 tree (absent children are the identity), in place in LDS (a register
        // array here would raise the whole step kernel's allocation)
        int cnt = npass < 16 ? npass : 16;
        auto pget = [&](int k) { return k < cnt ? WRec{s_pass[k][0], s_pass[k][1], s_pass[k][2]} : wrec_empty(); };
#pragma unroll 1
        while (cnt > 1) {
            const int nn = (cnt + 3) / 4;
#pragma unroll 1
            for (int k = 0; k < nn; ++k) {
                const WRec g = wrec_combine4(pget(4 * k), pget(4 * k + 1), pget(4 * k + 2), pget(4 * k + 3));
                s_pass[k][0] = g.m; s_pass[k][1] = g.s1; s_pass[k][2] = g.s2;
            }
            cnt = nn;
        }
        WRec root = cnt == 1 ? WRec{s_pass[0][0], s_pass[0][1], s_pass[0][2]} : wrec_empty();
        int err = s_perr;                                     // a workgroup's statistics never came: halt and report
        if (a.world > 1 && !xpeers && !err) {
            // the LEGACY exchange (no peers attached): every rank's root {m, s1, s2} through one page of pinned host memory that
            // every rank has mapped, two parities (a rank is at most one step ahead of the slowest); a record is three values
            // and a TAG = hash of their bit patterns xor key(step): the reader accepts a record only when its tag fits, so a
            // record that is stale, half arrived or torn is simply polled again.  The ranks' roots are combined in rank order:
            // NOT the canonical tree (a rank's root is not a node of it unless the slices are powers of four): this path agrees
            // with the one-rank filter to a few ulp, the peer path bit for bit.
            double* mine = a.xchg + ((size_t)par * a.world + a.rank) * 8;
            __hip_atomic_store(mine + 0, root.m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mine + 1, root.s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mine + 2, root.s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine + 3), wrec_hash(root) ^ xkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const double* page = a.xchg + (size_t)par * a.world * 8;
            const unsigned long long t0 = wall_clock64();
            double (*rv)[3] = s_rv;            // (LDS: a dynamically indexed local array would put the whole kernel on scratch)
            for (int r = 0; r < a.world && !err; ++r) {
                const double* slot = page + (size_t)r * 8;
                for (;;) {
                    rv[r][0] = __hip_atomic_load(slot + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    rv[r][1] = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    rv[r][2] = __hip_atomic_load(slot + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned long long tag = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(slot + 3), __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((wrec_hash(WRec{rv[r][0], rv[r][1], rv[r][2]}) ^ tag) == xkey) break;
                    __builtin_amdgcn_s_sleep(20);
                    if (wall_clock64() - t0 > 2000000000ull) { err = PF_ERR_EXCHANGE; break; }     // 20 s at 100 MHz: a rank is gone
                }
            }
            if (!err) {
                root = wrec_empty();
                for (int r = 0; r < a.world; ++r) root = wrec_combine4(root, WRec{rv[r][0], rv[r][1], rv[r][2]}, wrec_empty(), wrec_empty());
            }
        }
        // root: m = the largest log-weight, s1 = sum exp(logw - K ln 2), s2 = sum of its squares, K = ceil(m / ln 2)
        const double gM = root.m, gs1 = root.s1, gs2 = root.s2;
        const double kshift = wrec_k(gM) * PF_LN2;
        const double lg = log(gs1);
        s_g[0] = gM; s_g[1] = gs1; s_g[2] = gs2;
        s_g[3] = kshift + lg;                                               // the normalisation shift = log sum exp(logw)
        s_g[4] = gs1 * gs1 / gs2;                                           // Neff
        s_g[5] = (double)((T)gM - (T)(kshift + lg));                        // the largest log-weight after the shift, as stored
        // (a failed step: outcome 2 = halt, with the error code in the control block; its statistics are not published)
        const int want = err ? 1 : (a.force >= 0 ? a.force : (s_g[4] < a.neff_frac * (double)a.n_global ? 1 : 0));
        s_i[1] = want ? ((a.lazy_ok && !err) ? 1 : 2) : 0;
        s_i[2] = -1;
        s_i[3] = err;
    }
    __syncthreads();
    if (tid == 0) ctl->stamps[3] = wall_clock64();
    if (tid == 0 && s_i[1] == 1) {
        // lazy resampling: landmarks without a table share a fresh one (= the ancestor vector); live tables are composed
        int count = 0, free_idx = -1;
        for (int t = 0; t < PF_TAB_MAX; ++t) {
            if (s_tref[t] > 0) ctl->tl_idx[count++] = t;
            else if (free_idx < 0) free_idx = t;
        }
        if (s_i[0] > 0 && free_idx < 0) s_i[1] = 2;                         // no table left: the host resamples eagerly
        else {
            ctl->tl_count = count;
            ctl->tl_fresh = s_i[0] > 0 ? free_idx : -1;
            s_i[2] = s_i[0] > 0 ? free_idx : -1;
            if (s_i[0] > 0) s_tref[free_idx] = s_i[0];
        }
    }
    __syncthreads();
    if (s_i[1] == 1 && s_i[2] >= 0)
        for (int l = tid; l < a.nl; l += 256) {
            const int32_t st = a.lmstate[l];
            if ((st & LS_TAB) == 0) a.lmstate[l] = st | (s_i[2] + 1);
        }
    for (int t = tid; t < PF_TAB_MAX; t += 256) ctl->tref[t] = s_tref[t];
    if (tid == 0) {
        ctl->stamps[4] = wall_clock64();
        const int outcome = s_i[1];
        if (!s_i[3]) {                     // (a failed hand-over / exchange: the numbers are partial and are not recorded)
            ctl->stats[0] = s_g[0]; ctl->stats[1] = s_g[1]; ctl->stats[2] = s_g[2];
            ctl->stats[3] = ctl->stats[4] = ctl->stats[5] = ctl->stats[6] = 0.0;      // (not formed by a step: slam_pf_mean_pose_sums)
            ctl->stats[7] = s_g[4];
            ctl->shift_scan = s_g[3];
            ctl->gmax_norm = s_g[5];
            ctl->shift_next = outcome == 1 ? 0.0 : s_g[3];                  // a resampling leaves uniform weights behind
        } else {
            ctl->error = s_i[3];
        }
        ctl->seq = a.seq;
        ctl->identity = outcome == 1 ? 0 : s_i[0];                          // (a lazy resampling gives every landmark a table)
        if (outcome == 1) {
            ctl->u0 = resample_offset((uint32_t)nres0, a.seed);
            ctl->nresamples = nres0 + 1;
            ctl->pcur = pcur ^ 1;
            ctl->tside = tside ^ 1;
            ctl->lwcur = lwcur ^ 1;
            ctl->resample_seq = a.seq;
        } else if (outcome == 2) {
            ctl->halt_seq = a.seq;
        }
        // What the host may read without synchronising -- only every PF_PUBLISH_EVERY-th step, a halting or a failing one
        // (a.publish): two dependent PCIe writes at the very end of the kernel are ~2 us of every step otherwise, and the
        // host needs the mirror only to recycle its log (a quarter of the log's depth is granularity enough) and in
        // slam_pf_flush, which asks for the last step with pf_auto_publish_kernel.
        // Write-through system-scope stores, drained, then the sequence number: no fence (a system-scope release would
        // write back this XCD's whole L2, which the sweep has just filled with dirty landmark records).
        if (a.publish || outcome == 2 || s_i[3])
            pf_publish(a.mir, s_g[4], (long long)(nres0 + (outcome == 1 ? 1 : 0)), outcome == 1 ? a.seq : res0, s_i[3],
                       outcome == 2 ? a.seq : 0ll, a.seq);
        ctl->stamps[5] = wall_clock64();
    }
}

// (four waves per SIMD: at C4 the whole grid -- 1024 workgroups -- is then resident at once; one register more than 128
//  and a quarter of the workgroups start when the first ones end, which was measured as +10 us per step)
template <typename T, bool PROPOSAL, bool SH>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 4 : 2) void pf_auto_step_kernel(PfAutoArgs a) {
    PfCtl* ctl = a.ctl;
    PF_XS(0);
    PF_WG(0);
#if defined(SLAMHIP_EXPERIMENTS) && defined(PF_EXP_STAMPS)
    if (blockIdx.x == 0 && threadIdx.x == 0) g_xs[7] = wall_clock64();
#endif
    // the observed landmarks' state words are requested FIRST (their addresses need the kernel arguments only): the plan,
    // and with it the first record requests, then waits for one round trip (control block and state words together), not two
    typedef const __attribute__((address_space(4))) PfAutoArgs* KargPtr0;
    const KargPtr0 ka0 = (KargPtr0)__builtin_amdgcn_kernarg_segment_ptr();
    int l_pre = 0;
    int32_t st_pre = 0;
    if ((int)threadIdx.x < a.m) {
        l_pre = ka0->ids[threadIdx.x] - 1;
        st_pre = a.lmstate[l_pre];
    }
    // the control words this step needs, in one go (one cache line, one round trip)
    const long long halted = ctl->halt_seq;
    const int pcur = ctl->pcur, tside = ctl->tside, lwcur = ctl->lwcur;
    const double shift_next = ctl->shift_next;
    if (halted != 0 || ctl->error != 0) return;        // an earlier step waits for the host (which replays this one), or failed
    if constexpr (SH) {
        // a peer is destroying its handle: touch none of its memory (sweep_load, the inbox writes); the filter is dead
        if (pf_peer_gone(a.inbox, a.world)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                ctl->error = PF_ERR_PEER;
                pf_publish(a.mir, 0.0, (long long)ctl->nresamples, ctl->resample_seq, PF_ERR_PEER, a.seq, a.seq);
            }
            return;
        }
    }
    PF_XS(1);
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->stamps[0] = wall_clock64();
    __shared__ T s_obs[2 * PF_AUTO_MAXOBS];          // the observations in the state dtype: converted once per workgroup
    __shared__ int32_t s_ids[PF_AUTO_MAXOBS], s_meta[PF_AUTO_MAXOBS], s_l[PF_AUTO_MAXOBS], s_st[PF_AUTO_MAXOBS], s_first[PF_AUTO_MAXOBS];
    const int m = a.m;
    // the observations: read from the argument segment itself (constant address space, dynamic index) -- going through
    // the by-value copy `a` would put the whole 1.3 KB struct into scratch memory
    typedef const __attribute__((address_space(4))) PfAutoArgs* KargPtr;
    const KargPtr ka = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    for (int i = threadIdx.x; i < 2 * m; i += blockDim.x) s_obs[i] = (T)ka->z[i];
    const T pend = (T)shift_next;
    T* pose = (T*)(pcur ? a.pose1 : a.pose0);
    T* logw = (T*)(lwcur ? a.logw1 : a.logw0);
    const int32_t* tabs = tside ? a.tab1 : a.tab0;
    const int64_t n = a.n;
    const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pi < n;
    const int64_t p = valid ? pi : n - 1;              // idle lanes shadow the last particle, stores are masked
    PfShardCtx sc{};
    if constexpr (SH) sc = PfShardCtx{a.peers, (uint32_t)a.first, (uint32_t)a.n, a.rank, a.world};
    // the particle's pose and weight are requested, and its noise drawn, BEFORE the plan's two barriers and its dependent
    // loads (ids -> state words): the motion model then starts as soon as the plan stands
    T x = 0, y = 0, phi = 0, lw = 0, e1 = 0, e2 = 0;
    if (!PROPOSAL) {
        x = pose[p]; y = pose[n + p]; phi = pose[2 * n + p];
        lw = logw[p];
        normals2<T>((uint64_t)(a.first + p), a.step, STREAM_PREDICT, a.seed, e1, e2);
    }
    plan_obs(l_pre, st_pre, m, s_l, s_st, s_ids, s_meta, s_first);
    PF_XS(2);
    if (PROPOSAL)
        proposal_core<T, SH>(pose, LmView<T>{a.lmtab}, tabs, logw, n, a.first, a.step, a.seed, (T)a.V, (T)a.G, (T)a.wheelbase, (T)a.a0,
                             (T)a.a1, (T)a.a2, (T)a.dt, s_obs, s_ids, s_meta, m, (T)a.R00, (T)a.R10, (T)a.R01, (T)a.R11, pend, p, valid, x,
                             y, phi, lw, sc);
    else
        step_core<T, true, true, SH>(pose, LmView<T>{a.lmtab}, tabs, logw, n, a.first, a.step, a.seed, (T)a.V, (T)a.G,
                                     (T)a.wheelbase, (T)a.a0, (T)a.a1, (T)a.dt, s_obs, s_ids, s_meta, m, (T)a.R00, (T)a.R10, (T)a.R01,
                                     (T)a.R11, pend, p, valid, x, y, phi, lw, e1, e2, sc);
    PF_XS(4);
    PF_WG(1);
    wrec_block_line<T>(lw, valid, a.part, a.seq);          // the tree's 256-particle node as a tagged line, not waited for
    PF_XS(5);
    PF_WG(2);
    // the workgroup with the highest index collects the lines.  The wait cannot deadlock because NO other workgroup waits
    // for anything: each runs to its end on its own, whenever the dispatcher starts it (the dispatch order is not relied
    // on), and the collection ends on a time-out
    if (blockIdx.x == gridDim.x - 1) pf_auto_tail<T>(a, s_l, s_st, s_first, pcur, tside, lwcur, 1);
}

// ---- the same step with the OBSERVATIONS in parallel (small filters / shards) -----------------------------------------
// pf_auto_step_kernel gives a particle to a lane and walks the step's m observations one after the other: a dependent
// chain of m record loads, ~200 instructions and stores each.  On a full-size filter four such waves per SIMD hide each
// other's latency and the sweep runs at 5 TB/s; on a SMALL one -- a shard of an 8-rank filter has 32 768 particles, 128
// workgroups on 256 CUs -- the chain is what the step takes: 27 us for an eighth of the particles against 43 for all of them
// (tools/gpu_r3l.sh).  Here a workgroup owns 64 particles and wave w of its eight takes the observations w, w + 8, ...:
// different landmarks of a call are independent given the particle's pose (the host guarantees that no landmark occurs
// twice in the call, else the sequential kernel runs), and the log-weight is lw = (...((lw - shift) + t_0) + t_1 ...) with
// every term t_i formed without lw -- the waves leave their terms in LDS and wave 0 adds them IN OBSERVATION ORDER: the same
// particles and weights bit for bit.  One statistics line per workgroup of 64 particles; the collecting tail runs on
// the first four waves of the last workgroup (the others have ended: a barrier counts live waves only).
constexpr int PAR_WAVES = 8;
constexpr int PF_WAY4_MAX_N = 81920;         // one-box sweep (tools/gpu_r4c.sh, no resampling): 65536: seq 29.7, 2 ways 22.4, 4 ways 20.4 us; 98304: 32.4 / 27.1 / 31.3;
constexpr int PF_WAY2_MAX_N = 196608;        // 131072: 34.0 / 29.0 / 35.9; 196608: 38.4 / 36.1 / 47.9 (262144: the sequential sweep, 43.9)
constexpr int PF_PAR_MAX_N = 49152;          // one-box sweep (tools/gpu_r3m.sh): 16384: 26.4 -> 15.2 us, 32768: 27.2 -> 17.1, 65536: 28.4 -> 29.2, 98304: 31.9 -> 43.5
template <typename T, bool SH>
__global__ __launch_bounds__(64 * PAR_WAVES) void pf_auto_step_par_kernel(PfAutoArgs a) {
    PfCtl* ctl = a.ctl;
    typedef const __attribute__((address_space(4))) PfAutoArgs* KargPtr;
    const KargPtr ka = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    int l_pre = 0;
    int32_t st_pre = 0;
    if ((int)threadIdx.x < a.m) {
        l_pre = ka->ids[threadIdx.x] - 1;
        st_pre = a.lmstate[l_pre];
    }
    const long long halted = ctl->halt_seq;
    const int pcur = ctl->pcur, tside = ctl->tside, lwcur = ctl->lwcur;
    const double shift_next = ctl->shift_next;
    if (halted != 0 || ctl->error != 0) return;
    if constexpr (SH) {
        if (pf_peer_gone(a.inbox, a.world)) {              // (see pf_auto_step_kernel)
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                ctl->error = PF_ERR_PEER;
                pf_publish(a.mir, 0.0, (long long)ctl->nresamples, ctl->resample_seq, PF_ERR_PEER, a.seq, a.seq);
            }
            return;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->stamps[0] = wall_clock64();
    __shared__ T s_obs[2 * PF_AUTO_MAXOBS];
    __shared__ int32_t s_ids[PF_AUTO_MAXOBS], s_meta[PF_AUTO_MAXOBS], s_l[PF_AUTO_MAXOBS], s_st[PF_AUTO_MAXOBS], s_first[PF_AUTO_MAXOBS];
    __shared__ T s_pose[3][64];
    __shared__ T s_term[PF_AUTO_MAXOBS][64];
    const int m = a.m;
    for (int i = threadIdx.x; i < 2 * m; i += blockDim.x) s_obs[i] = (T)ka->z[i];
    const T pend = (T)shift_next;
    T* pose = (T*)(pcur ? a.pose1 : a.pose0);
    T* logw = (T*)(lwcur ? a.logw1 : a.logw0);
    const int32_t* tabs = tside ? a.tab1 : a.tab0;
    const int64_t n = a.n;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t pi = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = pi < n;
    const int64_t p = valid ? pi : n - 1;
    PfShardCtx sc{};
    if constexpr (SH) sc = PfShardCtx{a.peers, (uint32_t)a.first, (uint32_t)a.n, a.rank, a.world};
    T x = 0, y = 0, phi = 0, lw = 0;
    if (wave == 0) {                                     // F1: the motion model, once per particle
        x = pose[p]; y = pose[n + p]; phi = pose[2 * n + p];
        lw = logw[p];
        T e1, e2;
        normals2<T>((uint64_t)(a.first + p), a.step, STREAM_PREDICT, a.seed, e1, e2);
        const T Vn = (T)a.V + (T)a.a0 * e1;              // sim/sim-utils.jl:36
        const T Gn = (T)a.G + (T)a.a1 * e2;              // :37
        T sgp, cgp, sg, cg;
        m_sincos<T>(Gn + phi, sgp, cgp);
        m_sincos<T>(Gn, sg, cg);
        const T xn = x + Vn * (T)a.dt * cgp;             // src/ekf.jl:39-41
        const T yn = y + Vn * (T)a.dt * sgp;
        const T pn = wrap_pi<T>(phi + Vn * (T)a.dt * sg / (T)a.wheelbase);
        x = xn; y = yn; phi = pn;
        if (valid) { pose[p] = x; pose[n + p] = y; pose[2 * n + p] = phi; }
        s_pose[0][lane] = x; s_pose[1][lane] = y; s_pose[2][lane] = phi;
    }
    plan_obs(l_pre, st_pre, m, s_l, s_st, s_ids, s_meta, s_first);       // (two barriers: the pose is in LDS behind them)
    if (wave != 0) { x = s_pose[0][lane]; y = s_pose[1][lane]; phi = s_pose[2][lane]; }
    const T R00 = (T)a.R00, R10 = (T)a.R10, R01 = (T)a.R01, R11 = (T)a.R11;
    const LmView<T> lv{a.lmtab};
    for (int i = wave; i < m; i += PAR_WAVES) {          // F2 / F3: this wave's observations (uniform per wave)
        const int32_t code = __builtin_amdgcn_readfirstlane(s_ids[i]), meta = __builtin_amdgcn_readfirstlane(s_meta[i]);
        const int l = code & ID_MASK;
        const T r = s_obs[2 * i], b = s_obs[2 * i + 1];
        const BufRow<T, decltype(lm_rsrc<T>((const T*)nullptr, n))> row{lm_rsrc<T>(lv.rows((meta & META_WBUF) ? 1 : 0, l, n), n),
                                                          (uint32_t)p * (uint32_t)sizeof(T), (uint32_t)n * (uint32_t)sizeof(T)};
        T term = 0;
        if (code & NEW_FLAG) {
            lm_init<T>(row, n, x, y, phi, r, b, R00, R10, R01, R11, valid);
        } else {
            const LmRow<T> cur = sweep_load<T, 2, SH>(lv, tabs, n, (uint32_t)p, code, meta, sc);
            lm_update<T>(row, n, cur, x, y, phi, r, b, R00, R10, R01, R11, valid, term);      // term = 0 + (this observation's log-weight term)
        }
        s_term[i][lane] = term;
    }
    __syncthreads();
    if (wave == 0) {
        lw -= pend;
        for (int i = 0; i < m; ++i)
            if (!(__builtin_amdgcn_readfirstlane(s_ids[i]) & NEW_FLAG)) lw += s_term[i][lane];     // observation order
        if (valid) logw[p] = lw;
    }
    if (wave == 0) {                                       // the tree's leaf (this workgroup's 64 particles) as a tagged line
        const WRec leaf = wrec_wave<T>(lw, valid);
        if (lane == 0) wrec_store_line(a.part, (int)blockIdx.x, leaf, a.seq);
    }
    if (blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x >= 256) return;                  // the tail is written for four waves
        pf_auto_tail<T>(a, s_l, s_st, s_first, pcur, tside, lwcur, 0);
    }
}


// ---- the step with W-way observation parallelism on 256-particle workgroups (round 4; shards of 2 and 4 ranks) -------------
// The 8-way kernel above wins up to ~49 k particles and loses beyond (four times the workgroups, the plan and the pose
// hand-over per 64 particles); the sequential sweep needs ~262 k particles to hide its sixteen-update chain.  Between them --
// the shards of BASELINE.json's filter on two and four GPUs, 131 072 and 65 536 particles -- a workgroup keeps the sweep's 256
// particles and takes W = 2 or 4 WAYS: 256 W threads, wave w serves the particles 64 (w & 3) .. + 63 and the observations
// (w >> 2), (w >> 2) + W, ... with the sweep's record ring (PF_DEPTH requests in flight per way).  Way 0 runs the motion model
// and leaves the pose in LDS; every way leaves its log-weight terms in LDS and way 0 adds them IN OBSERVATION ORDER: particles
// and weights are the sequential kernel's bit for bit.  The first four waves hold the 256 particles' weights, so the workgroup
// stores the same 256-particle statistics line as the sequential sweep.  Needs: no landmark twice in the call (host-checked),
// m <= WAY_MAXOBS<T>.
template <typename T>
constexpr int WAY_MAXOBS = sizeof(T) == 4 ? 32 : 16;     // the term array [m][256] stays within 32 KB of LDS
template <typename T, bool SH, int W>
__global__ __launch_bounds__(256 * W) void pf_auto_step_way_kernel(PfAutoArgs a) {
    PfCtl* ctl = a.ctl;
    typedef const __attribute__((address_space(4))) PfAutoArgs* KargPtr;
    const KargPtr ka = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    int l_pre = 0;
    int32_t st_pre = 0;
    if ((int)threadIdx.x < a.m) {
        l_pre = ka->ids[threadIdx.x] - 1;
        st_pre = a.lmstate[l_pre];
    }
    const long long halted = ctl->halt_seq;
    const int pcur = ctl->pcur, tside = ctl->tside, lwcur = ctl->lwcur;
    const double shift_next = ctl->shift_next;
    if (halted != 0 || ctl->error != 0) return;
    if constexpr (SH) {
        if (pf_peer_gone(a.inbox, a.world)) {              // (see pf_auto_step_kernel)
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                ctl->error = PF_ERR_PEER;
                pf_publish(a.mir, 0.0, (long long)ctl->nresamples, ctl->resample_seq, PF_ERR_PEER, a.seq, a.seq);
            }
            return;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->stamps[0] = wall_clock64();
    __shared__ T s_obs[2 * PF_AUTO_MAXOBS];
    __shared__ int32_t s_ids[PF_AUTO_MAXOBS], s_meta[PF_AUTO_MAXOBS], s_l[PF_AUTO_MAXOBS], s_st[PF_AUTO_MAXOBS], s_first[PF_AUTO_MAXOBS];
    __shared__ T s_pose[3][256];
    __shared__ T s_term[WAY_MAXOBS<T>][256];
    const int m = a.m;
    for (int i = threadIdx.x; i < 2 * m; i += blockDim.x) s_obs[i] = (T)ka->z[i];
    const T pend = (T)shift_next;
    T* pose = (T*)(pcur ? a.pose1 : a.pose0);
    T* logw = (T*)(lwcur ? a.logw1 : a.logw0);
    const int32_t* tabs = tside ? a.tab1 : a.tab0;
    const int64_t n = a.n;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int way = wave >> 2;                               // 0 .. W-1 (wave-uniform)
    const int pl = (wave & 3) * 64 + lane;                   // the particle's place in the workgroup
    const int64_t pi = (int64_t)blockIdx.x * 256 + pl;
    const bool valid = pi < n;
    const int64_t p = valid ? pi : n - 1;
    PfShardCtx sc{};
    if constexpr (SH) sc = PfShardCtx{a.peers, (uint32_t)a.first, (uint32_t)a.n, a.rank, a.world};
    T x = 0, y = 0, phi = 0, lw = 0;
    if (way == 0) {                                          // F1: the motion model, once per particle
        x = pose[p]; y = pose[n + p]; phi = pose[2 * n + p];
        lw = logw[p];
        T e1, e2;
        normals2<T>((uint64_t)(a.first + p), a.step, STREAM_PREDICT, a.seed, e1, e2);
        const T Vn = (T)a.V + (T)a.a0 * e1;                  // sim/sim-utils.jl:36
        const T Gn = (T)a.G + (T)a.a1 * e2;                  // :37
        T sgp, cgp, sg, cg;
        m_sincos<T>(Gn + phi, sgp, cgp);
        m_sincos<T>(Gn, sg, cg);
        const T xn = x + Vn * (T)a.dt * cgp;                 // src/ekf.jl:39-41
        const T yn = y + Vn * (T)a.dt * sgp;
        const T pn = wrap_pi<T>(phi + Vn * (T)a.dt * sg / (T)a.wheelbase);
        x = xn; y = yn; phi = pn;
        if (valid) { pose[p] = x; pose[n + p] = y; pose[2 * n + p] = phi; }
        s_pose[0][pl] = x; s_pose[1][pl] = y; s_pose[2][pl] = phi;
    }
    plan_obs(l_pre, st_pre, m, s_l, s_st, s_ids, s_meta, s_first);       // (two barriers: the pose is in LDS behind them)
    if (way != 0) { x = s_pose[0][pl]; y = s_pose[1][pl]; phi = s_pose[2][pl]; }
    const T R00 = (T)a.R00, R10 = (T)a.R10, R01 = (T)a.R01, R11 = (T)a.R11;
    const LmView<T> lv{a.lmtab};
    auto uni = [](int32_t v) { return __builtin_amdgcn_readfirstlane(v); };
    // this way's observations i = way + W j, j = 0 .. cnt-1, with the sweep's ring: the records of the next PF_DEPTH of them in
    // flight (no landmark occurs twice in the call, so every record may be requested ahead)
    const int cnt = m > way ? (m - way + W - 1) / W : 0;
    LmRow<T> ring[PF_DEPTH];
    bool have[PF_DEPTH];
#pragma unroll
    for (int u = 0; u < PF_DEPTH; ++u) {
        have[u] = false;
        ring[u] = LmRow<T>{0, 0, 0, 0, 0};
        const int i = way + W * u;
        if (u < cnt && !(uni(s_ids[i]) & NEW_FLAG)) {
            ring[u] = sweep_load<T, 2, SH>(lv, tabs, n, (uint32_t)p, uni(s_ids[i]), uni(s_meta[i]), sc);
            have[u] = true;
        }
    }
    for (int j0 = 0; j0 < cnt; j0 += PF_DEPTH) {
#pragma unroll
        for (int u = 0; u < PF_DEPTH; ++u) {
            const int j = j0 + u;
            if (j >= cnt) break;                             // uniform
            const int i = way + W * j;
            const int32_t code = uni(s_ids[i]), meta = uni(s_meta[i]);
            const int l = code & ID_MASK;
            const T r = s_obs[2 * i], b = s_obs[2 * i + 1];
            const BufRow<T, decltype(lm_rsrc<T>((const T*)nullptr, n))> row{lm_rsrc<T>(lv.rows((meta & META_WBUF) ? 1 : 0, l, n), n),
                                                              (uint32_t)p * (uint32_t)sizeof(T), (uint32_t)n * (uint32_t)sizeof(T)};
            LmRow<T> cur = ring[u];
            const bool have_cur = have[u];
            have[u] = false;
            const int jn = j + PF_DEPTH;
            if (jn < cnt) {
                const int in = way + W * jn;
                if (!(uni(s_ids[in]) & NEW_FLAG)) {
                    ring[u] = sweep_load<T, 2, SH>(lv, tabs, n, (uint32_t)p, uni(s_ids[in]), uni(s_meta[in]), sc);
                    have[u] = true;
                }
            }
            T term = 0;
            if (code & NEW_FLAG) {                           // F3: first sighting
                lm_init<T>(row, n, x, y, phi, r, b, R00, R10, R01, R11, valid);
            } else {
                if (!have_cur) cur = sweep_load<T, 2, SH>(lv, tabs, n, (uint32_t)p, code, meta, sc);
                lm_update<T>(row, n, cur, x, y, phi, r, b, R00, R10, R01, R11, valid, term);      // term = 0 + (this observation's log-weight term)
            }
            s_term[i][pl] = term;
        }
    }
    __syncthreads();
    if (way == 0) {
        lw -= pend;
        for (int i = 0; i < m; ++i)
            if (!(uni(s_ids[i]) & NEW_FLAG)) lw += s_term[i][pl];                                 // observation order
        if (valid) logw[p] = lw;
    }
    wrec_block_line<T>(lw, valid, a.part, a.seq);            // (the first four waves = way 0 hold the 256 weights)
    if (blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x >= 256) return;                      // the tail is written for four waves
        pf_auto_tail<T>(a, s_l, s_st, s_first, pcur, tside, lwcur, 1);
    }
}

// ---- the resampling of a SHARDED filter on the device --------------------------------------------------------------
// Every rank takes the same decision from the same table of scalars (pf_auto_tail), so on a resampling step every rank
// runs the same three conditional kernels behind its step kernel:
//   gate      ONE workgroup: tells every peer "my step kernel of step s has COMPLETED" (stream order: its stores are in
//             memory, the end of a kernel writes the L2s back) and waits until every peer has said so.  One workgroup,
//             not a poll in every workgroup of the next kernel: ranks that share a card (the rehearsal) would fill it
//             with spinning workgroups and the peer's kernel that has to send the word would never start.
//   scan      the cdf of ALL n_global weights, every rank for itself: the all-gather of the log-weights is the kernel's
//             loads -- a rank's slice is read straight from its owner's buffer over xGMI (1 MiB in all at C4).  Same
//             blocks, same order of additions as on one rank: the ancestors are the same whatever the number of ranks.
//   resample  this rank's ancestors (global ids), their poses and table entries read from their owners' buffers; the
//             MAPS do not move: a table entry is a global particle id and the sweep reads a remote ancestor's record
//             from its owner when the landmark is next updated (sweep_load<SH>).  Uniform weights go to the OTHER
//             log-weight buffer, poses and tables to their other sides: a peer that is still reading this rank's
//             step-s state reads buffers nobody writes.  Why no further hand-shake is needed: a rank writes those old
//             sides again at its resampling s' > s at the earliest, which needs every rank's scalars of step s', which
//             a rank publishes only after its own resampling s has completed (stream order).
// one lane: "rank `rank` is going away" into every peer's inbox (slam_pf_destroy of a handle that is still attached)
__global__ void pf_peer_gone_kernel(const PfPeers* __restrict__ peers, int rank, int world) {
    const int r = threadIdx.x;
    if (r < world && r != rank) __hip_atomic_store(&peers->inbox[r]->gone[rank][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void pf_peer_gate_kernel(PfCtl* ctl, long long seq, const PfPeers* __restrict__ peers,
                                                          PfInbox* inbox, int rank, int world) {
    if (ctl->resample_seq != seq || ctl->error != 0) return;
    if (pf_peer_gone(inbox, world)) { ctl->error = PF_ERR_PEER; return; }       // (the kernels behind this one return on ctl->error)
    const int r = threadIdx.x;
    if (r < world) {
        __hip_atomic_store(&peers->inbox[r]->ready[rank][0], (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&inbox->ready[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)seq) {
            __builtin_amdgcn_s_sleep(20);
            if (wall_clock64() - t0 > 2000000000ull) { ctl->error = PF_ERR_PEER; break; }      // 20 s: a rank is gone
        }
    }
}

// A barrier among the ranks on their streams (materialise): every rank counts its calls, tells every peer, waits for all.
__global__ __launch_bounds__(64) void pf_peer_barrier_kernel(int32_t* err, unsigned long long count, const PfPeers* __restrict__ peers,
                                                             PfInbox* inbox, int rank, int world, unsigned long long timeout_ticks) {
    if (pf_peer_gone(inbox, world)) { *err = PF_ERR_PEER; return; }
    const int r = threadIdx.x;
    if (r < world) {
        __hip_atomic_store(&peers->inbox[r]->bar[rank][0], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&inbox->bar[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < count) {
            __builtin_amdgcn_s_sleep(20);
            if (wall_clock64() - t0 > timeout_ticks) { *err = PF_ERR_PEER; break; }
        }
    }
}

// The cdf of the step that resamples (pf_scan1_kernel behind the control block's gate), over the weights of the WHOLE
// filter: logw0 / logw1 are this rank's two buffers (n_local values each), the other slices come from `peers`.
template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void pf_auto_scan1_kernel(const T* __restrict__ logw0, const T* __restrict__ logw1,
                                                                    int64_t n_local, int64_t n_global, const PfCtl* __restrict__ ctl,
                                                                    long long seq, double* __restrict__ cdf,
                                                                    double* __restrict__ bsum, const PfPeers* __restrict__ peers,
                                                                    int rank, int world) {
    if (ctl->resample_seq != seq || ctl->error != 0) return;
    __shared__ double sh[SCAN_BLOCK];
    const int64_t i = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const T pend = (T)ctl->shift_scan;
    const double gmax = ctl->gmax_norm;
    const int old = ctl->lwcur ^ 1;                    // (the tail has flipped the live side: the step's weights are in the other)
    T v = 0;
    if (i < n_global) {
        const uint32_t owner = world > 1 ? pf_owner((uint32_t)i, (uint32_t)n_local, world) : 0u;
        if (world > 1 && owner != (uint32_t)rank) v = ld_sys((const T*)peers->logw[owner][old] + (i - (int64_t)owner * n_local));
        else v = (old ? logw1 : logw0)[i - (int64_t)owner * n_local];
    }
    sh[threadIdx.x] = i < n_global ? exp((double)(T)(v - pend) - gmax) : 0.0;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        const double u = threadIdx.x >= off ? sh[threadIdx.x - off] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += u;
        __syncthreads();
    }
    if (i < n_global) cdf[i] = sh[threadIdx.x];
    if (threadIdx.x == SCAN_BLOCK - 1) bsum[blockIdx.x] = sh[threadIdx.x];
}

// Block offsets (pf_scan2_kernel's serial order, redone by every workgroup out of LDS), ancestors (pf_ancestor_kernel)
// and the lazy apply (pf_lazy_apply_kernel) of the step that resamples, in one conditional launch.  n: this rank's
// particles, global ids [first, first + n); table entries and ancestors are GLOBAL ids (= local slots on one rank).
constexpr int AUTO_NB_MAX = 2048;            // scan blocks (of 1024 particles) the fused offsets support
template <typename T, bool SH>
__global__ __launch_bounds__(256) void pf_auto_resample_kernel(T* pose0, T* pose1, int32_t* tab0, int32_t* tab1,
                                                                T* logw0, T* logw1, int64_t n, int64_t first, int64_t n_global,
                                                                const PfCtl* __restrict__ ctl,
                                                                long long seq, const double* __restrict__ cdf,
                                                                const double* __restrict__ bsum, int nb,
                                                                int32_t* __restrict__ anc_out, T lw_uniform,
                                                                const PfPeers* __restrict__ peers, int rank, int world) {
    if (ctl->resample_seq != seq || ctl->error != 0) return;
    __shared__ double s_off[AUTO_NB_MAX + 1];
    __shared__ int s_tl[PF_TAB_MAX];                       // the live tables to compose (once per workgroup, not once per use)
    for (int i = threadIdx.x; i < nb; i += 256) s_off[i] = bsum[i];
    if (threadIdx.x < PF_TAB_MAX) s_tl[threadIdx.x] = ctl->tl_idx[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int i = 0; i < nb; ++i) {
            const double v = s_off[i];
            s_off[i] = run;
            run += v;
        }
        s_off[nb] = run;
    }
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double total = s_off[nb];
    const double target = ((double)(first + p) + ctl->u0) / (double)n_global * total;
    // first j with cdf[j] + offset(block of j) >= target, in two levels: the block out of LDS (the last element of block b
    // has exactly the value s_off[b + 1] = s_off[b] + bsum[b]), then ten steps inside it -- the same index as the plain
    // binary search of pf_ancestor_kernel over all n
    int bl = 0, bh = nb - 1;
    while (bl < bh) {
        const int bm = (bl + bh) >> 1;
        if (s_off[bm + 1] >= target) bh = bm; else bl = bm + 1;
    }
    int64_t lo = (int64_t)bl * SCAN_BLOCK, hi = lo + SCAN_BLOCK - 1;
    if (hi > n_global - 1) hi = n_global - 1;
    const double boff = s_off[bl];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] + boff >= target) hi = mid; else lo = mid + 1;
    }
    const int32_t a = (int32_t)lo;                         // the ancestor's GLOBAL id
    anc_out[p] = a;
    // the tail has already flipped the buffers: the live ones are the destination
    const int pnew = ctl->pcur, tnew = ctl->tside, lnew = ctl->lwcur;
    const uint32_t owner = SH ? pf_owner((uint32_t)a, (uint32_t)n, world) : 0u;
    const bool remote = SH && owner != (uint32_t)rank;
    const int64_t q = (int64_t)a - (int64_t)owner * n;     // the ancestor's slot on its owner
    const T* pose_old = remote ? (const T*)peers->pose[owner][pnew ^ 1] : (pnew ? pose0 : pose1);
    T* __restrict__ pose_new = pnew ? pose1 : pose0;
    const int32_t* tin = remote ? (const int32_t*)peers->tab[owner][tnew ^ 1] : (tnew ? tab0 : tab1);
    int32_t* __restrict__ tout = tnew ? tab1 : tab0;
    const int fresh = ctl->tl_fresh, count = ctl->tl_count;
    // Gathers in batches: ALL loads of a batch are issued before its first store.  (Written as load -> store per table,
    // with the table's index fetched from the control block each time, the compiler kept every pair in order behind a
    // full wait -- possible aliasing -- and the 31 live tables of the benchmark cost 31 serial round trips: 21 us.)
    // A remote ancestor's pose and entries are read with system-scope loads (its owner's kernels have completed: gate).
    auto ldp = [&](const T* ptr) {
        if constexpr (SH) return remote ? ld_sys(ptr) : *ptr;
        else return *ptr;
    };
    auto ldt = [&](const int32_t* ptr) {
        if constexpr (SH) return remote ? ld_sys(ptr) : *ptr;
        else return *ptr;
    };
    T pv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pv[r] = ldp(pose_old + (size_t)r * n + q);
    constexpr int TB = 8;
    int32_t tv[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) tv[u] = u < count ? ldt(tin + (size_t)s_tl[u] * n + q) : 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) pose_new[(size_t)r * n + p] = pv[r];
    if (fresh >= 0) tout[(size_t)fresh * n + p] = a;
    (lnew ? logw1 : logw0)[p] = lw_uniform;
    for (int i0 = 0; i0 < count; i0 += TB) {
        int32_t tn[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) tn[u] = i0 + TB + u < count ? ldt(tin + (size_t)s_tl[i0 + TB + u] * n + q) : 0;    // the next batch
#pragma unroll
        for (int u = 0; u < TB; ++u)
            if (i0 + u < count) tout[(size_t)s_tl[i0 + u] * n + p] = tv[u];
#pragma unroll
        for (int u = 0; u < TB; ++u) tv[u] = tn[u];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pf_weights_kernel(const T* __restrict__ logw, int64_t n, T pend, double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = exp((double)(T)(logw[p] - pend));
}

template <typename P>
int pf_alloc(P** p, size_t bytes, hipStream_t s) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc((void**)p, bytes));
    HIP_TRY(hipMemsetAsync(*p, 0, bytes, s));
    return SLAM_OK;
}

inline int grid_for(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

#define PF_DISPATCH(h, CALL_F, CALL_D) \
    do {                               \
        if ((h)->dtype == SLAM_F32) {  \
            typedef float T;           \
            CALL_F;                    \
        } else {                       \
            typedef double T;          \
            CALL_D;                    \
        }                              \
    } while (0)

static void pf_detach_peers_impl(slam_pf* h);
static const char* pf_error_text(long long code);
static int pf_auto_flush(slam_pf* h);      // wait for the steps slam_pf_step_auto has queued (resolving a halted one)
static int pf_auto_leave(slam_pf* h);      // auto mode -> legacy mode: wait for the queue, bring the bookkeeping back to the host
#define PF_LEGACY_ENTRY(h)                    \
    do {                                      \
        const int rc_leave_ = pf_auto_leave(h); \
        if (rc_leave_) return rc_leave_;      \
    } while (0)

extern "C" int slam_pf_destroy(slam_pf_t h) {
    if (!h) return SLAM_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->d_peers && h->xchg_world > 1) {
        // Destroyed while still attached (the orderly way is slam_pf_detach_peers on every rank first, then destroy): the peers
        // may have kernels queued that read THIS rank's buffers.  Tell them before anything is freed -- every kernel of a sharded
        // filter that touches peer memory first looks at its inbox's `gone` words and stops with PF_ERR_PEER -- and give
        // kernels already in flight (a step is tens of microseconds) time to end.  Best effort: a peer's kernel that started
        // between the word and the free can still fault; detach first.
        hipLaunchKernelGGL(pf_peer_gone_kernel, dim3(1), dim3(64), 0, h->stream, (const PfPeers*)h->d_peers, h->xchg_rank, h->xchg_world);
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->stream);
        usleep(5000);
    }
    pf_detach_peers_impl(h);                 // this rank's mappings of the peers' buffers are closed before anything is freed
    for (int b = 0; b < 2; ++b) {
        if (h->pose[b]) (void)hipFree(h->pose[b]);
        for (int k = 0; k < PF_LM_MAXC; ++k)
            if (h->lmtab.c[b][k]) (void)hipFree(h->lmtab.c[b][k]);
    }
    if (h->d_lmtab) (void)hipFree(h->d_lmtab);
    void* devs[] = {h->logw2[0], h->logw2[1], h->d_part, h->d_out, h->d_cdf, h->d_bsum, h->d_src, h->d_anc, h->d_tab[0], h->d_tab[1],
                    h->d_lmeta, h->d_ctl, h->d_lmstate, h->inbox};
    if (h->xchg_host) (void)hipHostUnregister(h->xchg_host);
    if (h->h_mir) (void)hipHostFree(h->h_mir);
    for (void* p : devs)
        if (p) (void)hipFree(p);
    if (h->h_ids) (void)hipHostFree(h->h_ids);
    if (h->h_obs) (void)hipHostFree(h->h_obs);
    if (h->h_out) (void)hipHostFree(h->h_out);
    for (int b = 0; b < 2; ++b)
        if (h->stage_ev[b]) (void)hipEventDestroy(h->stage_ev[b]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SLAM_OK;
}

static int pf_create_impl(slam_pf* h) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    int rc;
    const size_t n = (size_t)h->n;
    // the landmark records in chunks of 2^shift landmarks (PfLmTab): the largest power of two whose chunk stays within 1 GiB,
    // doubled while the buffer would need more than PF_LM_MAXC chunks (a chunk above 2047 MiB cannot be exported to another
    // process -- slam_pf_attach_peers refuses such a peer -- but works locally)
    {
        const size_t per_lm = h->esz * 5 * n;
        int shift = 0;
        while (shift < 20 && (per_lm << (shift + 1)) <= ((size_t)1 << 30)) ++shift;
        while ((((size_t)h->nl + ((size_t)1 << shift) - 1) >> shift) > (size_t)PF_LM_MAXC) ++shift;
        h->lmtab.shift = shift;
        h->lmtab.nchunks = (int)(((size_t)h->nl + ((size_t)1 << shift) - 1) >> shift);
        h->lm_chunk_bytes = per_lm << shift;
    }
    for (int b = 0; b < 2; ++b) {
        if ((rc = pf_alloc(&h->pose[b], h->esz * 3 * n, h->stream))) return rc;
        for (int k = 0; k < h->lmtab.nchunks; ++k) {
            // (the last chunk holds what is left of the nl landmarks)
            const size_t lms = (size_t)h->nl - ((size_t)k << h->lmtab.shift) < ((size_t)1 << h->lmtab.shift)
                                   ? (size_t)h->nl - ((size_t)k << h->lmtab.shift) : ((size_t)1 << h->lmtab.shift);
            if ((rc = pf_alloc(&h->lmtab.c[b][k], h->esz * 5 * n * lms, h->stream))) return rc;
        }
    }
    HIP_TRY(hipMalloc((void**)&h->d_lmtab, sizeof(PfLmTab)));
    HIP_TRY(hipMemcpy(h->d_lmtab, &h->lmtab, sizeof(PfLmTab), hipMemcpyHostToDevice));
    for (int b = 0; b < 2; ++b)
        if ((rc = pf_alloc(&h->logw2[b], h->esz * n, h->stream))) return rc;
    h->lwcur = 0;
    h->logw = h->logw2[0];
    // the inbox the peers of a sharded filter write into: fine-grained device memory (polled while a peer GPU writes it)
    // (header + the ranks' 1024-particle weight records of a step, two parities: 17 KB at 262144 particles)
    h->inbox_bytes = pf_inbox_bytes(h->n_global);
    if (hipExtMallocWithFlags((void**)&h->inbox, h->inbox_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        h->inbox = nullptr;
        HIP_TRY(hipMalloc((void**)&h->inbox, h->inbox_bytes));
    }
    HIP_TRY(hipMemsetAsync(h->inbox, 0, h->inbox_bytes, h->stream));
    h->ocap = PF_OCAP;
    h->red_blocks = grid_for(h->n);                  // one partial record per 256 particles
    // (room for one statistics line per 64 particles: the observation-parallel step kernel's workgroups)
    if ((rc = pf_alloc(&h->d_part, sizeof(double) * 8 * (size_t)((h->n + 63) / 64), h->stream))) return rc;
    if ((rc = pf_alloc(&h->d_out, sizeof(double) * 8, h->stream))) return rc;
    if ((rc = pf_alloc(&h->d_cdf, sizeof(double) * (size_t)h->n_global, h->stream))) return rc;
    const size_t nb = ((size_t)h->n_global + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if ((rc = pf_alloc(&h->d_bsum, sizeof(double) * (nb + 1), h->stream))) return rc;
    if ((rc = pf_alloc(&h->d_src, sizeof(int32_t) * n, h->stream))) return rc;
    if ((rc = pf_alloc(&h->d_anc, sizeof(int32_t) * n, h->stream))) return rc;
    HIP_TRY(hipHostMalloc((void**)&h->h_ids, sizeof(int32_t) * 4 * PF_OCAP, hipHostMallocDefault));
    memset(h->h_ids, 0, sizeof(int32_t) * 4 * PF_OCAP);
    for (int b = 0; b < 2; ++b)
        if ((rc = pf_alloc(&h->d_tab[b], sizeof(int32_t) * (size_t)PF_TAB_MAX * n, h->stream))) return rc;
    if ((rc = pf_alloc(&h->d_lmeta, sizeof(int32_t) * (size_t)h->nl, h->stream))) return rc;
    HIP_TRY(hipHostMalloc((void**)&h->h_obs, sizeof(double) * 4 * h->ocap, hipHostMallocDefault));
    for (int b = 0; b < 2; ++b) HIP_TRY(hipEventCreateWithFlags(&h->stage_ev[b], hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_TRY(hipHostGetDevicePointer((void**)&h->h_ids_dev, h->h_ids, 0));
    HIP_TRY(hipHostGetDevicePointer((void**)&h->h_obs_dev, h->h_obs, 0));
    HIP_TRY(hipHostMalloc((void**)&h->h_out, sizeof(double) * 8, hipHostMallocDefault));
    memset(h->h_out, 0, sizeof(double) * 8);
    HIP_TRY(hipHostGetDevicePointer((void**)&h->h_out_dev, h->h_out, 0));
    h->out_seq = 0;
    h->pending_shift = 0.0; h->has_pending = 0;
    // auto mode
    if ((rc = pf_alloc(&h->d_ctl, sizeof(PfCtl), h->stream))) return rc;
    if ((rc = pf_alloc(&h->d_lmstate, sizeof(int32_t) * (size_t)h->nl, h->stream))) return rc;
    HIP_TRY(hipHostMalloc((void**)&h->h_mir, sizeof(PfMirror), hipHostMallocDefault));
    memset(h->h_mir, 0, sizeof(PfMirror));
    HIP_TRY(hipHostGetDevicePointer((void**)&h->h_mir_dev, h->h_mir, 0));
    // uniform weights over the GLOBAL particle set
    const double lw = -log((double)h->n_global);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_set_pose_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->pose[0],
                                   (T*)h->logw, h->n, (T)0, (T)0, (T)0, (T)lw),
                hipLaunchKernelGGL(pf_set_pose_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->pose[0],
                                   (T*)h->logw, h->n, (T)0, (T)0, (T)0, (T)lw));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_pf_create(slam_pf_t* out, int dtype, int64_t n_local, int64_t n_global, int64_t first_id,
                              int max_landmarks, int device, uint64_t seed) {
    ARG_CHECK(out != nullptr, "handle pointer is null");
    *out = nullptr;
    ARG_CHECK(dtype == SLAM_F32 || dtype == SLAM_F64, "dtype must be SLAM_F32 or SLAM_F64");
    ARG_CHECK(n_local > 0 && n_global >= n_local && first_id >= 0 && first_id + n_local <= n_global,
              "particle range [first, first + n_local) must lie inside [0, n_global)");
    ARG_CHECK(n_global < (1ll << 31), "n_global must fit 31 bits");
    ARG_CHECK(5 * n_local * (dtype == SLAM_F32 ? 4 : 8) < (1ll << 32),
              "n_local: one landmark's five rows (5 n values) must fit a 4 GiB buffer descriptor");
    ARG_CHECK(max_landmarks > 0 && max_landmarks < (1 << 20), "max_landmarks out of range");
    const int ndev = slam_device_count();
    if (ndev <= 0) {
        slam_set_error("no HIP device available: libslamhip has no CPU fallback");
        return SLAM_E_HIP;
    }
    ARG_CHECK(device >= 0 && device < ndev, "device index out of range");
    slam_pf* h = new slam_pf();
    h->dtype = dtype; h->device = device; h->esz = dtype == SLAM_F32 ? 4 : 8;
    h->n = n_local; h->n_global = n_global; h->first = first_id; h->nl = max_landmarks;
    h->seed = seed; h->step = 0; h->cur = 0; h->pcur = 0; h->stream = nullptr;
    h->lbuf.assign(max_landmarks, 0); h->ltab.assign(max_landmarks, -1); h->tref.assign(PF_TAB_MAX, 0);
    h->prior.assign(max_landmarks, -1);
    h->d_tab[0] = h->d_tab[1] = nullptr; h->d_lmeta = nullptr; h->tside = 0; h->lazy_dirty = 0;
    h->lazy_off = getenv("SLAMHIP_PF_EAGER") && atoi(getenv("SLAMHIP_PF_EAGER")) ? 1 : 0;
    h->pose[0] = h->pose[1] = h->logw = h->logw2[0] = h->logw2[1] = nullptr;
    memset(&h->lmtab, 0, sizeof(h->lmtab)); h->d_lmtab = nullptr; h->lm_chunk_bytes = 0;
    h->lwcur = 0; h->d_peers = nullptr; h->inbox = nullptr; h->bar_count = 0; h->halts = 0;
    h->par_max_n = slam_exp_env("SLAMHIP_PF_PAR_MAX", PF_PAR_MAX_N);      // (the knobs are read by the experiments build only)
    h->way4_max_n = slam_exp_env("SLAMHIP_PF_WAY4_MAX", PF_WAY4_MAX_N);
    h->way2_max_n = slam_exp_env("SLAMHIP_PF_WAY2_MAX", PF_WAY2_MAX_N);
    memset(&h->peers, 0, sizeof(h->peers));
    memset(h->peer_open, 0, sizeof(h->peer_open));
    h->h_ids = nullptr; h->h_obs = nullptr; h->ocap = 0;
    h->stage_ev[0] = h->stage_ev[1] = nullptr; h->stage_used[0] = h->stage_used[1] = 0; h->stage_slot = 0;
    h->d_part = h->d_out = h->h_out = h->d_cdf = h->d_bsum = nullptr; h->d_src = nullptr; h->d_anc = nullptr;
    h->seen.assign(max_landmarks, 0);
    h->d_ctl = nullptr; h->d_lmstate = nullptr; h->h_mir = h->h_mir_dev = nullptr;
    h->auto_on = 0; h->auto_seq = 0; h->pub_seq = 0; h->nresamples = 0; h->halted = 0; h->halt_gmax = 0.0; h->last_resampled_seq = 0;
    h->d_xchg = nullptr; h->xchg_host = nullptr; h->xchg_rank = 0; h->xchg_world = 1;
    for (int i = 0; i < 4; ++i) h->last_out[i] = 0.0;
    const int rc = pf_create_impl(h);
    if (rc) { slam_pf_destroy(h); return rc; }
    *out = h;
    return SLAM_OK;
}

// ---- lazy resampling: host bookkeeping ----------------------------------------------------------------------
static void pf_release_table(slam_pf* h, int l) {
    if (h->ltab[l] >= 0) {
        h->tref[h->ltab[l]] -= 1;
        h->ltab[l] = -1;
    }
}

// Bring every landmark to (buffer h->cur, identity table): what the non-lazy kernels expect.  Two passes at most:
// a landmark that sits in h->cur behind a table cannot be gathered in place, it goes to the other buffer first.
static bool pf_sharded(const slam_pf* h) { return h->d_peers != nullptr && h->xchg_world > 1; }

static int pf_peer_barrier(slam_pf* h) {
    h->bar_count += 1;
    hipLaunchKernelGGL(pf_peer_barrier_kernel, dim3(1), dim3(64), 0, h->stream, &h->d_ctl->error, (unsigned long long)h->bar_count,
                       (const PfPeers*)h->d_peers, h->inbox, h->xchg_rank, h->xchg_world, 2000000000ull);       // 20 s
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

// (Sharded filter with peers: a COLLECTIVE call -- the ancestor tables hold global particle ids and a remote ancestor's
//  record is read from its owner, so every rank must be here, with barriers among the ranks' streams around the passes:
//  pass 0 reads what the peers' earlier kernels wrote, pass 1 overwrites what the peers' pass 0 reads.)
static int pf_materialise(slam_pf* h) {
    if (!h->lazy_dirty) return SLAM_OK;
    const int B = h->cur;
    const bool sh = pf_sharded(h);
    const PfShardCtx sc{h->d_peers, (uint32_t)h->first, (uint32_t)h->n, h->xchg_rank, h->xchg_world};
    std::vector<int32_t> work(h->nl);
    if (sh) { const int rcb = pf_peer_barrier(h); if (rcb) return rcb; }
    for (int pass = 0; pass < 2; ++pass) {
        if (sh && pass == 1) { const int rcb = pf_peer_barrier(h); if (rcb) return rcb; }
        bool any = false;
        for (int l = 0; l < h->nl; ++l) {
            const bool go = pass == 0 ? (h->lbuf[l] == B && h->ltab[l] >= 0) : (h->lbuf[l] != B);
            work[l] = -1;
            if (!go) continue;
            const int dst = pass == 0 ? (B ^ 1) : B;
            work[l] = (h->ltab[l] + 1) | (h->lbuf[l] ? META_RBUF : 0) | (dst ? META_WBUF : 0);
            any = true;
        }
        if (!any) continue;
        HIP_TRY(hipMemcpyAsync(h->d_lmeta, work.data(), sizeof(int32_t) * h->nl, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));             // `work` is pageable host memory
        const dim3 grid(grid_for(h->n), (h->nl + MAT_LMS - 1) / MAT_LMS);
        if (sh)
            PF_DISPATCH(h,
                        hipLaunchKernelGGL((pf_materialise_kernel<T, true>), grid, dim3(256), 0, h->stream, LmView<T>{h->d_lmtab},
                                           (const int32_t*)h->d_tab[h->tside], h->n, h->nl, (const int32_t*)h->d_lmeta, sc),
                        hipLaunchKernelGGL((pf_materialise_kernel<T, true>), grid, dim3(256), 0, h->stream, LmView<T>{h->d_lmtab},
                                           (const int32_t*)h->d_tab[h->tside], h->n, h->nl, (const int32_t*)h->d_lmeta, sc));
        else
            PF_DISPATCH(h,
                        hipLaunchKernelGGL((pf_materialise_kernel<T, false>), grid, dim3(256), 0, h->stream, LmView<T>{h->d_lmtab},
                                           (const int32_t*)h->d_tab[h->tside], h->n, h->nl, (const int32_t*)h->d_lmeta, sc),
                        hipLaunchKernelGGL((pf_materialise_kernel<T, false>), grid, dim3(256), 0, h->stream, LmView<T>{h->d_lmtab},
                                           (const int32_t*)h->d_tab[h->tside], h->n, h->nl, (const int32_t*)h->d_lmeta, sc));
        HIP_TRY(hipGetLastError());
        for (int l = 0; l < h->nl; ++l)
            if (work[l] >= 0) {
                pf_release_table(h, l);
                h->lbuf[l] = (int8_t)(pass == 0 ? (B ^ 1) : B);
            }
    }
    if (sh) {
        const int rcb = pf_peer_barrier(h);
        if (rcb) return rcb;
        // a barrier that timed out (a rank is gone, or -- several shards of ONE process -- two of their streams share a
        // hardware queue and the kernel that waits sits in front of the kernel it waits for) has let the passes run on
        // unfinished data: that must not pass silently
        int32_t err = 0;
        HIP_TRY(hipMemcpyAsync(&err, &h->d_ctl->error, sizeof(err), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (err) {
            slam_set_error("%s", pf_error_text(err));
            return SLAM_E_HIP;
        }
    }
    h->lazy_dirty = 0;
    return SLAM_OK;
}

// The lazy resampling step itself (whole filter local, d_anc = global = local ancestor ids).  Returns 1 in *done if it
// was performed, 0 if the caller must take the eager path (table pool exhausted).
static int pf_resample_lazy(slam_pf* h, const int32_t* d_anc, int* done, bool fused_fill = false) {
    *done = 0;
    TabList tl;
    tl.count = 0;
    tl.fresh = -1;
    int identity = 0;
    for (int l = 0; l < h->nl; ++l) identity += h->ltab[l] < 0;
    int free_idx = -1;
    for (int t = 0; t < PF_TAB_MAX; ++t) {
        if (h->tref[t] > 0) tl.idx[tl.count++] = (int16_t)t;
        else if (free_idx < 0) free_idx = t;
    }
    if (identity && free_idx < 0) return SLAM_OK;            // no table left: eager path (which resets all of this)
    if (identity) tl.fresh = free_idx;
    const int nxt = h->pcur ^ 1;
    if (fused_fill) {
        const double lw = -log((double)h->n_global);
        PF_DISPATCH(h,
                    hipLaunchKernelGGL(pf_lazy_apply_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                       (const T*)h->pose[h->pcur], (T*)h->pose[nxt], (const int32_t*)h->d_tab[h->tside],
                                       h->d_tab[h->tside ^ 1], (T*)h->logw, h->n, d_anc, tl, (T)lw),
                    hipLaunchKernelGGL(pf_lazy_apply_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                       (const T*)h->pose[h->pcur], (T*)h->pose[nxt], (const int32_t*)h->d_tab[h->tside],
                                       h->d_tab[h->tside ^ 1], (T*)h->logw, h->n, d_anc, tl, (T)lw));
    } else {
        PF_DISPATCH(h,
                    hipLaunchKernelGGL(pf_pose_gather_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                       (const T*)h->pose[h->pcur], (T*)h->pose[nxt], h->n, d_anc),
                    hipLaunchKernelGGL(pf_pose_gather_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                       (const T*)h->pose[h->pcur], (T*)h->pose[nxt], h->n, d_anc));
        hipLaunchKernelGGL(pf_compose_kernel, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (const int32_t*)h->d_tab[h->tside],
                           h->d_tab[h->tside ^ 1], h->n, d_anc, tl);
    }
    HIP_TRY(hipGetLastError());
    h->pcur = nxt;
    h->tside ^= 1;
    if (identity) {
        for (int l = 0; l < h->nl; ++l)
            if (h->ltab[l] < 0) h->ltab[l] = (int16_t)free_idx;
        h->tref[free_idx] = identity;
    }
    h->lazy_dirty = 1;
    *done = 1;
    return SLAM_OK;
}

extern "C" int slam_pf_set_pose(slam_pf_t h, const double pose[3]) {
    ARG_CHECK(h != nullptr && pose != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    h->has_pending = 0;                                // logw is overwritten: a deferred normalisation shift is moot
    h->pending_shift = 0.0;
    const double lw = -log((double)h->n_global);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_set_pose_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                   (T*)h->pose[h->pcur], (T*)h->logw, h->n, (T)pose[0], (T)pose[1], (T)pose[2], (T)lw),
                hipLaunchKernelGGL(pf_set_pose_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                   (T*)h->pose[h->pcur], (T*)h->logw, h->n, (T)pose[0], (T)pose[1], (T)pose[2], (T)lw));
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_pf_init_landmarks(slam_pf_t h, const double* lm_xy, int nl, double var, double jitter_sigma) {
    ARG_CHECK(h != nullptr && lm_xy != nullptr, "null argument");
    ARG_CHECK(nl >= 0 && nl <= h->nl, "more landmarks than capacity");
    if (nl == 0) return SLAM_OK;
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcm = pf_materialise(h); if (rcm) return rcm; }
    double* d_xy = nullptr;
    HIP_TRY(hipMalloc((void**)&d_xy, sizeof(double) * 2 * nl));
    HIP_TRY(hipMemcpyAsync(d_xy, lm_xy, sizeof(double) * 2 * nl, hipMemcpyHostToDevice, h->stream));
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_init_lm_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, LmView<T>{h->d_lmtab}, h->cur,
                                   h->n, h->first, h->seed, d_xy, nl, (T)var, (T)jitter_sigma),
                hipLaunchKernelGGL(pf_init_lm_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, LmView<T>{h->d_lmtab}, h->cur,
                                   h->n, h->first, h->seed, d_xy, nl, (T)var, (T)jitter_sigma));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void)hipFree(d_xy);
    for (int l = 0; l < nl; ++l) h->seen[l] = 1;
    return SLAM_OK;
}

extern "C" int slam_pf_predict(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && Q != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    const double sV = sqrt(Q[0]), sG = sqrt(Q[3]);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_predict_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->pose[h->pcur],
                                   h->n, h->first, h->step, h->seed, (T)V, (T)G, (T)wheelbase, (T)sV, (T)sG, (T)dt),
                hipLaunchKernelGGL(pf_predict_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->pose[h->pcur],
                                   h->n, h->first, h->step, h->seed, (T)V, (T)G, (T)wheelbase, (T)sV, (T)sG, (T)dt));
    HIP_TRY(hipGetLastError());
    h->step += 1;
    return SLAM_OK;
}

// slam_pf_normalize only RECORDS its shift; the next kernel that reads logw applies it (the fused step kernel takes it
// as a parameter, everything else flushes it first) -- one launch less per filter step.
static double pf_take_pending(slam_pf* h) {
    const double p = h->has_pending ? h->pending_shift : 0.0;
    h->has_pending = 0;
    h->pending_shift = 0.0;
    return p;
}

static int pf_flush_pending(slam_pf* h) {
    if (!h->has_pending) return SLAM_OK;
    const double shift = pf_take_pending(h);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_shift_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->logw, h->n, (T)shift),
                hipLaunchKernelGGL(pf_shift_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->logw, h->n, (T)shift));
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

// Stage m observations (ids recoded 0-based with the first-sighting flag) into the next staging slot and queue
// the copies; returns the slot's device addresses.  No stream synchronisation: a slot is reused only after the
// event behind its previous copies has fired.
static int pf_stage(slam_pf* h, const double* z, const int32_t* ids, int m, const double** d_z, const int32_t** d_i) {
    const int slot = h->stage_slot;
    h->stage_slot ^= 1;
    if (h->stage_used[slot]) HIP_TRY(hipEventSynchronize(h->stage_ev[slot]));
    int32_t* hi = h->h_ids + (size_t)slot * 2 * PF_OCAP;       // [codes | meta words]
    double* hz = h->h_obs + (size_t)slot * 2 * h->ocap;
    for (int i = 0; i < m; ++i) {
        if (ids) {                                       // (ids == nullptr: unknown correspondences, observations only)
            const int l = ids[i] - 1;
            hi[i] = l | (h->seen[l] == 0 ? NEW_FLAG : h->seen[l] == 2 ? FRESH_FLAG : 0);
            if (!h->seen[l]) h->seen[l] = 2;             // 2: first seen in this call
            // where the record is read and written: behind a table the update goes to the OTHER buffer (other particles
            // still read this slot) and the landmark is identity from then on; otherwise it is updated in place
            const int rb = h->lbuf[l];
            int wb = rb, tab = 0;
            if (h->ltab[l] >= 0) {
                tab = h->ltab[l] + 1;
                wb = rb ^ 1;
                pf_release_table(h, l);
                h->lbuf[l] = (int8_t)wb;
                if (wb != h->cur) h->lazy_dirty = 1;
            }
            if (h->prior[l] < 0) h->prior[l] = tab | (rb ? META_RBUF : 0);
            hi[PF_OCAP + i] = tab | (rb ? META_RBUF : 0) | (wb ? META_WBUF : 0) | (h->prior[l] << META_PRIOR_SHIFT);
        }
        hz[2 * i] = z[2 * i];
        hz[2 * i + 1] = z[2 * i + 1];
    }
    if (ids)
        for (int i = 0; i < m; ++i) {
            h->seen[ids[i] - 1] = 1;
            h->prior[ids[i] - 1] = -1;
        }
    // zero-copy: the kernel reads the pinned slot itself (once per workgroup, into LDS); the caller records the
    // slot's event behind that kernel (pf_stage_done)
    *d_z = h->h_obs_dev + (size_t)slot * 2 * h->ocap;
    *d_i = h->h_ids_dev + (size_t)slot * 2 * PF_OCAP;
    h->stage_last = slot;
    return SLAM_OK;
}

static int pf_stage_done(slam_pf* h) {
    HIP_TRY(hipEventRecord(h->stage_ev[h->stage_last], h->stream));
    h->stage_used[h->stage_last] = 1;
    return SLAM_OK;
}

static int pf_check_obs(slam_pf* h, const double* z, const int32_t* ids, int m, const double* R) {
    ARG_CHECK(m >= 0, "m < 0");
    if (m == 0) return SLAM_OK;
    ARG_CHECK(z != nullptr && ids != nullptr && R != nullptr, "null argument");
    ARG_CHECK(m <= h->ocap, "too many observations in one call (max 1024)");
    for (int i = 0; i < m; ++i) ARG_CHECK(ids[i] >= 1 && ids[i] <= h->nl, "landmark id out of range");
    return SLAM_OK;
}

extern "C" int slam_pf_update_known(slam_pf_t h, const double* z, const int32_t* ids, int m, const double R[4]) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    int rc = pf_check_obs(h, z, ids, m, R);
    if (rc || m == 0) return rc;
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    if (pf_sharded(h)) { const int rcm = pf_materialise(h); if (rcm) return rcm; }   // (collective: the legacy sweep reads no remote records)
    const double* dz;
    const int32_t* di;
    if ((rc = pf_stage(h, z, ids, m, &dz, &di))) return rc;
    const double pend = pf_take_pending(h);              // a deferred normalisation shift is applied on the way
    PF_DISPATCH(h,
                hipLaunchKernelGGL((pf_step_kernel<T, false, false>), dim3(grid_for(h->n)), dim3(256), (size_t)m * 24, h->stream,
                                   (T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->d_tab[h->tside], (T*)h->logw, h->n, h->first, 0u, h->seed, (T)0, (T)0,
                                   (T)1, (T)0, (T)0, (T)0, dz, di, m, (T)R[0], (T)R[1], (T)R[2], (T)R[3], (double*)nullptr, (T)pend),
                hipLaunchKernelGGL((pf_step_kernel<T, false, false>), dim3(grid_for(h->n)), dim3(256), (size_t)m * 24, h->stream,
                                   (T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->d_tab[h->tside], (T*)h->logw, h->n, h->first, 0u, h->seed, (T)0, (T)0,
                                   (T)1, (T)0, (T)0, (T)0, dz, di, m, (T)R[0], (T)R[1], (T)R[2], (T)R[3], (double*)nullptr, (T)pend));
    HIP_TRY(hipGetLastError());
    return pf_stage_done(h);
}

// fold the per-block partials in d_part and bring the seven numbers to the host
// wait (polling pinned memory) for the statistics published under sequence number h->out_seq
static int pf_wait_stats(slam_pf* h, double out[7]) {
    volatile long long* flag = reinterpret_cast<volatile long long*>(h->h_out + 7);
    unsigned long long spins = 0;
    while (*flag != h->out_seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xfffffull) == 0) {            // a failed kernel must not leave the host spinning
            const hipError_t q = hipStreamQuery(h->stream);
            if (q != hipErrorNotReady && *flag != h->out_seq) {
                slam_set_error("particle statistics were not published: %s", q == hipSuccess ? "kernel finished" : hipGetErrorString(q));
                return SLAM_E_HIP;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < 7; ++i) out[i] = h->h_out[i];
    return SLAM_OK;
}

static int pf_fold_and_read(slam_pf* h, int relative_to_max, double out[7]) {
    h->out_seq += 1;
    hipLaunchKernelGGL(pf_fold_kernel, dim3(1), dim3(256), 0, h->stream, (const double*)h->d_part, h->red_blocks,
                       relative_to_max, h->d_out, h->h_out_dev, h->out_seq);
    HIP_TRY(hipGetLastError());
    return pf_wait_stats(h, out);
}

/* F1 + F2/F3 + the local part of F4 as ONE sweep over the particles: predict, the m known-id updates and the weight
 * statistics {max logw, sum exp(logw - max), sum exp(2 (logw - max))}.  Same particles as slam_pf_predict +
 * slam_pf_update_known (bit for bit), same statistics as slam_pf_weight_stats.  Synchronises (the caller needs Neff). */
extern "C" int slam_pf_step(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt, const double* z,
                            const int32_t* ids, int m, const double R[4], double out[3]) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && Q != nullptr && out != nullptr, "null argument");
    double Rz[4] = {0, 0, 0, 0};
    int rc = pf_check_obs(h, z, ids, m, m ? R : Rz);
    if (rc) return rc;
    if (m) for (int i = 0; i < 4; ++i) Rz[i] = R[i];
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    if (pf_sharded(h)) { const int rcm = pf_materialise(h); if (rcm) return rcm; }   // (collective: the legacy sweep reads no remote records)
    const double* dz = h->h_obs_dev;
    const int32_t* di = h->h_ids_dev;
    if (m && (rc = pf_stage(h, z, ids, m, &dz, &di))) return rc;
    const double sV = sqrt(Q[0]), sG = sqrt(Q[3]);
    const double pend = pf_take_pending(h);
    PF_DISPATCH(h,
                hipLaunchKernelGGL((pf_step_kernel<T, true, true>), dim3(grid_for(h->n)), dim3(256), (size_t)m * 24, h->stream,
                                   (T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->d_tab[h->tside], (T*)h->logw, h->n, h->first, h->step, h->seed, (T)V,
                                   (T)G, (T)wheelbase, (T)sV, (T)sG, (T)dt, dz, di, m, (T)Rz[0], (T)Rz[1], (T)Rz[2], (T)Rz[3],
                                   h->d_part, (T)pend),
                hipLaunchKernelGGL((pf_step_kernel<T, true, true>), dim3(grid_for(h->n)), dim3(256), (size_t)m * 24, h->stream,
                                   (T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->d_tab[h->tside], (T*)h->logw, h->n, h->first, h->step, h->seed, (T)V,
                                   (T)G, (T)wheelbase, (T)sV, (T)sG, (T)dt, dz, di, m, (T)Rz[0], (T)Rz[1], (T)Rz[2], (T)Rz[3],
                                   h->d_part, (T)pend));
    HIP_TRY(hipGetLastError());
    if (m && (rc = pf_stage_done(h))) return rc;
    h->step += 1;
    double s[7];
    if ((rc = pf_fold_and_read(h, 1, s))) return rc;
    out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
    return SLAM_OK;
}

/* N4, FastSLAM 2.0: slam_pf_step with the pose drawn from the observation-aware proposal (pf_proposal_kernel).
 * Same arguments, same statistics; Q may be any symmetric positive definite 2 x 2 matrix (its Cholesky factor
 * shapes the control noise).  With m == 0 it is slam_pf_step bit for bit. */
extern "C" int slam_pf_step_proposal(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt,
                                     const double* z, const int32_t* ids, int m, const double R[4], double out[3]) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && Q != nullptr && out != nullptr, "null argument");
    double Rz[4] = {0, 0, 0, 0};
    int rc = pf_check_obs(h, z, ids, m, m ? R : Rz);
    if (rc) return rc;
    if (m) for (int i = 0; i < 4; ++i) Rz[i] = R[i];
    ARG_CHECK(Q[0] > 0.0, "Q is not positive definite");
    const double lq00 = sqrt(Q[0]), lq10 = 0.5 * (Q[1] + Q[2]) / lq00;
    ARG_CHECK(Q[3] - lq10 * lq10 > 0.0, "Q is not positive definite");
    const double lq11 = sqrt(Q[3] - lq10 * lq10);
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    if (pf_sharded(h)) { const int rcm = pf_materialise(h); if (rcm) return rcm; }   // (collective: the legacy sweep reads no remote records)
    const double* dz = h->h_obs_dev;
    const int32_t* di = h->h_ids_dev;
    if (m && (rc = pf_stage(h, z, ids, m, &dz, &di))) return rc;
    const double pend = pf_take_pending(h);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_proposal_kernel<T>, dim3(grid_for(h->n)), dim3(256), (size_t)m * 24, h->stream,
                                   (T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->d_tab[h->tside], (T*)h->logw, h->n, h->first, h->step, h->seed, (T)V,
                                   (T)G, (T)wheelbase, (T)lq00, (T)lq10, (T)lq11, (T)dt, dz, di, m, (T)Rz[0], (T)Rz[1], (T)Rz[2],
                                   (T)Rz[3], h->d_part, (T)pend),
                hipLaunchKernelGGL(pf_proposal_kernel<T>, dim3(grid_for(h->n)), dim3(256), (size_t)m * 24, h->stream,
                                   (T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->d_tab[h->tside], (T*)h->logw, h->n, h->first, h->step, h->seed, (T)V,
                                   (T)G, (T)wheelbase, (T)lq00, (T)lq10, (T)lq11, (T)dt, dz, di, m, (T)Rz[0], (T)Rz[1], (T)Rz[2],
                                   (T)Rz[3], h->d_part, (T)pend));
    HIP_TRY(hipGetLastError());
    if (m && (rc = pf_stage_done(h))) return rc;
    h->step += 1;
    double s[7];
    if ((rc = pf_fold_and_read(h, 1, s))) return rc;
    out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
    return SLAM_OK;
}

/* N4.  Every landmark slot of every particle unused (Pxx = -1 marks "no landmark here"). */
extern "C" int slam_pf_clear_landmarks(slam_pf_t h) {
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_clear_lm_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, LmView<T>{h->d_lmtab}, h->cur, h->n, h->nl),
                hipLaunchKernelGGL(pf_clear_lm_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, LmView<T>{h->d_lmtab}, h->cur, h->n, h->nl));
    HIP_TRY(hipGetLastError());
    for (int l = 0; l < h->nl; ++l) {
        h->seen[l] = 0;
        pf_release_table(h, l);          // every slot of buffer cur was just overwritten: plain maps again
        h->lbuf[l] = (int8_t)h->cur;
    }
    h->lazy_dirty = 0;
    return SLAM_OK;
}

/* N4.  m <= 16 (range, bearing) pairs with UNKNOWN correspondences: every particle associates them with its own
 * landmarks (gates as in associate(), src/data-association.jl:1-51), updates the matched ones, starts new landmarks
 * in its lowest unused slots.  d_assoc (device, [m][n] int32, may be NULL) receives the decisions: slot >= 0
 * matched, -1 new, -2 dropped.  Enqueued. */
extern "C" int slam_pf_update_unknown(slam_pf_t h, const double* z, int m, const double R[4], double gate1, double gate2,
                                      int32_t* d_assoc) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(m >= 0 && m <= UNK_MAX, "slam_pf_update_unknown takes at most 16 observations per call");
    if (m == 0) return SLAM_OK;
    ARG_CHECK(z != nullptr && R != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcm = pf_materialise(h); if (rcm) return rcm; }
    const double* dz;
    const int32_t* di;
    const int rc = pf_stage(h, z, nullptr, m, &dz, &di);      // the observation list goes through a staging slot
    if (rc) return rc;
    const double pend = pf_take_pending(h);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_update_unknown_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                   (const T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->cur, (T*)h->logw, h->n, h->nl, dz, m, (T)R[0], (T)R[1],
                                   (T)R[2], (T)R[3], (T)gate1, (T)gate2, (T)pend, d_assoc),
                hipLaunchKernelGGL(pf_update_unknown_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream,
                                   (const T*)h->pose[h->pcur], LmView<T>{h->d_lmtab}, h->cur, (T*)h->logw, h->n, h->nl, dz, m, (T)R[0], (T)R[1],
                                   (T)R[2], (T)R[3], (T)gate1, (T)gate2, (T)pend, d_assoc));
    HIP_TRY(hipGetLastError());
    return pf_stage_done(h);
}

/* slam_pf_step followed by slam_pf_normalize with the shard's OWN statistics, for a filter that lives on one GPU
 * (n == n_global): one library call per filter step.  out = {max logw, sum, sum2, Neff}. */
extern "C" int slam_pf_step_normalized(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt,
                                       const double* z, const int32_t* ids, int m, const double R[4], double out[4]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    ARG_CHECK(h->n == h->n_global, "slam_pf_step_normalized needs the whole filter on this shard");
    int rc = slam_pf_step(h, V, G, wheelbase, Q, dt, z, ids, m, R, out);
    if (rc) return rc;
    if ((rc = slam_pf_normalize(h, out[0], out[1]))) return rc;
    out[3] = out[1] * out[1] / out[2];
    return SLAM_OK;
}

// out = {max logw, sum exp(logw - max), sum exp(2(logw - max)),  sum w x, sum w y, sum w sin phi, sum w cos phi}
// with w = exp(logw - shift), shift = local max if relative_to_max else 0.
static int pf_stats(slam_pf* h, int relative_to_max, double out[7]) {
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcf = pf_flush_pending(h); if (rcf) return rcf; }
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_stats_kernel<T>, dim3(h->red_blocks), dim3(256), 0, h->stream, (const T*)h->logw,
                                   (const T*)h->pose[h->pcur], h->n, relative_to_max, h->d_part),
                hipLaunchKernelGGL(pf_stats_kernel<T>, dim3(h->red_blocks), dim3(256), 0, h->stream, (const T*)h->logw,
                                   (const T*)h->pose[h->pcur], h->n, relative_to_max, h->d_part));
    HIP_TRY(hipGetLastError());
    return pf_fold_and_read(h, relative_to_max, out);
}

extern "C" int slam_pf_weight_stats(slam_pf_t h, double out[3]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    double s[7];
    const int rc = pf_stats(h, 1, s);
    if (rc) return rc;
    out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
    return SLAM_OK;
}

extern "C" int slam_pf_mean_pose_sums(slam_pf_t h, double out[4]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    double s[7];
    const int rc = pf_stats(h, 0, s);
    if (rc) return rc;
    out[0] = s[3]; out[1] = s[4]; out[2] = s[5]; out[3] = s[6];
    return SLAM_OK;
}

extern "C" int slam_pf_normalize(slam_pf_t h, double gmax, double gsum) {
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(gsum > 0.0, "gsum must be positive");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    const double shift = gmax + log(gsum);
    const int rc = pf_flush_pending(h);          // (two normalisations in a row: the first shift is applied on its own)
    if (rc) return rc;
    h->pending_shift = shift;                    // applied by the next kernel that touches logw
    h->has_pending = 1;
    return SLAM_OK;
}

extern "C" int slam_pf_copy_logw(slam_pf_t h, void* d_dst) {
    ARG_CHECK(h != nullptr && d_dst != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcf = pf_flush_pending(h); if (rcf) return rcf; }
    HIP_TRY(hipMemcpyAsync(d_dst, h->logw, h->esz * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

static int pf_ancestors_impl(slam_pf_t h, const void* d_logw_all, double gmax, double u0, int32_t* d_anc, int64_t first,
                             int64_t count) {
    ARG_CHECK(h != nullptr && d_logw_all != nullptr && d_anc != nullptr, "null argument");
    ARG_CHECK(u0 >= 0.0 && u0 < 1.0, "u0 must be in [0, 1)");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    const int nb = (int)((h->n_global + SCAN_BLOCK - 1) / SCAN_BLOCK);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)d_logw_all,
                                   h->n_global, gmax, h->d_cdf, h->d_bsum, (T)0),
                hipLaunchKernelGGL(pf_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)d_logw_all,
                                   h->n_global, gmax, h->d_cdf, h->d_bsum, (T)0));
    hipLaunchKernelGGL(pf_scan2_kernel, dim3(1), dim3(256), 0, h->stream, h->d_bsum, nb);
    hipLaunchKernelGGL(pf_ancestor_kernel, dim3(grid_for(count)), dim3(256), 0, h->stream, h->d_cdf, h->d_bsum, nb, h->n_global,
                       first, count, u0, d_anc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_pf_ancestors(slam_pf_t h, const void* d_logw_all, double gmax, double u0, int32_t* d_anc) {
    ARG_CHECK(h != nullptr, "null handle");
    return pf_ancestors_impl(h, d_logw_all, gmax, u0, d_anc, h->first, h->n);
}

/* The ancestor of EVERY slot of the filter (n_global entries): every rank computes the same table from the
 * all-gathered weights, so each knows without further communication which of its particles every other rank needs. */
extern "C" int slam_pf_ancestors_all(slam_pf_t h, const void* d_logw_all, double gmax, double u0, int32_t* d_anc_all) {
    ARG_CHECK(h != nullptr, "null handle");
    return pf_ancestors_impl(h, d_logw_all, gmax, u0, d_anc_all, 0, h->n_global);
}

/* Resampling of a filter that lives WHOLLY on this shard, as one call: cdf of the stored weights (a pending
 * normalisation shift is applied on the fly), ancestors, then the lazy step -- poses, ancestor tables, uniform weights in
 * one kernel -- or, when the table pool is exhausted or SLAMHIP_PF_EAGER=1, the eager gather.  Same particles, bit for
 * bit, as slam_pf_copy_logw + slam_pf_ancestors + slam_pf_resample_apply.  gmax: the maximum of the (normalised)
 * log-weights.  Enqueued. */
extern "C" int slam_pf_resample_local(slam_pf_t h, double gmax, double u0) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(h->n == h->n_global, "slam_pf_resample_local needs the whole filter on this shard");
    ARG_CHECK(u0 >= 0.0 && u0 < 1.0, "u0 must be in [0, 1)");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    const double pend = pf_take_pending(h);
    const int nb = (int)((h->n_global + SCAN_BLOCK - 1) / SCAN_BLOCK);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)h->logw, h->n_global,
                                   gmax, h->d_cdf, h->d_bsum, (T)pend),
                hipLaunchKernelGGL(pf_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)h->logw, h->n_global,
                                   gmax, h->d_cdf, h->d_bsum, (T)pend));
    hipLaunchKernelGGL(pf_scan2_kernel, dim3(1), dim3(256), 0, h->stream, h->d_bsum, nb);
    hipLaunchKernelGGL(pf_ancestor_kernel, dim3(grid_for(h->n)), dim3(256), 0, h->stream, h->d_cdf, h->d_bsum, nb, h->n_global,
                       (int64_t)0, h->n, u0, h->d_anc);
    HIP_TRY(hipGetLastError());
    int done = 0;
    if (!h->lazy_off) {
        const int rc = pf_resample_lazy(h, h->d_anc, &done, true);
        if (rc) return rc;
    }
    if (done) return SLAM_OK;
    // eager: the stored weights must not carry the shift any more? they are overwritten by the apply -- nothing to flush
    return slam_pf_resample_apply(h, h->d_anc, nullptr, 0, nullptr);
}

extern "C" int slam_pf_record_rows(slam_pf_t h, int* rows) {
    ARG_CHECK(h != nullptr && rows != nullptr, "null argument");
    *rows = 3 + 5 * h->nl;
    return SLAM_OK;
}

extern "C" int slam_pf_pack(slam_pf_t h, const int32_t* d_local_idx, int cnt, void* d_records) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(cnt >= 0, "cnt < 0");
    if (cnt == 0) return SLAM_OK;
    ARG_CHECK(d_local_idx != nullptr && d_records != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcm = pf_materialise(h); if (rcm) return rcm; }
    const dim3 grid((cnt + 255) / 256, 3 + 5 * h->nl);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_pack_kernel<T>, grid, dim3(256), 0, h->stream, (const T*)h->pose[h->pcur],
                                   LmView<T>{h->d_lmtab}, h->cur, h->n, d_local_idx, cnt, (T*)d_records),
                hipLaunchKernelGGL(pf_pack_kernel<T>, grid, dim3(256), 0, h->stream, (const T*)h->pose[h->pcur],
                                   LmView<T>{h->d_lmtab}, h->cur, h->n, d_local_idx, cnt, (T*)d_records));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_pf_resample_apply(slam_pf_t h, const int32_t* d_anc, const int32_t* d_remote_ids, int nremote,
                                      const void* d_remote_records) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && d_anc != nullptr, "null argument");
    ARG_CHECK(nremote >= 0, "nremote < 0");
    ARG_CHECK(nremote == 0 || (d_remote_ids != nullptr && d_remote_records != nullptr), "remote buffers missing");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    const double lw = -log((double)h->n_global);      // uniform weights again
    if (nremote == 0 && h->n == h->n_global && !h->lazy_off) {
        // the whole filter is here: permute the poses, compose the ancestor tables, leave the maps where they are
        int done = 0;
        const int rcl = pf_resample_lazy(h, d_anc, &done);
        if (rcl) return rcl;
        if (done) {
            (void)pf_take_pending(h);
            PF_DISPATCH(h,
                        hipLaunchKernelGGL(pf_fill_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->logw, h->n, (T)lw),
                        hipLaunchKernelGGL(pf_fill_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->logw, h->n, (T)lw));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(h->stream));
            return SLAM_OK;
        }
    }
    { const int rcm = pf_materialise(h); if (rcm) return rcm; }
    hipLaunchKernelGGL(pf_src_kernel, dim3(grid_for(h->n)), dim3(256), 0, h->stream, d_anc, h->n, h->first, d_remote_ids,
                       nremote, h->d_src);
    const int nxt = h->cur ^ 1, pnxt = h->pcur ^ 1;
    const int nrows = 3 + 5 * h->nl;
    const dim3 grid(grid_for(h->n), (nrows + GATHER_ROWS - 1) / GATHER_ROWS);
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_gather_kernel<T>, grid, dim3(256), 0, h->stream, (const T*)h->pose[h->pcur],
                                   LmView<T>{h->d_lmtab}, h->cur, (T*)h->pose[pnxt], nxt, h->n, nrows, h->d_src,
                                   (const T*)d_remote_records, nremote),
                hipLaunchKernelGGL(pf_gather_kernel<T>, grid, dim3(256), 0, h->stream, (const T*)h->pose[h->pcur],
                                   LmView<T>{h->d_lmtab}, h->cur, (T*)h->pose[pnxt], nxt, h->n, nrows, h->d_src,
                                   (const T*)d_remote_records, nremote));
    (void)pf_take_pending(h);                          // logw is overwritten: a deferred shift is moot
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_fill_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->logw, h->n, (T)lw),
                hipLaunchKernelGGL(pf_fill_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (T*)h->logw, h->n, (T)lw));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->cur = nxt;
    h->pcur = pnxt;
    for (int l = 0; l < h->nl; ++l) h->lbuf[l] = (int8_t)nxt;      // (materialised above: identity tables, one buffer)
    return SLAM_OK;
}

extern "C" int slam_pf_download(slam_pf_t h, void* pose, void* logw, void* lm) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    const size_t n = (size_t)h->n;
    { const int rcf = pf_flush_pending(h); if (rcf) return rcf; }
    if (lm) { const int rcm = pf_materialise(h); if (rcm) return rcm; }
    if (pose) HIP_TRY(hipMemcpyAsync(pose, h->pose[h->pcur], h->esz * 3 * n, hipMemcpyDeviceToHost, h->stream));
    if (logw) HIP_TRY(hipMemcpyAsync(logw, h->logw, h->esz * n, hipMemcpyDeviceToHost, h->stream));
    if (lm)                          // chunk by chunk into the caller's contiguous [nl][5][n]
        for (int k = 0; k < h->lmtab.nchunks; ++k) {
            const size_t l0 = (size_t)k << h->lmtab.shift;
            const size_t lms = (size_t)h->nl - l0 < ((size_t)1 << h->lmtab.shift) ? (size_t)h->nl - l0 : ((size_t)1 << h->lmtab.shift);
            HIP_TRY(hipMemcpyAsync((char*)lm + h->esz * 5 * n * l0, h->lmtab.c[h->cur][k], h->esz * 5 * n * lms, hipMemcpyDeviceToHost,
                                   h->stream));
        }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_pf_sync(slam_pf_t h) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->auto_on) {                          // (steps queued by slam_pf_step_auto: a halted one is resolved on the way)
        const int rc = pf_auto_flush(h);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_pf_stream(slam_pf_t h, void** stream) {
    ARG_CHECK(h != nullptr && stream != nullptr, "null argument");
    *stream = (void*)h->stream;
    return SLAM_OK;
}

// ---- auto mode: host side -------------------------------------------------------------------------------------------
static int pf_auto_nb(const slam_pf* h) { return (int)((h->n_global + SCAN_BLOCK - 1) / SCAN_BLOCK); }

// may a step resample on the device?  The whole filter here, or a sharded one whose peers are attached.
static bool pf_auto_lazy_ok(const slam_pf* h) {
    if (h->lazy_off || pf_auto_nb(h) > AUTO_NB_MAX) return false;
    return h->xchg_world <= 1 ? h->n == h->n_global : pf_sharded(h);
}

static const char* pf_error_text(long long code) {
    switch (code) {
        case PF_ERR_HANDOVER: return "auto mode: a workgroup's statistics line never reached the collecting workgroup (2 s)";
        case PF_ERR_EXCHANGE: return "the scalar exchange between the ranks of the sharded filter timed out (a rank is gone)";
        case PF_ERR_PEER:
            return "a hand-shake between the ranks of the sharded filter timed out (a rank is gone; or, for shards of ONE process, "
                   "their streams share a hardware queue: set GPU_MAX_HW_QUEUES >= the number of shards + 2 before the first HIP call)";
        default: return "auto mode: the device reported an unknown error";
    }
}

// legacy mode -> auto mode: the host's bookkeeping becomes the device's
static int pf_auto_enter(slam_pf* h) {
    if (h->auto_on) return SLAM_OK;
    std::vector<int32_t> st(h->nl);
    for (int l = 0; l < h->nl; ++l)
        st[l] = (h->ltab[l] >= 0 ? h->ltab[l] + 1 : 0) | (h->lbuf[l] ? LS_BUF : 0) | (h->seen[l] ? LS_SEEN : 0);
    PfCtl c;
    memset(&c, 0, sizeof(c));
    c.shift_next = pf_take_pending(h);
    c.nresamples = (int32_t)h->nresamples;
    c.pcur = h->pcur;
    c.tside = h->tside;
    c.lwcur = h->lwcur;
    c.seq = h->auto_seq;
    for (int t = 0; t < PF_TAB_MAX; ++t) c.tref[t] = h->tref[t];
    for (int l = 0; l < h->nl; ++l) c.identity += h->ltab[l] < 0;
    HIP_TRY(hipMemcpyAsync(h->d_lmstate, st.data(), sizeof(int32_t) * h->nl, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_ctl, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));              // (pageable sources)
    const PfMirror keep = *h->h_mir;                        // (Neff and the statistics of the last confirmed step stay readable)
    memset(h->h_mir, 0, sizeof(PfMirror));
    h->h_mir->neff = keep.neff;
    for (int i = 0; i < 8; ++i) h->h_mir->stats[i] = keep.stats[i];
    h->h_mir->done_seq = h->log.empty() ? h->auto_seq : h->log.front().seq - 1;      // (a replay: the logged steps are still to come)
    h->pub_seq = h->h_mir->done_seq;                        // (publications asked for before a halt were not made)
    h->h_mir->nresamples = h->nresamples;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    h->auto_on = 1;
    return SLAM_OK;
}

// bring the device's bookkeeping back (the stream must be idle)
static int pf_auto_import(slam_pf* h, bool halted) {
    PfCtl c;
    std::vector<int32_t> st(h->nl);
    HIP_TRY(hipMemcpy(&c, h->d_ctl, sizeof(c), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(st.data(), h->d_lmstate, sizeof(int32_t) * h->nl, hipMemcpyDeviceToHost));
    h->lazy_dirty = 0;
    for (int l = 0; l < h->nl; ++l) {
        h->ltab[l] = (int16_t)((st[l] & LS_TAB) - 1);
        h->lbuf[l] = (int8_t)((st[l] & LS_BUF) ? 1 : 0);
        h->seen[l] = (st[l] & LS_SEEN) ? 1 : 0;
        if (h->ltab[l] >= 0 || h->lbuf[l] != h->cur) h->lazy_dirty = 1;
    }
    for (int t = 0; t < PF_TAB_MAX; ++t) h->tref[t] = c.tref[t];
    h->pcur = c.pcur;
    h->tside = c.tside;
    h->lwcur = c.lwcur;
    h->logw = h->logw2[h->lwcur];
    h->nresamples = c.nresamples;
    // a halted step has stored its weights but not yet normalised them: its shift is the pending one
    const double shift = halted ? c.shift_scan : c.shift_next;
    h->pending_shift = shift;
    h->has_pending = shift != 0.0;
    h->halt_gmax = c.gmax_norm;
    h->auto_on = 0;
    if (c.error) {
        slam_set_error("%s", pf_error_text(c.error));
        return SLAM_E_HIP;
    }
    return SLAM_OK;
}

static int pf_auto_enqueue(slam_pf* h, const PfStepRec& r) {
    PfAutoArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < r.m; ++i) { a.z[2 * i] = r.z[2 * i]; a.z[2 * i + 1] = r.z[2 * i + 1]; a.ids[i] = r.ids[i]; }
    a.pose0 = h->pose[0]; a.pose1 = h->pose[1]; a.lmtab = h->d_lmtab; a.logw0 = h->logw2[0]; a.logw1 = h->logw2[1];
    a.tab0 = h->d_tab[0]; a.tab1 = h->d_tab[1];
    a.n = h->n; a.first = h->first; a.n_global = h->n_global; a.seq = r.seq;
    a.seed = h->seed; a.step = r.rng_step;
    a.m = r.m; a.nl = h->nl; a.force = r.force; a.lazy_ok = pf_auto_lazy_ok(h) ? 1 : 0;
    a.rank = h->xchg_rank; a.world = h->xchg_world;
    a.rec_cap = (int)((h->n_global + 1023) / 1024) + PF_MAX_WORLD;
    a.publish = (r.seq % PF_PUBLISH_EVERY) == 0 ? 1 : 0;
    if (a.publish && r.seq > h->pub_seq) h->pub_seq = r.seq;
    a.V = r.V; a.G = r.G; a.wheelbase = r.wheelbase; a.dt = r.dt;
    if (r.proposal) {
        const double lq00 = sqrt(r.Q[0]), lq10 = 0.5 * (r.Q[1] + r.Q[2]) / lq00;
        a.a0 = lq00; a.a1 = lq10; a.a2 = sqrt(r.Q[3] - lq10 * lq10);
    } else {
        a.a0 = sqrt(r.Q[0]); a.a1 = sqrt(r.Q[3]);
    }
    a.R00 = r.R[0]; a.R10 = r.R[1]; a.R01 = r.R[2]; a.R11 = r.R[3];
    a.neff_frac = r.neff_frac;
    a.part = h->d_part; a.ctl = h->d_ctl; a.lmstate = h->d_lmstate; a.mir = h->h_mir_dev; a.xchg = h->d_xchg;
    const bool sh = pf_sharded(h);
    a.peers = sh ? h->d_peers : nullptr;
    a.inbox = h->inbox;
    const dim3 grid(grid_for(h->n));
    // small filter / shard: the observations in parallel (pf_auto_step_par_kernel) -- FastSLAM-1.0 step, no landmark twice
    // in the call; above PF_PAR_MAX_N particles the sequential sweep already fills the chip
    bool distinct = !r.proposal && r.m >= 2;
    for (int i = 1; i < r.m && distinct; ++i)
        for (int j = 0; j < i; ++j)
            if (r.ids[i] == r.ids[j]) { distinct = false; break; }
    const bool par = distinct && h->n <= h->par_max_n;
    // between the 8-way kernel's range and the size at which the sequential sweep fills the chip: 4 and 2 ways on
    // 256-particle workgroups (pf_auto_step_way_kernel).  fp64 keeps to 2 ways (170 registers: no 1024-thread workgroup).
    int ways = 0;
    if (distinct && !par) {
        const int mo = h->dtype == SLAM_F32 ? WAY_MAXOBS<float> : WAY_MAXOBS<double>;
        if (r.m <= mo && r.m >= 4) {
            if (h->n <= h->way4_max_n) ways = h->dtype == SLAM_F32 ? 4 : 2;
            else if (h->n <= h->way2_max_n) ways = 2;
        }
    }
    if (ways) {
        const dim3 wgrid((unsigned)((h->n + 255) / 256));
#define PF_WAY_LAUNCH(TT, WW)                                                                                                \
    do {                                                                                                                     \
        if (sh) hipLaunchKernelGGL((pf_auto_step_way_kernel<TT, true, WW>), wgrid, dim3(256 * WW), 0, h->stream, a);         \
        else hipLaunchKernelGGL((pf_auto_step_way_kernel<TT, false, WW>), wgrid, dim3(256 * WW), 0, h->stream, a);           \
    } while (0)
        if (h->dtype == SLAM_F32) { if (ways == 4) PF_WAY_LAUNCH(float, 4); else PF_WAY_LAUNCH(float, 2); }
        else PF_WAY_LAUNCH(double, 2);
#undef PF_WAY_LAUNCH
    } else if (par) {
        const dim3 pgrid((unsigned)((h->n + 63) / 64));
        if (h->dtype == SLAM_F32) {
            if (sh) hipLaunchKernelGGL((pf_auto_step_par_kernel<float, true>), pgrid, dim3(64 * PAR_WAVES), 0, h->stream, a);
            else hipLaunchKernelGGL((pf_auto_step_par_kernel<float, false>), pgrid, dim3(64 * PAR_WAVES), 0, h->stream, a);
        } else {
            if (sh) hipLaunchKernelGGL((pf_auto_step_par_kernel<double, true>), pgrid, dim3(64 * PAR_WAVES), 0, h->stream, a);
            else hipLaunchKernelGGL((pf_auto_step_par_kernel<double, false>), pgrid, dim3(64 * PAR_WAVES), 0, h->stream, a);
        }
    } else
#define PF_STEP_LAUNCH(TT)                                                                                                   \
    do {                                                                                                                     \
        if (sh) {                                                                                                            \
            if (r.proposal) hipLaunchKernelGGL((pf_auto_step_kernel<TT, true, true>), grid, dim3(256), 0, h->stream, a);     \
            else hipLaunchKernelGGL((pf_auto_step_kernel<TT, false, true>), grid, dim3(256), 0, h->stream, a);               \
        } else {                                                                                                             \
            if (r.proposal) hipLaunchKernelGGL((pf_auto_step_kernel<TT, true, false>), grid, dim3(256), 0, h->stream, a);    \
            else hipLaunchKernelGGL((pf_auto_step_kernel<TT, false, false>), grid, dim3(256), 0, h->stream, a);              \
        }                                                                                                                    \
    } while (0)
    {
        if (h->dtype == SLAM_F32) PF_STEP_LAUNCH(float);
        else PF_STEP_LAUNCH(double);
    }
#undef PF_STEP_LAUNCH
    HIP_TRY(hipGetLastError());
    if (a.lazy_ok && r.force != 0) {                        // (force == 0: this step never resamples, nothing to gate)
        const int nb = pf_auto_nb(h);
        const double lw = -log((double)h->n_global);
        const PfPeers* pp = sh ? h->d_peers : nullptr;
        const int rank = h->xchg_rank, world = sh ? h->xchg_world : 1;
        if (sh) hipLaunchKernelGGL(pf_peer_gate_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ctl, r.seq, pp, h->inbox, rank, world);
        PF_DISPATCH(h,
                    hipLaunchKernelGGL(pf_auto_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)h->logw2[0],
                                       (const T*)h->logw2[1], h->n, h->n_global, (const PfCtl*)h->d_ctl, r.seq, h->d_cdf, h->d_bsum, pp,
                                       rank, world),
                    hipLaunchKernelGGL(pf_auto_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)h->logw2[0],
                                       (const T*)h->logw2[1], h->n, h->n_global, (const PfCtl*)h->d_ctl, r.seq, h->d_cdf, h->d_bsum, pp,
                                       rank, world));
#define PF_RESAMPLE_LAUNCH(SHV)                                                                                              \
        PF_DISPATCH(h,                                                                                                       \
                    hipLaunchKernelGGL((pf_auto_resample_kernel<T, SHV>), grid, dim3(256), 0, h->stream, (T*)h->pose[0], (T*)h->pose[1], \
                                       h->d_tab[0], h->d_tab[1], (T*)h->logw2[0], (T*)h->logw2[1], h->n, h->first, h->n_global,     \
                                       (const PfCtl*)h->d_ctl, r.seq, (const double*)h->d_cdf, (const double*)h->d_bsum, nb,      \
                                       h->d_anc, (T)lw, pp, rank, world),                                                     \
                    hipLaunchKernelGGL((pf_auto_resample_kernel<T, SHV>), grid, dim3(256), 0, h->stream, (T*)h->pose[0], (T*)h->pose[1], \
                                       h->d_tab[0], h->d_tab[1], (T*)h->logw2[0], (T*)h->logw2[1], h->n, h->first, h->n_global,     \
                                       (const PfCtl*)h->d_ctl, r.seq, (const double*)h->d_cdf, (const double*)h->d_bsum, nb,      \
                                       h->d_anc, (T)lw, pp, rank, world))
        if (sh) PF_RESAMPLE_LAUNCH(true);
        else PF_RESAMPLE_LAUNCH(false);
#undef PF_RESAMPLE_LAUNCH
        HIP_TRY(hipGetLastError());
    }
    return SLAM_OK;
}

// wait (polling the pinned mirror) until step `target` is confirmed or a step has halted
static int pf_auto_wait(slam_pf* h, long long target) {
    volatile long long* done = &h->h_mir->done_seq;
    volatile long long* halt = &h->h_mir->halt_seq;
    unsigned long long spins = 0;
    while (*done < target && *halt == 0) {
        __builtin_ia32_pause();
        if ((++spins & 0xfffffull) == 0) {                // a failed kernel must not leave the host spinning
            const hipError_t q = hipStreamQuery(h->stream);
            if (q != hipErrorNotReady && *done < target && *halt == 0) {
                slam_set_error("auto mode: step %lld was not confirmed: %s", target,
                               q == hipSuccess ? "the stream is idle" : hipGetErrorString(q));
                return SLAM_E_HIP;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return SLAM_OK;
}

static void pf_auto_trim(slam_pf* h) {
    const long long done = h->h_mir->done_seq;
    size_t k = 0;
    while (k < h->log.size() && h->log[k].seq <= done) ++k;
    if (k) h->log.erase(h->log.begin(), h->log.begin() + k);
}

static int pf_auto_replay(slam_pf* h) {
    int rc = pf_auto_enter(h);
    if (rc) return rc;
    for (const PfStepRec& r : h->log)
        if ((rc = pf_auto_enqueue(h, r))) return rc;
    return SLAM_OK;
}

// A step has halted: its sweep is done, its resampling is not, everything queued behind it was skipped.  Returns
// SLAM_PF_HALTED when the caller has to resample (sharded filter); a filter that lives on this shard resamples here
// (the legacy path: lazy if a table is free, else the eager gather) and the skipped steps are enqueued again.
static int pf_auto_handle_halt(slam_pf* h) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    const long long s = h->h_mir->halt_seq;
    h->h_mir->done_seq = s;                                // (it is: the halting tail publishes both)
    pf_auto_trim(h);
    int rc = pf_auto_import(h, true);
    if (rc) return rc;
    h->last_resampled_seq = s;
    if (h->n != h->n_global) {
        h->halted = 1;
        h->halts += 1;
        return SLAM_PF_HALTED;
    }
    const double u0 = resample_offset((uint32_t)h->nresamples, h->seed);
    if ((rc = slam_pf_resample_local(h, h->halt_gmax, u0))) return rc;
    h->nresamples += 1;
    return pf_auto_replay(h);
}

static int pf_auto_flush(slam_pf* h) {
    while (h->auto_on) {
        if (h->pub_seq < h->auto_seq) {                    // the last step does not publish by itself: ask for it
            hipLaunchKernelGGL(pf_auto_publish_kernel, dim3(1), dim3(64), 0, h->stream, (const PfCtl*)h->d_ctl, h->h_mir_dev);
            HIP_TRY(hipGetLastError());
            h->pub_seq = h->auto_seq;
        }
        int rc = pf_auto_wait(h, h->auto_seq);
        if (rc) return rc;
        if (h->h_mir->halt_seq != 0) {
            if ((rc = pf_auto_handle_halt(h))) return rc;
            continue;
        }
        break;
    }
    if (h->auto_on) {
        pf_auto_trim(h);
        h->nresamples = h->h_mir->nresamples;
        if (h->h_mir->resampled_seq > h->last_resampled_seq) h->last_resampled_seq = h->h_mir->resampled_seq;
        h->last_out[0] = h->h_mir->neff;
        h->last_out[1] = h->last_resampled_seq == h->auto_seq ? 1.0 : 0.0;
        h->last_out[2] = (double)h->nresamples;
        h->last_out[3] = (double)h->auto_seq;
        if (h->h_mir->error) {
            slam_set_error("%s", pf_error_text(h->h_mir->error));
            return SLAM_E_HIP;
        }
    }
    return SLAM_OK;
}

static int pf_auto_leave(slam_pf* h) {
    if (!h->auto_on) return SLAM_OK;
    int rc = pf_auto_flush(h);
    if (rc) return rc;
    if (!h->auto_on) return SLAM_OK;                       // (a halt was handled on the way and left us in legacy mode)
    HIP_TRY(hipStreamSynchronize(h->stream));
    return pf_auto_import(h, false);
}

/* One filter step with NO answer needed from the host: predict (or the FastSLAM-2.0 proposal), the m <= 64 known-id
 * updates, the weight statistics, the normalisation, Neff, the decision to resample (force < 0: Neff < neff_frac *
 * n_global; 0 / 1: never / always) and -- for a filter that lives wholly on this shard -- the (lazy) resampling itself,
 * all on the device and all enqueued: the call returns at once and steps queue back to back.  Same particles as
 * slam_pf_step + slam_pf_normalize + slam_pf_resample_local.  Returns SLAM_PF_HALTED (1, nothing was enqueued by THIS
 * call) when an earlier step of a SHARDED filter decided to resample: the caller exchanges the weights and the
 * migrating records with the legacy entry points, calls slam_pf_resume and repeats the call. */
extern "C" int slam_pf_step_auto(slam_pf_t h, double V, double G, double wheelbase, const double Q[4], double dt, const double* z,
                                 const int32_t* ids, int m, const double R[4], double neff_frac, int force, int proposal) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && Q != nullptr, "null argument");
    ARG_CHECK(m >= 0 && m <= PF_AUTO_MAXOBS, "slam_pf_step_auto takes at most 64 observations per call");
    ARG_CHECK(m == 0 || (z != nullptr && ids != nullptr && R != nullptr), "null argument");
    for (int i = 0; i < m; ++i) ARG_CHECK(ids[i] >= 1 && ids[i] <= h->nl, "landmark id out of range");
    ARG_CHECK(!h->halted, "a halted step is waiting for slam_pf_resume");
    if (proposal) {
        ARG_CHECK(Q[0] > 0.0, "Q is not positive definite");
        const double lq10 = 0.5 * (Q[1] + Q[2]) / sqrt(Q[0]);
        ARG_CHECK(Q[3] - lq10 * lq10 > 0.0, "Q is not positive definite");
    }
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if (!h->auto_on && (rc = pf_auto_enter(h))) return rc;
    if (h->h_mir->halt_seq != 0 && (rc = pf_auto_handle_halt(h))) return rc;
    pf_auto_trim(h);
    while ((int)h->log.size() >= PF_LOG - 1) {             // the host is a whole log ahead: wait for the oldest step
        if ((rc = pf_auto_wait(h, h->log.front().seq))) return rc;
        if (h->h_mir->halt_seq != 0 && (rc = pf_auto_handle_halt(h))) return rc;
        pf_auto_trim(h);
    }
    PfStepRec r;
    memset(&r, 0, sizeof(r));
    r.seq = ++h->auto_seq;
    r.rng_step = h->step++;
    r.m = m; r.force = force; r.proposal = proposal ? 1 : 0;
    r.V = V; r.G = G; r.wheelbase = wheelbase; r.dt = dt; r.neff_frac = neff_frac;
    for (int i = 0; i < 4; ++i) { r.Q[i] = Q[i]; r.R[i] = m ? R[i] : 0.0; }
    for (int i = 0; i < m; ++i) { r.z[2 * i] = z[2 * i]; r.z[2 * i + 1] = z[2 * i + 1]; r.ids[i] = ids[i]; }
    h->log.push_back(r);
    return pf_auto_enqueue(h, h->log.back());
}

/* Wait for everything slam_pf_step_auto has queued.  out (may be NULL) = {Neff of the last step, 1 if it resampled,
 * resamplings so far, steps so far}.  SLAM_PF_HALTED as for slam_pf_step_auto. */
extern "C" int slam_pf_flush(slam_pf_t h, double out[4]) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->halted) return SLAM_PF_HALTED;
    const int rc = pf_auto_flush(h);
    if (rc) return rc;
    if (!h->auto_on) HIP_TRY(hipStreamSynchronize(h->stream));
    if (out) for (int i = 0; i < 4; ++i) out[i] = h->last_out[i];
    return SLAM_OK;
}

/* After SLAM_PF_HALTED and the caller's resampling (slam_pf_copy_logw ... slam_pf_resample_apply): the skipped steps are
 * enqueued again.  `resamplings`: the caller's count after its resampling (the offset of the next one derives from it). */
extern "C" int slam_pf_resume(slam_pf_t h, int64_t resamplings) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(h->halted, "nothing is halted");
    HIP_TRY(hipSetDevice(h->device));
    h->halted = 0;
    h->nresamples = resamplings;
    return pf_auto_replay(h);
}

/* The halted step's numbers for the caller's resampling: out = {largest normalised log-weight, resamplings so far}. */
extern "C" int slam_pf_halt_info(slam_pf_t h, double out[2]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    ARG_CHECK(h->halted, "nothing is halted");
    out[0] = h->halt_gmax;
    out[1] = (double)h->nresamples;
    return SLAM_OK;
}

extern "C" int slam_pf_resample_count(slam_pf_t h, int64_t* count) {
    ARG_CHECK(h != nullptr && count != nullptr, "null argument");
    if (h->auto_on && !h->halted) {                        // the mirror is current only after a publication: ask for one
        HIP_TRY(hipSetDevice(h->device));
        const int rc = pf_auto_flush(h);
        if (rc && rc != SLAM_PF_HALTED) return rc;
    }
    *count = h->auto_on ? (int64_t)h->h_mir->nresamples : (int64_t)h->nresamples;
    return SLAM_OK;
}

extern "C" int slam_pf_set_resample_count(slam_pf_t h, int64_t count) {
    ARG_CHECK(h != nullptr && count >= 0, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    h->nresamples = count;
    return SLAM_OK;
}

/* The ranks' shared scalar page of a sharded filter: `page` is host memory that EVERY rank has mapped (one file in
 * /dev/shm), at least 2 * world * 64 bytes, zero-filled before the first step.  It is registered with the HIP runtime
 * here; the step kernel's last workgroup writes this rank's (max, sum w, sum w^2, step) into its slot and polls the
 * others' -- the per-step all-gather of three scalars without a host in the loop. */
extern "C" int slam_pf_attach_exchange(slam_pf_t h, int rank, int world, void* page, size_t bytes) {
    ARG_CHECK(h != nullptr && page != nullptr, "null argument");
    ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    ARG_CHECK(bytes >= (size_t)2 * world * 64, "the page is too small");
    ARG_CHECK(h->n * world == h->n_global && h->first == (int64_t)rank * h->n, "ranks must own equal, contiguous slices in rank order");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    if (h->xchg_host) { (void)hipHostUnregister(h->xchg_host); h->xchg_host = nullptr; h->d_xchg = nullptr; }
    HIP_TRY(hipHostRegister(page, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    h->xchg_host = page;
    HIP_TRY(hipHostGetDevicePointer((void**)&h->d_xchg, page, 0));
    h->xchg_rank = rank;
    h->xchg_world = world;
    return SLAM_OK;
}

/* ---- sharding behind the C ABI: peers -----------------------------------------------------------------------------
 * One process per GPU.  Every rank exports a blob (slam_pf_export_peer: IPC handles of its state buffers and of its
 * inbox page, or -- same process, e.g. one host thread per GPU -- the raw device pointers), the caller moves the blobs
 * between the ranks by whatever it has (MPI, files, torch.distributed ...), and every rank attaches all of them in
 * rank order.  From then on slam_pf_step_auto resamples the sharded filter on the device (no SLAM_PF_HALTED). */
constexpr int PF_BLOB_FIXED = 7;                                  // pose0, pose1, logw0, logw1, tab0, tab1, inbox
constexpr int PF_BLOB_MAXH = PF_BLOB_FIXED + 2 * PF_LM_MAXC;      // ... then the landmark chunks: buffer 0's, buffer 1's
struct PfPeerBlob {
    uint64_t magic;
    int64_t pid;
    int32_t device, dtype, nl, lm_shift, lm_nchunks, reserved;
    int64_t n, n_global;
    uint64_t lm_chunk_bytes, inbox_bytes;
    void* raw[PF_BLOB_MAXH];
    hipIpcMemHandle_t ipc[PF_BLOB_MAXH];
};
static_assert(sizeof(PfPeerBlob) <= SLAM_PF_PEER_BLOB_BYTES, "peer blob");
constexpr uint64_t PF_BLOB_MAGIC = 0x534c414d50465034ull;      // "SLAMPFP4"
// What an IPC mapping may carry on this runtime (ROCm 7.2, dmabuf IPC; DESIGN section 7 has the records):
//   * hipIpcOpenMemHandle of an allocation above 2 GiB never returns (1.91 GiB opens in milliseconds, 2.50 GiB hangs both
//     processes): every exported buffer must stay below PF_IPC_MAX_BYTES -- the landmark records are chunked for that reason;
//   * the import of a FINE-GRAINED (hipExtMallocWithFlags) allocation larger than one 2 MiB fragment was seen with only its
//     first 2 MiB mapped (tools/ipc_probe.hip: page fault at import + 2 MiB in 3 of 7 runs; plain hipMalloc imports of the same
//     size never): the only fine-grained export is the inbox, which must stay within PF_IPC_FINE_MAX_BYTES.
// slam_pf_attach_peers checks both BEFORE opening anything and refuses with SLAM_E_CAPACITY (the caller keeps the halting flow).
constexpr uint64_t PF_IPC_MAX_BYTES = 2047ull << 20;
constexpr uint64_t PF_IPC_FINE_MAX_BYTES = 2ull << 20;

static int pf_blob_handles(const slam_pf* h, void* ptrs[PF_BLOB_MAXH]) {
    ptrs[0] = h->pose[0]; ptrs[1] = h->pose[1]; ptrs[2] = h->logw2[0]; ptrs[3] = h->logw2[1];
    ptrs[4] = h->d_tab[0]; ptrs[5] = h->d_tab[1]; ptrs[6] = h->inbox;
    int cnt = PF_BLOB_FIXED;
    for (int b = 0; b < 2; ++b)
        for (int k = 0; k < h->lmtab.nchunks; ++k) ptrs[cnt++] = h->lmtab.c[b][k];
    return cnt;
}

extern "C" int slam_pf_export_peer(slam_pf_t h, void* blob) {
    ARG_CHECK(h != nullptr && blob != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PfPeerBlob b;
    memset(&b, 0, sizeof(b));
    b.magic = PF_BLOB_MAGIC;
    b.pid = (int64_t)getpid();
    b.device = h->device; b.dtype = h->dtype; b.nl = h->nl; b.n = h->n; b.n_global = h->n_global;
    b.lm_shift = h->lmtab.shift; b.lm_nchunks = h->lmtab.nchunks; b.lm_chunk_bytes = h->lm_chunk_bytes; b.inbox_bytes = h->inbox_bytes;
    void* ptrs[PF_BLOB_MAXH];
    const int cnt = pf_blob_handles(h, ptrs);
    for (int i = 0; i < cnt; ++i) {
        b.raw[i] = ptrs[i];
        HIP_TRY(hipIpcGetMemHandle(&b.ipc[i], ptrs[i]));
    }
    memset(blob, 0, SLAM_PF_PEER_BLOB_BYTES);
    memcpy(blob, &b, sizeof(b));
    return SLAM_OK;
}

static void pf_detach_peers_impl(slam_pf* h) {
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int r = 0; r < PF_MAX_WORLD; ++r)
        for (int i = 0; i < PF_BLOB_MAXH; ++i)
            if (h->peer_open[r][i]) {
                (void)hipIpcCloseMemHandle(h->peer_open[r][i]);
                h->peer_open[r][i] = nullptr;
            }
    (void)hipDeviceSynchronize();       // the unmaps have taken effect before anybody frees (and re-exports) the memory behind them
    if (h->d_peers) { (void)hipFree(h->d_peers); h->d_peers = nullptr; }
    memset(&h->peers, 0, sizeof(h->peers));
}

extern "C" int slam_pf_detach_peers(slam_pf_t h) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcm = pf_materialise(h); if (rcm) return rcm; }      // (collective: no remote references may stay behind)
    pf_detach_peers_impl(h);
    if (!h->xchg_host) { h->xchg_rank = 0; h->xchg_world = 1; }
    return SLAM_OK;
}

extern "C" int slam_pf_attach_peers(slam_pf_t h, int rank, int world, const void* blobs) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && blobs != nullptr, "null argument");
    ARG_CHECK(world >= 1 && world <= PF_MAX_WORLD && rank >= 0 && rank < world, "rank / world out of range (at most 8 ranks)");
    ARG_CHECK(h->n * world == h->n_global && h->first == (int64_t)rank * h->n, "ranks must own equal, contiguous slices in rank order");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    ARG_CHECK(!pf_sharded(h), "peers are already attached (slam_pf_detach_peers first)");
    pf_detach_peers_impl(h);
    const int64_t me = (int64_t)getpid();
    PfPeers t;
    memset(&t, 0, sizeof(t));
    // pass 0 checks every blob (nothing is opened before all of them are acceptable), pass 1 opens
    for (int pass = 0; pass < 2; ++pass)
    for (int r = 0; r < world; ++r) {
        PfPeerBlob b;
        memcpy(&b, (const char*)blobs + (size_t)r * SLAM_PF_PEER_BLOB_BYTES, sizeof(b));
        const int cnt = PF_BLOB_FIXED + 2 * b.lm_nchunks;
        if (pass == 0) {
            ARG_CHECK(b.magic == PF_BLOB_MAGIC, "a peer blob is not one of slam_pf_export_peer's");
            ARG_CHECK(b.n == h->n && b.nl == h->nl && b.dtype == h->dtype && b.n_global == h->n_global, "the peers' shards differ in size or type");
            ARG_CHECK(b.lm_shift == h->lmtab.shift && b.lm_nchunks == h->lmtab.nchunks && b.lm_nchunks >= 1 && b.lm_nchunks <= PF_LM_MAXC,
                      "the peers' landmark chunking differs");
            if (r == rank) ARG_CHECK(b.pid == me && b.raw[0] == h->pose[0], "blob [rank] is not this handle's own export");
            if (r != rank && b.pid != me) {
                // the shapes this runtime's IPC mappings cannot carry (see PF_IPC_MAX_BYTES): refuse BEFORE opening anything
                const uint64_t n64 = (uint64_t)b.n, esz = (uint64_t)h->esz;
                const uint64_t largest = std::max<uint64_t>(std::max<uint64_t>(3 * n64 * esz, (uint64_t)PF_TAB_MAX * n64 * 4), b.lm_chunk_bytes);
                if (largest > PF_IPC_MAX_BYTES) {
                    slam_set_error("rank %d exports a buffer of %.2f GiB: above the 2 GiB an IPC mapping can carry on this runtime "
                                   "(use more ranks, or the halting flow)", r, (double)largest / 1073741824.0);
                    return SLAM_E_CAPACITY;
                }
                if (b.inbox_bytes > PF_IPC_FINE_MAX_BYTES) {
                    slam_set_error("rank %d's inbox is %.2f MiB: a fine-grained allocation above 2 MiB is not exported (it was seen "
                                   "half mapped on this runtime); use the halting flow for a filter of this size", r,
                                   (double)b.inbox_bytes / 1048576.0);
                    return SLAM_E_CAPACITY;
                }
            }
            continue;
        }
        void* ptr[PF_BLOB_MAXH];
        if (r == rank || b.pid == me) {                     // this handle, or a shard of this very process: plain pointers
            if (r != rank && b.device != h->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
                (void)hipGetLastError();
            }
            for (int i = 0; i < cnt; ++i) ptr[i] = b.raw[i];
        } else {
            for (int i = 0; i < cnt; ++i) {
                const hipError_t e = hipIpcOpenMemHandle(&ptr[i], b.ipc[i], hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) {
                    slam_set_error("hipIpcOpenMemHandle of rank %d's buffer %d failed: %s", r, i, hipGetErrorString(e));
                    pf_detach_peers_impl(h);
                    return SLAM_E_HIP;
                }
                h->peer_open[r][i] = ptr[i];
            }
        }
        t.pose[r][0] = ptr[0]; t.pose[r][1] = ptr[1]; t.logw[r][0] = ptr[2]; t.logw[r][1] = ptr[3];
        t.tab[r][0] = (int32_t*)ptr[4]; t.tab[r][1] = (int32_t*)ptr[5];
        t.inbox[r] = (PfInbox*)ptr[6];
        t.lm[r].shift = b.lm_shift; t.lm[r].nchunks = b.lm_nchunks;
        for (int bb = 0; bb < 2; ++bb)
            for (int k = 0; k < b.lm_nchunks; ++k) t.lm[r].c[bb][k] = ptr[PF_BLOB_FIXED + bb * b.lm_nchunks + k];
    }
    h->peers = t;
    HIP_TRY(hipMalloc((void**)&h->d_peers, sizeof(PfPeers)));
    HIP_TRY(hipMemcpy(h->d_peers, &t, sizeof(t), hipMemcpyHostToDevice));
    h->xchg_rank = rank;
    h->xchg_world = world;
    return SLAM_OK;
}

/* A barrier among the attached ranks through their inboxes (every rank writes a word into every peer's inbox and polls its
 * own): collective, synchronises.  SLAM_OK when every peer's word arrived within timeout_ms -- the caller's check that the
 * GPUs really see each other's writes before it relies on the device-side exchange (it can fall back to the halting
 * flow otherwise). */
extern "C" int slam_pf_peer_selftest(slam_pf_t h, int timeout_ms) {
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(pf_sharded(h), "no peers attached");
    ARG_CHECK(timeout_ms > 0, "timeout must be positive");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    int32_t* d_err = nullptr;
    HIP_TRY(hipMalloc((void**)&d_err, sizeof(int32_t)));
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(int32_t), h->stream);
    h->bar_count += 1;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pf_peer_barrier_kernel, dim3(1), dim3(64), 0, h->stream, d_err, (unsigned long long)h->bar_count,
                           (const PfPeers*)h->d_peers, h->inbox, h->xchg_rank, h->xchg_world, (unsigned long long)timeout_ms * 100000ull);
        e = hipGetLastError();
    }
    int32_t err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&err, d_err, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_err);
    if (e != hipSuccess) {
        slam_set_error("HIP error in slam_pf_peer_selftest: %s", hipGetErrorString(e));
        return SLAM_E_HIP;
    }
    if (err) {
        slam_set_error("peer self-test: a peer's word did not arrive within %d ms", timeout_ms);
        return SLAM_E_HIP;
    }
    return SLAM_OK;
}

/* out = {ranks of the filter, 1 if peers are attached (device-side resampling of the sharded filter), SLAM_PF_HALTED
 * returns so far, resamplings so far}. */
extern "C" int slam_pf_comm_info(slam_pf_t h, int64_t out[4]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    out[0] = h->xchg_world;
    out[1] = pf_sharded(h) ? 1 : 0;
    out[2] = h->halts;
    int64_t cnt = 0;
    const int rc = slam_pf_resample_count(h, &cnt);
    if (rc) return rc;
    out[3] = cnt;
    return SLAM_OK;
}

/* normalise, and resample if Neff < neff_frac * n (filter wholly on this shard).  *resampled (may be NULL)
 * tells whether it did.  The synchronous form of what slam_pf_step_auto decides on the device. */
extern "C" int slam_pf_resample(slam_pf_t h, double neff_frac, int* resampled) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(h->n == h->n_global, "slam_pf_resample needs the whole filter on this shard");
    double s[7];
    int rc = pf_stats(h, 1, s);
    if (rc) return rc;
    if ((rc = slam_pf_normalize(h, s[0], s[1]))) return rc;
    const double neff = s[1] * s[1] / s[2];
    const bool doit = neff < neff_frac * (double)h->n_global;
    if (resampled) *resampled = doit ? 1 : 0;
    if (!doit) return SLAM_OK;
    const double lg = log(s[1]);
    const double gmax = h->dtype == SLAM_F32 ? (double)((float)s[0] - (float)(s[0] + lg)) : s[0] - (s[0] + lg);
    const double u0 = resample_offset((uint32_t)h->nresamples, h->seed);
    if ((rc = slam_pf_resample_local(h, gmax, u0))) return rc;
    h->nresamples += 1;
    return SLAM_OK;
}

/* the weighted mean pose [x, y, phi] (phi = atan2 of the weighted sin / cos sums); filter wholly on this
 * shard (a sharded filter adds slam_pf_mean_pose_sums over its ranks). */
extern "C" int slam_pf_get_mean_pose(slam_pf_t h, double pose[3]) {
    ARG_CHECK(h != nullptr && pose != nullptr, "null argument");
    ARG_CHECK(h->n == h->n_global, "slam_pf_get_mean_pose needs the whole filter on this shard");
    double s[7];
    const int rc = pf_stats(h, 1, s);
    if (rc) return rc;
    pose[0] = s[3] / s[1];
    pose[1] = s[4] / s[1];
    pose[2] = atan2(s[5], s[6]);
    return SLAM_OK;
}

/* the weights w = exp(logw) of the local particles (double, n_local values; normalised if the filter is). */
extern "C" int slam_pf_get_weights(slam_pf_t h, double* w) {
    ARG_CHECK(h != nullptr && w != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcf = pf_flush_pending(h); if (rcf) return rcf; }
    double* d_w = nullptr;
    HIP_TRY(hipMalloc((void**)&d_w, sizeof(double) * (size_t)h->n));
    PF_DISPATCH(h,
                hipLaunchKernelGGL(pf_weights_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (const T*)h->logw, h->n, (T)0, d_w),
                hipLaunchKernelGGL(pf_weights_kernel<T>, dim3(grid_for(h->n)), dim3(256), 0, h->stream, (const T*)h->logw, h->n, (T)0, d_w));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(w, d_w, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_w);
    if (e != hipSuccess) {
        slam_set_error("HIP error in slam_pf_get_weights: %s", hipGetErrorString(e));
        return SLAM_E_HIP;
    }
    return SLAM_OK;
}

/* Diagnostics: 100 MHz wall-clock stamps of the LAST auto step: [0] kernel start, [1] every workgroup's statistics are in,
 * [2] statistics folded, [3] decision taken, [4] bookkeeping done, [5] published, [6] the collecting workgroup has done
 * its own share, [7] = [0] + 100 x its number of polls.  Waits for the queue. */
extern "C" int slam_pf_debug_stamps(slam_pf_t h, uint64_t out[8]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    const int rc = pf_auto_flush(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    PfCtl c;
    HIP_TRY(hipMemcpy(&c, h->d_ctl, sizeof(c), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out[i] = c.stamps[i];
#if defined(SLAMHIP_EXPERIMENTS) && defined(PF_EXP_STAMPS)
    unsigned long long xs[8];
    HIP_TRY(hipMemcpyFromSymbol(xs, HIP_SYMBOL(g_xs), sizeof(xs)));
    const unsigned long long t0 = c.stamps[0];
    fprintf(stderr, "[pf stamps, us from workgroup 0's start] mid workgroup:");
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %.2f", ((double)xs[i] - (double)t0) * 0.01);
    fprintf(stderr, "\n");
    {
        static unsigned long long wg[3][4096];
        HIP_TRY(hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_wg), sizeof(wg)));
        const int nb = grid_for(h->n) < 4096 ? grid_for(h->n) : 4096;
        for (int k = 0; k < 3; ++k) {
            fprintf(stderr, "[pf wg %s, us] by block index, every 64th:", k == 0 ? "start" : k == 1 ? "updates done" : "stats stored");
            for (int b = 0; b < nb; b += 64) fprintf(stderr, " %.1f", ((double)wg[k][b] - (double)t0) * 0.01);
            fprintf(stderr, " | last: %.1f", ((double)wg[k][nb - 1] - (double)t0) * 0.01);
            double mx = -1e30, mn = 1e30; int imx = 0;
            for (int b = 0; b < nb; ++b) { const double v = ((double)wg[k][b] - (double)t0) * 0.01; if (v > mx) { mx = v; imx = b; } if (v < mn) mn = v; }
            fprintf(stderr, " | min %.1f max %.1f (block %d)\n", mn, mx, imx);
        }
    }
#endif
    return SLAM_OK;
}
