// pf_internal.h -- the FastSLAM particle path: state, control blocks and the host-side pieces shared by its translation
// units (pf_legacy.hip: the rank-local kernels and entry points; pf_auto.hip: the step without the host; pf_peers.hip: the
// sharding behind the C ABI).  Device code common to the kernels: pf_device.h.
#pragma once
#include <stdlib.h>
#include <unistd.h>

#include <vector>

#include "common.h"

#define PF_PI 3.14159265358979323846

// ---- auto mode: device-resident control block, its pinned mirror, and the host's log of queued steps ----------
constexpr int PF_CTL_TABS = 64;             // = PF_TAB_MAX (asserted below)
constexpr int PF_CTL_MAXOBS = 64;           // = PF_AUTO_MAXOBS

// ---- sharded filter: every rank's buffers as THIS rank's GPU addresses them (slam_pf_attach_peers) ---------------------
// One process per GPU; at attach time the ranks exchange IPC handles of their state buffers and of an "inbox" page, so every
// rank's kernels can read every peer's log-weights, poses, ancestor tables and landmark records over xGMI and WRITE into
// every peer's inbox (per-step scalars, hand-shake words): posted writes to the peer, polls of local memory.
constexpr int PF_MAX_WORLD = 8;

// ---- the landmark records: [landmark][5][n] in CHUNKS of whole landmarks ----------------------------------------------------
// One allocation per chunk of 2^shift landmarks, every chunk below 2 GiB: a sharded filter with a larger exported buffer hung in
// its attach flow (cause unknown, fenced: pf_peers.hip), and BASELINE.json's weak-scaling shape (262144 particles x
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
    int32_t arrive;              // arrival counter of the cdf kernel's workgroups (its last one forms the block offsets; re-armed to 0)
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
    double* d_boff;      // [scan blocks + 1]: their offsets (auto mode: formed by the cdf kernel's last workgroup)
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
    double* d_pb_lines;          // pf_batch.hip: the workgroups' statistics lines of the persistent launch, two parities (allocated at the first one)
};

// ---- observation codes, per-landmark state words, sizes ------------------------------------------------------------------
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
constexpr int PF_LOG = 64;                  // steps the host may run ahead of the device (a persistent launch carries up to 16)
constexpr int PF_AUTO_PASS_MAX = 64;          // passes of 1024 statistics lines / records the tail of an auto step folds (16.7 M particles a rank)
constexpr int PF_PUBLISH_EVERY = 8;         // a step publishes to the host's mirror when its number is a multiple of this
                                            // (or when it halts / fails); slam_pf_flush asks for the last one
constexpr int32_t LS_TAB = 0xff, LS_BUF = 1 << 8, LS_SEEN = 1 << 9;     // per-landmark state word of the auto mode
static_assert(PF_CTL_TABS == PF_TAB_MAX && PF_CTL_MAXOBS == PF_AUTO_MAXOBS, "control-block sizes");

constexpr int SCAN_BLOCK = 1024;             // particles per block of the cdf (and per record of the statistics exchange)
constexpr int PF_BOFF_MIN_NB = 192;           // scan blocks from which the cdf kernel's last workgroup forms the block offsets (pf_auto.hip)
constexpr int AUTO_NB_MAX = 2048;            // scan blocks (of 1024 particles) the fused offsets support
constexpr int PAR_WAVES = 8;                 // the observation-parallel step kernel: ways per 64-particle workgroup
constexpr int PF_WAY4_MAX_N = 81920;         // one-box sweep (tools/gpu_r4c.sh, no resampling): 65536: seq 29.7, 2 ways 22.4, 4 ways 20.4 us; 98304: 32.4 / 27.1 / 31.3;
constexpr int PF_WAY2_MAX_N = 196608;        // 131072: 34.0 / 29.0 / 35.9; 196608: 38.4 / 36.1 / 47.9 (262144: the sequential sweep, 43.9)
constexpr int PF_PAR_MAX_N = 49152;          // one-box sweep (tools/gpu_r3m.sh): 16384: 26.4 -> 15.2 us, 32768: 27.2 -> 17.1, 65536: 28.4 -> 29.2, 98304: 31.9 -> 43.5

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

template <typename P>
inline int pf_alloc(P** p, size_t bytes, hipStream_t s) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc((void**)p, bytes));
    HIP_TRY(hipMemsetAsync(*p, 0, bytes, s));
    return SLAM_OK;
}
inline int grid_for(int64_t n) { return (int)((n + 255) / 256); }
inline bool pf_sharded(const slam_pf* h) { return h->d_peers != nullptr && h->xchg_world > 1; }

// ---- across the translation units ------------------------------------------------------------------------------------------
// pf_auto.hip
int pf_auto_flush(slam_pf* h);             // wait for the steps slam_pf_step_auto has queued (resolving a halted one)
int pf_auto_leave(slam_pf* h);             // auto mode -> legacy mode: wait for the queue, bring the bookkeeping back to the host
int pf_auto_passes(const slam_pf* h);      // passes of the statistics hand-over an auto step of this filter makes (<= PF_AUTO_PASS_MAX)
int pf_auto_enter(slam_pf* h);             // legacy mode -> auto mode: the host's bookkeeping becomes the device's
int pf_auto_handle_halt(slam_pf* h);       // a step has halted: resample the legacy way, enqueue the skipped steps again
void pf_auto_trim(slam_pf* h);             // drop the logged steps the device has confirmed
int pf_auto_wait(slam_pf* h, long long target);   // poll the mirror until step `target` is confirmed or a step has halted
const char* pf_error_text(long long code);
// pf_legacy.hip
double pf_take_pending(slam_pf* h);        // the normalisation shift slam_pf_normalize deferred (and forget it)
int pf_materialise(slam_pf* h);            // every landmark to (buffer h->cur, identity table); collective with peers attached
// pf_peers.hip
void pf_detach_peers_impl(slam_pf* h);
void pf_announce_gone(slam_pf* h);         // "this rank is going away" into every peer's inbox (destroy of an attached handle)
int pf_peer_barrier(slam_pf* h);           // a barrier among the ranks on their streams, through the inboxes (enqueued)
int pf_launch_peer_gate(slam_pf* h, long long seq);     // "my step kernel of step seq has completed" to every peer, wait for all
int pf_launch_peer_barrier(slam_pf* h, int32_t* d_err, unsigned long long count, unsigned long long timeout_ticks);

#define PF_LEGACY_ENTRY(h)                    \
    do {                                      \
        const int rc_leave_ = pf_auto_leave(h); \
        if (rc_leave_) return rc_leave_;      \
    } while (0)
