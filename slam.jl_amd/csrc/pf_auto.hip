// pf_auto.hip -- the FastSLAM filter step WITHOUT the host ("auto mode", slam_pf_step_auto): the step kernels (sequential sweep,
// 8-way observation-parallel, 2- / 4-way on 256-particle workgroups), the last workgroup's tail (statistics, decision,
// bookkeeping of the lazy resampling), the conditional cdf / resampling kernels, and the host's queue of steps.
// Reference: none (README.md:6 "FastSLAM is ongoing"; the types at src/common.jl:14-20,31-34); algorithm: SURVEY.md 8a F1-F4.
#include "pf_device.h"

namespace {

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
    __shared__ double s_pass[PF_AUTO_PASS_MAX][3];        // the tree nodes above each pass of 1024 lines / records (the host refuses
                                                  // a filter that needs more passes: slam_pf_step_auto, slam_pf_attach_peers)
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
    // (round 4) this thread's share of the per-landmark state words, for the resampling's "landmarks without a table take the
    // fresh one" pass: requested NOW, with the rest -- loaded where they are used they were a dependent global round trip at
    // the very end of every resampling step.  Maps of up to 1024 landmarks; larger ones take the loop below.
    constexpr int LS_PRE = 4;
    const bool ls_pre = a.nl <= 256 * LS_PRE;
    int32_t my_ls[LS_PRE];
#pragma unroll
    for (int j = 0; j < LS_PRE; ++j) my_ls[j] = (ls_pre && tid + 256 * j < a.nl) ? a.lmstate[tid + 256 * j] : 0;
    __shared__ unsigned s_obsbit[256 * LS_PRE / 32];         // the landmarks this step observes (their words change below)
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
    if (tid < 256 * LS_PRE / 32) s_obsbit[tid] = 0u;
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
            if (tid == 0 && ps < PF_AUTO_PASS_MAX) { s_pass[ps][0] = pr.m; s_pass[ps][1] = pr.s1; s_pass[ps][2] = pr.s2; }
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
        if (ls_pre) atomicOr(&s_obsbit[s_l[tid] >> 5], 1u << (s_l[tid] & 31));
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
            if (tid == 0 && ps < PF_AUTO_PASS_MAX) { s_pass[ps][0] = pr.m; s_pass[ps][1] = pr.s1; s_pass[ps][2] = pr.s2; }
        }
    }
    __syncthreads();
    if (tid == 0) ctl->stamps[7] = ctl->stamps[0] + 100ull * (unsigned long long)rounds;      // (diagnostic: polls of thread 0)
    if (tid == 0) {
        ctl->stamps[2] = wall_clock64();
        // the passes' nodes -> the root, still the radix-4 tree (absent children are the identity), in place in LDS (a register
        // array here would raise the whole step kernel's allocation)
        int cnt = npass < PF_AUTO_PASS_MAX ? npass : PF_AUTO_PASS_MAX;
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
    static_assert(PF_TAB_MAX == 64, "the table list is formed by one wave, a lane per table");
    if (tid < 64 && s_i[1] == 1) {
        // lazy resampling: landmarks without a table share a fresh one (= the ancestor vector); live tables are composed.
        // One lane per table (round 4: thread 0 used to walk the 64 tables, an LDS read and a global store each: 3 us of every
        // resampling step): the list position of a live table = the live tables before it, the fresh one = the first free.
        const int identity = s_i[0];
        const bool live = s_tref[tid] > 0;
        const unsigned long long mask = __ballot(live);
        const int free_idx = ~mask ? __ffsll((unsigned long long)~mask) - 1 : -1;
        const bool halt = identity > 0 && free_idx < 0;                      // no table left: the host resamples eagerly
        if (live && !halt) ctl->tl_idx[__popcll(mask & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) {
            if (halt) s_i[1] = 2;
            else {
                ctl->tl_count = __popcll(mask);
                ctl->tl_fresh = identity > 0 ? free_idx : -1;
                s_i[2] = identity > 0 ? free_idx : -1;
                if (identity > 0) s_tref[free_idx] = identity;
            }
        }
    }
    __syncthreads();
    if (s_i[1] == 1 && s_i[2] >= 0) {
        const int32_t fresh = s_i[2] + 1;
        if (ls_pre) {
            // the words requested at the tail's start; a landmark observed in THIS step has a new word (no table: it takes the
            // fresh one), written by its observation's thread above -- that thread adds the table to it
#pragma unroll
            for (int j = 0; j < LS_PRE; ++j) {
                const int l = tid + 256 * j;
                if (l < a.nl && !((s_obsbit[l >> 5] >> (l & 31)) & 1u) && (my_ls[j] & LS_TAB) == 0) a.lmstate[l] = my_ls[j] | fresh;
            }
            if (tid < a.m && s_first[tid]) {
                const int32_t st = s_st[tid];
                const int tab = st & LS_TAB, rb = (st & LS_BUF) ? 1 : 0;
                a.lmstate[s_l[tid]] = (LS_SEEN | ((tab ? (rb ^ 1) : rb) ? LS_BUF : 0)) | fresh;
            }
        } else {
            for (int l = tid; l < a.nl; l += 256) {
                const int32_t st = a.lmstate[l];
                if ((st & LS_TAB) == 0) a.lmstate[l] = st | fresh;
            }
        }
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


// The cdf of the step that resamples (pf_scan1_kernel behind the control block's gate), over the weights of the WHOLE
// filter: logw0 / logw1 are this rank's two buffers (n_local values each), the other slices come from `peers`.
template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void pf_auto_scan1_kernel(const T* __restrict__ logw0, const T* __restrict__ logw1,
                                                                    int64_t n_local, int64_t n_global, PfCtl* __restrict__ ctl,
                                                                    long long seq, double* __restrict__ cdf,
                                                                    double* __restrict__ bsum, double* __restrict__ boff,
                                                                    const PfPeers* __restrict__ peers, int rank, int world) {
    if (ctl->resample_seq != seq || ctl->error != 0) return;
    __shared__ double sh16[16];
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
    const double c = block_scan1024(i < n_global ? exp((double)(T)(v - pend) - gmax) : 0.0, sh16);
    if (i < n_global) cdf[i] = c;
    // Round 5: the block offsets (pf_scan2_kernel's serial order of additions) are formed ONCE, by the workgroup that finishes last,
    // not by thread 0 of every one of the resampling kernel's workgroups: the block sum goes out write-through, then the arrival
    // (from PF_BOFF_MIN_NB blocks on: below, thread 0 of a resampling workgroup adds the few sums up faster than this hand-over costs --
    //  one box, every step resampling: 32768 particles 35.3 against 37.6 us, 65536 42.3 / 44.3, 131072 56.4 / 56.7, 262144 85.6 / 82.8)
    __shared__ int s_last;
    if ((int)gridDim.x < PF_BOFF_MIN_NB) {
        if (threadIdx.x == SCAN_BLOCK - 1) bsum[blockIdx.x] = c;
        return;
    }
    if (threadIdx.x == SCAN_BLOCK - 1) {
        __hip_atomic_store(bsum + blockIdx.x, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(&ctl->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 64) {
        // 256 dependent additions at most are the cost: lane g holds the sixteen block sums of group g in registers, the groups are
        // added one after the other (the carry goes from lane to lane); offsets[b + 1] == offsets[b] + bsum[b] exactly
        const int nb = (int)gridDim.x, lane = threadIdx.x;
        for (int g0 = 0; g0 < nb; g0 += 1024) {                   // (1024 blocks = 1 M particles per round; the carry goes on)
            double vv[16], oo[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int b = g0 + 16 * lane + u;
                vv[u] = b < nb ? __hip_atomic_load(bsum + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
            double carry = g0 == 0 ? 0.0 : sh16[0], endv = 0.0;
            const int ng = (nb - g0 + 15) / 16 < 64 ? (nb - g0 + 15) / 16 : 64;
            for (int g = 0; g < ng; ++g) {                        // uniform
                if (lane == g) {
                    double run = carry;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        oo[u] = run;
                        run += vv[u];                             // (+ 0.0 beyond nb: exact)
                    }
                    endv = run;
                }
                carry = __shfl(endv, g);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int b = g0 + 16 * lane + u;
                if (lane < ng && b < nb) boff[b] = oo[u];
            }
            if (lane == 0) sh16[0] = carry;                       // (the same wave reads it in the next round)
            if (g0 + 1024 >= nb && lane == 0) boff[nb] = carry;
        }
        if (lane == 0) __hip_atomic_store(&ctl->arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed
    }
}

// Block offsets (pf_scan2_kernel's serial order, redone by every workgroup out of LDS), ancestors (pf_ancestor_kernel)
// and the lazy apply (pf_lazy_apply_kernel) of the step that resamples, in one conditional launch.  n: this rank's
// particles, global ids [first, first + n); table entries and ancestors are GLOBAL ids (= local slots on one rank).
template <typename T, bool SH>
__global__ __launch_bounds__(256) void pf_auto_resample_kernel(T* pose0, T* pose1, int32_t* tab0, int32_t* tab1,
                                                                T* logw0, T* logw1, int64_t n, int64_t first, int64_t n_global,
                                                                const PfCtl* __restrict__ ctl,
                                                                long long seq, const double* __restrict__ cdf,
                                                                const double* __restrict__ bsum, const double* __restrict__ boff, int nb,
                                                                int32_t* __restrict__ anc_out, T lw_uniform,
                                                                const PfPeers* __restrict__ peers, int rank, int world) {
    if (ctl->resample_seq != seq || ctl->error != 0) return;
    __shared__ double s_off[AUTO_NB_MAX + 1];
    __shared__ int s_tl[PF_TAB_MAX];                       // the live tables to compose (once per workgroup, not once per use)
    if (nb >= PF_BOFF_MIN_NB) {
        for (int i = threadIdx.x; i <= nb; i += 256) s_off[i] = boff[i];      // (formed once, by the cdf kernel's last workgroup)
        if (threadIdx.x < PF_TAB_MAX) s_tl[threadIdx.x] = ctl->tl_idx[threadIdx.x];
    } else {
        for (int i = threadIdx.x; i < nb; i += 256) s_off[i] = bsum[i];
        if (threadIdx.x < PF_TAB_MAX) s_tl[threadIdx.x] = ctl->tl_idx[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            // the same additions in the same (index) order as pf_scan2_kernel -- s_off[b + 1] == s_off[b] + bsum[b] exactly, which the
            // search below relies on -- sixteen entries at a time out of registers
            double run = 0.0;
            for (int i0 = 0; i0 < nb; i0 += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = i0 + u < nb ? s_off[i0 + u] : 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (i0 + u < nb) s_off[i0 + u] = run;
                    run += v[u];                               // (+ 0.0 beyond nb: exact)
                }
            }
            s_off[nb] = run;
        }
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
    const double bo = s_off[bl];
    // first j in [lo, hi] with cdf[j] + bo >= target, else hi -- what ten dependent halvings found, in THREE rounds of independent
    // probes (strides 128, 16, 1: the values are monotone, so the first probe that reaches the target brackets the answer)
    {
        const int64_t last = hi;
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            const int stride = round == 0 ? 128 : (round == 1 ? 16 : 1);
            constexpr int NP = 15;
            const int np = round == 2 ? 15 : 7;
            double c[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int64_t j = lo + (int64_t)stride * (u + 1) - 1;
                c[u] = (u < np && j <= last) ? cdf[j] : 0.0;
            }
            int sel = np;
#pragma unroll
            for (int u = NP - 1; u >= 0; --u) {
                const int64_t j = lo + (int64_t)stride * (u + 1) - 1;
                if (u < np && (j >= last || c[u] + bo >= target)) sel = u;
            }
            lo += (int64_t)stride * sel;
            if (lo > last) lo = last;
        }
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
    constexpr int TB = 16;
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

}  // namespace

// ---- auto mode: host side -------------------------------------------------------------------------------------------
static int pf_auto_nb(const slam_pf* h) { return (int)((h->n_global + SCAN_BLOCK - 1) / SCAN_BLOCK); }

// passes of 1024 lines / records the step kernel's tail makes (pf_auto_tail): over this rank's lines (one per 256 particles in the
// sweep kernels -- the observation-parallel kernel's 64-particle lines only run on small filters) and, with peers, over the
// ranks' 1024-particle records
int pf_auto_passes(const slam_pf* h) {
    const long long lines = (h->n + 255) / 256, recs = pf_sharded(h) ? (h->n_global + 1023) / 1024 + PF_MAX_WORLD : 0;
    const long long a = (lines + 1023) / 1024, b = (recs + 1023) / 1024;
    return (int)(a > b ? a : b);
}

// may a step resample on the device?  The whole filter here, or a sharded one whose peers are attached.
static bool pf_auto_lazy_ok(const slam_pf* h) {
    if (h->lazy_off || pf_auto_nb(h) > AUTO_NB_MAX) return false;
    return h->xchg_world <= 1 ? h->n == h->n_global : pf_sharded(h);
}

const char* pf_error_text(long long code) {
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
int pf_auto_enter(slam_pf* h) {
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
        if (sh) { const int rcg = pf_launch_peer_gate(h, r.seq); if (rcg) return rcg; }
        PF_DISPATCH(h,
                    hipLaunchKernelGGL(pf_auto_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)h->logw2[0],
                                       (const T*)h->logw2[1], h->n, h->n_global, h->d_ctl, r.seq, h->d_cdf, h->d_bsum, h->d_boff, pp,
                                       rank, world),
                    hipLaunchKernelGGL(pf_auto_scan1_kernel<T>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, (const T*)h->logw2[0],
                                       (const T*)h->logw2[1], h->n, h->n_global, h->d_ctl, r.seq, h->d_cdf, h->d_bsum, h->d_boff, pp,
                                       rank, world));
#define PF_RESAMPLE_LAUNCH(SHV)                                                                                              \
        PF_DISPATCH(h,                                                                                                       \
                    hipLaunchKernelGGL((pf_auto_resample_kernel<T, SHV>), grid, dim3(256), 0, h->stream, (T*)h->pose[0], (T*)h->pose[1], \
                                       h->d_tab[0], h->d_tab[1], (T*)h->logw2[0], (T*)h->logw2[1], h->n, h->first, h->n_global,     \
                                       (const PfCtl*)h->d_ctl, r.seq, (const double*)h->d_cdf, (const double*)h->d_bsum, (const double*)h->d_boff, nb,      \
                                       h->d_anc, (T)lw, pp, rank, world),                                                     \
                    hipLaunchKernelGGL((pf_auto_resample_kernel<T, SHV>), grid, dim3(256), 0, h->stream, (T*)h->pose[0], (T*)h->pose[1], \
                                       h->d_tab[0], h->d_tab[1], (T*)h->logw2[0], (T*)h->logw2[1], h->n, h->first, h->n_global,     \
                                       (const PfCtl*)h->d_ctl, r.seq, (const double*)h->d_cdf, (const double*)h->d_bsum, (const double*)h->d_boff, nb,      \
                                       h->d_anc, (T)lw, pp, rank, world))
        if (sh) PF_RESAMPLE_LAUNCH(true);
        else PF_RESAMPLE_LAUNCH(false);
#undef PF_RESAMPLE_LAUNCH
        HIP_TRY(hipGetLastError());
    }
    return SLAM_OK;
}

// wait (polling the pinned mirror) until step `target` is confirmed or a step has halted
int pf_auto_wait(slam_pf* h, long long target) {
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

void pf_auto_trim(slam_pf* h) {
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
int pf_auto_handle_halt(slam_pf* h) {
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

int pf_auto_flush(slam_pf* h) {
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

int pf_auto_leave(slam_pf* h) {
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
    if (pf_auto_passes(h) > PF_AUTO_PASS_MAX) {
        slam_set_error("slam_pf_step_auto: the step's statistics hand-over folds at most %d passes of 1024 lines (%lld particles on one rank, "
                       "%lld over the ranks of a filter with peers); use slam_pf_step / slam_pf_normalize / slam_pf_resample",
                       PF_AUTO_PASS_MAX, (long long)PF_AUTO_PASS_MAX * 1024 * 256, (long long)PF_AUTO_PASS_MAX * 1024 * 1024);
        return SLAM_E_CAPACITY;
    }
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
