// pf_batch.hip -- runs of FastSLAM filter steps that CANNOT resample (force = 0) as ONE launch (slam_pf_step_auto_batch): a
// persistent grid of one 1024-thread workgroup per compute unit that keeps the particles' poses and weights in registers from
// step to step, with the statistics tail of step t under the sweep of step t + 1.  Reference: none (README.md:6 "FastSLAM is
// ongoing"; the types at src/common.jl:14-20,31-34); algorithm: SURVEY.md 8a F1-F4; arithmetic: pf_device.h, shared with every
// other step kernel, so the particles are the same bit for bit.
//
// What a filter step costs beyond its sweep when every step is a launch of its own (pf_auto.hip; round 4's stamps at 262 144
// particles): the sweep ends at 33.8 us, the last statistics line is in at 35.2, collected 36.7, folded 38.4, decision and
// bookkeeping published 40.0, next kernel 43.6 -- ten microseconds of one workgroup's serial work and a kernel boundary per step.
// Here:
//   * NO collecting workgroup.  Every workgroup stores its node of the canonical statistics tree (WRec, pf_device.h) as a tagged
//     line and one or two WAVES of every workgroup read ALL lines (at most 512, four per lane and wave) and climb the same radix-4
//     tree to the same root: the same shift and Neff in every workgroup, bit for bit.
//   * The bookkeeping of the lazy resampling (per-landmark state words, table reference counts) is REPLAYED by every workgroup in
//     its LDS: without a resampling it is a deterministic function of the observation ids.  Workgroup 0 writes it back to the
//     control block when the launch ends.
//   * The log-weight of step t + 1 is lw = (...((lw_t - shift_t) + term_0) + term_1 ...) with every term formed WITHOUT lw (what
//     the observation-parallel kernels already rely on): the sweep of step t + 1 -- motion model, record loads, 2 x 2 updates,
//     record stores, the terms into LDS -- needs nothing of step t's statistics; shift_t is wanted only when the terms are added, at
//     the sweep's END.  A workgroup starts the next sweep at once; the previous step's lines are asked for in the MIDDLE of the
//     sweep (the loads come back behind the record loads: no poll, no drain of the asking wave's stores) and folded when that
//     wave's sweep is over, while the workgroup's other waves finish theirs.
// A step that may resample (force != 0) is NOT taken: the host enqueues it the old way (slam_pf_step_auto) and starts a new run
// behind it.  Round 5 built the resampling into the launch as well (cdf, ancestors and table composition as phases behind
// release / acquire hand-overs) and measured it slower than the three conditional kernels it replaces on every size -- DESIGN.md
// section 7 has the numbers: an L2 write-back per workgroup 76 us, per XCD still on the critical path; 8-byte write-through
// stores one fabric write each (60 us for the scan); the exchange hops cost what the kernel boundaries cost.
// The grid must be co-resident (workgroups wait for each other): one workgroup per compute unit -- each takes the unit's whole
// register file -- checked by the host against the occupancy query; launches of this kernel on one device are chained by an
// event so that two filters' grids never interleave.  It is therefore OPT-IN (flags bit 1 of slam_pf_step_auto_batch): the caller
// vouches that nothing else keeps the device's compute units busy meanwhile.  Measured (tests, round 5): a second filter
// enqueueing its 256-thread step kernels on another stream can keep a compute unit from ever falling wholly free, the persistent
// grid then never completes and its workgroups give up after 2 s (PF_ERR_HANDOVER: the filter is dead).
// Takes: fp32, the whole filter on this shard, FastSLAM-1.0 steps, <= 32 observations per step, maps of <= 2048 landmarks,
// <= 1024 x (compute units) particles.
#include <mutex>

#include "pf_device.h"

namespace {

constexpr int PB_THREADS = 1024;
constexpr int PB_MAXOBS = 32;            // observations per step
constexpr int PB_MAXSTEPS = 16;          // steps per launch
constexpr int PB_NL_MAX = 2048;          // landmarks whose state words a workgroup replays in LDS
constexpr int PB_WG_MAX = 256;           // workgroups (one per compute unit)
constexpr int PB_BLOB_WORDS = 840;       // the steps' descriptors, in the kernel arguments (3360 bytes)
constexpr int PB_LINE_CAP = 1024;        // statistics lines per parity
constexpr unsigned long long PB_TIMEOUT = 200000000ull;      // 2 s at 100 MHz

// blob: K headers of three words {V, G (floats), off | m << 16}, then per step m x {range, bearing (floats), id}
struct PbArgs {
    void *pose0, *pose1, *logw0, *logw1;
    const PfLmTab* lmtab;
    const int32_t *tab0, *tab1;
    long long n, first, n_global, seq0;
    unsigned long long seed;
    unsigned int step0;
    int K, nl, nlines, nwg;
    float wheelbase, a0, a1, dt, R00, R10, R01, R11;
    double* lines;           // [2][PB_LINE_CAP][8]: the workgroups' tree nodes {m, s1, s2, tag} of a step, two parities
    PfCtl* ctl;
    int32_t* lmstate;
    PfMirror* mir;
    unsigned int blob[PB_BLOB_WORDS];
};
static_assert(sizeof(PbArgs) <= 4096, "kernel argument segment");

// the compiler must not move memory operations across a hand-over inside the workgroup (the hardware keeps a wave's LDS
// operations in order).  The words are accessed through the LDS address space: a volatile access through a generic pointer is a
// FLAT instruction, which waits for the wave's outstanding record stores.
#define PB_CBAR() asm volatile("" ::: "memory")
typedef __attribute__((address_space(3))) int pb_lds_int;
__device__ __forceinline__ void pb_lds_store(int* p, int v) {
    PB_CBAR();
    *(volatile pb_lds_int*)p = v;
    PB_CBAR();
}
__device__ __forceinline__ int pb_lds_load(const int* p) {
    PB_CBAR();
    const int v = *(const volatile pb_lds_int*)p;
    PB_CBAR();
    return v;
}

#ifdef SLAMHIP_EXPERIMENTS
// per-workgroup, per-step 100 MHz stamps of the LAST launch (thread 0): [0] step start, [1] sweep done, [2] previous step's
// statistics in hand, [3] line stored
__device__ unsigned long long g_pb_tr[PB_WG_MAX][PB_MAXSTEPS][4];
#define PB_TR(k)                                                                                   \
    do {                                                                                           \
        if (tid == 0) g_pb_tr[blockIdx.x][t][k] = wall_clock64();                                  \
    } while (0)
#else
#define PB_TR(k) do { } while (0)
#endif

template <int W>
__global__ __launch_bounds__(PB_THREADS) void pf_batch_kernel(PbArgs a) {
    typedef float T;
    constexpr int PW = 16 / W;                                   // particle waves of a workgroup
    constexpr int PPW = 64 * PW;                                 // its particles
    constexpr int LL = PW == 16 ? 2 : (PW >= 4 ? 1 : 0);         // level of its statistics lines: 1024 / 256 / 64 particles
    constexpr int LPW = PPW / (64 << (2 * LL));                  // lines per workgroup (1 or 2)
    typedef const __attribute__((address_space(4))) PbArgs* KargPtr;
    const KargPtr ka = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    PfCtl* ctl = a.ctl;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int way = wave / PW, pw = wave % PW;                   // (wave-uniform)
    const int pl = pw * 64 + lane;                               // the particle's place in the workgroup
    const int64_t n = a.n;
    const int64_t pi = (int64_t)blockIdx.x * PPW + pl;
    const bool valid = pi < n;
    const int64_t p = valid ? pi : n - 1;                        // idle lanes shadow the last particle, stores are masked
    auto uni = [](int32_t v) { return __builtin_amdgcn_readfirstlane(v); };

    __shared__ T s_obs[2 * PB_MAXOBS];
    __shared__ int32_t s_ids[PB_MAXOBS], s_meta[PB_MAXOBS], s_l[PB_MAXOBS], s_st[PB_MAXOBS], s_first[PB_MAXOBS];
    __shared__ int32_t s_lm[PB_NL_MAX];                          // the per-landmark state words, replayed
    __shared__ int s_tref[PF_TAB_MAX];
    __shared__ int s_i[4];               // [0] identity landmarks, [1] error, [2] halves of the running fold handed in
    __shared__ double s_fold[8];         // the last fold: [0..2] the root, [3] shift, [4] Neff, [5] largest normalised log-weight
    __shared__ double s_half[2][3];      // the node over the lines 256 .. 511 (wave 1's half of a fold)
    __shared__ int s_half_seq;           // steps of this launch whose second half stands in s_half (monotonic)
    __shared__ double s_leaf[2][16][3];  // the waves' leaves of a step (parity)
    __shared__ int s_cnt_leaf[PB_MAXSTEPS];                      // waves that have handed in their leaf
    __shared__ T s_pose[W > 1 ? 3 : 1][W > 1 ? PPW : 1];
    __shared__ T s_term[PB_MAXOBS][PPW];                         // a step's log-weight terms, by observation

    // ---- the state this launch starts from (the control block is written only by workgroup 0, when the launch ends) ----
    if (ctl->halt_seq != 0 || ctl->error != 0) return;        // an earlier step waits for the host (which replays these), or failed
    double shift = ctl->shift_next;
    for (int l = tid; l < a.nl; l += PB_THREADS) s_lm[l] = a.lmstate[l];
    if (tid < PF_TAB_MAX) s_tref[tid] = ctl->tref[tid];
    if (tid < PB_MAXSTEPS) s_cnt_leaf[tid] = 0;
    if (tid == 0) { s_i[0] = ctl->identity; s_i[1] = 0; s_half_seq = 0; }
    T* const pose = (T*)(ctl->pcur ? a.pose1 : a.pose0);       // (the live sides: no resampling inside the launch)
    T* const logw = (T*)(ctl->lwcur ? a.logw1 : a.logw0);
    const int32_t* const tabs = ctl->tside ? a.tab1 : a.tab0;
    T x = 0, y = 0, phi = 0, lw = 0;
    if (way == 0) {
        x = pose[p]; y = pose[n + p]; phi = pose[2 * n + p];
        lw = logw[p];
    }
    __syncthreads();
    const LmView<T> lv{a.lmtab};
    const PfShardCtx sc{};
    const T R00 = a.R00, R10 = a.R10, R01 = a.R01, R11 = a.R11;
    const bool wg0 = blockIdx.x == 0;
    const int nhalves = a.nlines > 256 ? 2 : 1;
    const bool folder = wave < nhalves;                          // the waves that read the lines (way 0's first waves: PW >= 2)

    // ---- the fold of a step's statistics lines, one wave per half of the lines (lines 256 k + 4 lane .. + 3 in lane `lane`) ----
    auto lines_need = [&](int k) __attribute__((always_inline)) -> unsigned {
        unsigned need = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (256 * k + 4 * lane + j < a.nlines) need |= 1u << j;
        return need;
    };
    // issue: this lane's four lines of half k as raw words (nothing is waited for: the values are looked at later)
    auto lines_issue = [&](long long seq, int k, unsigned need, unsigned long long (&rw)[4][4]) __attribute__((always_inline)) {
        const double* src = a.lines + (size_t)(seq & 1) * PB_LINE_CAP * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(need & (1u << j))) continue;
            const unsigned long long* o = reinterpret_cast<const unsigned long long*>(src + (size_t)(256 * k + 4 * lane + j) * 8);
#pragma unroll
            for (int c = 0; c < 4; ++c) rw[j][c] = ld_sc1(o + c);
        }
    };
    // check: the lines whose tag fits their values and the step's key are taken (stale, half written or torn ones do not fit)
    auto lines_check = [&](long long seq, unsigned& need, const unsigned long long (&rw)[4][4], WRec (&q)[4]) __attribute__((always_inline)) {
        const unsigned long long key = part_key(seq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(need & (1u << j))) continue;
            WRec r;
            r.m = __longlong_as_double((long long)rw[j][0]); r.s1 = __longlong_as_double((long long)rw[j][1]);
            r.s2 = __longlong_as_double((long long)rw[j][2]);
            if ((wrec_hash(r) ^ rw[j][3]) == key) { q[j] = r; need &= ~(1u << j); }
        }
    };
    // this wave's half of the lines of step `seq` (asked for earlier into rw / need), polled for what is still missing, the tree above
    // them; wave 1 hands its node to wave 0 through LDS, wave 0 makes the root: shift / Neff into s_fold -- the same bits in every
    // workgroup.  `done_steps`: the value s_half_seq takes.  false on a time-out (uniform).
    auto fold_half = [&](long long seq, unsigned need, unsigned long long (&rw)[4][4], int done_steps) __attribute__((always_inline)) -> bool {
        WRec q[4] = {wrec_empty(), wrec_empty(), wrec_empty(), wrec_empty()};
        lines_check(seq, need, rw, q);
        const unsigned long long t0 = wall_clock64();
        bool ok = true;
        while (need) {                                            // (stragglers)
            __builtin_amdgcn_s_sleep(16);
            lines_issue(seq, wave, need, rw);
            lines_check(seq, need, rw, q);
            if (wall_clock64() - t0 > PB_TIMEOUT) { ok = false; break; }
        }
        if (__any(!ok)) return false;
        WRec r = wrec_combine4(q[0], q[1], q[2], q[3]);
#pragma unroll
        for (int s = 1; s < 64; s *= 4) {
            WRec b, c, d;
            b.m = __shfl_down(r.m, s); b.s1 = __shfl_down(r.s1, s); b.s2 = __shfl_down(r.s2, s);
            c.m = __shfl_down(r.m, 2 * s); c.s1 = __shfl_down(r.s1, 2 * s); c.s2 = __shfl_down(r.s2, 2 * s);
            d.m = __shfl_down(r.m, 3 * s); d.s1 = __shfl_down(r.s1, 3 * s); d.s2 = __shfl_down(r.s2, 3 * s);
            r = wrec_combine4(r, b, c, d);                        // valid in the lanes with lane % (4 s) == 0
        }
        if (wave == 1) {
            if (lane == 0) { s_half[done_steps & 1][0] = r.m; s_half[done_steps & 1][1] = r.s1; s_half[done_steps & 1][2] = r.s2; }
            pb_lds_store(&s_half_seq, done_steps);
            return true;
        }
        WRec h1 = wrec_empty();
        if (nhalves > 1) {
            const unsigned long long t1 = wall_clock64();
            while (uni(pb_lds_load(&s_half_seq)) < done_steps) {
                __builtin_amdgcn_s_sleep(2);
                if (pb_lds_load(&s_i[1]) != 0 || wall_clock64() - t1 > PB_TIMEOUT) return false;
            }
            h1 = WRec{s_half[done_steps & 1][0], s_half[done_steps & 1][1], s_half[done_steps & 1][2]};
        }
        if (lane == 0) {
            const WRec root = wrec_combine4(r, h1, wrec_empty(), wrec_empty());
            // root: m = the largest log-weight, s1 = sum exp(logw - K ln 2), s2 = sum of its squares, K = ceil(m / ln 2) (pf_auto_tail)
            const double kshift = wrec_k(root.m) * PF_LN2;
            const double lg = log(root.s1);
            s_fold[0] = root.m; s_fold[1] = root.s1; s_fold[2] = root.s2;
            s_fold[3] = kshift + lg;                                            // the normalisation shift = log sum exp(logw)
            s_fold[4] = root.s1 * root.s1 / root.s2;                            // Neff
            s_fold[5] = (double)((T)root.m - (T)(kshift + lg));                 // the largest log-weight after the shift, as stored
        }
        return true;
    };
    auto step_hdr = [&](int t, T& V, T& G, int& off, int& m) __attribute__((always_inline)) {
        V = __uint_as_float(ka->blob[3 * t]);
        G = __uint_as_float(ka->blob[3 * t + 1]);
        const unsigned w2 = ka->blob[3 * t + 2];
        off = (int)(w2 & 0xffffu); m = (int)((w2 >> 16) & 0xffu);
    };

    int t = 0;
    for (; t < a.K; ++t) {
        T V, G; int off, m;
        step_hdr(t, V, G, off, m);
        const long long seq = a.seq0 + t;
        PB_TR(0);
        // ---- the step's observations and their plan (pf_auto.hip: plan_obs), from the replayed state words ----
        if (tid < 2 * m) s_obs[tid] = __uint_as_float(ka->blob[off + 3 * (tid >> 1) + (tid & 1)]);
        int l_pre = 0;
        int32_t st_pre = 0;
        if (tid < m) {
            l_pre = (int)ka->blob[off + 3 * tid + 2] - 1;
            st_pre = s_lm[l_pre];
        }
        // F1, the motion model (sim/sim-utils.jl:36-37, src/ekf.jl:39-41): once per particle, by way 0
        auto motion = [&]() __attribute__((always_inline)) {
            T e1, e2;
            normals2<T>((uint64_t)(a.first + p), a.step0 + (unsigned)t, STREAM_PREDICT, a.seed, e1, e2);
            const T Vn = V + a.a0 * e1;
            const T Gn = G + a.a1 * e2;
            T sgp, cgp, sg, cg;
            m_sincos<T>(Gn + phi, sgp, cgp);
            m_sincos<T>(Gn, sg, cg);
            const T xn = x + Vn * a.dt * cgp;
            const T yn = y + Vn * a.dt * sgp;
            const T pn = wrap_pi<T>(phi + Vn * a.dt * sg / a.wheelbase);
            x = xn; y = yn; phi = pn;
            if (valid) { pose[p] = x; pose[n + p] = y; pose[2 * n + p] = phi; }
        };
        if constexpr (W > 1) {
            if (way == 0) {
                motion();
                s_pose[0][pl] = x; s_pose[1][pl] = y; s_pose[2][pl] = phi;
            }
        }
        plan_obs(l_pre, st_pre, m, s_l, s_st, s_ids, s_meta, s_first);        // (two barriers: the pose is in LDS behind them)
        // this step's state transitions (pf_auto_tail does them in the control block; here in every workgroup's copy)
        if (tid < m && s_first[tid]) {
            const int32_t st = s_st[tid];
            const int tab = st & LS_TAB, rb = (st & LS_BUF) ? 1 : 0;
            if (tab) {
                atomicSub(&s_tref[tab - 1], 1);
                atomicAdd(&s_i[0], 1);                            // released its table: a landmark without one ("identity")
            }
            s_lm[s_l[tid]] = LS_SEEN | ((tab ? (rb ^ 1) : rb) ? LS_BUF : 0);
        }
        if constexpr (W > 1) {
            if (way != 0) { x = s_pose[0][pl]; y = s_pose[1][pl]; phi = s_pose[2][pl]; }
        }
        // ---- F2 / F3: this way's observations i = way + W j with the sweep's record ring (PF_DEPTH requests in flight) ----
        const int cnt = m > way ? (m - way + W - 1) / W : 0;
        // the previous step's lines are asked for in the middle of this sweep, by the folding waves
        const bool want_prev = t > 0 && folder;
        const int jask = cnt >= 2 * PF_DEPTH ? (cnt / 2) / PF_DEPTH * PF_DEPTH : 0;
        unsigned need = 0;
        unsigned long long rw[4][4];
        {
            auto ahead = [&](int j) __attribute__((always_inline)) -> bool {                     // uniform: may observation j's record be requested before its turn?
                const int i = way + W * j;
                if constexpr (W == 1) return KnownRing<T, false>::ahead(s_ids, i);
                else return !(uni(s_ids[i]) & NEW_FLAG);          // (W > 1: no landmark twice in a step, host-checked)
            };
            LmRow<T> ring[PF_DEPTH];
            bool have[PF_DEPTH];
#pragma unroll
            for (int u = 0; u < PF_DEPTH; ++u) {
                have[u] = false;
                ring[u] = LmRow<T>{0, 0, 0, 0, 0};
                if (u < cnt && ahead(u)) {
                    const int i = way + W * u;
                    ring[u] = sweep_load<T, 2, false>(lv, tabs, n, (uint32_t)p, uni(s_ids[i]), uni(s_meta[i]), sc);
                    have[u] = true;
                }
            }
            if constexpr (W == 1) motion();                       // the first records are in flight during the motion model
            if (want_prev && cnt == 0) { need = lines_need(wave); lines_issue(seq - 1, wave, need, rw); }
            for (int j0 = 0; j0 < cnt; j0 += PF_DEPTH) {
                if (want_prev && j0 == jask) { need = lines_need(wave); lines_issue(seq - 1, wave, need, rw); }
#pragma unroll
                for (int u = 0; u < PF_DEPTH; ++u) {
                    const int j = j0 + u;
                    if (j >= cnt) break;                          // uniform
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
                    if (jn < cnt && ahead(jn)) {
                        const int in = way + W * jn;
                        ring[u] = sweep_load<T, 2, false>(lv, tabs, n, (uint32_t)p, uni(s_ids[in]), uni(s_meta[in]), sc);
                        have[u] = true;
                    }
                    T term = 0;
                    if (code & NEW_FLAG) {                        // F3: first sighting
                        lm_init<T>(row, n, x, y, phi, r, b, R00, R10, R01, R11, valid);
                    } else {
                        if (!have_cur) cur = sweep_load<T, 2, false>(lv, tabs, n, (uint32_t)p, code, meta, sc);
                        lm_update<T>(row, n, cur, x, y, phi, r, b, R00, R10, R01, R11, valid, term);      // term = 0 + (this observation's log-weight term)
                    }
                    s_term[i][pl] = term;
                }
            }
        }
        PB_TR(1);
        // ---- the previous step's statistics: folded by this workgroup's first wave(s) while the others finish their sweeps ----
        if (want_prev) {
            if (!fold_half(seq - 1, need, rw, t)) {
                pb_lds_store(&s_i[1], PF_ERR_HANDOVER);
            } else if (wg0 && wave == 0 && lane == 0 && ((seq - 1) % PF_PUBLISH_EVERY) == 0) {
                pf_publish(a.mir, s_fold[4], (long long)ctl->nresamples, ctl->resample_seq, 0, 0ll, seq - 1);
            }
        }
        __syncthreads();                                          // (the fold stands; every way's terms are in LDS)
        if (s_i[1]) break;
        if (t > 0) shift = s_fold[3];
        PB_TR(2);
        if (way == 0) {
            lw -= (T)shift;                                       // the normalisation deferred by the previous step
            for (int i = 0; i < m; ++i)
                if (!(uni(s_ids[i]) & NEW_FLAG)) lw += s_term[i][pl];          // observation order
            if (valid) logw[p] = lw;
            // this wave's leaf of the statistics tree; the last wave to hand its leaf in stores the workgroup's line(s), tagged, not
            // waited for
            double* lines = a.lines + (size_t)(seq & 1) * PB_LINE_CAP * 8;
            const WRec leaf = wrec_wave<T>(lw, valid);
            if constexpr (LL == 0) {
                if (lane == 0) wrec_store_line(lines, (int)blockIdx.x * LPW + pw, leaf, seq);
            } else {
                int last = 0;
                if (lane == 0) {
                    s_leaf[t & 1][pw][0] = leaf.m; s_leaf[t & 1][pw][1] = leaf.s1; s_leaf[t & 1][pw][2] = leaf.s2;
                    PB_CBAR();
                    last = atomicAdd(&s_cnt_leaf[t], 1) == PW - 1 ? 1 : 0;
                    PB_CBAR();
                }
                last = uni(last);
                if (last && lane < LPW) {
                    auto get = [&](int k) { return WRec{s_leaf[t & 1][k][0], s_leaf[t & 1][k][1], s_leaf[t & 1][k][2]}; };
                    WRec node;
                    if constexpr (LL == 1) {
                        node = wrec_combine4(get(4 * lane), get(4 * lane + 1), get(4 * lane + 2), get(4 * lane + 3));
                    } else {
                        const WRec q0 = wrec_combine4(get(0), get(1), get(2), get(3)), q1 = wrec_combine4(get(4), get(5), get(6), get(7)),
                                   q2 = wrec_combine4(get(8), get(9), get(10), get(11)), q3 = wrec_combine4(get(12), get(13), get(14), get(15));
                        node = wrec_combine4(q0, q1, q2, q3);
                    }
                    wrec_store_line(lines, (int)blockIdx.x * LPW + lane, node, seq);
                }
            }
        }
        PB_TR(3);
    }

    // ---- the launch ends: workgroup 0 folds the last step and brings the control block up to date ----
    if (!wg0) return;
    const long long last = a.seq0 + a.K - 1;
    if (!s_i[1]) {
        if (folder) {
            unsigned need = lines_need(wave);
            unsigned long long rw[4][4];
            lines_issue(last, wave, need, rw);
            if (!fold_half(last, need, rw, a.K)) pb_lds_store(&s_i[1], PF_ERR_HANDOVER);
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_i[1]) {
            ctl->error = s_i[1];
            pf_publish(a.mir, 0.0, (long long)ctl->nresamples, ctl->resample_seq, s_i[1], last, last);
        } else {
            ctl->stats[0] = s_fold[0]; ctl->stats[1] = s_fold[1]; ctl->stats[2] = s_fold[2];
            ctl->stats[3] = ctl->stats[4] = ctl->stats[5] = ctl->stats[6] = 0.0;      // (not formed by a step: slam_pf_mean_pose_sums)
            ctl->stats[7] = s_fold[4];
            ctl->shift_scan = s_fold[3];
            ctl->gmax_norm = s_fold[5];
            ctl->shift_next = s_fold[3];
            ctl->seq = last;
            ctl->identity = s_i[0];
            if ((last % PF_PUBLISH_EVERY) == 0) pf_publish(a.mir, s_fold[4], (long long)ctl->nresamples, ctl->resample_seq, 0, 0ll, last);
        }
        ctl->stamps[6] = wall_clock64();
    }
    if (s_i[1]) return;
    for (int l = tid; l < a.nl; l += PB_THREADS) a.lmstate[l] = s_lm[l];
    if (tid < PF_TAB_MAX) ctl->tref[tid] = s_tref[tid];
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct PbDevice {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    const void* last_owner = nullptr;
    int cus = 0, ok[4] = {-1, -1, -1, -1};      // occupancy answers for W = 1, 2, 4, 8
};
static PbDevice g_pb_dev[16];
constexpr int PB_MIN_RUN = 4;                   // shorter runs of force = 0 steps go step by step (a launch's fixed cost)

template <int W>
static int pb_occupancy(int* out) {
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pf_batch_kernel<W>, PB_THREADS, 0));
    *out = nb;
    return SLAM_OK;
}

// may this filter's steps go through the persistent launch at all?
static bool pb_filter_ok(const slam_pf* h) {
    return h->dtype == SLAM_F32 && h->xchg_world <= 1 && h->n == h->n_global && h->nl <= PB_NL_MAX && h->device >= 0 && h->device < 16;
}

static int pb_ensure_buffers(slam_pf* h) {
    if (h->d_pb_lines) return SLAM_OK;
    return pf_alloc(&h->d_pb_lines, sizeof(double) * 2 * PB_LINE_CAP * 8, h->stream);
}

// One persistent launch for kc steps that cannot resample; the steps are already in the host's log.
static int pb_launch(slam_pf* h, int widx, int kc, const PfStepRec* recs) {
    PbArgs a;
    memset(&a, 0, sizeof(a));
    const int W = 1 << widx;
    const int PPW = 1024 / W;
    a.pose0 = h->pose[0]; a.pose1 = h->pose[1]; a.logw0 = h->logw2[0]; a.logw1 = h->logw2[1];
    a.lmtab = h->d_lmtab; a.tab0 = h->d_tab[0]; a.tab1 = h->d_tab[1];
    a.n = h->n; a.first = h->first; a.n_global = h->n_global; a.seq0 = recs[0].seq;
    a.seed = h->seed; a.step0 = recs[0].rng_step;
    a.K = kc; a.nl = h->nl;
    const int PW = 16 / W, LL = PW == 16 ? 2 : (PW >= 4 ? 1 : 0);
    const int64_t per_line = (int64_t)64 << (2 * LL);
    a.nlines = (int)((h->n + per_line - 1) / per_line);
    a.nwg = (int)((h->n + PPW - 1) / PPW);
    a.wheelbase = (float)recs[0].wheelbase; a.dt = (float)recs[0].dt;
    a.a0 = (float)sqrt(recs[0].Q[0]); a.a1 = (float)sqrt(recs[0].Q[3]);
    a.R00 = (float)recs[0].R[0]; a.R10 = (float)recs[0].R[1]; a.R01 = (float)recs[0].R[2]; a.R11 = (float)recs[0].R[3];
    a.lines = h->d_pb_lines;
    a.ctl = h->d_ctl; a.lmstate = h->d_lmstate; a.mir = h->h_mir_dev;
    int off = 3 * kc;
    for (int t = 0; t < kc; ++t) {
        const PfStepRec& r = recs[t];
        const float V = (float)r.V, G = (float)r.G;
        memcpy(&a.blob[3 * t], &V, 4);
        memcpy(&a.blob[3 * t + 1], &G, 4);
        a.blob[3 * t + 2] = (unsigned)off | ((unsigned)r.m << 16);
        for (int i = 0; i < r.m; ++i) {
            const float rr = (float)r.z[2 * i], bb = (float)r.z[2 * i + 1];
            memcpy(&a.blob[off + 3 * i], &rr, 4);
            memcpy(&a.blob[off + 3 * i + 1], &bb, 4);
            a.blob[off + 3 * i + 2] = (unsigned)r.ids[i];
        }
        off += 3 * r.m;
        if ((r.seq % PF_PUBLISH_EVERY) == 0 && r.seq > h->pub_seq) h->pub_seq = r.seq;
    }
    // launches of the persistent kernel on one device never overlap: two co-resident grids could each hold compute units the
    // other one's unstarted workgroups wait for
    PbDevice& d = g_pb_dev[h->device];
    std::lock_guard<std::mutex> lock(d.mu);
    if (!d.ev) HIP_TRY(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
    if (d.last_owner && d.last_owner != (const void*)h) HIP_TRY(hipStreamWaitEvent(h->stream, d.ev, 0));
    const dim3 grid((unsigned)a.nwg), block(PB_THREADS);
    switch (W) {
        case 1: hipLaunchKernelGGL(pf_batch_kernel<1>, grid, block, 0, h->stream, a); break;
        case 2: hipLaunchKernelGGL(pf_batch_kernel<2>, grid, block, 0, h->stream, a); break;
        case 4: hipLaunchKernelGGL(pf_batch_kernel<4>, grid, block, 0, h->stream, a); break;
        default: hipLaunchKernelGGL(pf_batch_kernel<8>, grid, block, 0, h->stream, a); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(d.ev, h->stream));
    d.last_owner = (const void*)h;
    return SLAM_OK;
}

// the number of observation ways for this filter (index into {1, 2, 4, 8}), or -1: no persistent launch
static int pb_pick_ways(slam_pf* h, bool distinct) {
    PbDevice& d = g_pb_dev[h->device];
    {
        std::lock_guard<std::mutex> lock(d.mu);
        if (d.cus == 0) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) return -1;
            d.cus = prop.multiProcessorCount;
            int rc = pb_occupancy<1>(&d.ok[0]);
            rc |= pb_occupancy<2>(&d.ok[1]);
            rc |= pb_occupancy<4>(&d.ok[2]);
            rc |= pb_occupancy<8>(&d.ok[3]);
            if (rc) { d.cus = -1; (void)hipGetLastError(); }
        }
    }
    if (d.cus <= 0) return -1;
    const int cus = d.cus < PB_WG_MAX ? d.cus : PB_WG_MAX;
    const int forced = slam_exp_env("SLAMHIP_PB_W", 0);          // (experiments build)
    int best = -1;
    for (int widx = 0; widx < 4; ++widx) {
        const int W = 1 << widx;
        if (W > 1 && !distinct) break;
        if (d.ok[widx] < 1) continue;
        const int64_t wgs = (h->n * W + 1023) / 1024;
        if (wgs > cus) continue;
        if (forced && W != forced) continue;
        best = widx;                                              // the widest that still fits: fills the chip
    }
    return best;
}

/* K consecutive slam_pf_step_auto calls as ONE call: step k = control (VG[2k], VG[2k+1]), observations z + 2 zstride k (m[k]
 * (range, bearing) pairs), their landmark ids at ids + zstride k, force[k] (NULL: the Neff rule at every step).  Same filter as
 * the K calls, bit for bit.  Runs of at least four consecutive steps that cannot resample (force[k] == 0) go, where the filter
 * allows it (see the head of pf_batch.hip) and the caller ALLOWS it (flags bit 1: nothing else competes for the device's compute
 * units meanwhile), as persistent launches of up to 16 steps; every other step is enqueued as slam_pf_step_auto enqueues it.  *enqueued (may be NULL): the steps taken; less than K
 * only with SLAM_PF_HALTED (sharded halting flow: resolve, then call again with the rest). */
extern "C" int slam_pf_step_auto_batch(slam_pf_t h, int K, const double* VG, double wheelbase, const double Q[4], double dt,
                                       const double* z, const int32_t* ids, const int32_t* m, int zstride, const double R[4],
                                       double neff_frac, const int32_t* force, int proposal, int flags, int* enqueued) {
    SLAM_RANGE();
    if (enqueued) *enqueued = 0;
    ARG_CHECK(h != nullptr && Q != nullptr && VG != nullptr && m != nullptr, "null argument");
    ARG_CHECK(K >= 0 && zstride >= 0, "negative count");
    int mmax = 0;
    for (int k = 0; k < K; ++k) {
        ARG_CHECK(m[k] >= 0 && m[k] <= PF_AUTO_MAXOBS && m[k] <= zstride, "m[k] out of range (at most 64 observations per step, at most zstride)");
        if (m[k] > mmax) mmax = m[k];
    }
    ARG_CHECK(mmax == 0 || (z != nullptr && ids != nullptr && R != nullptr), "null argument");
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < m[k]; ++i) ARG_CHECK(ids[(size_t)zstride * k + i] >= 1 && ids[(size_t)zstride * k + i] <= h->nl, "landmark id out of range");
    ARG_CHECK(!h->halted, "a halted step is waiting for slam_pf_resume");
    const double R0[4] = {0, 0, 0, 0};
    const bool persistent = (flags & 2) && !proposal && force != nullptr && pb_filter_ok(h);
    int k = 0;
    while (k < K) {
        int kc = 0, widx = -1;
        if (persistent && force[k] == 0) {
            // the run: consecutive steps that cannot resample, as many as the argument blob holds; the ways by whether every step's
            // landmarks are distinct
            int words = 0;
            bool distinct = true;
            while (k + kc < K && kc < PB_MAXSTEPS && force[k + kc] == 0 && m[k + kc] <= PB_MAXOBS && words + 3 + 3 * m[k + kc] <= PB_BLOB_WORDS) {
                words += 3 + 3 * m[k + kc];
                const int32_t* id = ids + (size_t)zstride * (k + kc);
                for (int i = 1; i < m[k + kc] && distinct; ++i)
                    for (int j = 0; j < i; ++j)
                        if (id[i] == id[j]) { distinct = false; break; }
                ++kc;
            }
            widx = kc >= PB_MIN_RUN ? pb_pick_ways(h, distinct) : -1;
        }
        if (widx < 0) {                                           // step by step
            const int rc = slam_pf_step_auto(h, VG[2 * k], VG[2 * k + 1], wheelbase, Q, dt, z ? z + (size_t)2 * zstride * k : nullptr,
                                             ids ? ids + (size_t)zstride * k : nullptr, m[k], R ? R : R0, neff_frac, force ? force[k] : -1, proposal);
            if (rc) return rc;
            ++k;
            if (enqueued) *enqueued = k;
            continue;
        }
        HIP_TRY(hipSetDevice(h->device));
        int rc;
        if (!h->auto_on && (rc = pf_auto_enter(h))) return rc;
        if (h->h_mir->halt_seq != 0 && (rc = pf_auto_handle_halt(h))) return rc;
        pf_auto_trim(h);
        while ((int)h->log.size() + kc > PF_LOG - 1) {            // the host is a whole log ahead: wait for the oldest step
            if ((rc = pf_auto_wait(h, h->log.front().seq))) return rc;
            if (h->h_mir->halt_seq != 0 && (rc = pf_auto_handle_halt(h))) return rc;
            pf_auto_trim(h);
        }
        if ((rc = pb_ensure_buffers(h))) return rc;
        const size_t first_rec = h->log.size();
        for (int t = 0; t < kc; ++t) {
            PfStepRec r;
            memset(&r, 0, sizeof(r));
            r.seq = ++h->auto_seq;
            r.rng_step = h->step++;
            r.m = m[k + t]; r.force = 0; r.proposal = 0;
            r.V = VG[2 * (k + t)]; r.G = VG[2 * (k + t) + 1]; r.wheelbase = wheelbase; r.dt = dt; r.neff_frac = neff_frac;
            for (int i = 0; i < 4; ++i) { r.Q[i] = Q[i]; r.R[i] = (r.m && R) ? R[i] : 0.0; }
            for (int i = 0; i < r.m; ++i) {
                r.z[2 * i] = z[(size_t)2 * zstride * (k + t) + 2 * i];
                r.z[2 * i + 1] = z[(size_t)2 * zstride * (k + t) + 2 * i + 1];
                r.ids[i] = ids[(size_t)zstride * (k + t) + i];
            }
            h->log.push_back(r);
        }
        // (R of the launch: the first step's, or -- a run whose first step has no observation -- the call's)
        if (R) for (int i = 0; i < 4; ++i) h->log[first_rec].R[i] = R[i];
        if ((rc = pb_launch(h, widx, kc, &h->log[first_rec]))) return rc;
        k += kc;
        if (enqueued) *enqueued = k;
    }
    return SLAM_OK;
}

#ifdef SLAMHIP_EXPERIMENTS
/* Experiments build: the per-workgroup, per-step stamps of the last persistent launch ([256][16][4], 100 MHz). */
extern "C" int slam_pf_debug_batch_trace(slam_pf_t h, uint64_t* out) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pb_tr), sizeof(unsigned long long) * PB_WG_MAX * PB_MAXSTEPS * 4));
    return SLAM_OK;
}
#endif
