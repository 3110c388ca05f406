// pf_legacy.hip -- K9..K12: the FastSLAM-1.0 particle path (known correspondences): rank-local kernels and the synchronous C ABI.
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
// This file: the rank-local kernels and the synchronous C ABI (the host decides after every step), creation and destruction,
// the lazy resampling's host bookkeeping.  The step without the host: pf_auto.hip; the sharding: pf_peers.hip.
#include "pf_device.h"

namespace {

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


template <typename T>
__global__ __launch_bounds__(256) void pf_stats_kernel(const T* __restrict__ logw, const T* __restrict__ pose, int64_t n,
                                                        int relative, double* __restrict__ part) {
    const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pi < n;
    const int64_t p = valid ? pi : n - 1;
    block_weight_stats<T>(logw[p], pose[p], pose[n + p], pose[2 * n + p], valid, relative, part);
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



// per-block inclusive scan of w = exp(logw - max) (double) + block totals
template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void pf_scan1_kernel(const T* __restrict__ logw_all, int64_t n, double gmax,
                                                               double* __restrict__ cdf, double* __restrict__ bsum, T pend) {
    __shared__ double sh16[16];
    const int64_t i = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    // `pend`: a normalisation shift not yet applied to the stored values (rounded as pf_shift_kernel would store it)
    const double c = block_scan1024(i < n ? exp((double)(T)(logw_all[i] - pend) - gmax) : 0.0, sh16);
    if (i < n) cdf[i] = c;
    if (threadIdx.x == SCAN_BLOCK - 1) bsum[blockIdx.x] = c;
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



template <typename T>
__global__ __launch_bounds__(256) void pf_weights_kernel(const T* __restrict__ logw, int64_t n, T pend, double* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = exp((double)(T)(logw[p] - pend));
}

}  // namespace

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
        pf_announce_gone(h);
        usleep(5000);
    }
    pf_detach_peers_impl(h);                 // this rank's mappings of the peers' buffers are closed before anything is freed
    for (int b = 0; b < 2; ++b) {
        if (h->pose[b]) (void)hipFree(h->pose[b]);
        for (int k = 0; k < PF_LM_MAXC; ++k)
            if (h->lmtab.c[b][k]) (void)hipFree(h->lmtab.c[b][k]);
    }
    if (h->d_lmtab) (void)hipFree(h->d_lmtab);
    void* devs[] = {h->logw2[0], h->logw2[1], h->d_part, h->d_out, h->d_cdf, h->d_bsum, h->d_boff, h->d_src, h->d_anc, h->d_tab[0], h->d_tab[1],
                    h->d_lmeta, h->d_ctl, h->d_lmstate, h->inbox, h->d_pb_lines};
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
    if ((rc = pf_alloc(&h->d_boff, sizeof(double) * (nb + 2), h->stream))) return rc;
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
    h->d_part = h->d_out = h->h_out = h->d_cdf = h->d_bsum = h->d_boff = nullptr; h->d_src = nullptr; h->d_anc = nullptr;
    h->seen.assign(max_landmarks, 0);
    h->d_ctl = nullptr; h->d_lmstate = nullptr; h->h_mir = h->h_mir_dev = nullptr;
    h->auto_on = 0; h->auto_seq = 0; h->pub_seq = 0; h->nresamples = 0; h->halted = 0; h->halt_gmax = 0.0; h->last_resampled_seq = 0;
    h->d_xchg = nullptr; h->xchg_host = nullptr; h->xchg_rank = 0; h->xchg_world = 1;
    h->d_pb_lines = nullptr;
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

// (Sharded filter with peers: a COLLECTIVE call -- the ancestor tables hold global particle ids and a remote ancestor's
//  record is read from its owner, so every rank must be here, with barriers among the ranks' streams around the passes:
//  pass 0 reads what the peers' earlier kernels wrote, pass 1 overwrites what the peers' pass 0 reads.)
int pf_materialise(slam_pf* h) {
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
double pf_take_pending(slam_pf* h) {
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

