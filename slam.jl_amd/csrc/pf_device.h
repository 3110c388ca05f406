// pf_device.h -- device code shared by the particle path's kernels (included inside each translation unit's anonymous
// namespace): Philox, the fp32 arithmetic of the sweep, the 2 x 2 landmark EKF, the record ring of the sweep, the step cores,
// the weight statistics (legacy block partials and the canonical tree of the auto mode).
#pragma once
#include "pf_internal.h"

// No FMA contraction in the particle path: the arithmetic is specified operation by operation (the oracle is NumPy, which
// rounds every operation), and the fused step kernels must reproduce the separate kernels bit for bit whatever the compiler
// would otherwise fuse across the predict/update boundary.  The kernels are HBM-bound.
#pragma clang fp contract(off)

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

template <typename T>
__device__ __forceinline__ T ld_sc1(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Inclusive scan of one double per thread over a 1024-thread workgroup (the cdf blocks of the systematic resampling): inside a
// wave by shuffles (Hillis-Steele over 64 lanes), the waves' totals added in wave order -- ONE barrier instead of the twenty of
// a Hillis-Steele scan over 1024 LDS words (round 4: the conditional cdf kernel 5.0 -> ~2 us).  Both cdf kernels (the legacy one
// and the auto mode's) use it, so their ancestors stay identical; the order of additions is fixed, hence the same cdf for any
// number of ranks.  sh16: 16 doubles of LDS.
__device__ __forceinline__ double block_scan1024(double v, double* sh16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double u = __shfl_up(v, off);
        if (lane >= off) v += u;
    }
    if (lane == 63) sh16[wave] = v;
    __syncthreads();
    double offs = 0.0;
    for (int w = 0; w < wave; ++w) offs += sh16[w];
    return offs + v;
}

// the host's philox_uniform(step, stream, seed) (pf.py): counter (0, 0, step, stream)
__host__ __device__ inline double resample_offset(uint32_t count, uint64_t seed) {
    uint32_t r[4];
    philox(0u, 0u, count, 2u /* STREAM_RESAMPLE */, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return ((double)(r[0] >> 8) + 0.5) * (1.0 / 16777216.0);
}

// has a peer announced that it is going away?  (uniform: the words sit in this rank's own inbox)
__device__ __forceinline__ bool pf_peer_gone(const PfInbox* inbox, int world) {
    unsigned long long g = 0;
    for (int r = 0; r < world; ++r) g |= __hip_atomic_load(&inbox->gone[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return g != 0;
}


// ---- the auto mode's step plan and publication (shared by pf_auto.hip and pf_batch.hip) -------------------------------------
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


}  // namespace
