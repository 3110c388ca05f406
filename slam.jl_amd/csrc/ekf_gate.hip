// ekf_gate.hip -- K1: Mahalanobis gating sweep over all map landmarks.
//
// Takes over associate()/compute_association() (src/data-association.jl:1-63)
// and predict_observation() (src/common.jl:139-165).
//
// One thread owns one landmark j.  Everything that does not depend on the
// observation -- predicted observation zp, the 5 non-zero Jacobian columns,
// S = H P H' + R, det S -- is evaluated ONCE per landmark from 12 state values
// (x_f 2, P_fv 6 read from the contiguous column strip P[3:,0:3], P_ff 4) plus
// the wave-uniform pose / P_vv.  The loop over the nz observations then costs a
// 2-vector innovation and a 2x2 quadratic form per pair.  The reference instead
// forms a dense 2 x n H and a dense H*P*H' per pair (:59).
//
// Per-observation decision, order-independent form of the scan at :21-50
// (SURVEY.md 3.2):  jbest = argmin_{j : nis_j < gate1, nd_j < Inf} nd_j, lowest j
// on ties; if none: new feature iff no j has nis_j <= gate2 (outer > gate2).
// Only waves that contain an in-gate landmark pay for a shuffle reduction; the
// rest contribute two ballots per observation.  A workgroup is eight waves over the SAME
// 64 landmarks, each taking every eighth observation: N/64 one-wave workgroups left
// most SIMDs idle while each wave ground through all nz observations alone.
#include "common.h"
#include "device_math.h"

namespace {

constexpr int GATE_BLOCK = 64;      // landmarks per workgroup (one per lane): N/64 workgroups spread the sweep over the CUs
constexpr int GATE_WAVES = 1;       // (partials per workgroup and observation)
constexpr int OBS_WAVES = 8;        // waves per workgroup: all own the same 64 landmarks, wave w the observations w, w+8, ...

struct PairConst {        // per-landmark, observation-independent
    double zp0, zp1;
    double s00, s01, s10, s11;
    double det, logdet;
    double qa, qb, qc;    // nis = qa*v0^2 + qb*v0*v1 + qc*v1^2  (inv(S) folded in once per landmark)
};

// S = Hv Pvv Hv' + Hv Pvf Hf' + Hf Pfv Hv' + Hf Pff Hf' + R  (2x2), from the
// 5x5 sub-block of P.  pvv is row-major 3x3; pfv[a][c] = P[f+a][c]; pff row-major.
__device__ inline PairConst pair_const(const ObsModel& om, const double* pvv, const double pfv[2][3],
                                       const double pff[4], const double R[4]) {
    PairConst pc;
    pc.zp0 = om.zp[0];
    pc.zp1 = om.zp[1];
    double A[2][3];   // Hv * Pvv
    double B[2][2];   // Hv * Pvf,  Pvf[c][a] = pfv[a][c]
    double Cc[2][3];  // Hf * Pfv
    double D[2][2];   // Hf * Pff
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A[r][c] = om.Hv[r * 3 + 0] * pvv[0 * 3 + c] + om.Hv[r * 3 + 1] * pvv[1 * 3 + c] +
                      om.Hv[r * 3 + 2] * pvv[2 * 3 + c];
            Cc[r][c] = om.Hf[r * 2 + 0] * pfv[0][c] + om.Hf[r * 2 + 1] * pfv[1][c];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            B[r][a] = om.Hv[r * 3 + 0] * pfv[a][0] + om.Hv[r * 3 + 1] * pfv[a][1] + om.Hv[r * 3 + 2] * pfv[a][2];
            D[r][a] = om.Hf[r * 2 + 0] * pff[0 * 2 + a] + om.Hf[r * 2 + 1] * pff[1 * 2 + a];
        }
    }
    double S[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double s = A[r][0] * om.Hv[q * 3 + 0] + A[r][1] * om.Hv[q * 3 + 1] + A[r][2] * om.Hv[q * 3 + 2];
            s += B[r][0] * om.Hf[q * 2 + 0] + B[r][1] * om.Hf[q * 2 + 1];
            s += Cc[r][0] * om.Hv[q * 3 + 0] + Cc[r][1] * om.Hv[q * 3 + 1] + Cc[r][2] * om.Hv[q * 3 + 2];
            s += D[r][0] * om.Hf[q * 2 + 0] + D[r][1] * om.Hf[q * 2 + 1];
            S[r][q] = s + R[q * 2 + r];       // R column-major: R[r][q] = R[q*2+r]
        }
    pc.s00 = S[0][0]; pc.s01 = S[0][1]; pc.s10 = S[1][0]; pc.s11 = S[1][1];
    pc.det = pc.s00 * pc.s11 - pc.s01 * pc.s10;
    pc.logdet = log(pc.det);
    const double rdet = 1.0 / pc.det;
    pc.qa = pc.s11 * rdet;
    pc.qb = -(pc.s01 + pc.s10) * rdet;
    pc.qc = pc.s00 * rdet;
    return pc;
}

// nis = v' inv(S) v  (src/data-association.jl:60), nd = nis + log det S (:61)
__device__ inline void pair_eval(const PairConst& pc, double z0, double z1, double& nis, double& nd) {
    const double v0 = z0 - pc.zp0;
    const double v1 = mpi_to_pi_d(z1 - pc.zp1);        // :57
    nis = pc.qa * v0 * v0 + pc.qb * v0 * v1 + pc.qc * v1 * v1;
    nd = nis + pc.logdet;
}

// side (may be null: the single-pair entry points read the matrix): the packed 2 x 2 diagonal blocks, three coalesced rows
template <typename T>
__device__ inline PairConst landmark_const(const T* __restrict__ x, const T* __restrict__ P, int ld, int tlog, int j0,
                                           const double pose[3], const double* pvv, const double R[4],
                                           const T* __restrict__ side = nullptr, int side_n = 0) {
    const int f = 3 + 2 * j0;
    const double lx = (double)x[f], ly = (double)x[f + 1];
    const ObsModel om = obs_model(pose[0], pose[1], pose[2], lx, ly);
    double pfv[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pfv[0][c] = (double)P[p_off(ld, tlog, f, c)];
        pfv[1][c] = (double)P[p_off(ld, tlog, f + 1, c)];
    }
    double pff[4];
    if (side) {
        pff[0] = (double)side[j0];
        pff[2] = (double)side[(size_t)side_n + j0];
        pff[3] = (double)side[(size_t)2 * side_n + j0];
    } else {
        pff[0] = (double)P[p_off(ld, tlog, f, f)];           // P[f][f]
        pff[2] = (double)P[p_off(ld, tlog, f + 1, f)];       // P[f+1][f]  (lower: always stored)
        pff[3] = (double)P[p_off(ld, tlog, f + 1, f + 1)];   // P[f+1][f+1]
    }
    pff[1] = pff[2];                                     // P[f][f+1]  = its mirror (it may lie in a tile above the diagonal)
    return pair_const(om, pvv, pfv, pff, R);
}

// ---- N2: the spatial pre-gate (the reference's TODO, src/data-association.jl:18-20: "a quick bounding-box threshold to
// remove distant features") -- with IDENTICAL decisions.  For S positive definite, nis = v' inv(S) v >= v_k^2 / S_kk for
// either component k (Cauchy-Schwarz), and for P positive semi-definite |P_ij| <= sqrt(P_ii P_jj), so
//     S_kk = h_k P5 h_k' + R_kk <= (sum_j |h_kj| sqrt(P_jj))^2 + R_kk =: B_k
// where the landmark's own two variances are replaced by the filter-wide bound pmax >= max diag(P_ff) (kept on the device:
// set at upload, raised by add_features, never raised by the down-date, which only lowers a diagonal).  B_k needs the
// landmark's MEAN only (two coalesced values); if v_k^2 > gate2 * B_k for k = 0 or 1, the pair has nis > gate2: it is
// neither a candidate (nis < gate1 <= gate2) nor "near" (nis <= gate2) and contributes nothing -- exactly what the
// full evaluation would find.  A landmark that is out of reach of all the wave's observations never loads its ten
// covariance values (the stride-(2 ld + 2) gathers that made the sweep fetch 6.5x its algorithmic bytes).
struct PreGate {
    double b0, b1;        // gate2 * B_k * (1 + slack); +inf where the bound is not usable (then nothing is skipped)
};

__device__ inline PreGate pre_gate(double dx, double dy, const double* pvv, double pmax, const double R[4], double gate2) {
    PreGate g;
    const double d2 = dx * dx + dy * dy, d = sqrt(d2);
    const double sx = sqrt(pvv[0]), sy = sqrt(pvv[4]), sp = sqrt(pvv[8]), sm = sqrt(pmax);
    const double ax = fabs(dx), ay = fabs(dy);
    // h_0 = [-dx/d, -dy/d, 0, dx/d, dy/d],  h_1 = [dy/d2, -dx/d2, -1, -dy/d2, dx/d2]   (src/common.jl:161-162)
    const double a0 = (ax * sx + ay * sy + (ax + ay) * sm) / d;
    const double a1 = (ay * sx + ax * sy + (ax + ay) * sm) / d2 + sp;
    const double slack = 1.0 + 1e-6;
    g.b0 = gate2 * (a0 * a0 + R[0]) * slack;
    g.b1 = gate2 * (a1 * a1 + R[3]) * slack;
    if (!(g.b0 > 0.0) || !(g.b0 < __builtin_inf())) g.b0 = __builtin_inf();      // NaN, d = 0, a negative variance: no skipping
    if (!(g.b1 > 0.0) || !(g.b1 < __builtin_inf())) g.b1 = __builtin_inf();
    return g;
}

// observe(): turn the association vector into the update's and add_features' inputs without leaving the device.
// One wave walks the observations in order, 64 at a time: matched ones (assoc >= 1) are compacted to the front
// of zbuf/idf (zsrc may BE zbuf: a write position never passes the read position of a later chunk), new ones
// (assoc < 0) go to zn.  count = {matched, new}.  Order is the observation order, as in data-association.jl:43-47.
template <bool ZAGENT = false>
__device__ __forceinline__ void compact_wave(const int32_t* assoc, int nz, const double* zsrc,
                                             double* zbuf, int32_t* __restrict__ idf,
                                             double* __restrict__ zn, int32_t* __restrict__ count,
                                             int32_t* __restrict__ assoc_host, int lane, int32_t* __restrict__ flag_host,
                                             int32_t seq) {
    int m = 0, nn = 0;
    for (int base = 0; base < nz; base += 64) {
        const int i = base + lane;
        // the decisions were stored write-through by the other workgroups of this launch: read them at agent scope
        const int a = i < nz ? __hip_atomic_load(assoc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        // ZAGENT: the observations were stored (write-through) by other workgroups of this very launch
        double z0 = 0.0, z1 = 0.0;
        if (i < nz) {
            if (ZAGENT) {
                z0 = __hip_atomic_load(zsrc + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                z1 = __hip_atomic_load(zsrc + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                z0 = zsrc[2 * i];
                z1 = zsrc[2 * i + 1];
            }
        }
        if (i < nz) __hip_atomic_store(assoc_host + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host memory
        const unsigned long long mm = __ballot(a > 0), mn = __ballot(a < 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        __builtin_amdgcn_s_waitcnt(0);           // every lane holds its z before any lane overwrites the front
        __builtin_amdgcn_wave_barrier();
        if (a > 0) {
            const int pos = m + __popcll(mm & below);
            idf[pos] = a;
            zbuf[2 * pos] = z0;
            zbuf[2 * pos + 1] = z1;
        } else if (a < 0) {
            const int pos = nn + __popcll(mn & below);
            zn[2 * pos] = z0;
            zn[2 * pos + 1] = z1;
        }
        m += __popcll(mm);
        nn += __popcll(mn);
    }
    if (lane == 0) {
        count[0] = m;
        count[1] = nn;
    }
    // the host polls this word in pinned memory instead of waiting on an event (an event record costs ~6 us of
    // stream time): the decisions go out as write-through system-scope stores, every lane drains its own, then the
    // sequence number of this call.  No fence: a system-scope release writes back this XCD's whole L2.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The observations of one launch travel IN the kernel arguments (2 KB of the 4 KB a launch may carry): every workgroup
// needs all of them at its start, and reading them from the pinned host page the caller's z was staged in meant a few
// thousand 64-byte reads across PCIe per sweep, on the critical path of a 9 us kernel.  First parameter: the kernel
// indexes the argument segment itself (a dynamic index into the by-value copy would move the struct to scratch).
constexpr int GATE_CHUNK = 128;        // observations per launch
__device__ __forceinline__ const double* zdev_base(const double*, const double* zbuf) { return zbuf; }     // (the chunks' copies sit in the one buffer)
struct GateObs {
    double z[2 * GATE_CHUNK];
};

template <typename T>
__global__ __launch_bounds__(GATE_BLOCK * OBS_WAVES) void gate_kernel(GateObs zarg, const T* __restrict__ x, const T* __restrict__ P,
                                                           int ld, int N, double* __restrict__ zdev, int nz, double R0,
                                                           double R1, double R2, double R3, double gate1, double gate2,
                                                           double* __restrict__ part, const double* __restrict__ pmax_ptr,
                                                           int pregate, int tlog, const T* __restrict__ side, int side_n,
                                                           int part_cap, int32_t* assoc, int32_t* assoc_all, int compact_total,
                                                           double* zbuf, int32_t* __restrict__ idf, double* __restrict__ zn,
                                                           int32_t* __restrict__ count, int32_t* __restrict__ assoc_host, int32_t* arrive,
                                                           int32_t* __restrict__ flag_host, int32_t seq) {
    extern __shared__ double smem[];
    double* zs = smem;                 // [nz][2]
    double* red = smem + 2 * nz;       // [nz][3]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int NT = GATE_BLOCK * OBS_WAVES;
    typedef const __attribute__((address_space(4))) GateObs* KargPtr;
    const KargPtr ka = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    for (int i = tid; i < 2 * nz; i += NT) {
        const double v = ka->z[i];
        zs[i] = v;
        // the device copy the compaction (this launch's last workgroup), the update and add_features read: write-through
        if (blockIdx.x == 0) __hip_atomic_store(zdev + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Round 5: ONE launch.  What used to be the partials of every workgroup for every observation (nz x N / 64 records, folded by a
    // second kernel) is now SPARSE: almost no workgroup has a landmark inside an observation's gates, and the few that do append
    // {nd, landmark} to the observation's list (a slot from an atomic counter) and raise its `near` word; the workgroup that
    // arrives LAST folds the lists -- a couple of entries per observation -- takes the decisions and, for observe(), compacts
    // them.  part = [list: chunk observations x part_cap entries x 2 doubles][cnt: int32 per observation][near: int32 per observation].
    double* const list = part;
    int32_t* const cnt = reinterpret_cast<int32_t*>(part + (size_t)2 * GATE_CHUNK * part_cap);
    int32_t* const nearw = cnt + GATE_CHUNK;

    const double R[4] = {R0, R1, R2, R3};
    double pose[3] = {(double)x[0], (double)x[1], (double)x[2]};
    double pvv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) pvv[r * 3 + c] = (double)P[p_off(ld, tlog, r, c)];

    // the landmark constants are evaluated by every wave (cheaper than a hand-over through LDS: ~200 flops)
    const int j0 = blockIdx.x * GATE_BLOCK + lane;
    bool valid = j0 < N;
    const double INF = __builtin_inf();
    if (pregate) {
        // N2: which of this wave's landmarks are within reach of ANY of this wave's observations?  Needs the landmark's
        // mean only; the others are provably outside both gates for every one of them and skip the covariance loads.
        __syncthreads();                                        // the observations are in LDS
        bool need = false;
        if (valid) {
            const int f = 3 + 2 * j0;
            const double dx = (double)x[f] - pose[0], dy = (double)x[f + 1] - pose[1];
            const PreGate g = pre_gate(dx, dy, pvv, *pmax_ptr, R, gate2);
            const double zp0 = sqrt(dx * dx + dy * dy), zp1 = atan2(dy, dx) - pose[2];
            for (int i = wave; i < nz; i += OBS_WAVES) {
                const double v0 = zs[2 * i] - zp0, v1 = mpi_to_pi_d(zs[2 * i + 1] - zp1);
                if (!(v0 * v0 > g.b0) && !(v1 * v1 > g.b1)) need = true;
            }
        }
        valid = need;
    }
    PairConst pc;
    if (valid) pc = landmark_const(x, P, ld, tlog, j0, pose, pvv, R, side, side_n);
    __syncthreads();                                            // (the observations are in LDS)

    // Almost every (observation, landmark) pair is far outside both gates: the common path is a
    // 2-vector innovation, a 2x2 quadratic form and two ballots, with no LDS traffic at all.
    for (int i = wave; i < nz; i += OBS_WAVES) {
        double nis = INF, nd = INF;
        if (valid) pair_eval(pc, zs[2 * i], zs[2 * i + 1], nis, nd);
        const bool cand = valid && (nis < gate1) && (nd < INF);
        const bool near = valid && (nis <= gate2);
        const unsigned long long cand_mask = __ballot(cand);
        const unsigned long long near_mask = __ballot(near);
        if ((cand_mask | near_mask) != 0ull) {       // wave-uniform, rare
            double nd_c = INF;
            int j_c = 0x7fffffff;
            if (cand_mask != 0ull) {
                if (cand) { nd_c = nd; j_c = j0 + 1; }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const double o_nd = __shfl_xor(nd_c, off);
                    const int o_j = __shfl_xor(j_c, off);
                    if (o_nd < nd_c || (o_nd == nd_c && o_j < j_c)) { nd_c = o_nd; j_c = o_j; }
                }
            }
            if (lane == 0) {
                (void)__hip_atomic_fetch_or(nearw + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // near_mask != 0 or a candidate (also near-or-matched)
                if (cand_mask != 0ull) {
                    const int e = __hip_atomic_fetch_add(cnt + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    double* o = list + ((size_t)i * part_cap + e) * 2;
                    __hip_atomic_store(o, nd_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(o + 1, (double)j_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    // every wave's write-through stores and atomics are out; then this workgroup's arrival.  The workgroup whose add comes last
    // reads the lists with agent-scope loads (no release / acquire fence = no L2 write-back + invalidate)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const s_last = reinterpret_cast<int*>(red);
    if (tid == 0) *s_last = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!*s_last) return;
    // ---- the last workgroup: per observation the best candidate (smallest nd, the lower landmark on a tie -- whatever order the
    //      entries arrived in), the decision of src/data-association.jl:43-47; the words are re-armed for the next launch ----
    for (int i = tid; i < nz; i += NT) {
        const int c = __hip_atomic_load(cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nr = __hip_atomic_load(nearw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double nd_c = INF, j_c = (double)0x7fffffff;
        for (int e = 0; e < c; ++e) {
            const double* o = list + ((size_t)i * part_cap + e) * 2;
            const double r0 = __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double r1 = __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r0 < nd_c || (r0 == nd_c && r1 < j_c)) { nd_c = r0; j_c = r1; }
        }
        int32_t a;
        if (nd_c < INF) a = (int32_t)j_c;            // jbest != 0          (:43)
        else if (!nr) a = -1;                        // outer > gate2       (:46)
        else a = 0;                                  // dropped
        __hip_atomic_store(assoc + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cnt + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(nearw + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // re-armed
    if (compact_total > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // observe(), last chunk: assoc[0 .. compact_total) -> the update's inputs, no extra launch
        if (wave == 0) compact_wave<true>(assoc_all, compact_total, zdev_base(zdev, zbuf), zbuf, idf, zn, count, assoc_host, lane, flag_host, seq);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// N2, the O(candidates) form (the reference's TODO, src/data-association.jl:18-20: "a quick bounding-box threshold to
// remove distant features; or, better yet, a balanced k-d tree lookup"): a UNIFORM GRID over the landmark means, kept
// on the device, so that an observation evaluates only the landmarks of the cells its gate can reach.  Decisions are
// IDENTICAL to the sweep's (asserted by the tests), by the pre-gate's inequalities made position-independent:
//     nis <= gate2  =>  v_k^2 <= gate2 * S_kk (k = 0, 1),   S_00 <= A0^2 + R_00,   S_11 <= (A0 / d + sp)^2 + R_11,
//     A0 = sqrt(Pvv_xx + Pvv_yy) + sqrt(2 pmax),  sp = sqrt(Pvv_phiphi),  d = distance pose -> landmark
// (|h_0j| <= 1 with sum over x, y of |h_0j| sqrt(P_jj) <= sqrt(sx^2 + sy^2) by Cauchy-Schwarz, (|dx| + |dy|) / d <= sqrt 2; the
// bearing row is the range row / d plus the heading).  Hence a landmark that matters to observation (r, b) lies in the
// annulus r - rho <= d <= r + rho, rho = sqrt(gate2 (A0^2 + R_00)), and in the sector |bearing - b| <= beta, beta =
// sqrt(gate2 ((A0 / (r - rho) + sp)^2 + R_11)) (the whole annulus when r <= rho or beta >= pi): the kernel takes the
// bounding box of that sector, in cells, and of the landmarks in those cells the ones inside the annulus.
//
// The means MOVE with every update.  The grid is not rebuilt for that: every update records the largest displacement
// of a landmark coordinate (one atomicMax per WORKGROUP of the kernel that applies x += W v), a query adds the bounds up
// (`drift`) and widens box and annulus by it; landmarks appended since the build sit in a tail that every query scans.
// The item list carries each landmark's mean AT BUILD TIME, so the annulus test needs no second round trip (the exact
// evaluation reads the current mean together with the covariance entries).  Now and then (every 16th update, a tail
// of min(512, 32 + a quarter of the grid's landmarks), a state upload) the host puts a one-workgroup kernel in front
// of the query that folds the bounds and REBUILDS the grid -- histogram in LDS, scan, scatter: a counting sort by
// cell -- if drift exceeds a quarter of a cell or the tail has that size.  A query costs O(landmarks in the box +
// tail), whatever N is, in ONE launch: the per-observation decision and the compaction of the sweep's second kernel
// ride in the same kernel.
constexpr int GRID_MAX_G = 128;          // cells per axis: the build's histogram (G^2 ints) sits in LDS
constexpr int GRID_SLOTS = SLAM_GRID_SLOTS;   // updates whose displacement bounds are kept apart until the next fold
constexpr int GRID_SUBS = 16;            // an update's bound arrives as 16 partial maxima (its workgroups spread their atomics)
constexpr int GRID_FOLD_UPDATES = 16;    // the host enqueues the fold / rebuild check after this many updates ...
constexpr int GRID_TAIL_MAX = 512;       // ... or when the tail has reached min(this, 32 + a quarter of the grid's population)
constexpr int GRID_AUTO_N = 16384;       // SLAM_GATE_AUTO: the grid from this many landmarks on (measured)
constexpr int GRID_BUILD_THREADS = 1024;
constexpr int GRID_QUERY_THREADS = 256;

struct GridItem {                        // 24 bytes
    double bx, by;                       // the landmark's mean when the grid was built
    int j0, pad;
};

struct GridMeta {
    double x0, y0, invx, invy;           // cell of a point: floor((p - p0) * inv), clamped to [0, G)
    double cellmin;                      // the shorter cell edge
    double drift;                        // >= |mean now - mean at build time| of any coordinate, as of the last fold
    int G, n_built, valid, pad;
    unsigned long long rebuilds, queries, boxed, evals;      // counters (slam_ekf_gate_info)
    unsigned long long slot[GRID_SUBS][GRID_SLOTS];      // bit patterns of doubles: slot[.][u] = partial maxima of the largest
                                         // displacement of update u since the last fold
};

__device__ __forceinline__ int grid_coord(double p, double p0, double inv, int G) {
    const double t = floor((p - p0) * inv);          // monotone in p
    if (t >= (double)(G - 1)) return G - 1;
    return t > 0.0 ? (int)t : 0;                     // NaN -> 0
}

template <typename T>
__global__ __launch_bounds__(GRID_BUILD_THREADS) void grid_prepare_kernel(const T* __restrict__ x, int N, GridMeta* __restrict__ meta,
                                                                          int32_t* __restrict__ cell_start,
                                                                          GridItem* __restrict__ items, int nslots, int force) {
    extern __shared__ int g_hist[];                  // [G * G]
    __shared__ double s_red[4][GRID_BUILD_THREADS / 64];
    __shared__ int s_scan[GRID_BUILD_THREADS / 64];
    __shared__ int s_do;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = GRID_BUILD_THREADS / 64;
    const double INF = __builtin_inf();
    if (wave == 0) {
        // the displacement bounds of the updates since the last fold: add them up (each is a max over the landmarks)
        double d = 0.0;
        if (lane < nslots) {
            unsigned long long m = 0ull;
            for (int sub = 0; sub < GRID_SUBS; ++sub) {
                const unsigned long long v = meta->slot[sub][lane];
                m = v > m ? v : m;
                meta->slot[sub][lane] = 0ull;
            }
            d = __longlong_as_double((long long)m);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
        if (lane == 0) {
            d = (meta->drift + d) * (1.0 + 1e-12);
            const int tail_max = min(GRID_TAIL_MAX, 32 + meta->n_built / 4);
            const bool rebuild = force || !meta->valid || !(d <= 0.25 * meta->cellmin) || N < meta->n_built ||
                                 N - meta->n_built >= tail_max;
            meta->drift = d;
            s_do = rebuild ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_do) return;

    // bounding box of the means
    double xmin = INF, xmax = -INF, ymin = INF, ymax = -INF;
    for (int j = tid; j < N; j += GRID_BUILD_THREADS) {
        const double lx = (double)x[3 + 2 * j], ly = (double)x[4 + 2 * j];
        if (lx < xmin) xmin = lx;
        if (lx > xmax) xmax = lx;
        if (ly < ymin) ymin = ly;
        if (ly > ymax) ymax = ly;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        xmin = fmin(xmin, __shfl_xor(xmin, off));
        xmax = fmax(xmax, __shfl_xor(xmax, off));
        ymin = fmin(ymin, __shfl_xor(ymin, off));
        ymax = fmax(ymax, __shfl_xor(ymax, off));
    }
    if (lane == 0) { s_red[0][wave] = xmin; s_red[1][wave] = xmax; s_red[2][wave] = ymin; s_red[3][wave] = ymax; }
    __syncthreads();
    xmin = s_red[0][0]; xmax = s_red[1][0]; ymin = s_red[2][0]; ymax = s_red[3][0];
    for (int w = 1; w < NW; ++w) {
        xmin = fmin(xmin, s_red[0][w]); xmax = fmax(xmax, s_red[1][w]);
        ymin = fmin(ymin, s_red[2][w]); ymax = fmax(ymax, s_red[3][w]);
    }
    if (!(xmin <= xmax) || !(xmax < INF) || !(xmin > -INF)) { xmin = 0.0; xmax = 0.0; }      // no (finite) landmark
    if (!(ymin <= ymax) || !(ymax < INF) || !(ymin > -INF)) { ymin = 0.0; ymax = 0.0; }
    int G = (int)sqrt((double)N / 4.0);              // ~4 landmarks per cell
    G = G < 4 ? 4 : (G > GRID_MAX_G ? GRID_MAX_G : G);
    const double scale = fmax(fmax(fabs(xmin), fabs(xmax)), fmax(fabs(ymin), fabs(ymax)));
    const double tiny = fmax(1e-9, 1e-12 * scale);
    double cwx = (xmax - xmin) / G, cwy = (ymax - ymin) / G;
    if (!(cwx > tiny)) cwx = fmax(1.0, tiny);        // all landmarks on one line: one row / column of cells is used
    if (!(cwy > tiny)) cwy = fmax(1.0, tiny);
    const double invx = 1.0 / cwx, invy = 1.0 / cwy;
    const int cells = G * G;
    for (int c = tid; c < cells; c += GRID_BUILD_THREADS) g_hist[c] = 0;
    __syncthreads();
    for (int j = tid; j < N; j += GRID_BUILD_THREADS) {
        const int cx = grid_coord((double)x[3 + 2 * j], xmin, invx, G), cy = grid_coord((double)x[4 + 2 * j], ymin, invy, G);
        atomicAdd(&g_hist[cy * G + cx], 1);
    }
    __syncthreads();
    // exclusive scan: a thread owns `per` consecutive cells
    const int per = (cells + GRID_BUILD_THREADS - 1) / GRID_BUILD_THREADS;
    const int c0 = tid * per;
    int sum = 0;
    for (int u = 0; u < per; ++u)
        if (c0 + u < cells) sum += g_hist[c0 + u];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_scan[w];
    int run = base + incl - sum;
    for (int u = 0; u < per; ++u)
        if (c0 + u < cells) {
            const int cnt = g_hist[c0 + u];
            cell_start[c0 + u] = run;
            g_hist[c0 + u] = run;                    // the scatter's cursor
            run += cnt;
        }
    if (tid == 0) cell_start[cells] = N;
    __syncthreads();
    for (int j = tid; j < N; j += GRID_BUILD_THREADS) {
        const double lx = (double)x[3 + 2 * j], ly = (double)x[4 + 2 * j];
        const int cx = grid_coord(lx, xmin, invx, G), cy = grid_coord(ly, ymin, invy, G);
        GridItem it;
        it.bx = lx; it.by = ly; it.j0 = j; it.pad = 0;
        items[atomicAdd(&g_hist[cy * G + cx], 1)] = it;      // order inside a cell: any (the decision does not depend on it)
    }
    if (tid == 0) {
        meta->x0 = xmin; meta->y0 = ymin; meta->invx = invx; meta->invy = invy;
        meta->cellmin = fmin(cwx, cwy);
        meta->drift = 0.0;
        meta->G = G; meta->n_built = N; meta->valid = 1;
        meta->rebuilds += 1ull;
    }
}

// does the closed interval of angles [a, b] (b - a < 2 pi) contain t + 2 pi k for some integer k?
__device__ __forceinline__ bool arc_has(double a, double b, double t) {
    const double k = ceil((a - t) / (2.0 * SLAM_PI_D));
    return t + 2.0 * SLAM_PI_D * k <= b;
}

// One workgroup per observation: the cells of the gate's bounding box (wave w the cell rows w, w + 4, ...: the items of
// one row of cells are one contiguous run of the item list), then the tail.  Per item the annulus test on the mean at
// build time (widened by the drift), then the sweep's own evaluation (landmark_const / pair_eval: the same values, bit
// for bit).  Then what gate_final_kernel does for the sweep: the decision (data-association.jl:43-47), and the
// workgroup that arrives last compacts the decisions into the update's inputs.
template <typename T>
__global__ __launch_bounds__(GRID_QUERY_THREADS) void gate_grid_kernel(
    GateObs zarg, const T* __restrict__ x, const T* __restrict__ P, int ld, int N, double* zdev, int nz, double R0, double R1, double R2,
    double R3, double gate1, double gate2, const double* __restrict__ pmax_ptr, int tlog, const T* __restrict__ side, int side_n,
    GridMeta* meta, int nslots, const int32_t* __restrict__ cell_start, const GridItem* __restrict__ items,
    int32_t* assoc, int32_t* assoc_all, int compact_total, double* zall, int32_t* __restrict__ idf, double* __restrict__ zn,
    int32_t* __restrict__ count, int32_t* __restrict__ assoc_host, int32_t* arrive, int32_t* __restrict__ flag_host, int32_t seq) {
    __shared__ double s_nd[GRID_QUERY_THREADS / 64];
    __shared__ int s_j[GRID_QUERY_THREADS / 64], s_near[GRID_QUERY_THREADS / 64];
    __shared__ unsigned s_cnt[2][GRID_QUERY_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = GRID_QUERY_THREADS / 64;
    const int i = blockIdx.x;
    typedef const __attribute__((address_space(4))) GateObs* KargPtr;
    const KargPtr ka = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const double z0 = ka->z[2 * i], z1 = ka->z[2 * i + 1];
    // first round trip: everything the reach needs, requested together
    // (the host folds after GRID_FOLD_UPDATES = 16 updates: a query sees at most 15 slots; lane = 16 * slot group + sub)
    unsigned long long dm[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int sl = 4 * it + (lane >> 4);
        dm[it] = sl < nslots ? meta->slot[lane & 15][sl] : 0ull;
    }
    const double R[4] = {R0, R1, R2, R3};
    const double pose[3] = {(double)x[0], (double)x[1], (double)x[2]};
    double pvv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) pvv[r * 3 + c] = (double)P[p_off(ld, tlog, r, c)];
    const double pmax = *pmax_ptr;
    const int G = meta->G, n_built = meta->n_built < N ? meta->n_built : N;
    const double gx0 = meta->x0, gy0 = meta->y0, invx = meta->invx, invy = meta->invy, drift0 = meta->drift;
    if (tid == 0) {      // the device copy the compaction, the update and add_features read (write-through: the compacting
                         // workgroup of THIS launch reads it at agent scope)
        __hip_atomic_store(zdev + 2 * i, z0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(zdev + 2 * i + 1, z1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double dsum = 0.0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int off = 1; off <= 8; off <<= 1) {                       // max over the 16 partial maxima (bit patterns order like the values)
            const unsigned long long o = __shfl_xor(dm[it], off);
            dm[it] = o > dm[it] ? o : dm[it];
        }
        dsum += __longlong_as_double((long long)dm[it]);
    }
    dsum += __shfl_xor(dsum, 16);
    dsum += __shfl_xor(dsum, 32);
    const double drift = (drift0 + dsum) * (1.0 + 1e-12);           // NaN: every test below fails open
    const double INF = __builtin_inf();

    // the reach of this observation (header comment); anything not finite: the whole grid, every item
    int cx0 = 0, cx1 = G - 1, cy0 = 0, cy1 = G - 1;
    double d_lo = -INF, d_hi = INF;                                // the annulus, widened: build-time distances that can matter
    {
        const double A0 = sqrt(fmax(pvv[0], 0.0) + fmax(pvv[4], 0.0)) + sqrt(2.0 * fmax(pmax, 0.0));
        const double sp = sqrt(fmax(pvv[8], 0.0));
        const double rho = sqrt(gate2 * (A0 * A0 + R0)) * (1.0 + 1e-6) + 1e-9;
        const double dmax = z0 + rho;
        double dmin = z0 - rho;
        if (dmin < 0.0) dmin = 0.0;
        if (dmax < INF && dmax == dmax && drift == drift && drift < INF) {
            if (dmax < 0.0) { cx0 = 1; cx1 = 0; d_hi = -INF; }          // a negative range beyond the gate: nothing can match
            else {
                double cmin = -1.0, cmax = 1.0, smin = -1.0, smax = 1.0;
                if (dmin > 0.0) {
                    const double a1 = A0 / dmin + sp;
                    const double beta = sqrt(gate2 * (a1 * a1 + R3)) * (1.0 + 1e-6) + 1e-9;
                    const double th = z1 + pose[2];
                    if (beta < SLAM_PI_D && th == th && fabs(th) < 1e6) {
                        const double a = th - beta, b = th + beta;
                        const double ca = cos(a), cb = cos(b), sa = sin(a), sb = sin(b);
                        cmax = arc_has(a, b, 0.0) ? 1.0 : fmax(ca, cb);
                        cmin = arc_has(a, b, SLAM_PI_D) ? -1.0 : fmin(ca, cb);
                        smax = arc_has(a, b, 0.5 * SLAM_PI_D) ? 1.0 : fmax(sa, sb);
                        smin = arc_has(a, b, -0.5 * SLAM_PI_D) ? -1.0 : fmin(sa, sb);
                    }
                }
                const double eps = 1e-9 * (dmax + fabs(pose[0]) + fabs(pose[1]) + 1.0);
                const double pad = drift + eps;
                const double xlo = fmin(cmin * dmax, cmin * dmin), xhi = fmax(cmax * dmax, cmax * dmin);
                const double ylo = fmin(smin * dmax, smin * dmin), yhi = fmax(smax * dmax, smax * dmin);
                cx0 = grid_coord(pose[0] + xlo - pad, gx0, invx, G);
                cx1 = grid_coord(pose[0] + xhi + pad, gx0, invx, G);
                cy0 = grid_coord(pose[1] + ylo - pad, gy0, invy, G);
                cy1 = grid_coord(pose[1] + yhi + pad, gy0, invy, G);
                d_lo = dmin - 1.4142135623730951 * drift - eps;      // a mean has moved by at most sqrt(2) drift
                d_hi = dmax + 1.4142135623730951 * drift + eps;
            }
        }
    }

    double best_nd = INF;
    int best_j = 0x7fffffff;
    bool near_any = false;
    unsigned n_box = 0, n_eval = 0;
    auto evaluate = [&](int j0) {
        ++n_eval;
        const PairConst pc = landmark_const(x, P, ld, tlog, j0, pose, pvv, R, side, side_n);
        double nis, nd;
        pair_eval(pc, z0, z1, nis, nd);
        if (nis <= gate2) near_any = true;
        if (nis < gate1 && nd < INF && (nd < best_nd || (nd == best_nd && j0 + 1 < best_j))) {
            best_nd = nd;
            best_j = j0 + 1;
        }
    };
    if (cx0 <= cx1)
        for (int cy = cy0 + wave; cy <= cy1; cy += NW) {
            const int k0 = cell_start[cy * G + cx0], k1 = cell_start[cy * G + cx1 + 1];
            for (int k = k0 + lane; k < k1; k += 64) {
                const GridItem it = items[k];
                ++n_box;
                const double dx = it.bx - pose[0], dy = it.by - pose[1];
                const double db = sqrt(dx * dx + dy * dy);
                if (db < d_lo || db > d_hi) continue;               // (a NaN mean is evaluated)
                if (it.j0 < N) evaluate(it.j0);
            }
        }
    for (int j0 = n_built + tid; j0 < N; j0 += GRID_QUERY_THREADS) {               // appended since the build
        ++n_box;
        evaluate(j0);
    }

#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o_nd = __shfl_xor(best_nd, off);
        const int o_j = __shfl_xor(best_j, off);
        if (o_nd < best_nd || (o_nd == best_nd && o_j < best_j)) { best_nd = o_nd; best_j = o_j; }
        n_box += __shfl_xor(n_box, off);
        n_eval += __shfl_xor(n_eval, off);
    }
    const bool wave_near = __ballot(near_any) != 0ull;
    if (lane == 0) { s_nd[wave] = best_nd; s_j[wave] = best_j; s_near[wave] = wave_near ? 1 : 0; s_cnt[0][wave] = n_box; s_cnt[1][wave] = n_eval; }
    __syncthreads();
    if (wave != 0) return;
    int last = 0;
    if (lane == 0) {
        bool near = false;
        unsigned long long nb = 0, ne = 0;
        for (int w = 0; w < NW; ++w) {
            if (s_nd[w] < best_nd || (s_nd[w] == best_nd && s_j[w] < best_j)) { best_nd = s_nd[w]; best_j = s_j[w]; }
            near = near || s_near[w] != 0;
            nb += s_cnt[0][w]; ne += s_cnt[1][w];
        }
        int32_t a;
        if (best_nd < INF) a = best_j;               // jbest != 0          (data-association.jl:43)
        else if (!near) a = -1;                      // outer > gate2       (:46)
        else a = 0;                                  // dropped
        __hip_atomic_store(assoc + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (compact_total > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        }
        atomicAdd(&meta->boxed, nb);                 // the counters of slam_ekf_gate_info: behind the arrival, nothing waits for them
        atomicAdd(&meta->evals, ne);
        if (i == 0) atomicAdd(&meta->queries, 1ull);
    }
    if (compact_total > 0 && __shfl(last, 0)) {
        compact_wave<true>(assoc_all, compact_total, zall, zall, idf, zn, count, assoc_host, lane, flag_host, seq);
        if (lane == 0) __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    }
}

// compute_association for ONE pair and predict_observation for ONE landmark.
template <typename T>
__global__ void single_pair_kernel(const T* __restrict__ x, const T* __restrict__ P, int ld, int j0, double z0, double z1,
                                   double R0, double R1, double R2, double R3, int mode, double* __restrict__ out, int tlog) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double R[4] = {R0, R1, R2, R3};
    double pose[3] = {(double)x[0], (double)x[1], (double)x[2]};
    if (mode == 1) {   // predict_observation: zp[2], Hv (col-major 2x3), Hf (col-major 2x2)
        const int f = 3 + 2 * j0;
        const ObsModel om = obs_model(pose[0], pose[1], pose[2], (double)x[f], (double)x[f + 1]);
        out[0] = om.zp[0]; out[1] = om.zp[1];
        for (int c = 0; c < 3; ++c) { out[2 + 2 * c] = om.Hv[c]; out[2 + 2 * c + 1] = om.Hv[3 + c]; }
        for (int c = 0; c < 2; ++c) { out[8 + 2 * c] = om.Hf[c]; out[8 + 2 * c + 1] = om.Hf[2 + c]; }
        return;
    }
    double pvv[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) pvv[r * 3 + c] = (double)P[p_off(ld, tlog, r, c)];
    const PairConst pc = landmark_const(x, P, ld, tlog, j0, pose, pvv, R);
    double nis, nd;
    pair_eval(pc, z0, z1, nis, nd);
    out[0] = nis;
    out[1] = nd;
}

}  // namespace

// pmax = max over the landmarks' diagonal entries of P (>= 0), as the bit pattern of a non-negative double (which orders
// like the integer); a negative or NaN variance makes it +inf: the pre-gate then skips nothing.
template <typename T>
__global__ __launch_bounds__(256) void diag_max_kernel(const T* __restrict__ P, int ld, int n, unsigned long long* __restrict__ pmax, int tlog) {
    const int i = 3 + blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (i < n) {
        v = (double)P[p_off(ld, tlog, i, i)];
        if (!(v >= 0.0)) v = __builtin_inf();
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    if ((threadIdx.x & 63) == 0 && v > 0.0) atomicMax(pmax, (unsigned long long)__double_as_longlong(v));
}

// (re)compute the bound when the state was uploaded or a Joseph-form update may have raised a diagonal by a rounding
int ensure_pmax(slam_ekf* h) {
    if (h->pmax_valid) return SLAM_OK;
    const int n = 3 + 2 * h->N;
    HIP_TRY(hipMemsetAsync(h->d_pmax, 0, sizeof(double), h->stream));
    if (h->N > 0) {
        const int blocks = (2 * h->N + 255) / 256;
        if (h->dtype == SLAM_F32)
            hipLaunchKernelGGL(diag_max_kernel<float>, dim3(blocks), dim3(256), 0, h->stream, (const float*)h->P, h->ld, n,
                               (unsigned long long*)h->d_pmax, 7);
        else
            hipLaunchKernelGGL(diag_max_kernel<double>, dim3(blocks), dim3(256), 0, h->stream, (const double*)h->P, h->ld, n,
                               (unsigned long long*)h->d_pmax, 6);
        HIP_TRY(hipGetLastError());
    }
    h->pmax_valid = 1;
    return SLAM_OK;
}

// ---- the grid form's host side ----
static int grid_alloc_zero(void** p, size_t bytes, hipStream_t s) {
    *p = nullptr;
    HIP_TRY(hipMalloc(p, bytes));
    HIP_TRY(hipMemsetAsync(*p, 0, bytes, s));
    return SLAM_OK;
}

static int ensure_grid(slam_ekf* h) {
    if (h->grid_meta) return SLAM_OK;
    int rc;
    if ((rc = grid_alloc_zero(&h->grid_meta, sizeof(GridMeta), h->stream))) return rc;
    if ((rc = grid_alloc_zero((void**)&h->grid_cells, sizeof(int32_t) * (GRID_MAX_G * GRID_MAX_G + 1), h->stream))) return rc;
    if ((rc = grid_alloc_zero(&h->grid_items, sizeof(GridItem) * (size_t)(h->maxN > 0 ? h->maxN : 1), h->stream))) return rc;
    h->grid_force = 1;
    h->grid_upd = 0;
    return SLAM_OK;
}

// the update that is about to be enqueued: where its kernel leaves the largest displacement of a landmark mean (null: no
// grid in use).  More than GRID_SLOTS updates between two queries: the next query rebuilds.
unsigned long long* grid_drift_slot(slam_ekf* h) {
    if (!h->grid_live) return nullptr;
    if (h->grid_upd >= GRID_SLOTS) {
        h->grid_force = 1;
        return nullptr;
    }
    return &((GridMeta*)h->grid_meta)->slot[0][h->grid_upd++];
}

static int launch_gate_grid(slam_ekf* h, int nz, const double R[4], double gate1, double gate2, const double* z_host, bool compact) {
    int rc;
    if ((rc = ensure_pmax(h))) return rc;
    if ((rc = ensure_grid(h))) return rc;
    GridMeta* meta = (GridMeta*)h->grid_meta;
    GridItem* items = (GridItem*)h->grid_items;
    const int tlog = h->dtype == SLAM_F32 ? 7 : 6;
    KTimer t(h, SLAM_K_GATE);
    const int tail_max = GRID_TAIL_MAX < 32 + h->grid_n_seen / 4 ? GRID_TAIL_MAX : 32 + h->grid_n_seen / 4;
    if (h->grid_force || !h->grid_live || h->grid_upd >= GRID_FOLD_UPDATES || h->N - h->grid_n_seen >= tail_max || h->N < h->grid_n_seen) {
        // folds the updates' displacement bounds; rebuilds the grid when they, the tail or the host say so
        const size_t lds = sizeof(int) * GRID_MAX_G * GRID_MAX_G;
        if (h->dtype == SLAM_F32)
            hipLaunchKernelGGL(grid_prepare_kernel<float>, dim3(1), dim3(GRID_BUILD_THREADS), lds, h->stream, (const float*)h->x, h->N,
                               meta, h->grid_cells, items, h->grid_upd, h->grid_force);
        else
            hipLaunchKernelGGL(grid_prepare_kernel<double>, dim3(1), dim3(GRID_BUILD_THREADS), lds, h->stream, (const double*)h->x, h->N,
                               meta, h->grid_cells, items, h->grid_upd, h->grid_force);
        HIP_TRY(hipGetLastError());
        h->grid_upd = 0;
        h->grid_force = 0;
        h->grid_live = 1;
        h->grid_n_seen = h->N;
    }
    constexpr int CHUNK = GATE_CHUNK;
    for (int o = 0; o < nz; o += CHUNK) {
        const int cz = nz - o < CHUNK ? nz - o : CHUNK;
        GateObs zarg;
        memcpy(zarg.z, z_host + 2 * (size_t)o, sizeof(double) * 2 * (size_t)cz);
        double* zc = h->obsbuf + 2 * (size_t)o;
        const bool last = o + CHUNK >= nz;
        const int ctot = (compact && last) ? nz : 0;
        if (h->dtype == SLAM_F32)
            hipLaunchKernelGGL(gate_grid_kernel<float>, dim3(cz), dim3(GRID_QUERY_THREADS), 0, h->stream, zarg, (const float*)h->x,
                               (const float*)h->P, h->ld, h->N, zc, cz, R[0], R[1], R[2], R[3], gate1, gate2, (const double*)h->d_pmax,
                               tlog, (const float*)h->Pside, h->npad / 2, meta, h->grid_upd, (const int32_t*)h->grid_cells,
                               (const GridItem*)items, h->d_assoc + o, h->d_assoc, ctot, h->obsbuf, h->idfbuf, h->znbuf, h->d_count,
                               h->h_assoc_dev, h->d_count + 2, h->h_flag_dev, h->obs_seq);
        else
            hipLaunchKernelGGL(gate_grid_kernel<double>, dim3(cz), dim3(GRID_QUERY_THREADS), 0, h->stream, zarg, (const double*)h->x,
                               (const double*)h->P, h->ld, h->N, zc, cz, R[0], R[1], R[2], R[3], gate1, gate2, (const double*)h->d_pmax,
                               tlog, (const double*)h->Pside, h->npad / 2, meta, h->grid_upd, (const int32_t*)h->grid_cells,
                               (const GridItem*)items, h->d_assoc + o, h->d_assoc, ctot, h->obsbuf, h->idfbuf, h->znbuf, h->d_count,
                               h->h_assoc_dev, h->d_count + 2, h->h_flag_dev, h->obs_seq);
        HIP_TRY(hipGetLastError());
    }
    h->gate_last = SLAM_GATE_GRID;
    return SLAM_OK;
}

int gate_kernels_init() {
    const int lds = (int)sizeof(int) * GRID_MAX_G * GRID_MAX_G + 4096;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&grid_prepare_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&grid_prepare_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return SLAM_OK;
}

// {form of the last gating (SLAM_GATE_SWEEP / SLAM_GATE_GRID), cells per axis, landmarks in the grid, tail, rebuilds,
//  queries (launches), landmarks visited, landmarks fully evaluated}; synchronises the stream
int gate_info(slam_ekf* h, int64_t out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[0] = h->gate_last;
    if (!h->grid_meta) return SLAM_OK;
    GridMeta m;
    HIP_TRY(hipMemcpyAsync(&m, h->grid_meta, sizeof(GridMeta), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    out[1] = m.G;
    out[2] = m.n_built;
    out[3] = h->N - m.n_built;
    out[4] = (int64_t)m.rebuilds;
    out[5] = (int64_t)m.queries;
    out[6] = (int64_t)m.boxed;
    out[7] = (int64_t)m.evals;
    return SLAM_OK;
}

// z_host: the caller's observations (host memory).  The sweep leaves a device copy in h->obsbuf.
int launch_gate(slam_ekf* h, int nz, const double R[4], double gate1, double gate2, const double* z_host, bool compact) {
    const int nblocks = (h->N + GATE_BLOCK - 1) / GATE_BLOCK;
    if (nblocks > h->gate_blocks_cap) {
        slam_set_error("internal: gate partial buffer too small");
        return SLAM_E_HIP;
    }
    // N2 pre-gate: needs S positive definite for every landmark, i.e. (P being a covariance) R positive definite, and
    // gate2 >= gate1 -- otherwise the plain sweep.
    const bool r_pd = R[0] > 0.0 && R[3] > 0.0 && R[0] * R[3] - 0.25 * (R[1] + R[2]) * (R[1] + R[2]) > 0.0 && R[1] == R[2];
    // Measured (tools/bench_gate.py, fp32, 64 observations): the sweep is LATENCY-bound, and the pre-gate puts the
    // covariance loads behind the test on the mean -- 8.1 against 5.9 us at N = 1k, 8.9 against 7.1 us at N = 10k; it pays
    // only where the saved gathers outweigh the longer chain: 14.7 against 15.2 us at N = 50k.  So it is used from
    // 32768 landmarks on (SLAMHIP_X bit 64: always, bit 32: never).
    const bool big = h->N >= 32768 || (h->xflags & 64);
    const int pregate = (big && r_pd && gate2 >= gate1 && gate2 < __builtin_inf() && !(h->xflags & 32)) ? 1 : 0;
    if (pregate) {
        const int rcp = ensure_pmax(h);
        if (rcp) return rcp;
    }
    const bool grid_ok = r_pd && gate2 >= gate1 && gate2 < __builtin_inf();
    if (grid_ok && (h->gate_mode == SLAM_GATE_GRID || (h->gate_mode == SLAM_GATE_AUTO && h->N >= GRID_AUTO_N)))
        return launch_gate_grid(h, nz, R, gate1, gate2, z_host, compact);
    h->gate_last = SLAM_GATE_SWEEP;
    if (h->grid_live) {                // the sweep is in use: updates stop recording displacements; a later grid query rebuilds
        h->grid_live = 0;
        h->grid_force = 1;
    }
    // observations are swept in chunks so the LDS footprint stays bounded for any nz
    constexpr int CHUNK = GATE_CHUNK;
    for (int o = 0; o < nz; o += CHUNK) {
        const int cz = nz - o < CHUNK ? nz - o : CHUNK;
        const size_t shmem = (size_t)(2 * cz + 3 * GATE_WAVES * cz) * sizeof(double);
        GateObs zarg;
        memcpy(zarg.z, z_host + 2 * (size_t)o, sizeof(double) * 2 * (size_t)cz);
        double* zc = h->obsbuf + 2 * (size_t)o;
        {
            KTimer t(h, SLAM_K_GATE);
            const bool last = o + CHUNK >= nz;
            const int ctot = (compact && last) ? nz : 0;
            if (h->dtype == SLAM_F32)
                hipLaunchKernelGGL(gate_kernel<float>, dim3(nblocks), dim3(GATE_BLOCK * OBS_WAVES), shmem, h->stream, zarg,
                                   (const float*)h->x, (const float*)h->P, h->ld, h->N, zc, cz, R[0], R[1], R[2], R[3],
                                   gate1, gate2, h->gate_part, (const double*)h->d_pmax, pregate, 7, (const float*)h->Pside, h->npad / 2,
                                   h->gate_blocks_cap, h->d_assoc + o, h->d_assoc, ctot, h->obsbuf, h->idfbuf, h->znbuf, h->d_count,
                                   h->h_assoc_dev, h->d_count + 2, h->h_flag_dev, h->obs_seq);
            else
                hipLaunchKernelGGL(gate_kernel<double>, dim3(nblocks), dim3(GATE_BLOCK * OBS_WAVES), shmem, h->stream, zarg,
                                   (const double*)h->x, (const double*)h->P, h->ld, h->N, zc, cz, R[0], R[1], R[2], R[3],
                                   gate1, gate2, h->gate_part, (const double*)h->d_pmax, pregate, 6, (const double*)h->Pside, h->npad / 2,
                                   h->gate_blocks_cap, h->d_assoc + o, h->d_assoc, ctot, h->obsbuf, h->idfbuf, h->znbuf, h->d_count,
                                   h->h_assoc_dev, h->d_count + 2, h->h_flag_dev, h->obs_seq);
        }
        HIP_TRY(hipGetLastError());
    }
    return SLAM_OK;
}

static int launch_single(slam_ekf* h, int j0, double z0, double z1, const double R[4], int mode) {
    if (h->dtype == SLAM_F32)
        hipLaunchKernelGGL(single_pair_kernel<float>, dim3(1), dim3(64), 0, h->stream, (const float*)h->x,
                           (const float*)h->P, h->ld, j0, z0, z1, R[0], R[1], R[2], R[3], mode, h->d_small, 7);
    else
        hipLaunchKernelGGL(single_pair_kernel<double>, dim3(1), dim3(64), 0, h->stream, (const double*)h->x,
                           (const double*)h->P, h->ld, j0, z0, z1, R[0], R[1], R[2], R[3], mode, h->d_small, 6);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

int launch_nis(slam_ekf* h, const double z1[2], int j, const double R[4]) {
    return launch_single(h, j - 1, z1[0], z1[1], R, 0);
}

int launch_obs_model(slam_ekf* h, int j) {
    const double R[4] = {0, 0, 0, 0};
    return launch_single(h, j - 1, 0.0, 0.0, R, 1);
}
