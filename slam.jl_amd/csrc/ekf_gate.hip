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

// The observations of one launch travel IN the kernel arguments (2 KB of the 4 KB a launch may carry): every workgroup
// needs all of them at its start, and reading them from the pinned host page the caller's z was staged in meant a few
// thousand 64-byte reads across PCIe per sweep, on the critical path of a 9 us kernel.  First parameter: the kernel
// indexes the argument segment itself (a dynamic index into the by-value copy would move the struct to scratch).
constexpr int GATE_CHUNK = 128;        // observations per launch
struct GateObs {
    double z[2 * GATE_CHUNK];
};

template <typename T>
__global__ __launch_bounds__(GATE_BLOCK * OBS_WAVES) void gate_kernel(GateObs zarg, const T* __restrict__ x, const T* __restrict__ P,
                                                           int ld, int N, double* __restrict__ zdev, int nz, double R0,
                                                           double R1, double R2, double R3, double gate1, double gate2,
                                                           double* __restrict__ part, const double* __restrict__ pmax_ptr,
                                                           int pregate, int tlog, const T* __restrict__ side, int side_n) {
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
        if (blockIdx.x == 0) zdev[i] = v;            // the device copy the compaction, the update and add_features read
    }

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
    // per-observation result of this wave, default "nothing in either gate"
    for (int i = tid; i < nz; i += NT) {
        red[3 * i] = INF;
        red[3 * i + 1] = (double)0x7fffffff;
        red[3 * i + 2] = 0.0;
    }
    __syncthreads();

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
                red[3 * i] = nd_c;
                red[3 * i + 1] = (double)j_c;
                red[3 * i + 2] = 1.0;                // near_mask != 0 or a candidate (which is also near-or-matched)
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * nz; i += NT) part[(size_t)blockIdx.x * 3 * nz + i] = red[i];
}

// observe(): turn the association vector into the update's and add_features' inputs without leaving the device.
// One wave walks the observations in order, 64 at a time: matched ones (assoc >= 1) are compacted to the front
// of zbuf/idf (zsrc may BE zbuf: a write position never passes the read position of a later chunk), new ones
// (assoc < 0) go to zn.  count = {matched, new}.  Order is the observation order, as in data-association.jl:43-47.
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
        const double z0 = i < nz ? zsrc[2 * i] : 0.0, z1 = i < nz ? zsrc[2 * i + 1] : 0.0;
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

// One wave per observation folds the per-workgroup partials into assoc[i].  With `compact_total` > 0 (observe(),
// last chunk) the LAST workgroup to finish -- an arrival counter in device memory -- then turns
// assoc[0 .. compact_total) into the update's inputs: no extra launch, and the fold stays nz-way parallel.
__global__ __launch_bounds__(64) void gate_final_kernel(
    const double* __restrict__ part, int nblocks, int nz, int32_t* assoc, int32_t* assoc_all,     // (the two alias)
    int compact_total, const double* zsrc, double* zbuf, int32_t* __restrict__ idf,
    double* __restrict__ zn, int32_t* __restrict__ count, int32_t* __restrict__ assoc_host, int32_t* arrive,
    int32_t* __restrict__ flag_host, int32_t seq) {
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const double INF = __builtin_inf();
    double nd_c = INF, j_c = (double)0x7fffffff;
    bool near = false;
    for (int b = lane; b < nblocks; b += 64) {
        const double* r = part + ((size_t)b * nz + i) * 3;
        const double r0 = r[0], r1 = r[1], r2 = r[2];
        if (r0 < nd_c || (r0 == nd_c && r1 < j_c)) { nd_c = r0; j_c = r1; }
        near = near || (r2 != 0.0);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o_nd = __shfl_xor(nd_c, off);
        const double o_j = __shfl_xor(j_c, off);
        if (o_nd < nd_c || (o_nd == nd_c && o_j < j_c)) { nd_c = o_nd; j_c = o_j; }
    }
    const bool any_near = __ballot(near) != 0ull;
    int last = 0;
    if (lane == 0) {
        int32_t a;
        if (nd_c < INF) a = (int32_t)j_c;            // jbest != 0          (:43)
        else if (!any_near) a = -1;                  // outer > gate2       (:46)
        else a = 0;                                  // dropped
        // write-through at agent scope, drained, then the arrival: the workgroup whose add comes last reads the
        // decisions with agent-scope loads (no release/acquire fence = no L2 write-back + invalidate)
        __hip_atomic_store(assoc + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (compact_total > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        }
    }
    if (compact_total > 0 && __shfl(last, 0)) {
        compact_wave(assoc_all, compact_total, zsrc, zbuf, idf, zn, count, assoc_host, lane, flag_host, seq);
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
            if (h->dtype == SLAM_F32)
                hipLaunchKernelGGL(gate_kernel<float>, dim3(nblocks), dim3(GATE_BLOCK * OBS_WAVES), shmem, h->stream, zarg,
                                   (const float*)h->x, (const float*)h->P, h->ld, h->N, zc, cz, R[0], R[1], R[2], R[3],
                                   gate1, gate2, h->gate_part, (const double*)h->d_pmax, pregate, 7, (const float*)h->Pside, h->npad / 2);
            else
                hipLaunchKernelGGL(gate_kernel<double>, dim3(nblocks), dim3(GATE_BLOCK * OBS_WAVES), shmem, h->stream, zarg,
                                   (const double*)h->x, (const double*)h->P, h->ld, h->N, zc, cz, R[0], R[1], R[2], R[3],
                                   gate1, gate2, h->gate_part, (const double*)h->d_pmax, pregate, 6, (const double*)h->Pside, h->npad / 2);
        }
        HIP_TRY(hipGetLastError());
        {
            KTimer t(h, SLAM_K_GATE_FIN);
            const bool last = o + CHUNK >= nz;
            hipLaunchKernelGGL(gate_final_kernel, dim3(cz), dim3(64), 0, h->stream, (const double*)h->gate_part, nblocks, cz,
                               h->d_assoc + o, h->d_assoc, (compact && last) ? nz : 0, h->obsbuf, h->obsbuf, h->idfbuf, h->znbuf,
                               h->d_count, h->h_assoc_dev, h->d_count + 2, h->h_flag_dev, h->obs_seq);
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
