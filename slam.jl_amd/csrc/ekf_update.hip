// ekf_update.hip -- K2..K5: the front half of update()  (src/ekf.jl:46-77).
//
//   K2/K3  pht_kernel      PHt = P*H'  from the 5 non-zero columns of each H row pair
//                          (the reference multiplies by a dense 2m x n H, :67)
//   K4     factor_kernel   S = H*PHt + RR, S = (S+S')/2, C = inv(chol(S))   (:68-70)
//                          one workgroup, double precision, register/LDS resident
//   K5     panel_gemm      W1 = PHt*C (:71);   x_update: x += W*v = PHt*(C*C'*v)   (:72,:74)
//
// The covariance down-date P -= W1*W1' (:75) is in ekf_syrk.hip.
//
// Precision: everything in this file is evaluated in DOUBLE whatever the state
// dtype.  S inherits the low-rank structure of P, so C has large entries of mixed
// sign and PHt*C cancels by one to two orders of magnitude; doing that product in
// fp32 costs ~1e-3 relative error on the updated covariance blocks.  W1 is rounded
// to the state dtype once, at the end.
//
// Panel buffers are ROW-major [npad][pitch], zero in rows >= n (never written)
// and in columns k..kp-1 (written as zeros here), so the down-date needs no
// guards on its operand loads.
#include "common.h"
#include "device_math.h"

int launch_downdate(slam_ekf* h, int kp_total, const void* X, const void* Y, int pitch, const int32_t* dcount, int joseph, int k16,
                    const void* img);   // ekf_syrk.hip

namespace {

// observe(): the host launches with an upper bound of the matched count; the kernels read the real one.
#define SLAM_DEVICE_COUNT(dcount, m, k, kp)          \
    if (dcount) {                                    \
        m = dcount[0];                               \
        k = 2 * m;                                   \
        kp = (k + SLAM_KPAD - 1) / SLAM_KPAD * SLAM_KPAD; \
    }

// ---------------------------------------------------------------------------
// K2/K3: PHt[r, 2i:2i+2] = P[r,0:3]*Hv_i' + P[r,f_i:f_i+2]*Hf_i'
// grid.x: 256-row blocks, grid.y: chunks of PHT_OBS observations
// ---------------------------------------------------------------------------
constexpr int PHT_OBS = 16;
// One entry of P H': h . (P[r, 0], P[r, 1], P[r, 2], P[r, f], P[r, f + 1]) with the roundings SPELLED OUT (one product, four
// fused multiply-adds, left to right): the four kernels that form entries of P H' must produce the same number for the same
// entry whatever the compiler would have contracted in each of them (round 5: the streamed front half keeps its operands in
// registers and is compared bit for bit with the two-launch form, which reads them back from memory).
__device__ __forceinline__ double pht_entry(double h0, double h1, double h2, double h3, double h4, double p0, double p1, double p2,
                                            double q0, double q1) {
    return __builtin_fma(h4, q1, __builtin_fma(h3, q0, __builtin_fma(h2, p2, __builtin_fma(h1, p1, h0 * p0))));
}
constexpr int HB_STRIDE = 10, HB_MAXOBS = 64;      // Jacobian blocks of the kp <= 128 path: [HB_MAXOBS][10] doubles, then HB_MAXOBS state indices

// (a device function: it runs as the workgroups 1.. of the fused factor kernel, next to the one-workgroup
//  factorisation; row0 = first row of this workgroup, by = its chunk of PHT_OBS observations)
template <typename T>
__device__ __forceinline__ void pht_body(const T* __restrict__ x, const T* __restrict__ P, int ld, int n,
                                         const int32_t* __restrict__ idf, int m, int k, int kp, double* __restrict__ PHt,
                                         int pitch, int tile_log2, const int32_t* __restrict__ dcount, int row0, int by,
                                         const double* __restrict__ hblk = nullptr) {
    __shared__ double sh[PHT_OBS][10];
    __shared__ int sf[PHT_OBS];
    SLAM_DEVICE_COUNT(dcount, m, k, kp)
    const int tid = threadIdx.x;
    const int i0 = by * PHT_OBS;
    if (2 * i0 >= kp) return;                                   // (device count: a chunk beyond the padded width)
    const int mc = (m - i0 < PHT_OBS) ? m - i0 : PHT_OBS;       // observations in this chunk (may be <= 0 for pure padding)
    if (tid < mc && hblk) {                                     // (kp <= 128: the blocks s_build_kernel left -- the same numbers the streamed form uses)
#pragma unroll
        for (int q = 0; q < 10; ++q) sh[tid][q] = hblk[HB_STRIDE * (i0 + tid) + q];
        sf[tid] = reinterpret_cast<const int*>(hblk + HB_STRIDE * HB_MAXOBS)[i0 + tid];
    } else if (tid < mc) {
        const double xv = (double)x[0], yv = (double)x[1], phi = (double)x[2];
        const int f = 3 + 2 * (idf[i0 + tid] - 1);
        const ObsModel om = obs_model(xv, yv, phi, (double)x[f], (double)x[f + 1]);
#pragma unroll
        for (int q = 0; q < 6; ++q) sh[tid][q] = om.Hv[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) sh[tid][6 + q] = om.Hf[q];
        sf[tid] = f;
    }
    __syncthreads();
    const int r = row0 + tid;
    if (r >= n) return;
    double* out = PHt + (size_t)r * pitch + 2 * i0;
    if (mc > 0) {
        const double p0 = (double)P[p_off(ld, tile_log2, r, 0)];
        const double p1 = (double)P[p_off(ld, tile_log2, r, 1)];
        const double p2 = (double)P[p_off(ld, tile_log2, r, 2)];
        // all gathers of the chunk are issued before the first use: one memory latency, not sixteen
        T q0[PHT_OBS], q1[PHT_OBS];
#pragma unroll
        for (int i = 0; i < PHT_OBS; ++i) {
            const int f = sf[i < mc ? i : 0];
            q0[i] = sym_at(P, ld, tile_log2, r, f);          // P[r, f]: from row f where (r, f) lies above the diagonal tiles
            q1[i] = sym_at(P, ld, tile_log2, r, f + 1);
        }
#pragma unroll
        for (int i = 0; i < PHT_OBS; ++i) {
            if (i < mc) {
                const double* hb = sh[i];
                out[2 * i] = pht_entry(hb[0], hb[1], hb[2], hb[6], hb[7], p0, p1, p2, (double)q0[i], (double)q1[i]);
                out[2 * i + 1] = pht_entry(hb[3], hb[4], hb[5], hb[8], hb[9], p0, p1, p2, (double)q0[i], (double)q1[i]);
            }
        }
    }
    // zero the padding columns k..kp-1 that fall into this chunk
    const int c_begin = (2 * i0 > k) ? 2 * i0 : k;
    const int c_end = (2 * (i0 + PHT_OBS) < kp) ? 2 * (i0 + PHT_OBS) : kp;
    for (int c = c_begin; c < c_end; ++c) PHt[(size_t)r * pitch + c] = 0.0;
}

// ---------------------------------------------------------------------------
// K2c: the COMPACT panel -- the 3 + 2m rows of P*H' the factorisation needs (pose rows, then each observed
// landmark's two rows), written to rows 0 .. 3+2m-1 of PHtS.  One workgroup per row, one thread per observation:
// all 2m gathers of a row are in flight at once and the rows are spread over the CUs (a few workgroups gathering
// 4000 scattered values each are limited by one CU's miss queue).  The full n-row panel is formed meanwhile on
// the second stream.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void pht_compact_kernel(const T* __restrict__ x, const T* __restrict__ P, int ld, int n,
                                                           const int32_t* __restrict__ idf, int m, int k, int kp,
                                                           double* __restrict__ PHtS, int pitch, int tile_log2,
                                                           const int32_t* __restrict__ dcount) {
    SLAM_DEVICE_COUNT(dcount, m, k, kp)
    const int slot = blockIdx.x;
    if (slot >= 3 + 2 * m) return;
    const int r = slot < 3 ? slot : 3 + 2 * (idf[(slot - 3) >> 1] - 1) + ((slot - 3) & 1);
    const double xv = (double)x[0], yv = (double)x[1], phi = (double)x[2];
    const double p0 = (double)sym_at(P, ld, tile_log2, r, 0), p1 = (double)sym_at(P, ld, tile_log2, r, 1),
                 p2 = (double)sym_at(P, ld, tile_log2, r, 2);
    double* out = PHtS + (size_t)slot * pitch;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int f = 3 + 2 * (idf[i] - 1);
        const double q0 = (double)sym_at(P, ld, tile_log2, r, f), q1 = (double)sym_at(P, ld, tile_log2, r, f + 1);
        const ObsModel om = obs_model(xv, yv, phi, (double)x[f], (double)x[f + 1]);
        out[2 * i] = pht_entry(om.Hv[0], om.Hv[1], om.Hv[2], om.Hf[0], om.Hf[1], p0, p1, p2, q0, q1);
        out[2 * i + 1] = pht_entry(om.Hv[3], om.Hv[4], om.Hv[5], om.Hf[2], om.Hf[3], p0, p1, p2, q0, q1);
    }
    for (int c = k + threadIdx.x; c < kp; c += blockDim.x) out[c] = 0.0;
}

// ---------------------------------------------------------------------------
// K2s (kp <= 128): S = H*P*H' + RR (ekf.jl:68) formed DIRECTLY, one workgroup per observation i (rows 2i, 2i + 1 of S),
// one thread per column b -- the same two steps as the compact panel followed by the factor kernel's build_S, with the
// same expressions (PHt[r][b] for the five rows r = 0, 1, 2, f_i, f_i + 1 that row pair needs, then h_i times them), but
// spread over kp/2 CUs instead of 7 us of ONE workgroup's gathers at the head of the serial factorisation.  The pose
// rows are recomputed by every workgroup (3 of its 5 rows): 2.5x the compact panel's gathers, all L2 hits.  Rows and
// columns >= k: the identity (the factor kernel used to pad in LDS).  Written to the compact panel's buffer.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void s_build_kernel(const T* __restrict__ x, const T* __restrict__ P, int ld,
                                                       const int32_t* __restrict__ idf, int m, int k, int kp,
                                                       double* __restrict__ Sg, int pitch, int tile_log2,
                                                       const int32_t* __restrict__ dcount, double R0, double R1, double R2, double R3,
                                                       double* __restrict__ hblk, unsigned* __restrict__ ready) {
    // (the word factor_w1_kernel's workgroups meet on starts every update at zero, whatever the last update left)
    if (ready && blockIdx.x == 0 && threadIdx.x == 0) ready[0] = 0u;
    SLAM_DEVICE_COUNT(dcount, m, k, kp)
    const int i = blockIdx.x;
    if (2 * i >= kp) return;
    double* row0 = Sg + (size_t)(2 * i) * pitch;
    double* row1 = row0 + pitch;
    if (i >= m) {                                                      // padding rows
        for (int b = threadIdx.x; b < kp; b += blockDim.x) {
            row0[b] = (b == 2 * i) ? 1.0 : 0.0;
            row1[b] = (b == 2 * i + 1) ? 1.0 : 0.0;
        }
        return;
    }
    const double xv = (double)x[0], yv = (double)x[1], phi = (double)x[2];
    const int fi = 3 + 2 * (idf[i] - 1);
    const ObsModel oi = obs_model(xv, yv, phi, (double)x[fi], (double)x[fi + 1]);
    if (hblk && threadIdx.x == 0) {
        // the observation's Jacobian blocks and state index, for the workgroups of factor_w1_kernel that form P H' (they must not
        // evaluate the model themselves: x is updated inside that launch)
#pragma unroll
        for (int q = 0; q < 6; ++q) hblk[HB_STRIDE * i + q] = oi.Hv[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) hblk[HB_STRIDE * i + 6 + q] = oi.Hf[q];
        reinterpret_cast<int*>(hblk + HB_STRIDE * HB_MAXOBS)[i] = fi;
    }
    const int rows[5] = {0, 1, 2, fi, fi + 1};
    for (int b = threadIdx.x; b < kp; b += blockDim.x) {
        if (b >= k) {                                                  // padding columns
            row0[b] = 0.0;
            row1[b] = 0.0;
            continue;
        }
        const int j = b >> 1, bb = b & 1;
        const int fj = 3 + 2 * (idf[j] - 1);
        const ObsModel oj = obs_model(xv, yv, phi, (double)x[fj], (double)x[fj + 1]);
        double pr[5][5];                                               // all 25 gathers before the first use
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            pr[t][0] = (double)sym_at(P, ld, tile_log2, rows[t], 0);
            pr[t][1] = (double)sym_at(P, ld, tile_log2, rows[t], 1);
            pr[t][2] = (double)sym_at(P, ld, tile_log2, rows[t], 2);
            pr[t][3] = (double)sym_at(P, ld, tile_log2, rows[t], fj);
            pr[t][4] = (double)sym_at(P, ld, tile_log2, rows[t], fj + 1);
        }
        double ph[5];                                                  // PHt[rows[t]][b]   (pht_compact_kernel's expression)
#pragma unroll
        for (int t = 0; t < 5; ++t)
            ph[t] = bb == 0 ? pht_entry(oj.Hv[0], oj.Hv[1], oj.Hv[2], oj.Hf[0], oj.Hf[1], pr[t][0], pr[t][1], pr[t][2], pr[t][3], pr[t][4])
                            : pht_entry(oj.Hv[3], oj.Hv[4], oj.Hv[5], oj.Hf[2], oj.Hf[3], pr[t][0], pr[t][1], pr[t][2], pr[t][3], pr[t][4]);
        // Round 5: S leaves this kernel SYMMETRISED, S = (S + S') * 0.5 (ekf.jl:69) -- it used to be a pass of the factorising
        // workgroup over its LDS copy, 2.8 us at the head of the serial chain.  The mirrored entry S[b][2i + ra] is formed HERE as
        // workgroup j's thread 2i + ra forms it: from the transposed 5 x 5 block of P (the same 25 values: the stored matrix is
        // symmetric entry for entry, sym_at(r, c) == sym_at(c, r)), with the same expressions in the same order.
#pragma unroll
        for (int ra = 0; ra < 2; ++ra) {
            double sv = pht_entry(oi.Hv[3 * ra + 0], oi.Hv[3 * ra + 1], oi.Hv[3 * ra + 2], oi.Hf[2 * ra + 0], oi.Hf[2 * ra + 1], ph[0], ph[1], ph[2],
                                  ph[3], ph[4]);                        // S[2i + ra][b]
            if (j == i) sv += ra ? (bb ? R3 : R1) : (bb ? R2 : R0);     // RR block = R (column-major args)
            double pt[5];                                               // PHt[rows of observation j][2i + ra]
#pragma unroll
            for (int t = 0; t < 5; ++t)
                pt[t] = pht_entry(oi.Hv[3 * ra + 0], oi.Hv[3 * ra + 1], oi.Hv[3 * ra + 2], oi.Hf[2 * ra + 0], oi.Hf[2 * ra + 1], pr[0][t], pr[1][t],
                                  pr[2][t], pr[3][t], pr[4][t]);
            // (row bb of observation j's Jacobian picked by selects: indexed with bb the whole model went to scratch)
            double st = pht_entry(bb ? oj.Hv[3] : oj.Hv[0], bb ? oj.Hv[4] : oj.Hv[1], bb ? oj.Hv[5] : oj.Hv[2], bb ? oj.Hf[2] : oj.Hf[0],
                                  bb ? oj.Hf[3] : oj.Hf[1], pt[0], pt[1], pt[2], pt[3], pt[4]);      // S[b][2i + ra]
            if (j == i) st += bb ? (ra ? R3 : R1) : (ra ? R2 : R0);
            (ra ? row1 : row0)[b] = (b == 2 * i + ra) ? sv : (sv + st) * 0.5;
        }
    }
}

// ---------------------------------------------------------------------------
// K4: one workgroup builds S, factors it and emits C = inv(chol(S)).
//
// Factorisation: in-place Gauss-Jordan elimination without pivoting on the SPD
// matrix (an LDL' factorisation): after step j the strictly-lower part of row i
// holds row i of inv(L) (unit lower), the diagonal holds D.  Then
//   chol(S) = U = sqrt(D) L'   and   C = inv(U) = inv(L)' / sqrt(D)  (upper),
// which is the reference's inv(chol(S)) (unique: upper, positive diagonal,
// C*C' = inv(S)).
//
// For kp <= 128 the matrix lives in REGISTERS during the elimination: 256
// threads form a 16 x 16 grid and thread (ty, tx) owns the elements
// (ty + 16u, tx + 16v) -- a cyclic distribution, so every thread stays busy as the
// active rows shrink.  Per step the owners publish row j and column j through a
// double-buffered LDS line, one barrier, and everyone updates its registers.
// Larger k falls back to an elimination in global memory (two barriers per step).
// ---------------------------------------------------------------------------
constexpr int FACTOR_THREADS = 512;      // 8 waves: 256 VGPRs each, so the 4 worker waves can hold 64 doubles

template <int NB>      // NB = kp / 16
__device__ __forceinline__ bool eliminate_in_registers(double* M, int mp, int k, int kp, double* rowbuf, double* colbuf) {
    // Four waves (one per SIMD) hold the matrix: thread (ty, tx) of a 16 x 16 grid owns the elements
    // (ty + 16u, tx + 16v).  More waves only multiply the per-step overhead (the pivot reciprocal, the
    // LDS reads, the predicates) that every wave pays; the other twelve waves just keep the barriers.
    const int tid = threadIdx.x;
    const bool worker = tid < 256;
    const int tx = tid & 15, ty = (tid >> 4) & 15;
    double a[NB][NB];
    if (worker) {
        // load with the symmetrisation S = (S + S')*0.5 (ekf.jl:69) folded in
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int v = 0; v < NB; ++v) {
                const int i = ty + 16 * u, c = tx + 16 * v;
                a[u][v] = (M[(size_t)i * mp + c] + M[(size_t)c * mp + i]) * 0.5;
            }
    }
    bool bad = false;
#pragma unroll
    for (int ub = 0; ub < NB; ++ub) {
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * ub + jj;
            if (j >= k || bad) break;                     // uniform
            double* rb = rowbuf + (j & 1) * kp;
            double* cb = colbuf + (j & 1) * kp;
            if (worker && ty == jj) {
#pragma unroll
                for (int v = 0; v < NB; ++v) rb[tx + 16 * v] = a[ub][v];
            }
            if (worker && tx == jj) {
#pragma unroll
                for (int u = 0; u < NB; ++u) cb[ty + 16 * u] = a[u][ub];
            }
            __syncthreads();
            const double piv = rb[j];
            if (!(piv > 0.0) || piv == __builtin_inf()) { bad = true; break; }   // uniform: same LDS word
            if (worker) {
                // 1/piv: hardware estimate + two Newton steps
                double rp = __builtin_amdgcn_rcp(piv);
                rp = __builtin_fma(rp, __builtin_fma(-piv, rp, 1.0), rp);
                rp = __builtin_fma(rp, __builtin_fma(-piv, rp, 1.0), rp);
                // every LDS operand of the step is requested up front (one latency, not one per row block)
                double rowv[NB], colv[NB];
#pragma unroll
                for (int v = 0; v < NB; ++v) rowv[v] = rb[tx + 16 * v];
#pragma unroll
                for (int u = 0; u < NB; ++u) colv[u] = cb[ty + 16 * u];
                // rows i = ty + 16u: u < ub is finished (i < j), u > ub is active, u == ub is active iff ty > jj.
                // Branch-free: an inactive row gets the multiplier 0.  Padding rows (i >= k) carry a zero in
                // column j, so their multiplier is zero by itself.
                const bool fix = tx == jj;
#pragma unroll
                for (int u = ub; u < NB; ++u) {
                    const bool act = (u > ub) || (ty > jj);
                    const double mu = act ? colv[u] * rp : 0.0;
#pragma unroll
                    for (int v = 0; v < NB; ++v) a[u][v] = __builtin_fma(-mu, rowv[v], a[u][v]);
                    a[u][ub] = (act && fix) ? -mu : a[u][ub];      // column j now holds column j of inv(L)
                }
            }
        }
    }
    __syncthreads();
    if (worker) {
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int v = 0; v < NB; ++v) M[(size_t)(ty + 16 * u) * mp + tx + 16 * v] = a[u][v];
    }
    __syncthreads();
    return !bad;
}

// ---------------------------------------------------------------------------
// Blocked elimination for kp <= 128 (same result contract as eliminate_in_registers: afterwards the strictly lower
// part of M holds inv(L), unit lower, and the diagonal holds D of S = L D L').
//
// 16 x 16 blocks, right-looking LDL' with the inverse carried along; everything but the diagonal blocks runs on
// the fp64 matrix cores (v_mfma_f64_16x16x4_f64: A[i][kk] on lane i + 16 kk, B[kk][j] on lane j + 16 kk,
// D[4r + lane/16][lane%16] in register r -- so an MFMA RESULT is, register by register, already the B operand of
// the next product).  Per block step J, TWO barriers instead of 16:
//   A. wave 0 applies step J-1's trailing update to block (J, J) and factors it in registers (one MFMA per pivot,
//      no LDS traffic, no barrier inside); meanwhile waves 1.. do the rest of step J-1's trailing update
//      A_IK -= (L_I,J-1 D_J-1) L_K,J-1' and form T_JC = sum_{K=C}^{J-1} L_JK X_KC for the inverse.
//   B. panel: L_IJ = A_IJ Linv_JJ' D_J^-1 (I > J), and the inverse's row block X_JC = -Linv_JJ T_JC (C < J).
// X_JC (J > C) is kept transposed in the unused upper block (C, J) and moved below the diagonal at the end.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

// 1/p: hardware estimate + two Newton steps
__device__ __forceinline__ double pivot_rcp(double piv) {
    double rp = __builtin_amdgcn_rcp(piv);
    rp = __builtin_fma(rp, __builtin_fma(-piv, rp, 1.0), rp);
    rp = __builtin_fma(rp, __builtin_fma(-piv, rp, 1.0), rp);
    return rp;
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), srclane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double bpermute_f64(double v, int srclane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(4 * srclane, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(4 * srclane, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Diagonal block J: Gauss-Jordan of a 16 x 16 SPD block held by ONE wave in the MFMA result layout (element
// (4r + q, c) in register r of lane c + 16 q), one v_mfma_f64_16x16x4_f64 per pivot: the rank-1 update
// a_ic -= mu_i a_jc of all 256 elements is the product of the column (-mu_i) and the row (a_jc) in the k = 0 slice.
// Row j reaches every lane with one bpermute; the multipliers use a_ji for a_ij (the active part is symmetric);
// column j gets -mu_i exactly (its C entries are zeroed and its row entry is 1).  Leaves the CLEAN unit-lower
// inv(L_JJ) in M (ones on the diagonal, zeros above), D in dvec, 1/D in dinv.  False on a non-positive pivot.
//
// Round 4: FOUR pivots per matrix-core instruction.  The sixteen sequential steps above each wait for a cross-lane row
// broadcast, a reciprocal and a dependent MFMA (~460 cycles measured per pivot; the chain is what the factorisation costs).
// The four rows of a pivot GROUP s (rows 4s .. 4s+3) are register s of the four quarter-waves, so
//   * inside the group the four pivots are eliminated with VALU work only (row t reaches the other quarter-waves with one
//     bpermute, the group's multipliers a_{j, 4s+q} with a second one issued beside it; no MFMA, no hazard wait);
//   * the rows BELOW the group then take all four pivots in ONE v_mfma_f64_16x16x4_f64 -- and need no data movement at all:
//     with R_kk the finished group row kk, the B operand B[kk][c] = R_kk[c] is register s of lane c + 16 kk AS IT STANDS
//     (diagonal replaced by 1; the group's own columns right of the diagonal by 0: those entries of the rows below are
//     overwritten by later pivots of the group), and the A operand A[i][kk] = -R_kk[i] / d_kk (the multiplier of row i at
//     pivot kk, by the symmetry of the active part) is the same register of the same lane, scaled.
// Same contract as the one-pivot form; worked out and checked against it in NumPy first (the emulation is in the round's log).
__device__ __forceinline__ bool factor_diag_block4(double* M, int mp, int J, double* dvec, double* dinv) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;
    double* blk = M + (16 * J) * mp + 16 * J;
    f64x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = blk[(4 * r + q) * mp + c];
    bool bad = false;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int j0 = 4 * s;
        double rp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + t;
            const double piv = readlane_f64(acc[s], j + 16 * t);
            bad = bad || !(piv > 0.0) || piv == __builtin_inf();
            rp[t] = pivot_rcp(piv);
            if (t < 3) {
                const double rowv = bpermute_f64(acc[s], c + 16 * t);             // a_{j, c} on every lane
                const double muv = bpermute_f64(acc[s], (j0 + q) + 16 * t);       // a_{j, 4s+q}: row q of the group at the pivot's column
                const double mu = muv * rp[t];
                if (q > t) acc[s] = (c == j) ? -mu : acc[s] - mu * rowv;
            }
        }
        if (s < 3) {
            const double rpq = q == 0 ? rp[0] : (q == 1 ? rp[1] : (q == 2 ? rp[2] : rp[3]));
            const double aval = (c > j0 + 3) ? -(acc[s] * rpq) : 0.0;              // A[i = c][kk = q]: rows below the group only
            const double bval = (c == j0 + q) ? 1.0 : ((c > j0 + q && c <= j0 + 3) ? 0.0 : acc[s]);      // B[kk = q][c]
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r > s) acc[r] = (c >= j0 && c <= j0 + 3) ? 0.0 : acc[r];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aval, bval, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * r + q;
        blk[row * mp + c] = (c < row) ? acc[r] : (c == row ? 1.0 : 0.0);
        if (c == row) {
            dvec[16 * J + row] = acc[r];
            dinv[16 * J + row] = pivot_rcp(acc[r]);
        }
    }
    return !bad;
}

__device__ __forceinline__ bool factor_diag_block(double* M, int mp, int J, double* dvec, double* dinv) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;
    double* blk = M + (16 * J) * mp + 16 * J;
    f64x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = blk[(4 * r + q) * mp + c];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int rj = j >> 2, qj = j & 3;
        const double rowv = bpermute_f64(acc[rj], c + 16 * qj);           // a_{j,c} on every lane
        const double piv = readlane_f64(acc[rj], j + 16 * qj);
        bad = bad || !(piv > 0.0) || piv == __builtin_inf();
        const double rp = pivot_rcp(piv);
        const double aval = (q == 0 && c > j) ? -(rowv * rp) : 0.0;       // A[i = c][kk = 0] = -mu_i (a_ji for a_ij)
        const double bval = (q == 0) ? (c == j ? 1.0 : rowv) : 0.0;       // B[kk = 0][c]
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + 3 > j) acc[r] = (c == j && 4 * r + q > j) ? 0.0 : acc[r];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aval, bval, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * r + q;
        blk[row * mp + c] = (c < row) ? acc[r] : (c == row ? 1.0 : 0.0);
        if (c == row) {
            dvec[16 * J + row] = acc[r];
            dinv[16 * J + row] = pivot_rcp(acc[r]);
        }
    }
    return !bad;
}

// Round 5: the inverse factor leaves the factorisation BLOCK COLUMN BY BLOCK COLUMN (factor_w1_kernel): after step J of the blocked
// elimination the block row J of inv(L) is final -- its off-diagonal blocks X_JC sit transposed in the blocks (C, J) above the
// diagonal, i.e. where C = inv(L)' / sqrt(D) has them, its diagonal block inv(L)_JJ in block (J, J) -- so C's columns 16 J .. 16 J +
// 15 can go out while the elimination works on step J + 1.  They are stored write-through (16 bytes per lane: two adjacent
// columns) and the ready word is raised once they have drained: the workgroups that form W1 = P H' C take block column after
// block column.
struct ProgC {
    double* C;               // [kp][pitch] row-major, as factor_body's emit_C leaves it
    int pitch, k;
    unsigned* ready;         // block columns out so far; PC_DONE: all of them and g; PC_FAIL: the factorisation failed
    int mode;                // (experiments build, SLAMHIP_FW1: 1 = no drain and no ready word inside the elimination, 2 = no block column leaves before its end,
                             //  16 = wave 0 stamps step 3 of the elimination into stamps[8..13] instead of the panel workgroup's stamps)
    unsigned long long* stamps;
};
constexpr unsigned PC_DONE = 15u, PC_FAIL = 31u;
constexpr unsigned long long FW1_TIMEOUT = 200000000ull;      // ticks of wall_clock64 (100 MHz): 2 s
typedef unsigned pc_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_pc __attribute__((ext_vector_type(2)));
// Block column Jc of C from the elimination's working matrix (see above), one 16 x 16 block per WAVE (waves 1 .. Jc: the blocks
// above the diagonal, wave Jc + 1 -- wave 0 when there is no such wave -- the diagonal block): the same expressions as emit_C,
// C[a][b] = inv(L)[b][a] / sqrt(D_b) for a < b, 1 / sqrt(D_b) on the diagonal, 0 below it and in the padding.  Two 16-byte
// write-through stores per lane and NO wait for them: a store drained inside the elimination's step would lengthen the step
// (measured: 1.5 us per step with one wave emitting and draining); the caller drains one step later, when it costs nothing.
__device__ __forceinline__ void emit_c_part(const double* M, int mp, int Jc, const double* dvec, const ProgC& pc, int wave, int lane,
                                            int nwaves) {
    const int blk = wave - 1;                                    // this wave's row block of the column
    const bool diag = blk == Jc || (wave == 0 && Jc + 1 >= nwaves);
    if (!diag && !(blk >= 0 && blk < Jc)) return;
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(pc.C, (short)0, 0x7fffffff, 0x00020000);
    const int b0 = 16 * Jc + 2 * (lane & 7);                     // this lane's two columns
    const double m0 = b0 < pc.k ? 1.0 / sqrt(dvec[b0]) : 0.0, m1 = b0 + 1 < pc.k ? 1.0 / sqrt(dvec[b0 + 1]) : 0.0;
    const int rblk = diag ? Jc : blk;
    double v[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                // (both rows' LDS reads before the first use)
        const int a = 16 * rblk + (lane >> 3) + 8 * u;
        if (!diag) {                                             // X_JcC, transposed in block (C, Jc)
            v[u][0] = M[a * mp + b0];
            v[u][1] = M[a * mp + b0 + 1];
        } else {                                                 // inv(L)_JcJc, unit lower
            v[u][0] = a < b0 ? M[b0 * mp + a] : (a == b0 ? 1.0 : 0.0);
            v[u][1] = a < b0 + 1 ? M[(b0 + 1) * mp + a] : (a == b0 + 1 ? 1.0 : 0.0);
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int a = 16 * rblk + (lane >> 3) + 8 * u;
        double c0, c1;
        if (!diag) {
            c0 = v[u][0] * m0;
            c1 = v[u][1] * m1;
        } else {
            c0 = a == b0 ? m0 : v[u][0] * m0;
            c1 = a == b0 + 1 ? m1 : v[u][1] * m1;
        }
        if (a >= pc.k || b0 >= pc.k) c0 = 0.0;
        if (a >= pc.k || b0 + 1 >= pc.k) c1 = 0.0;
        pc_u32x4 w4;
        const unsigned long long u0 = (unsigned long long)__double_as_longlong(c0), u1 = (unsigned long long)__double_as_longlong(c1);
        w4.x = (unsigned)u0; w4.y = (unsigned)(u0 >> 32); w4.z = (unsigned)u1; w4.w = (unsigned)(u1 >> 32);
        __builtin_amdgcn_raw_buffer_store_b128(w4, rsc, (unsigned)(((size_t)a * pc.pitch + b0) * 8), 0u, 16);      // (aux 16 = sc1: write-through)
    }
}

__device__ __forceinline__ bool eliminate_blocked(double* M, int mp, int k, int kp, double* dvec, double* dinv, double* flag,
                                                  bool four = true, const ProgC pc = ProgC{nullptr, 0, 0, nullptr, 0, nullptr}) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int nbk = (k + 15) >> 4;                    // blocks that hold real rows; the rest is identity padding
    // (S arrives symmetrised: s_build_kernel forms S = (S + S')/2, ekf.jl:69, entry by entry.)  D = 1 on the padding
    const int tc = tid & 31, tr = tid >> 5, trs = nt >> 5;
    for (int j = tid; j < kp; j += nt) { dvec[j] = 1.0; dinv[j] = 1.0; }
    if (tid == 0) flag[0] = 0.0;
    __syncthreads();
    // trailing update of step Jp on the lower blocks e = e0, e0 + stride, ... (linear index into the lower triangle
    // of the (nbk-1-Jp)^2 trailing blocks, row by row): A_IK -= (L_IJp D_Jp) L_KJp'.  Up to four blocks at a time with
    // all their LDS operands requested before the first MFMA.
    auto trailing = [&](int Jp, int e0, int stride, int e_end) {
        for (int first = e0; first < e_end; first += 4 * stride) {
            f64x4 acc[4];
            double av[4][4], bv[4][4];
            int bi[4], bk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = first + q * stride;
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= e) ++I;
                const int K = e - I * (I + 1) / 2;
                bi[q] = e < e_end ? Jp + 1 + I : -1;
                bk[q] = Jp + 1 + K;
                if (bi[q] < 0) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] = M[(16 * bi[q] + 4 * r + kk) * mp + 16 * bk[q] + li];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kx = 4 * s + kk;
                    av[q][s] = -M[(16 * bi[q] + li) * mp + 16 * Jp + kx] * dvec[16 * Jp + kx];
                    bv[q][s] = M[(16 * bk[q] + li) * mp + 16 * Jp + kx];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (bi[q] < 0) continue;
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][s], bv[q][s], acc[q], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) M[(16 * bi[q] + 4 * r + kk) * mp + 16 * bk[q] + li] = acc[q][r];
            }
        }
    };
#define ESTAMP(i)                                                                                             \
    do {                                                                                                      \
        if ((pc.mode & 16) && pc.stamps && J == 3 && tid == 0) pc.stamps[8 + (i)] = wall_clock64();           \
    } while (0)
    if ((pc.mode & 16) && pc.stamps && tid == 0) pc.stamps[14] = wall_clock64();      // (the loop starts)
    for (int J = 0; J < nbk; ++J) {
        ESTAMP(0);
        if (pc.ready && J > 0 && !(pc.mode & 2)) {
            // the block column the previous step completed goes out (the stores of the one before have had a step to drain)
            if (!(pc.mode & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave > 0) emit_c_part(M, mp, J - 1, dvec, pc, wave, lane, nwaves);
        }
        // ---- [A] wave 0: the trailing update of block (J, J) from step J-1, then its factorisation.
        //      waves 1..: the rest of step J-1's trailing update, then T_JC = sum_{K=C}^{J-1} L_JK X_KC.
        f64x4 T = {0.0, 0.0, 0.0, 0.0};
        const int C = wave - 1;                       // this wave's block column of the inverse (waves 1..J)
        const int tp = nbk - J;                       // trailing block rows of step J-1
        if (wave == 0) {
            if (J > 0) trailing(J - 1, 0, 1, 1);
            ESTAMP(1);
            const bool okb = four ? factor_diag_block4(M, mp, J, dvec, dinv) : factor_diag_block(M, mp, J, dvec, dinv);
            if (!okb && lane == 0) flag[0] = 1.0;
            ESTAMP(2);
        } else {
            if (J > 0) trailing(J - 1, wave, nwaves - 1, tp * (tp + 1) / 2);
            if (C < J) {
                for (int K = C; K < J; ++K) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int kx = 4 * s + kk;
                        const double av = M[(16 * J + li) * mp + 16 * K + kx];                       // L_JK[li][kx]
                        const double bv = (K == C) ? M[(16 * C + kx) * mp + 16 * C + li]             // Linv_CC[kx][li]
                                                   : M[(16 * C + li) * mp + 16 * K + kx];            // X_KC[kx][li]
                        T = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, T, 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        ESTAMP(3);
        // (every wave has drained its stores of block column J - 2 before the barrier: the columns 0 .. J - 2 are out)
        if (pc.ready && !(pc.mode & 3) && J >= 2 && tid == nt - 1 && flag[0] == 0.0) __hip_atomic_store(pc.ready, (unsigned)(J - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- [B] (waves 1..nbk-1: at most 7 tasks, nbk <= 8)
        if (wave >= 1 && C < J) {                     // X_JC = -Linv_JJ * T_JC, stored transposed in block (C, J)
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(16 * J + li) * mp + 16 * J + 4 * s + kk], T[s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(16 * C + li) * mp + 16 * J + 4 * r + kk] = -acc[r];
        } else if (wave > J && wave < nbk) {          // L_IJ = A_IJ * Linv_JJ' * D_J^-1, I = wave
            const int I = wave;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kx = 4 * s + kk;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(16 * I + li) * mp + 16 * J + kx],
                                                           M[(16 * J + li) * mp + 16 * J + kx], acc, 0, 0, 0);
            }
            const double di = dinv[16 * J + li];
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(16 * I + 4 * r + kk) * mp + 16 * J + li] = acc[r] * di;
        }
        ESTAMP(4);
        __syncthreads();
        ESTAMP(5);
    }
#undef ESTAMP
    if ((pc.mode & 16) && pc.stamps && tid == 0) pc.stamps[15] = wall_clock64();      // (the loop has ended)
    if (pc.ready) {
        // Streamed form: the last block column that holds real columns goes out (the padding's are zero: the readers do not wait for
        // them) and its drain is the one that is paid for.  The inverse's off-diagonal blocks STAY above the diagonal, transposed
        // (the caller's y / g products read them there: no pass over the matrix between the last pivot and the ready word); D goes
        // on the diagonal -- the emission reads only strictly lower entries of the diagonal blocks.
        if (flag[0] == 0.0) {
            if (pc.mode & 2)
                for (int Jc = 0; Jc < nbk - 1; ++Jc) emit_c_part(M, mp, Jc, dvec, pc, wave, lane, nwaves);
            emit_c_part(M, mp, nbk - 1, dvec, pc, wave, lane, nwaves);
        }
        for (int j = tid; j < kp; j += nt) M[(size_t)j * mp + j] = dvec[j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool good = flag[0] == 0.0;
        if (good && tid == 0) __hip_atomic_store(pc.ready, (unsigned)nbk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return good;
    }
    // the inverse's off-diagonal blocks move below the diagonal (over L, no longer needed); D goes on the diagonal
    for (int b = tr; b < kp; b += trs) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = tc + 32 * u;
            const bool low = a < kp && (b >> 4) > (a >> 4);
            v[u] = (low && (b >> 4) < nbk) ? M[(size_t)a * mp + b] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = tc + 32 * u;
            if (a < kp && (b >> 4) > (a >> 4)) M[(size_t)b * mp + a] = v[u];
        }
    }
    for (int j = tid; j < kp; j += nt) M[(size_t)j * mp + j] = dvec[j];
    __syncthreads();
    return flag[0] == 0.0;
}

__device__ inline bool eliminate_in_memory(double* M, int mp, int k, double* mvec) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = 0; j < k; ++j) {
        const double piv = M[(size_t)j * mp + j];
        if (!(piv > 0.0) || piv == __builtin_inf()) return false;      // uniform
        const double rp = 1.0 / piv;
        for (int i = j + 1 + tid; i < k; i += nt) mvec[i] = M[(size_t)i * mp + j] * rp;
        __syncthreads();
        const int rows = k - j - 1;
        for (int idx = tid; idx < rows * k; idx += nt) {
            const int ii = idx / k;
            const int c = idx - ii * k;
            const int i = j + 1 + ii;
            const double mi = mvec[i];
            if (c == j) M[(size_t)i * mp + j] = -mi;
            else M[(size_t)i * mp + c] -= mi * M[(size_t)j * mp + c];
        }
        __syncthreads();
    }
    return true;
}

// INLDS = true: M lives in LDS and is addressed as LDS (ds_* instructions).  A single kernel that picked
// "LDS or global" at run time made every access a FLAT instruction -- 3x slower per elimination step.
template <typename T, bool INLDS>
__device__ __forceinline__ void factor_body(
    const T* __restrict__ x, const double* __restrict__ PHt, int pht_pitch, const double* __restrict__ z,
    const int32_t* __restrict__ idf, int m, int k, int kp, double R0, double R1, double R2, double R3,
    double* __restrict__ Cout, int c_pitch, double* __restrict__ gvec, double* __restrict__ Sout, int want_sinv,
    double* __restrict__ Mglobal, int32_t* __restrict__ status, unsigned long long* __restrict__ stamps,
    const int32_t* __restrict__ dcount, int blocked, unsigned* __restrict__ ready = nullptr, int mode = 0) {
#define STAMP(i)                                                  \
    do {                                                          \
        if (stamps && threadIdx.x == 0) stamps[i] = wall_clock64(); \
    } while (0)
    extern __shared__ double lds[];
    STAMP(0);
    SLAM_DEVICE_COUNT(dcount, m, k, kp)
    if (m == 0) {                                            // (device count) nothing matched: the update is the identity
        if (threadIdx.x == 0) status[0] = 0;
        return;
    }
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const int mp = kp + 1;                                   // odd pitch: conflict-free column walks
    double* M;
    double* aux;
    if constexpr (INLDS) {
        M = lds;
        aux = lds + (size_t)kp * mp;
    } else {
        M = Mglobal;
        aux = lds;
    }
    double* mvec = aux;                                      // [kp]  multipliers, later 1/sqrt(D)
    double* vvec = aux + kp;                                 // [kp]  innovation
    double* yvec = aux + 2 * kp;                             // [kp]
    double* rowbuf = aux + 3 * kp;                           // [2][kp]
    double* colbuf = aux + 5 * kp;                           // [2][kp]
    double* hb = aux + 7 * kp;                               // [m][10]
    int* sf = reinterpret_cast<int*>(hb + (size_t)10 * m);   // [m]

    if (tid == 0) status[0] = 0;
    // innovation and Jacobian blocks (ekf.jl:55-61)
    const double xv = (double)x[0], yv = (double)x[1], phi = (double)x[2];
    for (int i = tid; i < m; i += nt) {
        const int f = 3 + 2 * (idf[i] - 1);
        const ObsModel om = obs_model(xv, yv, phi, (double)x[f], (double)x[f + 1]);
        for (int q = 0; q < 6; ++q) hb[10 * i + q] = om.Hv[q];
        for (int q = 0; q < 4; ++q) hb[10 * i + 6 + q] = om.Hf[q];
        sf[i] = f;
        vvec[2 * i] = z[2 * i] - om.zp[0];
        vvec[2 * i + 1] = mpi_to_pi_d(z[2 * i + 1] - om.zp[1]);
    }
    for (int a = k + tid; a < kp; a += nt) vvec[a] = 0.0;
    __syncthreads();
    STAMP(1);

    // S = H*PHt + RR (ekf.jl:68); rows/cols >= k are padded with the identity
    if constexpr (INLDS) {
        // kp <= 128: S was formed by s_build_kernel (kp/2 workgroups) and sits where the compact panel used to: one
        // coalesced copy into LDS, every load in flight before the first LDS store
        const int a0 = tid >> 5, bcol = tid & 31;                     // rows a0 + 16 u, columns bcol + 32 v
        for (int u0 = 0; u0 < kp; u0 += 8 * (nt >> 5)) {              // (one pass at 512 threads: 32 loads per thread)
            double v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int a = u0 + a0 + u * (nt >> 5), b = bcol + 32 * w;
                    v[u][w] = (a < kp && b < kp) ? PHt[(size_t)a * pht_pitch + b] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int a = u0 + a0 + u * (nt >> 5), b = bcol + 32 * w;
                    if (a < kp && b < kp) M[(size_t)a * mp + b] = v[u][w];
                }
        }
    } else {
    for (int a = tid >> 5; a < kp; a += nt >> 5)      // 32 consecutive columns per row visit: no integer division
        for (int b = tid & 31; b < kp; b += 32) {
        double s;
        if (a < k && b < k) {
            const int i = a >> 1, ra = a & 1;
            const double* h = hb + 10 * i;
            s = h[3 * ra + 0] * PHt[(size_t)0 * pht_pitch + b] + h[3 * ra + 1] * PHt[(size_t)1 * pht_pitch + b] +
                h[3 * ra + 2] * PHt[(size_t)2 * pht_pitch + b] + h[6 + 2 * ra + 0] * PHt[(size_t)(3 + 2 * i) * pht_pitch + b] +
                h[6 + 2 * ra + 1] * PHt[(size_t)(4 + 2 * i) * pht_pitch + b];
            if ((b >> 1) == i) s += ra ? ((b & 1) ? R3 : R1) : ((b & 1) ? R2 : R0);     // RR block = R (column-major args)
        } else {
            s = (a == b) ? 1.0 : 0.0;
        }
        M[(size_t)a * mp + b] = s;
    }
    }
    __syncthreads();
    STAMP(2);
    // S = (S + S')*0.5 (ekf.jl:69): folded into the register load of the elimination (LDS path)
    if constexpr (!INLDS) {
        for (int idx = tid; idx < k * k; idx += nt) {
            const int a = idx / k, b = idx - a * k;
            if (a < b) {
                const double s = (M[(size_t)a * mp + b] + M[(size_t)b * mp + a]) * 0.5;
                M[(size_t)a * mp + b] = s;
                M[(size_t)b * mp + a] = s;
            }
        }
        __syncthreads();
    }
    if (Sout) {
        for (int a = tid >> 5; a < kp; a += nt >> 5)      // 32 consecutive columns per row visit: no integer division
            for (int b = tid & 31; b < kp; b += 32) {
            Sout[(size_t)a * c_pitch + b] =
                (a < k && b < k) ? (M[(size_t)a * mp + b] + M[(size_t)b * mp + a]) * 0.5 : 0.0;
        }
        __syncthreads();
    }
    STAMP(3);
    bool ok;
    if constexpr (!INLDS) ok = eliminate_in_memory(M, mp, k, mvec);
    else if (blocked) {
        const ProgC pcs{Cout, c_pitch, k, ready, mode, stamps};      // (ready: C leaves block column by block column, factor_w1_kernel)
        ok = eliminate_blocked(M, mp, k, kp, rowbuf, rowbuf + kp, colbuf, blocked != 2, pcs);      // (2: one pivot per MFMA, experiments build)
    }
    else if (kp == 32) ok = eliminate_in_registers<2>(M, mp, k, kp, rowbuf, colbuf);
    else if (kp == 64) ok = eliminate_in_registers<4>(M, mp, k, kp, rowbuf, colbuf);
    else if (kp == 96) ok = eliminate_in_registers<6>(M, mp, k, kp, rowbuf, colbuf);
    else ok = eliminate_in_registers<8>(M, mp, k, kp, rowbuf, colbuf);
    if (!ok) {
        if (tid == 0) {
            status[0] = 1; status[1] = 1;                   // [1] is sticky until slam_ekf_sync reads it
            if (ready) __hip_atomic_store(ready, PC_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    STAMP(4);
    // mvec <- 1/sqrt(D)
    for (int b = tid; b < kp; b += nt) mvec[b] = (b < k) ? 1.0 / sqrt(M[(size_t)b * mp + b]) : 0.0;
    __syncthreads();
    // (streamed form: inv(L)[b][a] of an earlier block column a sits transposed above the diagonal -- the same numbers in the same order)
    const bool up = ready != nullptr;
    // y = C'*v :  y[b] = (v[b] + sum_{a<b} Linv[b][a] v[a]) / sqrt(D_b);  8 lanes per row
    for (int b0 = 0; b0 < kp; b0 += nt / 8) {
        const int b = b0 + (tid >> 3), part = tid & 7;
        double s = 0.0;
        if (b < k)       // (all 16 reads of a lane's share requested together, masked ones included: 4.8 against 3.4 us -- dropped)
            for (int a = part; a < b; a += 8) s += ((up && (a >> 4) < (b >> 4)) ? M[(size_t)a * mp + b] : M[(size_t)b * mp + a]) * vvec[a];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (part == 0 && b < kp) yvec[b] = (b < k) ? (s + vvec[b]) * mvec[b] : 0.0;
    }
    __syncthreads();
    // g = C*y = inv(S)*v  (x += PHt*g  ==  x += W*v, ekf.jl:72,74)
    for (int a0 = 0; a0 < kp; a0 += nt / 8) {
        const int a = a0 + (tid >> 3), part = tid & 7;
        double s = 0.0;
        if (a < k)
            for (int b = a + 1 + part; b < k; b += 8) s += ((up && (a >> 4) < (b >> 4)) ? M[(size_t)a * mp + b] : M[(size_t)b * mp + a]) * mvec[b] * yvec[b];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (part == 0 && a < kp) {
            const double gv = (a < k) ? s + yvec[a] * mvec[a] : 0.0;
            if (ready) __hip_atomic_store(gvec + a, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (write-through: read by other XCDs inside this launch)
            else gvec[a] = gv;
        }
    }
    STAMP(5);
    if (ready) {                                           // C went out during the elimination; g is out now
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ready, PC_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (!want_sinv) {
        // C[a][b] = Linv[b][a]/sqrt(D_b) (a<b), 1/sqrt(D_b) (a==b), 0 below and in the padding.  Consecutive threads:
        // consecutive b (coalesced store, conflict-free LDS column walk thanks to the odd pitch); four rows' worth of
        // LDS reads are requested before the first is used (the loop used to wait for every element's read in turn)
        if (kp < 96) {                                    // small k: the plain loop (a 4 x 4 batch would be mostly idle)
            for (int a = tid >> 5; a < kp; a += nt >> 5)
                for (int b = tid & 31; b < kp; b += 32) {
                    double c = 0.0;
                    if (a < k && b < k) {
                        if (a == b) c = mvec[b];
                        else if (a < b) c = M[(size_t)b * mp + a] * mvec[b];
                    }
                    Cout[(size_t)a * c_pitch + b] = c;
                }
        } else
        for (int bb = 0; bb < kp; bb += 128)              // (kp <= 128 on the LDS path: one pass)
        for (int a0 = tid >> 5; a0 < kp; a0 += 4 * (nt >> 5)) {
            double v[4][4], mv[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int b = bb + (tid & 31) + 32 * w;
                mv[w] = b < kp ? mvec[b] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int a = a0 + u * (nt >> 5);
                    v[u][w] = (a < b && b < k) ? M[(size_t)b * mp + a] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a0 + u * (nt >> 5);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int b = bb + (tid & 31) + 32 * w;
                    double c = 0.0;
                    if (a < k && b < k) c = a == b ? mv[w] : (a < b ? v[u][w] * mv[w] : 0.0);
                    if (a < kp && b < kp) Cout[(size_t)a * c_pitch + b] = c;
                }
            }
        }
    } else {
        // inv(S) = C*C' :  Sinv[a][b] = sum_{c >= max(a,b)} C[a][c] C[b][c]   (Joseph form needs K = PHt*inv(S))
        for (int a = tid >> 5; a < kp; a += nt >> 5)      // 32 consecutive columns per row visit: no integer division
            for (int b = tid & 31; b < kp; b += 32) {
            double s = 0.0;
            if (a < k && b < k) {
                const int c0 = a > b ? a : b;
                for (int c = c0; c < k; ++c) {
                    const double ca = (c == a) ? 1.0 : M[(size_t)c * mp + a];
                    const double cb = (c == b) ? 1.0 : M[(size_t)c * mp + b];
                    s += ca * cb * mvec[c] * mvec[c];
                }
            }
            Cout[(size_t)a * c_pitch + b] = s;
        }
    }
    STAMP(6);
#undef STAMP
}

// The launched kernel: workgroup 0 factors S (it needs only the COMPACT panel formed just before), workgroups 1.. form
// the FULL n-row panel P*H' meanwhile -- which only W1 = PHt*C and x += PHt*g consume -- on the CUs the one-workgroup
// factorisation leaves idle.  One launch, one stream: a fork/join over two streams cost two event hops (~12 us) and
// was measured.  (Every workgroup is given the factorisation's LDS, so the panel workgroups run one per CU.)
template <typename T, bool INLDS>
__global__ __launch_bounds__(FACTOR_THREADS) void factor_kernel(
    const T* __restrict__ x, const double* __restrict__ PHtS, int pht_pitch, const double* __restrict__ z,
    const int32_t* __restrict__ idf, int m, int k, int kp, double R0, double R1, double R2, double R3,
    double* __restrict__ Cout, int c_pitch, double* __restrict__ gvec, double* __restrict__ Sout, int want_sinv,
    double* __restrict__ Mglobal, int32_t* __restrict__ status, unsigned long long* __restrict__ stamps,
    const int32_t* __restrict__ dcount, int blocked, const T* __restrict__ P, int ld, int n, double* __restrict__ PHt,
    int tile_log2, int nbx, const double* __restrict__ hblk) {
    if (blockIdx.x == 0) {
        factor_body<T, INLDS>(x, PHtS, pht_pitch, z, idf, m, k, kp, R0, R1, R2, R3, Cout, c_pitch, gvec, Sout, want_sinv, Mglobal,
                              status, stamps, dcount, blocked);
    } else {
        const int b = blockIdx.x - 1;
        pht_body<T>(x, P, ld, n, idf, m, k, kp, PHt, pht_pitch, tile_log2, dcount, (b % nbx) * FACTOR_THREADS, b / nbx, hblk);
    }
}

// ---------------------------------------------------------------------------
// K5: OUT = beta*ADD + alpha*(IN * MAT)   (n x kp) = (n x kp)(kp x kp), in double.
// 64 rows x 64 columns per block, 2 x 8 outputs per thread (rows tr + 32u so the
// LDS column walk is conflict-free).  The inner dimension is staged through LDS in
// chunks of 32 with the NEXT chunk's global loads in flight during the FMAs.
// `upper` skips the structurally zero part of an upper-triangular MAT.  The
// result is written in the state dtype to OUT1 (and OUT2), optionally also in
// double to OUTD.
// ---------------------------------------------------------------------------
constexpr int PG_ROWS = 64;
constexpr int PG_COLS = 64;
constexpr int PG_KC = 32;

template <typename TO>
__global__ __launch_bounds__(256) void panel_gemm_kernel(const double* __restrict__ IN, int in_pitch,
                                                          const double* __restrict__ MAT, int mat_pitch, int kp, int upper,
                                                          double alpha, const double* __restrict__ ADD, int add_pitch,
                                                          double beta, TO* __restrict__ OUT1, int out1_pitch, int out1_col,
                                                          TO* __restrict__ OUT2, int out2_pitch, int out2_col,
                                                          double* __restrict__ OUTD, int outd_pitch,
                                                          const int32_t* __restrict__ status,
                                                          const int32_t* __restrict__ dcount) {
    if (status[0] != 0) return;
    if (dcount) {                                            // out*_col is 0 or the (host upper bound of) kp
        const int kp_host = kp;
        int m = 0, k = 0;
        SLAM_DEVICE_COUNT(dcount, m, k, kp)
        if (out1_col == kp_host) out1_col = kp;
        if (out2_col == kp_host) out2_col = kp;
        if ((int)(blockIdx.y * PG_COLS) >= kp) return;
    }
    __shared__ double sIn[PG_ROWS][PG_KC + 1];
    __shared__ __attribute__((aligned(16))) double sMat[PG_KC][PG_COLS];
    const int tid = threadIdx.x;
    const int tr = tid & 31;          // rows tr + 32u
    const int tc = tid >> 5;          // cols tc*8 + q
    const int row0 = blockIdx.x * PG_ROWS;
    const int b0 = blockIdx.y * PG_COLS;
    const int bw = (kp - b0 < PG_COLS) ? kp - b0 : PG_COLS;            // valid columns of this block (multiple of 32)
    double acc[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[u][q] = 0.0;
    const int a_end = upper ? (b0 + bw) : kp;
    // staging: IN chunk 64 x 32 and MAT chunk 32 x 64 = 8 + 8 doubles per thread
    double gi[8], gm[8];
    const int irow = tid >> 5, icol = tid & 31;        // + 8*s rows
    const int mrow = tid >> 6, mcol = tid & 63;        // + 4*s rows
    const double* isrc = IN + (size_t)(row0 + irow) * in_pitch + icol;
    const double* msrc = MAT + (size_t)mrow * mat_pitch + b0 + mcol;
    const bool mvalid = mcol < bw;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        gi[s] = isrc[(size_t)(8 * s) * in_pitch];
        gm[s] = mvalid ? msrc[(size_t)(4 * s) * mat_pitch] : 0.0;
    }
    for (int a0 = 0; a0 < a_end; a0 += PG_KC) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            sIn[irow + 8 * s][icol] = gi[s];
            sMat[mrow + 4 * s][mcol] = gm[s];
        }
        __syncthreads();
        if (a0 + PG_KC < a_end) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                gi[s] = isrc[(size_t)(8 * s) * in_pitch + a0 + PG_KC];
                gm[s] = mvalid ? msrc[(size_t)(a0 + PG_KC + 4 * s) * mat_pitch] : 0.0;
            }
        }
#pragma unroll 8
        for (int a = 0; a < PG_KC; ++a) {
            const double p0 = sIn[tr][a], p1 = sIn[tr + 32][a];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double mv = sMat[a][tc * 8 + q];
                acc[0][q] = __builtin_fma(p0, mv, acc[0][q]);
                acc[1][q] = __builtin_fma(p1, mv, acc[1][q]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const size_t row = (size_t)(row0 + tr + 32 * u);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = b0 + tc * 8 + q;
            if (b >= kp) continue;
            double v = alpha * acc[u][q];
            if (ADD) v += beta * ADD[row * add_pitch + b];
            OUT1[row * out1_pitch + out1_col + b] = (TO)v;
            if (OUT2) OUT2[row * out2_pitch + out2_col + b] = (TO)v;
            if (OUTD) OUTD[row * outd_pitch + b] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// K5 (reference form), kp <= 128:  W1 = PHt*C and x += PHt*g in ONE kernel on the fp64 matrix cores.
//
// v_mfma_f64_16x16x4_f64 (exact fp64 FMAs; layout probed with tools/mfma_f64_layout.hip):
//   A[i][kk]: lane = i + 16*kk      B[kk][j]: lane = j + 16*kk      D[4r + lane/16][lane%16] in register r.
// A wave owns 16 rows of PHt, held in registers for the whole product (and for x += PHt*g); C sits in LDS.
// The k index inside one MFMA is only a label: lane group kk takes k = 8s + 2kk + t (t = 0, 1) so that a lane
// loads PAIRS of consecutive doubles (16-byte loads).  C is upper triangular: column block cb needs k < 16(cb+1)
// only -- 144 instead of 256 MFMAs per wave at kp = 128.  A workgroup = 8 waves = 128 rows.
// ---------------------------------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int W1_THREADS = 512;

// three-way bf16 split of an fp32 value (exact: v = h + m + l), round to nearest even at every level
__device__ __forceinline__ void split_bf16(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    typedef __bf16 bf1;
    // (the split is OF THE FLOAT -- of the number W1 holds, as the down-date's own split2 would make it.  Handed an
    //  fp64 -> fp32 conversion, the compiler folded it into the bf16 conversion in one kernel and not in another: where the float
    //  is a tie between two bf16 values the two rounded differently -- both splits exact, the down-dates one ulp apart.  Round 5.)
    asm volatile("" : "+v"(v));
    const bf1 bh = (bf1)v;
    const float r1 = v - (float)bh;
    const bf1 bm = (bf1)r1;
    const float r2 = r1 - (float)bm;
    const bf1 bl = (bf1)r2;
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}

// max that keeps a NaN (fmax drops it)
__device__ __forceinline__ double fmax_nan(double a, double b) { return (a != a || a > b) ? a : b; }

template <typename TO, int NCB>      // NCB = kp / 16
__device__ __forceinline__ void w1_mfma_body(const double* __restrict__ PHt, int pitchA, const double* __restrict__ Cmat,
                                             int pitchC, double* sC, TO* __restrict__ W1, int pitchW, TO* __restrict__ x,
                                             int n, const double* __restrict__ g, char* __restrict__ img, int img_nch,
                                             unsigned long long* __restrict__ drift, int dbg) {
    constexpr int kp = 16 * NCB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int r0 = (blockIdx.x * (W1_THREADS / 64) + wave) * 16;
    // the wave's 16 x kp block of PHt: all loads in flight at once, BEFORE the staging of C (one memory latency for both)
    f64x2 a[NCB * 2];
    const double* arow = PHt + (size_t)(r0 + i) * pitchA + 2 * kk;
#pragma unroll
    for (int s = 0; s < NCB * 2; ++s) a[s] = *reinterpret_cast<const f64x2*>(arow + 8 * s);
    // stage C (upper triangular) into LDS: up to 16 loads per thread in flight (kp is a compile-time constant here:
    // no integer division, static trip counts)
    constexpr int PER = kp * kp / W1_THREADS;          // 2, 8, 18, 32 elements per thread
    constexpr int BATCH = PER < 16 ? PER : 16;
#pragma unroll
    for (int base = 0; base < PER; base += BATCH) {
        double v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int idx = (base + u) * W1_THREADS + threadIdx.x;
            const int ar = idx / kp, c = idx % kp;
            v[u] = (base + u < PER && ar <= c) ? Cmat[(size_t)ar * pitchC + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
            if (base + u < PER) sC[(base + u) * W1_THREADS + threadIdx.x] = v[u];
    }
    f64x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
    __syncthreads();                 // C is in LDS
#pragma unroll
    for (int s = 0; s < NCB * 2; ++s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const double* brow = sC + (size_t)(8 * s + 2 * kk + t) * kp + i;     // B[kk][j = i]
#pragma unroll
            for (int cb = s / 2; cb < NCB; ++cb)                                 // C[k][c] = 0 for k > c
                acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][t], brow[16 * cb], acc[cb], 0, 0, 0);
        }
    }
    // (dbg: SLAMHIP_W1DBG of the experiments build -- 1: no image stores, 2: no W1 stores; timing only, wrong numbers)
    if (!(dbg & 2)) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            W1[(size_t)(r0 + 4 * r + kk) * pitchW + 16 * cb + i] = (TO)acc[cb][r];
    }
    if (img && !(dbg & 1)) {
        // the same panel, split into bf16 (h, m, l) and laid out as the LDS image of the split-bf16 down-date
        // (ekf_syrk.hip, "PRE-SPLIT panel"): [row block = this workgroup][chunk cb][split][row][32 B, halves swizzled]
        // (96 two-byte stores per lane; packed into 24 eight-byte stores by two DPP exchanges per quad they cost the
        //  same -- 25.6 against 24.6 us for the kernel, tools/gpu_r3v.sh: what the stores cost, 5.4 us with them switched
        //  off, is their 7.7 MB draining at the end of a short kernel, not their instruction count)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = (r0 & 127) + 4 * r + kk;                       // row inside the 128-row block
                unsigned short hh, mm, ll;
                split_bf16((float)acc[cb][r], hh, mm, ll);
                char* d = img + ((size_t)(blockIdx.x * img_nch + cb) * 3) * 4096 + rr * 32 + ((((i >> 3) ^ (rr >> 3)) & 1) * 16) + (i & 7) * 2;
                *reinterpret_cast<unsigned short*>(d) = hh;
                *reinterpret_cast<unsigned short*>(d + 4096) = mm;
                *reinterpret_cast<unsigned short*>(d + 8192) = ll;
            }
    }
    // x += PHt*g  (ekf.jl:74 with W*v = PHt*(C*C'*v)): lane (i, kk) holds a quarter of row i
    double sx = 0.0;
#pragma unroll
    for (int s = 0; s < NCB * 2; ++s) {
        const f64x2 gg = *reinterpret_cast<const f64x2*>(g + 8 * s + 2 * kk);
        sx = __builtin_fma(a[s][0], gg[0], sx);
        sx = __builtin_fma(a[s][1], gg[1], sx);
    }
    sx += __shfl_xor(sx, 16);
    sx += __shfl_xor(sx, 32);
    double moved = 0.0;
    if (kk == 0 && r0 + i < n) {
        const double xo = (double)x[r0 + i];
        const TO xn = (TO)(xo + sx);
        x[r0 + i] = xn;
        if (r0 + i >= 3) moved = fabs((double)xn - xo);        // a landmark coordinate, as stored
    }
    if (drift) {
        // the grid form of the gating (ekf_gate.hip) widens its boxes by how far the means have moved since the grid
        // was built: the largest displacement of this update (bit pattern of a non-negative double orders like the
        // integer; a NaN sorts above everything and makes the next fold rebuild).  One atomic per WORKGROUP, spread over
        // 16 words in different cache lines: one atomic per wave to one address cost this kernel 12 us at n = 20k.
        __shared__ double s_moved[W1_THREADS / 64];
        moved = fmax_nan(moved, __shfl_xor(moved, 1));
        moved = fmax_nan(moved, __shfl_xor(moved, 2));
        moved = fmax_nan(moved, __shfl_xor(moved, 4));
        moved = fmax_nan(moved, __shfl_xor(moved, 8));
        if (lane == 0) s_moved[wave] = moved;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < W1_THREADS / 64; ++w) moved = fmax_nan(moved, s_moved[w]);
            if (!(moved == 0.0)) atomicMax(drift + (blockIdx.x & 15) * SLAM_GRID_SLOTS, (unsigned long long)__double_as_longlong(moved));
        }
    }
}

template <typename TO>
__global__ __launch_bounds__(W1_THREADS) void w1_mfma_kernel(const double* __restrict__ PHt, int pitchA,
                                                             const double* __restrict__ Cmat, int pitchC, int kp,
                                                             TO* __restrict__ W1, int pitchW, TO* __restrict__ x, int n,
                                                             const double* __restrict__ g,
                                                             const int32_t* __restrict__ status,
                                                             const int32_t* __restrict__ dcount, char* __restrict__ img,
                                                             int img_nch, unsigned* __restrict__ dd_claim,
                                                             unsigned long long* __restrict__ drift, int dbg) {
    // the tile counters of the down-date that follows (its persistent grid claims tiles from them): zeroed here, one
    // launch ahead, instead of by a memset node of its own in front of the dominant kernel
    if (dd_claim && blockIdx.x == 0 && threadIdx.x < 128) dd_claim[threadIdx.x] = 0u;
    if (status[0] != 0) return;
    if (dcount) {
        int m = 0, k = 0;
        SLAM_DEVICE_COUNT(dcount, m, k, kp)
        if (m == 0) return;
    }
    extern __shared__ double sC[];                    // [kp][kp], staged inside the body
    // (the barrier is inside the body, after the wave's PHt loads have been issued)
    if (kp == 32) w1_mfma_body<TO, 2>(PHt, pitchA, Cmat, pitchC, sC, W1, pitchW, x, n, g, img, img_nch, drift, dbg);
    else if (kp == 64) w1_mfma_body<TO, 4>(PHt, pitchA, Cmat, pitchC, sC, W1, pitchW, x, n, g, img, img_nch, drift, dbg);
    else if (kp == 96) w1_mfma_body<TO, 6>(PHt, pitchA, Cmat, pitchC, sC, W1, pitchW, x, n, g, img, img_nch, drift, dbg);
    else w1_mfma_body<TO, 8>(PHt, pitchA, Cmat, pitchC, sC, W1, pitchW, x, n, g, img, img_nch, drift, dbg);
}


// ---------------------------------------------------------------------------
// Round 5 -- K4 + K2/K3 + K5 in ONE launch with C STREAMED (kp <= 128, reference form, blocked factorisation).
//
// Before: factor_kernel (workgroup 0 factors S, the others form the panel P H' and write it to memory) and then
// w1_mfma_kernel (reads the panel and C back, W1 = P H' C, x += P H' g): the second launch could not start before the
// first had ended, and all it did between its launch and its first MFMA was to wait for the panel it had just written.
// Here every WAVE of the workgroups 1.. keeps its 16 rows of P H' in REGISTERS (the MFMA A operands, formed exactly as
// pht_body forms them) and takes C block column by block column as the elimination of workgroup 0 completes them (ProgC
// above): column block cb of W1 needs the rows 0 .. 16 cb + 15 of C's column block cb and nothing else.  The full panel is
// never written.  What remains after the factorisation's last step is one block column's worth of work, not a launch.
// `wpw` waves of a workgroup work (the launch spreads the 16-row groups over all CUs; every workgroup carries the
// factorisation's LDS, so there is one per CU).
//
// Why waiting inside the launch is safe for ANY grid size: the only wait is for workgroup 0, and workgroup 0 waits for nobody.
// Workgroups go to the XCDs round-robin and every XCD places its share in index order, so workgroup 0 is the first of this
// launch that XCD 0 places: no workgroup of this launch can hold a CU of XCD 0 before it, and the panel workgroups spinning on
// the other XCDs do not keep it from a CU.  (Another stream's kernel on XCD 0 delays it, and with it everyone; it does not
// depend on this launch, so it ends.)  The ready word: block columns out (PC_DONE: g too, PC_FAIL: not positive definite);
// s_build_kernel, the launch before, zeroes it.
// The panel workgroups must not evaluate the observation model: x is updated by THIS launch (a workgroup placed late would
// see moved means).  They read the Jacobian blocks s_build_kernel left (hblk).
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename TO, int NCB>      // NCB = kp / 16
__device__ __forceinline__ void w1_stream_body(const TO* __restrict__ P, int ld, int tile_log2, int n, int npad, int m, int k,
                                               const double* __restrict__ hblk, const double* __restrict__ Cmat, int pitchC,
                                               double* lds, TO* __restrict__ W1, int pitchW, TO* __restrict__ x,
                                               const double* __restrict__ g, char* __restrict__ img, int img_nch,
                                               unsigned long long* __restrict__ drift, const unsigned* __restrict__ ready,
                                               unsigned long long* __restrict__ stamps, int wpw, int mode, int32_t* __restrict__ status) {
#define CSTAMP(i)                                                                        \
    do {                                                                                 \
        if (stamps && !(mode & 16) && blockIdx.x == 1 && threadIdx.x == 0) stamps[8 + (i)] = wall_clock64(); \
    } while (0)
    CSTAMP(0);
    double* sh = lds;                                           // [HB_MAXOBS][10], then the state indices
    const int* sfs = reinterpret_cast<const int*>(sh + HB_STRIDE * HB_MAXOBS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int rb = blockIdx.x - 1;
    const int r0 = (rb * wpw + wave) * 16;                      // this wave's 16 rows
    const bool act = wave < wpw && r0 < npad;                   // (wave-uniform; the other waves only keep the two barriers)
    for (int t = tid; t < HB_STRIDE * HB_MAXOBS + HB_MAXOBS / 2; t += FACTOR_THREADS) sh[t] = hblk[t];
    __syncthreads();
    double moved = 0.0;
    if (act) {
        // the wave's 16 x kp block of P H': lane (i, kk) holds row r0 + i, the two columns of the observations 4 s + kk
        // (pht_body's expressions, so the operands are the numbers the two-launch form read back from memory)
        f64x2 a[NCB * 2];
        {
            const int r = r0 + i;
            const bool live = r < n;
            const int rr = live ? r : 0;
            const double p0 = (double)P[p_off(ld, tile_log2, rr, 0)];
            const double p1 = (double)P[p_off(ld, tile_log2, rr, 1)];
            const double p2 = (double)P[p_off(ld, tile_log2, rr, 2)];
            TO q0[NCB * 2], q1[NCB * 2];
#pragma unroll
            for (int s = 0; s < NCB * 2; ++s) {                 // all gathers before the first use
                const int ob = 4 * s + kk;
                const int f = sfs[ob < m ? ob : 0];
                q0[s] = sym_at(P, ld, tile_log2, rr, f);
                q1[s] = sym_at(P, ld, tile_log2, rr, f + 1);
            }
#pragma unroll
            for (int s = 0; s < NCB * 2; ++s) {
                const int ob = 4 * s + kk;
                const double* hb = sh + HB_STRIDE * (ob < m ? ob : 0);
                const double v0 = pht_entry(hb[0], hb[1], hb[2], hb[6], hb[7], p0, p1, p2, (double)q0[s], (double)q1[s]);
                const double v1 = pht_entry(hb[3], hb[4], hb[5], hb[8], hb[9], p0, p1, p2, (double)q0[s], (double)q1[s]);
                double a0 = (live && ob < m) ? v0 : 0.0, a1 = (live && ob < m) ? v1 : 0.0;
                // (evaluated HERE: left alone, the compiler sinks the second column's arithmetic to its first use, the MFMAs, and keeps
                //  its seven inputs per observation alive until then -- 200 registers, spilled)
                asm volatile("" : "+v"(a0), "+v"(a1));
                a[s][0] = a0;
                a[s][1] = a1;
            }
        }
        CSTAMP(1);
        const int nbk = (k + 15) >> 4;                          // block columns with real columns; the others of W1 are zero
        const auto rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Cmat), (short)0, 0x7fffffff, 0x00020000);
        // the ready word, read by ONE lane at agent scope (sc1) until it covers `need`.  Bounded: a factorising workgroup that has not
        // published after FW1_TIMEOUT (2 s of the 100 MHz clock; the whole kernel takes 50 us) is a defect, and the wave then gives up
        // LOUDLY -- status 2, the down-date behind this launch does not run, slam_ekf_sync reports it -- instead of hanging the device.
        auto wait_for = [&](unsigned need) __attribute__((always_inline)) {
            unsigned v = 0;
            if (lane == 0) {
                const unsigned long long t0 = wall_clock64();
                while ((v = ld_agent(ready)) < need) {
                    __builtin_amdgcn_s_sleep(16);
                    if (wall_clock64() - t0 > FW1_TIMEOUT) {
                        status[0] = 2; status[1] = 2;
                        v = PC_FAIL;
                        break;
                    }
                }
            }
            return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        };
        // column block cb of W1 and of its bf16 image (layout: see w1_mfma_body)
        auto put = [&](int cb, const f64x4& v) __attribute__((always_inline)) {
            if (mode & 4) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) W1[(size_t)(r0 + 4 * r + kk) * pitchW + 16 * cb + i] = (TO)v[r];
            if (img) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = (r0 & 127) + 4 * r + kk;
                    unsigned short hh, mm, ll;
                    split_bf16((float)v[r], hh, mm, ll);
                    char* d = img + ((size_t)((r0 >> 7) * img_nch + cb) * 3) * 4096 + rr * 32 + ((((i >> 3) ^ (rr >> 3)) & 1) * 16) + (i & 7) * 2;
                    *reinterpret_cast<unsigned short*>(d) = hh;
                    *reinterpret_cast<unsigned short*>(d + 4096) = mm;
                    *reinterpret_cast<unsigned short*>(d + 8192) = ll;
                }
            }
        };
        unsigned rdy = 0;
        bool failed = false;
        f64x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
            if (cb < nbk && !failed) {                          // (wave-uniform)
                if (rdy < (unsigned)(cb + 1)) rdy = wait_for((unsigned)(cb + 1));
                failed = rdy == PC_FAIL;
                if (cb == 0) CSTAMP(2);
                if (cb == nbk - 1) CSTAMP(3);
                if (!failed) {
                    // the B operands straight from C (rows 0 .. 16 cb + 15 of block column cb: one 128-byte line per quarter-wave and
                    // load).  PLAIN loads: a line of C is written once per launch, write-through, and read only after the ready
                    // word covers it, so no cache on the way can hold an older copy -- and the CU's L1 and the XCD's L2 serve
                    // the other waves.  No LDS, no workgroup barrier: a wave whose gathers came back early does not wait for
                    // its neighbours'.
                    double bv[2 * NCB][2];
#pragma unroll
                    for (int s = 0; s < 2 * cb + 2; ++s)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const u32x2_pc w2 = __builtin_amdgcn_raw_buffer_load_b64(rsc, (unsigned)(((size_t)(8 * s + 2 * kk + t) * pitchC + 16 * cb + i) * 8), 0u, 0);
                            bv[s][t] = __longlong_as_double((long long)((unsigned long long)w2.x | ((unsigned long long)w2.y << 32)));
                        }
                    if (!(mode & 8)) {
#pragma unroll
                        for (int s = 0; s < 2 * cb + 2; ++s)    // (the order of w1_mfma_body: s, then t, ascending)
#pragma unroll
                            for (int t = 0; t < 2; ++t) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][t], bv[s][t], acc[cb], 0, 0, 0);
                    }
                }
            }
            // the stores lag one column block behind (vmcnt counts stores too: the wait for the next block column's loads then
            // covers stores that have had a step of the elimination to drain)
            if (cb > 0 && !failed) put(cb - 1, acc[cb - 1]);
        }
        if (!failed) put(NCB - 1, acc[NCB - 1]);
        CSTAMP(4);
        // g = inv(S) v follows the last block column
        if (!failed && rdy != PC_DONE) {
            rdy = wait_for(PC_DONE);
            failed = rdy == PC_FAIL;
        }
        CSTAMP(5);
        if (!failed) {
            // x += PHt*g  (ekf.jl:74 with W*v = PHt*(C*C'*v)): lane (i, kk) holds a quarter of row i
            const auto rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g), (short)0, 0x7fffffff, 0x00020000);
            double sx = 0.0;
            pc_u32x4 gw[NCB * 2];
#pragma unroll
            for (int s = 0; s < NCB * 2; ++s) gw[s] = __builtin_amdgcn_raw_buffer_load_b128(rsg, (unsigned)((8 * s + 2 * kk) * 8), 0u, 0);      // (plain loads, as for C)
#pragma unroll
            for (int s = 0; s < NCB * 2; ++s) {
                const double g0 = __longlong_as_double((long long)((unsigned long long)gw[s].x | ((unsigned long long)gw[s].y << 32)));
                const double g1 = __longlong_as_double((long long)((unsigned long long)gw[s].z | ((unsigned long long)gw[s].w << 32)));
                sx = __builtin_fma(a[s][0], g0, sx);
                sx = __builtin_fma(a[s][1], g1, sx);
            }
            sx += __shfl_xor(sx, 16);
            sx += __shfl_xor(sx, 32);
            if (kk == 0 && r0 + i < n) {
                const double xo = (double)x[r0 + i];
                const TO xn = (TO)(xo + sx);
                x[r0 + i] = xn;
                if (r0 + i >= 3) moved = fabs((double)xn - xo);
            }
        }
    }
    if (drift) {                // (see w1_mfma_body)
        __shared__ double s_moved[FACTOR_THREADS / 64];
        moved = fmax_nan(moved, __shfl_xor(moved, 1));
        moved = fmax_nan(moved, __shfl_xor(moved, 2));
        moved = fmax_nan(moved, __shfl_xor(moved, 4));
        moved = fmax_nan(moved, __shfl_xor(moved, 8));
        if (lane == 0) s_moved[wave] = moved;
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < FACTOR_THREADS / 64; ++w) moved = fmax_nan(moved, s_moved[w]);
            if (!(moved == 0.0)) atomicMax(drift + (rb & 15) * SLAM_GRID_SLOTS, (unsigned long long)__double_as_longlong(moved));
        }
    }
    CSTAMP(6);
#undef CSTAMP
}

template <typename T>
__global__ __launch_bounds__(FACTOR_THREADS) void factor_w1_kernel(
    T* __restrict__ x, const double* __restrict__ Sg, int s_pitch, const double* __restrict__ z,
    const int32_t* __restrict__ idf, int m, int k, int kp, double R0, double R1, double R2, double R3,
    double* __restrict__ Cout, int c_pitch, double* __restrict__ gvec, int32_t* __restrict__ status,
    unsigned long long* __restrict__ stamps, const int32_t* __restrict__ dcount, int blocked, const T* __restrict__ P, int ld,
    int n, int npad, int tile_log2, const double* __restrict__ hblk, T* __restrict__ W1, int pitchW, char* __restrict__ img, int img_nch,
    unsigned* __restrict__ dd_claim, unsigned long long* __restrict__ drift, unsigned* __restrict__ sync, int wpw, int mode) {
    if (blockIdx.x == 0) {
        if (mode & 32) return;      // (experiments build, SLAMHIP_FW1 bit 32: the factorising workgroup publishes NOTHING -- what the panel waves' timeout is for)
        factor_body<T, true>(x, Sg, s_pitch, z, idf, m, k, kp, R0, R1, R2, R3, Cout, c_pitch, gvec, (double*)nullptr, 0, (double*)nullptr,
                             status, stamps, dcount, blocked, sync, mode);
        return;
    }
    // (the tile counters of the down-date that follows: see w1_mfma_kernel)
    if (dd_claim && blockIdx.x == 1 && threadIdx.x < 128) dd_claim[threadIdx.x] = 0u;
    SLAM_DEVICE_COUNT(dcount, m, k, kp)
    if (m == 0) return;                                          // (workgroup 0 returns the same way: nobody waits)
    extern __shared__ double lds[];
    if (kp == 32) w1_stream_body<T, 2>(P, ld, tile_log2, n, npad, m, k, hblk, Cout, c_pitch, lds, W1, pitchW, x, gvec, img, img_nch, drift, sync, stamps, wpw, mode, status);
    else if (kp == 64) w1_stream_body<T, 4>(P, ld, tile_log2, n, npad, m, k, hblk, Cout, c_pitch, lds, W1, pitchW, x, gvec, img, img_nch, drift, sync, stamps, wpw, mode, status);
    else if (kp == 96) w1_stream_body<T, 6>(P, ld, tile_log2, n, npad, m, k, hblk, Cout, c_pitch, lds, W1, pitchW, x, gvec, img, img_nch, drift, sync, stamps, wpw, mode, status);
    else w1_stream_body<T, 8>(P, ld, tile_log2, n, npad, m, k, hblk, Cout, c_pitch, lds, W1, pitchW, x, gvec, img, img_nch, drift, sync, stamps, wpw, mode, status);
}

// x += PHt * g      (ekf.jl:74 with W*v = PHt*(C*C'*v)); 8 lanes per row
template <typename T>
__global__ __launch_bounds__(256) void x_update_kernel(T* __restrict__ x, const double* __restrict__ PHt, int pitch, int n,
                                                        int k, const double* __restrict__ g,
                                                        const int32_t* __restrict__ status,
                                                        const int32_t* __restrict__ dcount,
                                                        unsigned long long* __restrict__ drift) {
    if (status[0] != 0) return;
    if (dcount) k = 2 * dcount[0];
    const int r = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int part = threadIdx.x & 7;
    double s = 0.0;
    if (r < n) {
        const double* row = PHt + (size_t)r * pitch;
        for (int a = part; a < k; a += 8) s += row[a] * g[a];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    double moved = 0.0;
    if (part == 0 && r < n) {
        const double xo = (double)x[r];
        const T xn = (T)(xo + s);
        x[r] = xn;
        if (r >= 3) moved = fabs((double)xn - xo);
    }
    if (drift) {                // (see w1_mfma_body)
        __shared__ double s_moved[4];
        moved = fmax_nan(moved, __shfl_xor(moved, 8));
        moved = fmax_nan(moved, __shfl_xor(moved, 16));
        moved = fmax_nan(moved, __shfl_xor(moved, 32));
        if ((threadIdx.x & 63) == 0) s_moved[threadIdx.x >> 6] = moved;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) moved = fmax_nan(moved, s_moved[w]);
            if (!(moved == 0.0)) atomicMax(drift + (blockIdx.x & 15) * SLAM_GRID_SLOTS, (unsigned long long)__double_as_longlong(moved));
        }
    }
}

template <typename T>
int update_typed(slam_ekf* h, int m, const double R[4], int form, const int32_t* dcount) {
    const int n = 3 + 2 * h->N;
    const int k = 2 * m;
    const int kp = round_up(k, SLAM_KPAD);
    const int pitchA = h->kcap;         // PHt, Cmat, Smat, Kd
    const int pitchW = 2 * h->kcap;     // W1, W2
    T* x = (T*)h->x;
    T* P = (T*)h->P;
    T* W1 = (T*)h->W1;
    T* W2 = (T*)h->W2;
    const bool joseph = form == SLAM_FORM_JOSEPH;
    unsigned long long* drift = grid_drift_slot(h);      // the grid form of the gating follows the means (ekf_gate.hip)

    // K2c: the 3 + 2m rows of P*H' the factorisation needs (compact panel PHtS); the full panel is formed by the
    // fused kernel below, next to the factorisation
    const int tlog = h->dtype == SLAM_F32 ? 7 : 6;
    // round 5: factorisation, panel and W1 in one launch, C streamed (factor_w1_kernel; SLAMHIP_X bit 128: the two launches of round 4)
    const bool fused = !joseph && kp <= 128 && h->factor_blocked == 1 && !(h->xflags & 128);
    double* hblk = h->Smat;      // (free in the reference form: the observations' Jacobian blocks, HB_MAXOBS * (HB_STRIDE + 1/2) doubles <= 32 * 32,
                                 //  written by s_build_kernel for kp <= 128)
    {
        KTimer t(h, SLAM_K_PHT);
        if (kp <= 128)      // S itself, kp/2 workgroups (the factor kernel copies it into LDS)
            hipLaunchKernelGGL(s_build_kernel<T>, dim3(kp / 2), dim3(128), 0, h->stream, x, P, h->ld, h->idfbuf, m, k, kp, h->PHtS,
                               pitchA, tlog, dcount, R[0], R[1], R[2], R[3], joseph ? (double*)nullptr : hblk,
                               fused ? reinterpret_cast<unsigned*>(h->d_small + 56) : (unsigned*)nullptr);
        else                // the compact panel; the factor kernel forms S from it
            hipLaunchKernelGGL(pht_compact_kernel<T>, dim3(3 + 2 * m), dim3(128), 0, h->stream, x, P, h->ld, n, h->idfbuf, m, k, kp,
                               h->PHtS, pitchA, tlog, dcount);
    }
    HIP_TRY(hipGetLastError());
    {   // K4 + K2/K3 (full panel), one launch
        KTimer t(h, SLAM_K_FACTOR);
        const bool in_lds = kp <= 128;
        const int nbx = (n + FACTOR_THREADS - 1) / FACTOR_THREADS, nby = kp / (2 * PHT_OBS);
        const dim3 fgrid(1 + nbx * nby);
        const size_t aux = (size_t)7 * kp * sizeof(double) + (size_t)m * (10 * sizeof(double) + sizeof(int));
        const size_t shm = aux + (in_lds ? (size_t)kp * (kp + 1) * sizeof(double) : 0);
        unsigned long long* stamps = h->debug_stamps ? (unsigned long long*)(h->d_small + 40) : (unsigned long long*)nullptr;
        if (fused) {
            const bool img = h->dtype == SLAM_F32 && h->Wimg && !(h->xflags & 16);
            const size_t need = (size_t)(HB_STRIDE * HB_MAXOBS + HB_MAXOBS / 2) * sizeof(double);
            // the 16-row groups spread over the CUs: `wpw` working waves per workgroup.  Over 7/8 of the chip, not all of it:
            // at C3 (1256 groups) 6 waves on 210 CUs end at 40.1 us, 5 waves on 252 CUs at 44.1 us and slow the factorisation's own
            // workgroup by 3 us, 8 waves on 157 CUs at 44.0 us (profiles/r05_front_half_timeline.txt)
            const int groups = h->npad / 16, room = h->num_cus >= 16 ? h->num_cus * 7 / 8 : 1;
            int wpw = (groups + room - 1) / room;
            wpw = wpw < 1 ? 1 : (wpw > FACTOR_THREADS / 64 ? FACTOR_THREADS / 64 : wpw);
            wpw = slam_exp_env("SLAMHIP_FW1_WPW", wpw);
            hipLaunchKernelGGL(factor_w1_kernel<T>, dim3(1 + (groups + wpw - 1) / wpw), dim3(FACTOR_THREADS), shm > need ? shm : need, h->stream, x,
                               (const double*)h->PHtS, pitchA, (const double*)h->obsbuf, (const int32_t*)h->idfbuf, m, k, kp, R[0], R[1],
                               R[2], R[3], h->Cmat, pitchA, h->gvec, h->d_status, stamps, dcount, h->factor_blocked, (const T*)P, h->ld, n,
                               h->npad, tlog, (const double*)hblk, W1, pitchW, img ? (char*)h->Wimg : (char*)nullptr, h->kcap / 16,
                               img ? h->dd_claim : (unsigned*)nullptr, drift, reinterpret_cast<unsigned*>(h->d_small + 56), wpw,
                               slam_exp_env("SLAMHIP_FW1", 0));
        } else if (in_lds)
            hipLaunchKernelGGL((factor_kernel<T, true>), fgrid, dim3(FACTOR_THREADS), shm, h->stream, x, h->PHtS, pitchA,
                               h->obsbuf, h->idfbuf, m, k, kp, R[0], R[1], R[2], R[3], h->Cmat, pitchA, h->gvec,
                               joseph ? h->Smat : (double*)nullptr, joseph ? 1 : 0, (double*)nullptr, h->d_status, stamps, dcount, h->factor_blocked,
                               (const T*)P, h->ld, n, h->PHt, tlog, nbx, joseph ? (const double*)nullptr : (const double*)hblk);
        else
            hipLaunchKernelGGL((factor_kernel<T, false>), fgrid, dim3(FACTOR_THREADS), shm, h->stream, x, h->PHtS, pitchA,
                               h->obsbuf, h->idfbuf, m, k, kp, R[0], R[1], R[2], R[3], h->Cmat, pitchA, h->gvec,
                               joseph ? h->Smat : (double*)nullptr, joseph ? 1 : 0, h->Mwork, h->d_status, stamps, dcount, 0,
                               (const T*)P, h->ld, n, h->PHt, tlog, nbx, (const double*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    const dim3 pg_grid(h->npad / PG_ROWS, (kp + PG_COLS - 1) / PG_COLS);
    int kp_total;
    bool use_img = false;      // the W1 kernel also leaves the panel pre-split for the split-bf16 down-date (SLAMHIP_X bit 16: off)
    use_img = !joseph && kp <= 128 && h->dtype == SLAM_F32 && h->Wimg && !(h->xflags & 16);
    if (fused) {
        kp_total = round_up(k, 16);
    } else {   // K5
        KTimer t(h, SLAM_K_W1);
        if (!joseph && kp <= 128) {
            // W1 = PHt*C and x += PHt*g on the fp64 matrix cores
            hipLaunchKernelGGL(w1_mfma_kernel<T>, dim3(h->npad / 128), dim3(W1_THREADS), (size_t)kp * kp * sizeof(double),
                               h->stream, (const double*)h->PHt, pitchA, (const double*)h->Cmat, pitchA, kp, W1, pitchW, x, n,
                               (const double*)h->gvec, h->d_status, dcount, use_img ? (char*)h->Wimg : (char*)nullptr, h->kcap / 16,
                               use_img ? h->dd_claim : (unsigned*)nullptr, drift, slam_exp_env("SLAMHIP_W1DBG", 0));
            kp_total = round_up(k, 16);
        } else if (!joseph) {
            // W1 = PHt*C
            hipLaunchKernelGGL(panel_gemm_kernel<T>, pg_grid, dim3(256), 0, h->stream, h->PHt, pitchA, h->Cmat, pitchA, kp, 1,
                               1.0, (const double*)nullptr, 0, 0.0, W1, pitchW, 0, (T*)nullptr, 0, 0, (double*)nullptr, 0,
                               h->d_status, dcount);
            kp_total = round_up(k, 16);      // W1 is zero in columns k..kp-1: the down-date stops at the next multiple of 16
        } else {
            // K = PHt*inv(S)            -> W1[:, 0:kp], W2[:, kp:2kp], Kd (double)
            hipLaunchKernelGGL(panel_gemm_kernel<T>, pg_grid, dim3(256), 0, h->stream, h->PHt, pitchA, h->Cmat, pitchA, kp, 0,
                               1.0, (const double*)nullptr, 0, 0.0, W1, pitchW, 0, W2, pitchW, kp, h->Kd, pitchA,
                               h->d_status, dcount);
            // T = PHt - 0.5 * K * S     -> W1[:, kp:2kp], W2[:, 0:kp]
            hipLaunchKernelGGL(panel_gemm_kernel<T>, pg_grid, dim3(256), 0, h->stream, (const double*)h->Kd, pitchA,
                               (const double*)h->Smat, pitchA, kp, 0, -0.5, (const double*)h->PHt, pitchA, 1.0, W1, pitchW,
                               kp, W2, pitchW, 0, (double*)nullptr, 0, h->d_status, dcount);
            kp_total = 2 * kp;
        }
        if (joseph || kp > 128)
            hipLaunchKernelGGL(x_update_kernel<T>, dim3((n + 31) / 32), dim3(256), 0, h->stream, x, h->PHt, pitchA, n, k,
                               h->gvec, h->d_status, dcount, drift);
    }
    HIP_TRY(hipGetLastError());
    return launch_downdate(h, kp_total, W1, joseph ? (const void*)W2 : (const void*)W1, pitchW, dcount, joseph ? 1 : 0, round_up(k, 16),
                           use_img ? h->Wimg : nullptr);
}

}  // namespace

int launch_update(slam_ekf* h, int m, const double R[4], int form, bool device_count) {
    // the reference form only LOWERS a diagonal entry (P_ii - sum_k w_ik^2, every term non-negative as computed); the
    // Joseph form may raise one by a rounding: the pre-gate's variance bound is recomputed before the next sweep
    if (form == SLAM_FORM_JOSEPH) h->pmax_valid = 0;
    const int32_t* dcount = device_count ? h->d_count : (const int32_t*)nullptr;
    return h->dtype == SLAM_F32 ? update_typed<float>(h, m, R, form, dcount) : update_typed<double>(h, m, R, form, dcount);
}

int update_kernels_init() {
    // the factor kernel keeps a 128 x 129 double matrix in LDS: raise the dynamic-LDS cap
    const int big = 160 * 1024;
    // (the W1 kernel needs 128 KiB for C at kp = 128, plus 64 bytes of static LDS: static + dynamic <= 160 KiB)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&w1_mfma_kernel<float>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big - 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&w1_mfma_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big - 1024));
    // (the fused factor kernel also holds ~1.4 KiB of static LDS for its panel workgroups: static + dynamic <= 160 KiB)
    const int fbig = big - 4096;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_kernel<float, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, fbig));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_kernel<double, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, fbig));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_w1_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, fbig));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_w1_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, fbig));
    return SLAM_OK;
}
