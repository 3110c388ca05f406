// ekf_syrk.hip -- K6 / K6J: the covariance down-date  P -= X * Y'.
//
//   reference form   P -= W1*W1'            (src/ekf.jl:75)      X = Y = W1   (n x kp)
//   Joseph form      P -= K*T' + T*K'       (not in reference)   X = [K|T], Y = [T|K]  (n x 2kp)
//
// P is n x n column-major with leading dimension ld; X, Y are row-major
// [npad][pitch] panels, zero in rows >= n and in padding columns.  This kernel
// is where >= 95 % of an update's time goes: algorithmic traffic is one read and
// one write of P (2*n^2*sizeof(T)), algorithmic work 2*n^2*k flops.
//
// fp32: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A 256-thread
// workgroup owns a 128 x 128 tile of P; its four waves own 64 x 64 quadrants as
// 2 x 2 MFMA blocks (64 accumulator VGPRs).  The panels are staged through LDS in
// k-chunks of 32 with 16-byte loads; each lane then pulls FOUR consecutive k of
// its row with one ds_read_b128 and feeds four MFMAs -- the k index inside an
// MFMA is only a label, so lane half h takes k = kc+4h..kc+4h+3 for both operands.
// The product is accumulated from zero and subtracted from P once (the
// reference's order: form W1*W1', then subtract), so P's magnitude never enters
// the accumulation error.
//
// MFMA orientation: D[i][j] = sum_k A[i][k] B[k][j], j on lanes, i in registers.
// P is column-major, so rows of P go on the LANES (j) and columns in the
// registers (i): each accumulator register then covers 32 consecutive rows of
// one column = one full 128-byte line per half-wave for the P load and store.
//
// fp64: LDS-tiled VALU kernel (64 x 64 tile, 4 x 4 per thread); at the k used by
// BASELINE.json's fp64 configuration the down-date is HBM-bound.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = SLAM_TILE;      // 128
constexpr int KC = 32;               // k-chunk staged per barrier pair
constexpr int LDSP = KC + 4;         // LDS row pitch in floats (144 B: 16-B aligned, conflict-free b128)

__global__ __launch_bounds__(256) void downdate_f32_mfma(float* __restrict__ P, int ld, int n, const float* __restrict__ X,
                                                         const float* __restrict__ Y, int pitch, int kp,
                                                         const int32_t* __restrict__ status) {
    if (status[0] != 0) return;
    __shared__ __attribute__((aligned(16))) float sX[TILE][LDSP];   // rows of P  (lanes,  "B" operand)
    __shared__ __attribute__((aligned(16))) float sY[TILE][LDSP];   // cols of P  (regs,   "A" operand)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave & 1;          // row half of the tile
    const int wc = wave >> 1;         // column half
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int R0 = blockIdx.x * TILE;
    const int C0 = blockIdx.y * TILE;

    f32x16 acc[2][2];                 // [cb][rb]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][rb][r] = 0.0f;

    for (int kc = 0; kc < kp; kc += KC) {
        // stage 128 x 32 floats of each panel: 1024 float4 per panel, 4 per thread
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int idx = tid + 256 * s;
            const int row = idx >> 3;
            const int c4 = idx & 7;
            const f32x4 vx = *reinterpret_cast<const f32x4*>(X + (size_t)(R0 + row) * pitch + kc + 4 * c4);
            const f32x4 vy = *reinterpret_cast<const f32x4*>(Y + (size_t)(C0 + row) * pitch + kc + 4 * c4);
            *reinterpret_cast<f32x4*>(&sX[row][4 * c4]) = vx;
            *reinterpret_cast<f32x4*>(&sY[row][4 * c4]) = vy;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 8) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                a[cb] = *reinterpret_cast<const f32x4*>(&sY[64 * wc + 32 * cb + l31][kk + 4 * lh]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                b[rb] = *reinterpret_cast<const f32x4*>(&sX[64 * wr + 32 * rb + l31][kk + 4 * lh]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
                        acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][t], b[rb][t], acc[cb][rb], 0, 0, 0);
        }
        __syncthreads();
    }

    // P[row, col] -= acc ;  row on the lane, col = register-mapped
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = R0 + 64 * wr + 32 * rb + l31;
            const int colb = C0 + 64 * wc + 32 * cb + 4 * lh;
            if (row < n) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = colb + (r & 3) + 8 * (r >> 2);
                    if (col < n) {
                        float* p = P + (size_t)col * ld + row;
                        *p = *p - acc[cb][rb][r];
                    }
                }
            }
        }
}

// ---- fp64 (and generic) VALU down-date ---------------------------------------
constexpr int DT = 64;     // tile edge
constexpr int DK = 16;     // k-chunk

template <typename T>
__global__ __launch_bounds__(256) void downdate_valu(T* __restrict__ P, int ld, int n, const T* __restrict__ X,
                                                     const T* __restrict__ Y, int pitch, int kp,
                                                     const int32_t* __restrict__ status) {
    if (status[0] != 0) return;
    __shared__ T sX[DT][DK + 1];
    __shared__ T sY[DT][DK + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15;          // rows  tx + 16u
    const int ty = tid >> 4;          // cols  ty + 16v
    const int R0 = blockIdx.x * DT;
    const int C0 = blockIdx.y * DT;
    T acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = (T)0;
    for (int kc = 0; kc < kp; kc += DK) {
        // 64 x 16 elements per panel, 4 per thread, coalesced along k
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int idx = tid + 256 * s;
            const int row = idx >> 4;
            const int cc = idx & 15;
            sX[row][cc] = X[(size_t)(R0 + row) * pitch + kc + cc];
            sY[row][cc] = Y[(size_t)(C0 + row) * pitch + kc + cc];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) {
            T xr[4], yc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xr[u] = sX[tx + 16 * u][kk];
#pragma unroll
            for (int v = 0; v < 4; ++v) yc[v] = sY[ty + 16 * v][kk];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += xr[u] * yc[v];
        }
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int col = C0 + ty + 16 * v;
        if (col >= n) continue;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = R0 + tx + 16 * u;
            if (row < n) {
                T* p = P + (size_t)col * ld + row;
                *p = *p - acc[u][v];
            }
        }
    }
}

}  // namespace

int launch_downdate(slam_ekf* h, int kp_total, const void* X, const void* Y, int pitch) {
    const int n = 3 + 2 * h->N;
    KTimer t(h, SLAM_K_SYRK);
    if (h->dtype == SLAM_F32) {
        const int tiles = (n + TILE - 1) / TILE;
        hipLaunchKernelGGL(downdate_f32_mfma, dim3(tiles, tiles), dim3(256), 0, h->stream, (float*)h->P, h->ld, n,
                           (const float*)X, (const float*)Y, pitch, kp_total, h->d_status);
    } else {
        const int tiles = (n + DT - 1) / DT;
        hipLaunchKernelGGL(downdate_valu<double>, dim3(tiles, tiles), dim3(256), 0, h->stream, (double*)h->P, h->ld, n,
                           (const double*)X, (const double*)Y, pitch, kp_total, h->d_status);
    }
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}
