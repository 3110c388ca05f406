// ekf_syrk.hip -- K6 / K6J: the covariance down-date  P -= X * Y'.
//
//   reference form   P -= W1*W1'            (src/ekf.jl:75)      X = Y = W1   (n x kp)
//   Joseph form      P -= K*T' + T*K'       (not in reference)   X = [K|T], Y = [T|K]  (n x 2kp)
//
// P is n x n column-major with leading dimension ld; X, Y are row-major
// [npad][pitch] panels, zero in rows >= n and in padding columns.  This kernel
// is where >= 90 % of an update's time goes.
//
// Symmetry.  X*Y' is symmetric in both forms and P is kept EXACTLY symmetric,
// so only the tiles on and below the diagonal are computed: tile (I,J), I >= J,
// reads P(I,J) once, forms P(I,J) - X_I*Y_J', and stores it to (I,J) and,
// transposed through LDS so that both stores are full 128-byte lines, to (J,I).
// Algorithmic traffic per update: n^2/2 elements read + n^2 written; algorithmic
// work n^2*k flops (half of the full product).  Diagonal tiles store their lower
// triangle directly and their upper triangle from the mirror, which keeps P
// bit-for-bit symmetric also in the Joseph form.
//
// fp32: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A 256-thread
// workgroup owns a 128 x 128 tile; its four waves own 64 x 64 quadrants as 2 x 2
// MFMA blocks (64 accumulator registers).  The product is accumulated from zero
// and subtracted from P once (the reference's order: form W1*W1', then subtract).
// The k-loop is software pipelined: panels travel global -> registers -> LDS in
// chunks of 32 columns with two LDS buffers and ONE barrier per chunk; the loads
// of chunk c+1 and of the P tile itself are in flight while chunk c feeds the
// MFMAs.  Each lane pulls FOUR consecutive k of its row with one ds_read_b128 and
// feeds four MFMAs -- the k index inside an MFMA is only a label, so lane half h
// takes k = kc+4h..kc+4h+3 for both operands.
//
// MFMA orientation: D[i][j] = sum_k A[i][k] B[k][j], j on lanes, i in registers.
// P is column-major, so rows of P go on the LANES (j) and columns in the
// registers (i): each accumulator register then covers 32 consecutive rows of one
// column = one full 128-byte line per half-wave for the P load and store.
//
// Tile order.  Workgroups are dispatched round-robin over the 8 XCDs (observed,
// not contractual: only speed depends on it), each with a private 4 MiB L2.  The
// host builds a tile list in which XCD x walks "super-rows" of four tile rows,
// column by column, so the four row panels stay L2-resident and every column
// panel fetched is used four times.
//
// fp64: the same triangular / mirrored scheme with an LDS-tiled VALU kernel
// (64 x 64 tile, 4 x 4 per thread); at the k of BASELINE.json's fp64
// configuration the down-date is HBM-bound.
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = SLAM_TILE;      // 128
constexpr int KC = 32;               // k-chunk per barrier
constexpr int LDSP = KC + 4;         // LDS row pitch in floats (144 B: 16-B aligned, conflict-free b128)
constexpr int TP = 33;               // pitch of the per-wave 32 x 32 transpose scratch

// The transpose scratch is private to a wave; LDS executes one wave's instructions in order, so a
// compiler-level fence is all that is needed between its writes and reads.  (A __syncthreads()
// here would also drain every outstanding global store of the epilogue: vmcnt(0) per barrier.)
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// P tile <-> registers.
//
// Off-diagonal tiles need no masks at all (P is allocated in whole 128 x 128 tiles: rows/columns
// >= n are padding that stays zero because the panels are zero there), and every address is
// "wave-uniform column offset + per-lane row offset".  They go through buffer loads/stores whose
// column offset rides in an SGPR (soffset): no per-element 64-bit VALU address arithmetic, no
// branches.  A buffer descriptor covers one 128-column band of P (128*ld*4 bytes < 4 GiB).
// Diagonal tiles (1 in 80 at N = 10k) take the masked path: lower triangle stored directly,
// upper triangle from the mirror.
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int rsrc_t;

// The per-element scalar offsets (column * ld) are invariant across tiles; left alone, the compiler
// hoists all ~200 of them out of the persistent loop and spills hundreds of SGPRs.  Pretending that
// `ld` changes before each use keeps every offset a two-instruction SALU computation next to its use.
#define KEEP_SCALAR_LOCAL(v) ((void)0)
// ... which is done by re-reading `ld` through a volatile LDS word at the top of every load/store group:
__device__ __forceinline__ int fresh_ld(const volatile int* ld_lds) { return __builtin_amdgcn_readfirstlane(*ld_lds); }

__device__ inline auto band_rsrc(const float* band, int ld) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(band), (short)0, TILE * ld * 4, 0x00020000);
}

__device__ __forceinline__ void load_p_tile_fast(const float* __restrict__ P, int ld, const volatile int* ld_lds, int R0,
                                                 int C0, int wr, int wc, int l31, int lh, float (&pold)[2][2][16]) {
    const auto rs = band_rsrc(P + (size_t)C0 * ld, ld);
    const int voff = (l31 + lh * 4 * ld) * 4;                 // bytes: row on the lane, upper half-wave 4 columns on
    const int lds = fresh_ld(ld_lds);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci0 = (r & 3) + 8 * (r >> 2);
                KEEP_SCALAR_LOCAL(lds);
                const int soff = ((64 * wc + 32 * cb + ci0) * lds + R0 + 64 * wr + 32 * rb) * 4;
                pold[cb][rb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
            }
}

// Epilogue of ONE 32 x 32 sub-block q = (cb, rb) of a wave's quadrant: direct store, transpose through the
// wave's private LDS scratch, mirrored store.
template <int CB, int RB>
__device__ __forceinline__ void store_sub_fast(float* __restrict__ P, int ld, const volatile int* ld_lds, int R0, int C0, int wr,
                                               int wc, int l31, int lh, const float (&pold)[2][2][16],
                                               const f32x16 (&acc)[2][2], float* sT, int dbg) {
    const auto rs = band_rsrc(P + (size_t)C0 * ld, ld);       // direct:   columns C0.., rows R0..
    const auto rsm = band_rsrc(P + (size_t)R0 * ld, ld);      // mirrored: columns R0.., rows C0..
    const int voff = (l31 + lh * 4 * ld) * 4;
    const int voff_m = (l31 + lh * ld) * 4;                   // columns on lanes, upper half-wave 1 row on
    const int lds = fresh_ld(ld_lds);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci0 = (r & 3) + 8 * (r >> 2);
        const float val = pold[CB][RB][r] - acc[CB][RB][r];
        KEEP_SCALAR_LOCAL(lds);
        const int soff = ((64 * wc + 32 * CB + ci0) * lds + R0 + 64 * wr + 32 * RB) * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs, voff, soff, 0);
        sT[l31 * TP + 4 * lh + ci0] = val;
    }
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float val = sT[(2 * s + lh) * TP + l31];
        KEEP_SCALAR_LOCAL(lds);
        const int soff = ((64 * wr + 32 * RB + 2 * s) * lds + C0 + 64 * wc + 32 * CB) * 4;
        if (!(dbg & 1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsm, voff_m, soff, 0);
    }
    wave_lds_fence();
}

template <int CB, int RB>
__device__ __forceinline__ void store_sub_masked(float* __restrict__ P, int ld, int n, int R0, int C0, int wr, int wc, int l31, int lh,
                                        const float (&pold)[2][2][16], const f32x16 (&acc)[2][2], float* sT) {
    const int rowb = R0 + 64 * wr + 32 * RB;
    const int colb = C0 + 64 * wc + 32 * CB;
    const int row = rowb + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci = 4 * lh + (r & 3) + 8 * (r >> 2);
        const int col = colb + ci;
        const float val = pold[CB][RB][r] - acc[CB][RB][r];
        if (row < n && col < n && row >= col) P[(size_t)col * ld + row] = val;
        sT[l31 * TP + ci] = val;
    }
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int rr = 2 * s + lh;
        const float val = sT[rr * TP + l31];
        const int rowI = rowb + rr, colJ = colb + l31;
        if (rowI < n && colJ < n && rowI > colJ) P[(size_t)rowI * ld + colJ] = val;
    }
    wave_lds_fence();
}

template <int Q, bool MASKED>
__device__ __forceinline__ void store_sub(bool diag, float* __restrict__ P, int ld, const volatile int* ld_lds, int n, int R0,
                                          int C0, int wr, int wc, int l31, int lh, const float (&pold)[2][2][16],
                                          const f32x16 (&acc)[2][2], float* sT, int dbg) {
    if constexpr (MASKED) store_sub_masked<(Q >> 1), (Q & 1)>(P, ld, n, R0, C0, wr, wc, l31, lh, pold, acc, sT);
    else store_sub_fast<(Q >> 1), (Q & 1)>(P, ld, ld_lds, R0, C0, wr, wc, l31, lh, pold, acc, sT, dbg);
}

__device__ __forceinline__ void load_p_tile_masked(const float* __restrict__ P, int ld, int n, int R0, int C0, int wr, int wc,
                                          int l31, int lh, float (&pold)[2][2][16]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = R0 + 64 * wr + 32 * rb + l31;
            const int colb = C0 + 64 * wc + 32 * cb + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = colb + (r & 3) + 8 * (r >> 2);
                pold[cb][rb][r] = (row < n && col < n) ? P[(size_t)col * ld + row] : 0.0f;
            }
        }
}

// One k-chunk of MFMAs out of LDS buffer `buf`; `kend` (16 or 32) columns are live.
__device__ __forceinline__ void mfma_chunk(const float (&sm)[2][2][TILE][LDSP], int buf, int kend, int wr, int wc, int l31, int lh,
                                  f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int kk = 0; kk < KC; kk += 8) {
        if (kk < kend) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                a[cb] = *reinterpret_cast<const f32x4*>(&sm[buf][1][64 * wc + 32 * cb + l31][kk + 4 * lh]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                b[rb] = *reinterpret_cast<const f32x4*>(&sm[buf][0][64 * wr + 32 * rb + l31][kk + 4 * lh]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
                        acc[cb][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][t], b[rb][t], acc[cb][rb], 0, 0, 0);
        }
    }
}

struct XcdLens { int len[8]; };

// Per-thread constants of the down-date kernel (all scalars: passed by value, lives in registers).
struct DdCtx {
    float* P;
    const float* X;
    const float* Y;
    int ld, n, pitch, kp, nchunks, nslots, dbg;
    const volatile int* ld_lds;
    int wr, wc, l31, lh, srow, sc4;
};

typedef float smem_t[2][2][TILE][LDSP];

// request the operands of tile `t`: first panel chunk, then the P tile
template <bool MASKED>
__device__ __forceinline__ void dd_prologue(const DdCtx& c, int2 t, f32x4 (&gx)[4], f32x4 (&gy)[4], float (&pold)[2][2][16]) {
    const float* xs = c.X + (size_t)(t.x * TILE + c.srow) * c.pitch + 4 * c.sc4;
    const float* ys = c.Y + (size_t)(t.y * TILE + c.srow) * c.pitch + 4 * c.sc4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        gx[s] = *reinterpret_cast<const f32x4*>(xs + (size_t)(32 * s) * c.pitch);
        gy[s] = *reinterpret_cast<const f32x4*>(ys + (size_t)(32 * s) * c.pitch);
    }
    if constexpr (MASKED) load_p_tile_masked(c.P, c.ld, c.n, t.x * TILE, t.y * TILE, c.wr, c.wc, c.l31, c.lh, pold);
    else load_p_tile_fast(c.P, c.ld, c.ld_lds, t.x * TILE, t.y * TILE, c.wr, c.wc, c.l31, c.lh, pold);
}

// k-loop of tile `t` (or, if !active, just this phase's nslots barriers)
__device__ __forceinline__ void dd_compute_phase(const DdCtx& c, bool active, int2 t, smem_t& smem, f32x4 (&gx)[4],
                                                 f32x4 (&gy)[4], f32x16 (&acc)[2][2]) {
    const float* xsrc = c.X + (size_t)(t.x * TILE + c.srow) * c.pitch + 4 * c.sc4;
    const float* ysrc = c.Y + (size_t)(t.y * TILE + c.srow) * c.pitch + 4 * c.sc4;
    if (active) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][rb][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *reinterpret_cast<f32x4*>(&smem[0][0][c.srow + 32 * s][4 * c.sc4]) = gx[s];
            *reinterpret_cast<f32x4*>(&smem[0][1][c.srow + 32 * s][4 * c.sc4]) = gy[s];
        }
    }
    __syncthreads();
    for (int ch = 0; ch < c.nchunks; ++ch) {
        if (active) {
            if (ch + 1 < c.nchunks) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    gx[s] = *reinterpret_cast<const f32x4*>(xsrc + (size_t)(32 * s) * c.pitch + (ch + 1) * KC);
                    gy[s] = *reinterpret_cast<const f32x4*>(ysrc + (size_t)(32 * s) * c.pitch + (ch + 1) * KC);
                }
            }
            if (!(c.dbg & 2))
                mfma_chunk(smem, ch & 1, (c.kp - ch * KC < KC) ? c.kp - ch * KC : KC, c.wr, c.wc, c.l31, c.lh, acc);
            if (ch + 1 < c.nchunks) {
                const int nb = (ch + 1) & 1;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    *reinterpret_cast<f32x4*>(&smem[nb][0][c.srow + 32 * s][4 * c.sc4]) = gx[s];
                    *reinterpret_cast<f32x4*>(&smem[nb][1][c.srow + 32 * s][4 * c.sc4]) = gy[s];
                }
            }
        }
        __syncthreads();
    }
}

template <bool MASKED>
__device__ __forceinline__ void dd_store_slot(const DdCtx& c, int slot, int2 te, const float (&pold)[2][2][16],
                                              const f32x16 (&acc)[2][2], float* sT) {
    const bool ediag = te.x == te.y;
    const int eR0 = te.x * TILE, eC0 = te.y * TILE;
    if (0 * c.nslots / 4 == slot) store_sub<0, MASKED>(ediag, c.P, c.ld, c.ld_lds, c.n, eR0, eC0, c.wr, c.wc, c.l31, c.lh, pold, acc, sT, c.dbg);
    if (1 * c.nslots / 4 == slot) store_sub<1, MASKED>(ediag, c.P, c.ld, c.ld_lds, c.n, eR0, eC0, c.wr, c.wc, c.l31, c.lh, pold, acc, sT, c.dbg);
    if (2 * c.nslots / 4 == slot) store_sub<2, MASKED>(ediag, c.P, c.ld, c.ld_lds, c.n, eR0, eC0, c.wr, c.wc, c.l31, c.lh, pold, acc, sT, c.dbg);
    if (3 * c.nslots / 4 == slot) store_sub<3, MASKED>(ediag, c.P, c.ld, c.ld_lds, c.n, eR0, eC0, c.wr, c.wc, c.l31, c.lh, pold, acc, sT, c.dbg);
}

// store tile `te` (a quarter per slot), then request the operands of tile `tp`
template <bool MASKED>
__device__ __forceinline__ void dd_memory_phase(const DdCtx& c, bool have_e, int2 te, bool have_p, int2 tp, f32x4 (&gx)[4],
                                                f32x4 (&gy)[4], float (&pold)[2][2][16], const f32x16 (&acc)[2][2],
                                                float* sT) {
    if (have_e) dd_store_slot<MASKED>(c, 0, te, pold, acc, sT);
    __syncthreads();
    for (int ch = 0; ch < c.nchunks; ++ch) {
        if (have_e) dd_store_slot<MASKED>(c, ch + 1, te, pold, acc, sT);
        if (have_p && ch == c.nchunks - 1) dd_prologue<MASKED>(c, tp, gx, gy, pold);
        __syncthreads();
    }
}

// PERSISTENT, ROLE-SPLIT kernel.  gridDim.x = 8 * nper workgroups of 512 threads, one per CU; workgroup b
// walks the tile list of XCD b % 8 with stride nper.
//
// Measured on the plain two-workgroups-per-CU version: the memory side of a tile (64 KB read, 128 KB
// written) and its MFMA side do not overlap -- the whole chip falls into a convoy in which every
// workgroup computes, then every workgroup waits for HBM (0.29 ms + 0.49 ms ~ the 0.70 ms observed).
// Here the two halves of a workgroup (waves 0-3 / 4-7, one wave of each half per SIMD) alternate ROLES
// phase by phase: while one half runs the k-loop of its tile, the other stores the tile it finished in
// the previous phase -- a quarter of it between every two barriers of the k-loop, so the stores trickle
// out during the MFMAs -- and requests the operands of its next tile.  Every wave executes exactly
// nchunks + 1 barriers per phase whatever its role.
// MASKED = false: off-diagonal tiles only (no masks, buffer addressing).  MASKED = true: the T diagonal
// tiles, in a launch of their own -- keeping the masked code out of the main kernel is what lets that
// one fit its 160 long-lived registers without spilling.
template <bool MASKED>
__global__ __launch_bounds__(512, 2) void downdate_f32_mfma(float* __restrict__ P, int ld, int n,
                                                            const float* __restrict__ X, const float* __restrict__ Y,
                                                            int pitch, int kp, const int2* __restrict__ tiles, int L,
                                                            XcdLens lens, const int32_t* __restrict__ status, int dbg) {
    if (status[0] != 0) return;
    __shared__ __attribute__((aligned(16))) float smem[2][2][TILE][LDSP];   // [buffer][X|Y][row][k]  73,728 B
    __shared__ float scratch[4][32 * TP];                                   // transpose scratch, one per wave of the storing half
    __shared__ int s_ld;
    const int tid = threadIdx.x;
    if (tid == 0) s_ld = ld;
    __syncthreads();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // provably wave-uniform, 0..7
    const int group = wave >> 2;
    const int w4 = wave & 3;
    const int gtid = tid & 255;
    DdCtx c;
    c.P = P; c.X = X; c.Y = Y; c.ld = ld; c.n = n; c.pitch = pitch; c.kp = kp; c.dbg = dbg;
    c.ld_lds = &s_ld;
    c.nchunks = (kp + KC - 1) / KC;                                         // kp multiple of 16: last chunk may be half
    c.nslots = c.nchunks + 1;
    c.wr = w4 & 1;                    // row half of the tile
    c.wc = w4 >> 1;                   // column half
    c.l31 = lane & 31;
    c.lh = lane >> 5;
    c.srow = gtid >> 3;               // staging: 128 x 32 floats per panel = 1024 float4, 4 per thread per panel
    c.sc4 = gtid & 7;
    const int xcd = blockIdx.x & 7;
    const int rk = blockIdx.x >> 3;
    const int nper = gridDim.x >> 3;
    const int2* list = tiles + (size_t)xcd * L;
    const int mylen = lens.len[xcd];
    const int m = mylen > rk ? (mylen - rk + nper - 1) / nper : 0;          // tiles of this workgroup
    float* sT = scratch[w4];

    f32x4 gx[4], gy[4];
    float pold[2][2][16];
    f32x16 acc[2][2];

    // phases 0..m: the half with parity p runs the k-loop of tile p, the other half stores tile p-1 and
    // requests tile p+1.  Both halves execute nslots barriers in every phase.
    const int2 none = make_int2(0, 0);
    if (group == 0) {
        if (m > 0) dd_prologue<MASKED>(c, list[rk], gx, gy, pold);
        for (int p = 0; p <= m; p += 2) {
            dd_compute_phase(c, p < m, p < m ? list[rk + p * nper] : none, smem, gx, gy, acc);
            if (p + 1 <= m)
                dd_memory_phase<MASKED>(c, p < m, p < m ? list[rk + p * nper] : none, p + 2 < m,
                                p + 2 < m ? list[rk + (p + 2) * nper] : none, gx, gy, pold, acc, sT);
        }
    } else {
        for (int p = 0; p <= m; p += 2) {
            dd_memory_phase<MASKED>(c, p >= 1, p >= 1 ? list[rk + (p - 1) * nper] : none, p + 1 < m,
                            p + 1 < m ? list[rk + (p + 1) * nper] : none, gx, gy, pold, acc, sT);
            if (p + 1 <= m) dd_compute_phase(c, p + 1 < m, p + 1 < m ? list[rk + (p + 1) * nper] : none, smem, gx, gy, acc);
        }
    }
}

// ---- fp64 VALU down-date -----------------------------------------------------
constexpr int DT = 64;     // tile edge
constexpr int DK = 16;     // k-chunk

template <typename T>
__global__ __launch_bounds__(256) void downdate_valu(T* __restrict__ P, int ld, int n, const T* __restrict__ X,
                                                     const T* __restrict__ Y, int pitch, int kp,
                                                     const int2* __restrict__ tiles, int L,
                                                     const int32_t* __restrict__ status) {
    if (status[0] != 0) return;
    const int2 tile = tiles[(size_t)(blockIdx.x & 7) * L + (blockIdx.x >> 3)];     // workgroup b -> list b % 8, slot b / 8
    if (tile.x < 0) return;
    __shared__ T sX[DT][DK + 1];
    __shared__ T sY[DT][DK + 1];
    __shared__ T sT[DT][DT + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15;          // rows  tx + 16u
    const int ty = tid >> 4;          // cols  ty + 16v
    const int R0 = tile.x * DT;
    const int C0 = tile.y * DT;
    const bool diag = tile.x == tile.y;
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
        const int cl = ty + 16 * v;
        const int col = C0 + cl;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rl = tx + 16 * u;
            const int row = R0 + rl;
            T val = (T)0;
            if (row < n && col < n) {
                T* p = P + (size_t)col * ld + row;
                val = *p - acc[u][v];
                if (!diag || row >= col) *p = val;
            }
            sT[rl][cl] = val;
        }
    }
    __syncthreads();
    // mirror: thread (tx, ty) stores element (row = ty+16v, col = tx+16u) to P[col, row]; col on tx
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int rl = ty + 16 * v;
        const int rowI = R0 + rl;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cl = tx + 16 * u;
            const int colJ = C0 + cl;
            if (rowI < n && colJ < n && (!diag || rowI > colJ)) P[(size_t)rowI * ld + colJ] = sT[rl][cl];
        }
    }
}

// Tile lists: eight lists of equal length L (padded with -1), laid out [xcd][slot].  XCD x
// (= workgroup id % 8 under round-robin dispatch) walks super-rows of SR tile rows, largest
// first, column by column; its workgroups take slots rk, rk + nper, ...
void build_tile_order(int T, bool with_diag, std::vector<int2>& out, int xlen[8]) {
    constexpr int NX = 8, SR = 4;
    const int nsr = (T + SR - 1) / SR;
    std::vector<long> load(NX, 0);
    std::vector<std::vector<int>> mine(NX);
    for (int s = nsr - 1; s >= 0; --s) {
        const int I0 = s * SR, I1 = std::min(T, I0 + SR);
        long cnt = 0;
        for (int I = I0; I < I1; ++I) cnt += I + 1;
        int best = 0;
        for (int xcd = 1; xcd < NX; ++xcd)
            if (load[xcd] < load[best]) best = xcd;
        load[best] += cnt;
        mine[best].push_back(s);
    }
    std::vector<std::vector<int2>> lists(NX);
    size_t L = 1;
    for (int xcd = 0; xcd < NX; ++xcd) {
        for (int s : mine[xcd]) {
            const int I0 = s * SR, I1 = std::min(T, I0 + SR);
            for (int J = 0; J < I1; ++J)
                for (int I = std::max(I0, J); I < I1; ++I)
                    if (with_diag || I != J) lists[xcd].push_back(make_int2(I, J));
        }
        L = std::max(L, lists[xcd].size());
    }
    out.assign(L * NX, make_int2(-1, -1));
    for (int xcd = 0; xcd < NX; ++xcd) {
        xlen[xcd] = (int)lists[xcd].size();
        for (size_t slot = 0; slot < lists[xcd].size(); ++slot) out[xcd * L + slot] = lists[xcd][slot];
    }
}

// the T diagonal tiles, dealt round-robin to the eight lists
void build_diag_order(int T, std::vector<int2>& out, int xlen[8]) {
    const size_t L = (size_t)(T + 7) / 8;
    out.assign(L * 8, make_int2(-1, -1));
    for (int i = 0; i < 8; ++i) xlen[i] = 0;
    for (int I = 0; I < T; ++I) {
        const int xcd = I & 7;
        out[xcd * L + xlen[xcd]++] = make_int2(I, I);
    }
}

int ensure_tile_order(slam_ekf* h, int T) {
    if (h->tiles && h->tiles_T == T) return SLAM_OK;
    const bool f32 = h->dtype == SLAM_F32;
    std::vector<int2> order, diag;
    build_tile_order(T, /*with_diag=*/!f32, order, h->tiles_xlen);      // the fp64 kernel masks its diagonal tiles itself
    if (f32) build_diag_order(T, diag, h->diag_xlen);
    HIP_TRY(hipStreamSynchronize(h->stream));          // earlier down-dates may still read the old lists
    const size_t total = order.size() + diag.size();
    if ((int)total > h->tiles_cap) {
        if (h->tiles) (void)hipFree(h->tiles);
        h->tiles = nullptr;
        h->tiles_cap = 0;
        HIP_TRY(hipMalloc((void**)&h->tiles, sizeof(int2) * total));
        h->tiles_cap = (int)total;
    }
    HIP_TRY(hipMemcpy(h->tiles, order.data(), sizeof(int2) * order.size(), hipMemcpyHostToDevice));
    if (!diag.empty())
        HIP_TRY(hipMemcpy(h->tiles + order.size(), diag.data(), sizeof(int2) * diag.size(), hipMemcpyHostToDevice));
    h->tiles_T = T;
    h->tiles_len = (int)order.size() / 8;          // L: entries per XCD list
    h->diag_off = (int)order.size();
    h->diag_len = (int)diag.size() / 8;
    return SLAM_OK;
}

}  // namespace

int launch_downdate(slam_ekf* h, int kp_total, const void* X, const void* Y, int pitch) {
    const int n = 3 + 2 * h->N;
    const int edge = h->dtype == SLAM_F32 ? TILE : DT;
    const int rc = ensure_tile_order(h, (n + edge - 1) / edge);
    if (rc) return rc;
    KTimer t(h, SLAM_K_SYRK);
    if (h->dtype == SLAM_F32) {
        // persistent: one 512-thread workgroup per CU, never more per XCD than its list is long
        int per_xcd = h->num_cus / 8;
        if (per_xcd > h->tiles_len) per_xcd = h->tiles_len;
        if (per_xcd < 1) per_xcd = 1;
        XcdLens lens, dlens;
        for (int i = 0; i < 8; ++i) { lens.len[i] = h->tiles_xlen[i]; dlens.len[i] = h->diag_xlen[i]; }
        if (h->tiles_len > 0 && lens.len[0] + lens.len[1] + lens.len[2] + lens.len[3] + lens.len[4] + lens.len[5] +
                                    lens.len[6] + lens.len[7] > 0)
            hipLaunchKernelGGL(downdate_f32_mfma<false>, dim3(8 * per_xcd), dim3(512), 0, h->stream, (float*)h->P, h->ld, n,
                               (const float*)X, (const float*)Y, pitch, kp_total, (const int2*)h->tiles, h->tiles_len, lens,
                               h->d_status, h->debug_flags);
        int dper = h->num_cus / 8;
        if (dper > h->diag_len) dper = h->diag_len;
        if (dper < 1) dper = 1;
        hipLaunchKernelGGL(downdate_f32_mfma<true>, dim3(8 * dper), dim3(512), 0, h->stream, (float*)h->P, h->ld, n,
                           (const float*)X, (const float*)Y, pitch, kp_total, (const int2*)(h->tiles + h->diag_off),
                           h->diag_len, dlens, h->d_status, h->debug_flags);
    } else {
        hipLaunchKernelGGL(downdate_valu<double>, dim3(8 * h->tiles_len), dim3(256), 0, h->stream, (double*)h->P, h->ld, n,
                           (const double*)X, (const double*)Y, pitch, kp_total, (const int2*)h->tiles, h->tiles_len,
                           h->d_status);
    }
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}
