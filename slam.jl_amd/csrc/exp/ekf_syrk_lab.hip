// ekf_syrk.hip -- K6 / K6J: the covariance down-date  P -= X * Y'.
//
//   reference form   P -= W1*W1'            (src/ekf.jl:75)      X = Y = W1   (n x kp)
//   Joseph form      P -= K*T' + T*K'       (not in reference)   X = [K|T], Y = [T|K]  (n x 2kp)
//
// P is n x n column-major with leading dimension ld; X, Y are row-major
// [npad][pitch] panels, zero in rows >= n and in padding columns.  This kernel
// is where >= 90 % of an update's time goes.
//
// Symmetry.  X*Y' is symmetric in both forms, so -- like BLAS syrk, which is what
// Julia's W1*W1' dispatches to -- only ONE triangle is updated: the tiles on and
// below the diagonal.  Tile (I,J), I > J, is read once, P(I,J) - X_I*Y_J' is stored
// once; the tiles ABOVE the diagonal are not maintained (every reader of P goes
// through sym_at() in device_math.h, and slam_ekf_get_state mirrors the lower tiles
// on download).  Diagonal tiles stay complete: they store their lower triangle
// directly and their upper triangle from an in-tile mirror through LDS, which keeps
// them bit-for-bit symmetric also in the Joseph form.  Algorithmic traffic per
// update: n^2/2 elements read + n^2/2 written; algorithmic work n^2*k flops.
//
// fp32: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A 512-thread
// workgroup owns a 128 x 128 tile; wave (wr, wc) of its eight owns rows 64 wr..,
// columns 32 wc.. as 2 x 1 MFMA blocks (32 accumulator registers; 128 VGPRs per
// wave, so two workgroups = four waves per SIMD are resident).  The product is
// accumulated from zero and subtracted from P once (the reference's order: form
// W1*W1', then subtract).  Panels travel global -> registers -> LDS in chunks of
// 32 columns with two LDS buffers and ONE barrier per chunk; the loads of chunk
// c+1 are in flight while chunk c feeds the MFMAs, the P tile is requested behind
// the tile's last panel request.  Each lane pulls FOUR consecutive k of its row
// with one ds_read_b128 and feeds four MFMAs -- the k index inside an MFMA is
// only a label, so lane half h takes k = kc+4h..kc+4h+3 for both operands.
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
// fp64: the same triangular scheme on v_mfma_f64_16x16x4_f64 (64 x 64 tiles); at the k of
// BASELINE.json's fp64 configuration the down-date is HBM-bound.
#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include <type_traits>
#include "device_math.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = SLAM_TILE;      // 128
constexpr int KC = 32;               // k-chunk per barrier
constexpr int LDSP = KC + 4;         // LDS row pitch in floats (144 B: 16-B aligned, conflict-free b128)

// The epilogue scratch is private to a wave; LDS executes one wave's instructions in order, so a
// compiler-level fence is all that is needed between its writes and reads.  (A __syncthreads()
// here would also drain every outstanding global store of the epilogue: vmcnt(0) per barrier.)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// P tile <-> registers, 16 bytes per lane per instruction.
//
// P is allocated in whole 128 x 128 tiles (rows/columns >= n are padding that stays zero because
// the panels are zero there), so no element of an off-diagonal tile needs a mask.  Every access is
// a buffer instruction "descriptor of a 128-column band + per-lane offset (VGPR) + wave-uniform
// offset (SGPR)": no per-element 64-bit VALU address arithmetic.  In the "x4" layout lane
// (q = lane & 7, cl = lane >> 3) owns rows 4q..4q+3 of column cl + 8s (s = 0..3) of a 32 x 32
// sub-block: eight lanes cover one full 128-byte line, a wave instruction eight lines.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct DdCtxP {      // where P lives (first members of DdCtx)
    float* P;
    int ld;
    float* side;     // the packed 2 x 2 diagonal blocks (device_math.h: side_note), kept by the diagonal tiles' epilogue
    int side_n;
};

// P is TILE-MAJOR (device_math.h): tile (I, J) is one contiguous 64 KiB column-major block, the tiles of a column band
// follow each other -- the band-major walk of the split-bf16 path reads and writes P as one linear stream.
__device__ __forceinline__ auto tile_rsrc(const DdCtxP& c, int R0, int C0) {
    const float* tile = c.P + tile_base(R0 >> 7, C0 >> 7, c.ld >> 7, 7);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), (short)0, TILE * TILE * 4, 0x00020000);
}

constexpr int SP = 36;                 // pitch of the per-wave 32 x 32 epilogue scratches (16-byte aligned rows)
constexpr int NWAVE = 8;               // waves per workgroup: wave (wr, wc) owns rows 64*wr.., columns 32*wc.. of the tile
constexpr int NTHREADS = 64 * NWAVE;

struct DdCtx : DdCtxP {       // per-thread constants of the down-date kernel
    const float* X;
    const float* Y;
    int pitch, kp, nchunks, dbg, xflags;
    const char* img;     // pre-split bf16 image of the panel (split-bf16 path, reference form), or null
    int img_nch;         // chunks of 16 columns per 128-row block in the image
    int wr, wc, l31, lh, q, cl, srow, sc4;
    unsigned long long t_head, t_wait, t_epi, t_total;     // DBG instantiation only (shader clocks, summed over tiles)
};

__device__ __forceinline__ void load_p_tile(const DdCtx& c, int R0, int C0, f32x4 (&pold)[2][4]) {
    const auto rs = tile_rsrc(c, R0, C0);
    const int voff = (c.cl * TILE + 4 * c.q) * 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int soff = ((32 * c.wc + 8 * s) * TILE + 64 * c.wr + 32 * rb) * 4;
            pold[rb][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
        }
}

// Epilogue of one tile.  Per 32 x 32 sub-block: the product leaves the MFMA layout through LDS
// ([col][row]), comes back in the x4 layout, and P_old - product is stored to tile (I,J) with full
// 128-byte lines.  `diag`: tile (I,I) -- the direct store keeps row >= col; the values also go to a
// second scratch transposed ([row][col]) and from there to the upper triangle of the same tile
// (row > col of the mirror), element by element where a 4-group straddles the diagonal, so P stays
// bit-for-bit symmetric.
template <bool diag>
__device__ __forceinline__ void store_p_tile(const DdCtx& c, int R0, int C0, const f32x4 (&pold)[2][4],
                                             const f32x16 (&acc)[2], float* sD, float* sV, int dbg) {
    const auto rs = tile_rsrc(c, R0, C0);       // the tile (the in-tile mirror of a diagonal tile goes to the same one)
    const int voff = (c.cl * TILE + 4 * c.q) * 4;
    const int q = c.q, cl = c.cl;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int rowb = R0 + 64 * c.wr + 32 * rb, colb = C0 + 32 * c.wc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sD[(4 * c.lh + (r & 3) + 8 * (r >> 2)) * SP + c.l31] = acc[rb][r];
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = cl + 8 * s;
            const f32x4 prod = *reinterpret_cast<const f32x4*>(&sD[col * SP + 4 * q]);
            const f32x4 val = pold[rb][s] - prod;
            const int soff = ((32 * c.wc + 8 * s) * TILE + 64 * c.wr + 32 * rb) * 4;
            if (dbg & 1) continue;                                 // experiment: no stores (P stays as it is)
            if (!diag) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rs, voff, soff, 0);
                // the tile below a diagonal tile holds P[f+1, f] of the landmark straddling the boundary: (row 0, column 127) of the
                // tile = wave (0, 3), (rb, s) = (0, 3), lane (q, cl) = (0, 7), element 0
                if (rb == 0 && s == 3 && R0 == C0 + TILE && c.wr == 0 && c.wc == 3 && q == 0 && cl == 7)
                    c.side[(size_t)c.side_n + ((C0 + TILE - 4) >> 1)] = val.x;
            } else {
                // (scalar copies: bit-casting val[t] directly made hipcc 7.2 store element 0 four times)
                const float ve[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (rowb + 4 * q + t >= colb + col) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ve[t]), rs, voff + 4 * t, soff, 0);
                        side_note(c.side, c.side_n, rowb + 4 * q + t, colb + col, ve[t]);
                    }
#pragma unroll
                for (int t = 0; t < 4; ++t) sV[(4 * q + t) * SP + col] = ve[t];
            }
        }
        wave_lds_fence();
        if (diag && !(dbg & 1)) {                                 // in-tile mirror: upper triangle of a diagonal tile
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int rr = cl + 8 * s;                         // row of the sub-block -> column of the mirror
                const f32x4 val = *reinterpret_cast<const f32x4*>(&sV[rr * SP + 4 * q]);
                const int soff = ((64 * c.wr + 32 * rb + 8 * s) * TILE + 32 * c.wc) * 4;
                const float ve[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (rowb + rr > colb + 4 * q + t)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ve[t]), rs, voff + 4 * t, soff, 0);
            }
            wave_lds_fence();
        }
    }
}

// One "step" = 8 columns of k: each lane holds four consecutive k of its row for the wave's column
// block and its two row blocks (3 x ds_read_b128) and feeds 8 MFMAs.
struct Frag {
    f32x4 a, b[2];
};

typedef float smem_t[2][2][TILE][LDSP];

// panel staging: 128 x 32 floats per panel and chunk = 1024 float4, two per thread per panel
__device__ __forceinline__ void request_chunk(const DdCtx& c, int2 t, int chunk, f32x4 (&gx)[2], f32x4 (&gy)[2]) {
    // descriptor of the whole panel + one per-lane offset + wave-uniform offsets: no 64-bit per-lane pointers
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.X), (short)0, c.ld * c.pitch * 4, 0x00020000);
    const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.Y), (short)0, c.ld * c.pitch * 4, 0x00020000);
    const int voff = (c.srow * c.pitch + 4 * c.sc4) * 4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int sx = ((t.x * TILE + 64 * s) * c.pitch + chunk * KC) * 4;
        const int sy = ((t.y * TILE + 64 * s) * c.pitch + chunk * KC) * 4;
        gx[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, sx, 0));
        gy[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, voff, sy, 0));
    }
}

__device__ __forceinline__ void fill_lds(const DdCtx& c, smem_t& smem, int buf, const f32x4 (&gx)[2], const f32x4 (&gy)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        *reinterpret_cast<f32x4*>(&smem[buf][0][c.srow + 64 * s][4 * c.sc4]) = gx[s];
        *reinterpret_cast<f32x4*>(&smem[buf][1][c.srow + 64 * s][4 * c.sc4]) = gy[s];
    }
}

__device__ __forceinline__ void read_frag(const DdCtx& c, const smem_t& sm, int buf, int kk, Frag& f) {
    f.a = *reinterpret_cast<const f32x4*>(&sm[buf][1][32 * c.wc + c.l31][kk + 4 * c.lh]);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
        f.b[rb] = *reinterpret_cast<const f32x4*>(&sm[buf][0][64 * c.wr + 32 * rb + c.l31][kk + 4 * c.lh]);
}

template <bool DBG>
__device__ __forceinline__ void mfma_step(const DdCtx& c, const Frag& f, f32x16 (&acc)[2]) {
    if (DBG && (c.dbg & 2)) return;
    if (!(c.xflags & 2)) __builtin_amdgcn_s_setprio(3);     // the wave that has its operands keeps the matrix pipe (A/B: -0.6 %)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t], f.b[rb][t], acc[rb], 0, 0, 0);
    if (!(c.xflags & 2)) __builtin_amdgcn_s_setprio(0);
}

// One tile, start to finish.  On entry gx/gy hold (or are about to receive) the first panel chunk of
// `tile`; on exit they hold the request for the first chunk of `next` (if valid), issued BEFORE this
// tile's stores.
//
// Four waves per SIMD cover each other's LDS and barrier latencies, so the k-loop is kept simple (one
// fragment set: the 128-VGPR budget of that occupancy has no room for two).  Vector-memory results return in issue
// order, so anything waited for after the 64 KiB P tile has been requested also waits for the P tile:
// the P loads are therefore issued right after the LAST panel request of the tile and nothing but the
// epilogue ever waits behind them.
template <bool DIAG, bool DBG>
__device__ __forceinline__ void dd_tile(DdCtx& c, int2 tile, int2 next, smem_t& smem, float* sD, float* sV,
                                        f32x4 (&gx)[2], f32x4 (&gy)[2]) {
    const int R0 = tile.x * TILE;     // rows  (I)
    const int C0 = tile.y * TILE;     // cols  (J <= I)
    const int nch = c.nchunks;
    f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    f32x4 pold[2][4];                 // the wave's 64 x 32 part of the P tile
    Frag f0;

    constexpr bool PROF = DBG;
    unsigned long long tp0 = 0, tp1 = 0;
    if (PROF) tp0 = __builtin_amdgcn_s_memtime();
    fill_lds(c, smem, 0, gx, gy);
    __syncthreads();
    if (PROF) { tp1 = __builtin_amdgcn_s_memtime(); c.t_head += tp1 - tp0; }
    if (nch > 1) request_chunk(c, tile, 1, gx, gy);
    if (nch <= 2 && !(DBG && (c.dbg & 4))) load_p_tile(c, R0, C0, pold);
    int ch = 0;
    // chunk ch is followed by chunk ch+1: four steps, then the hand-over of the other LDS buffer
#define DD_STEPS(BUF, N)                                   \
    _Pragma("unroll") for (int st = 0; st < (N); ++st) {   \
        read_frag(c, smem, BUF, 8 * st, f0);               \
        mfma_step<DBG>(c, f0, acc);                        \
    }
#define DD_CHUNK(REQUEST, LOADP)                                                                   \
    {                                                                                              \
        const int buf = ch & 1;                                                                    \
        DD_STEPS(buf, 4)                                                                           \
        if (PROF) tp1 = __builtin_amdgcn_s_memtime();                                              \
        fill_lds(c, smem, buf ^ 1, gx, gy);                                                        \
        __syncthreads();                                                                           \
        if (PROF) c.t_wait += __builtin_amdgcn_s_memtime() - tp1;                                  \
        if (REQUEST) request_chunk(c, tile, ch + 2, gx, gy);                                       \
        if ((LOADP) && !(DBG && (c.dbg & 4))) load_p_tile(c, R0, C0, pold);                        \
        ++ch;                                                                                      \
    }
    while (ch + 3 < nch) DD_CHUNK(true, false)
    if (nch >= 3) DD_CHUNK(true, true)          // ch = nch-3: the last panel request, then the P tile
    if (nch >= 2) DD_CHUNK(false, false)        // ch = nch-2
#undef DD_CHUNK
    {   // last chunk: two or four steps (kp is a multiple of 16)
        const int buf = ch & 1;
        DD_STEPS(buf, 2)
        if (c.kp - ch * KC > 16) {
            read_frag(c, smem, buf, 16, f0);
            mfma_step<DBG>(c, f0, acc);
            read_frag(c, smem, buf, 24, f0);
            mfma_step<DBG>(c, f0, acc);
        }
    }
#undef DD_STEPS
    if (PROF) tp1 = __builtin_amdgcn_s_memtime();
    __syncthreads();                  // every wave is done with the panels: LDS becomes epilogue scratch
    if (next.x >= 0) request_chunk(c, next, 0, gx, gy);
    store_p_tile<DIAG>(c, R0, C0, pold, acc, sD, sV, DBG ? c.dbg : 0);
    __syncthreads();                  // scratch free again before the next tile's LDS fill
    if (PROF) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        c.t_epi += t - tp1;
        c.t_total += t - tp0;
    }
}

// ---- streaming path for runs of OFF-DIAGONAL tiles -----------------------------------------------------------------
// The LDS pipeline does not know about tiles: every chunk, whichever tile it belongs to, is [steps on buffer pb] ->
// [fill buffer pb^1 with the following chunk] -> barrier -> [request the chunk after that].  A tile boundary is then
// only register work: P_old - acc is formed in the MFMA layout and stored with dword buffer stores (an accumulator
// register covers 32 consecutive rows of one column = a full 128-byte line per half-wave, so nothing has to pass
// through LDS), acc is cleared and the k-loop of the next tile is already fed.  Compared with dd_tile this removes
// two of the six barriers per tile, the per-wave LDS transposes, and the pipeline drain at the tile boundary.
template <int AUX = 2>        // 2 = non-temporal: the P tile is streamed (see dd_stream_b)
__device__ __forceinline__ void load_p_mfma(const DdCtx& c, int R0, int C0, float (&pold)[2][16]) {
    const auto rs = tile_rsrc(c, R0, C0);
    const int voff = (4 * c.lh * TILE + c.l31) * 4;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int soff = ((32 * c.wc + 8 * (r >> 2) + (r & 3)) * TILE + 64 * c.wr + 32 * rb) * 4;
            pold[rb][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX));
        }
}

template <int AUX = 2>
__device__ __forceinline__ void store_p_mfma(const DdCtx& c, int R0, int C0, const float (&pold)[2][16], f32x16 (&acc)[2]) {
    const auto rs = tile_rsrc(c, R0, C0);
    const int voff = (4 * c.lh * TILE + c.l31) * 4;
    // the tile BELOW a diagonal tile holds P[f+1, f] of the landmark that straddles the tile boundary (f = 127 mod 128):
    // that entry of the packed 2 x 2 blocks (device_math.h: side_note) is kept here -- 1 tile in 78, a wave-uniform branch
    const bool adj = R0 == C0 + TILE;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int soff = ((32 * c.wc + 8 * (r >> 2) + (r & 3)) * TILE + 64 * c.wr + 32 * rb) * 4;
            const float v = pold[rb][r] - acc[rb][r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, AUX);
            // the one entry of such a tile that belongs to the side array: (row 0, column 127) = wave (0, 3), lane 32, (rb, r) = (0, 15)
            if (rb == 0 && r == 15 && adj && c.wr == 0 && c.wc == 3 && c.lh == 1 && c.l31 == 0) c.side[(size_t)c.side_n + ((C0 + TILE - 4) >> 1)] = v;
            acc[rb][r] = 0.0f;
        }
}

// Processes list[slot], list[slot + nper], ... while they are off-diagonal.  NCH = chunks per tile (2..4), a
// compile-time constant: the body of a tile is straight-line code, so the compiler's vmcnt bookkeeping is exact (a
// panel chunk is waited for with vmcnt(#younger requests), never with vmcnt(0) behind the P tile or the stores).
// On entry gx/gy hold the request for chunk 0 of the first tile; on return `slot` is the first unprocessed position
// and, if that tile exists, gx/gy hold the request for ITS chunk 0 -- the contract of dd_tile, which takes over for
// the diagonal tiles.
template <bool DBG, int NCH, int POFF = 2>
__device__ __forceinline__ void dd_stream(DdCtx& c, const int2* __restrict__ list, int L, int nper, int& slot, smem_t& smem,
                                          f32x4 (&gx)[2], f32x4 (&gy)[2]) {
    auto fetch = [&](int sl) { return sl < L ? list[sl] : make_int2(-1, -1); };
    constexpr int PCH = NCH - POFF > 0 ? NCH - POFF : 0;  // the chunk at whose start the P tile is requested
    int2 tile = fetch(slot);
    int2 next = fetch(slot + nper);
    f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    float pold[2][16];
    Frag f0;
    const int last_steps = (c.kp - (NCH - 1) * KC <= 16) ? 2 : 4;
    fill_lds(c, smem, 0, gx, gy);
    __syncthreads();
    request_chunk(c, tile, 1, gx, gy);
    // buffer parity: chunk ch of a tile sits in buffer (base + ch) & 1; with NCH odd the base flips per tile
    int base = 0;
    for (;;) {
        const bool next_off = next.x >= 0 && next.x != next.y;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int pb = (base + ch) & 1;
            if (ch == PCH) load_p_mfma(c, tile.x * TILE, tile.y * TILE, pold);
            if (ch < NCH - 1) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    read_frag(c, smem, pb, 8 * st, f0);
                    mfma_step<DBG>(c, f0, acc);
                }
                fill_lds(c, smem, pb ^ 1, gx, gy);               // chunk ch + 1 of this tile
                __syncthreads();
                if (ch + 2 < NCH) request_chunk(c, tile, ch + 2, gx, gy);
                else if (next.x >= 0) request_chunk(c, next, 0, gx, gy);      // (also when `next` is diagonal: hand-over)
            } else {
                for (int st = 0; st < last_steps; ++st) {
                    read_frag(c, smem, pb, 8 * st, f0);
                    mfma_step<DBG>(c, f0, acc);
                }
                if (next_off) fill_lds(c, smem, pb ^ 1, gx, gy); // chunk 0 of the next tile
                __syncthreads();
                if (next_off) request_chunk(c, next, 1, gx, gy);
                store_p_mfma(c, tile.x * TILE, tile.y * TILE, pold, acc);
            }
        }
        slot += nper;
        if (!next_off) return;
        tile = next;
        next = fetch(slot + nper);
        base = (base + NCH) & 1;
    }
}

// ---- split-bf16 streaming path ------------------------------------------------------------------------------------
// The fp32 matrix core (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) is what bounds the path above; the bf16 one is 16 times
// faster.  Every fp32 number is EXACTLY the sum of three bf16 numbers, v = h + m + l (h = bf16(v), m = bf16(v - h),
// l = bf16(v - h - m): 3 x 8 significant bits + signs cover the 24 of fp32), and a product of two bf16 numbers is
// exact in fp32, so x*y = sum over the nine split products, each of them one exact term of an fp32 accumulation.
// Six of them are formed -- (h,h), (h,m), (m,h), (h,l), (l,h), (m,m); the three left out, (m,l), (l,m), (l,l), are
// below 2^-25 |x y|, half a unit in the last place of the product itself, i.e. the result carries the rounding of an
// UNFUSED fp32 multiply-add instead of a fused one.  Measured against the fp64 oracle (tools/acc_downdate.py, N =
// 1500, 64 observations, rms error of P after one / six updates relative to max|P|): fp32 MFMA path 8.80e-8 /
// 1.32e-7, this path 7.08e-8 / 1.05e-7, all nine terms 7.08e-8 / 1.05e-7 -- the three small terms change nothing
// and cost 11 % of the kernel's time; tests/test_gpu_ekf.py holds the comparison.  Six bf16 MFMAs do the work of
// sixteen fp32 ones: the down-date becomes HBM-bound.
//
// Same pipeline as dd_stream with 16 columns of k per chunk (kp is a multiple of 16: no partial chunk).  The panels
// stay fp32 in global memory; the split happens once per element and tile where the chunk is written to LDS:
// [buffer][X|Y][h|m|l][128 rows][16 bf16 + pad] with a 48-byte row pitch, which makes the ds_read_b128 of a fragment
// (lane = row, 16 bytes = 8 consecutive k) conflict-free.  2 x 36,864 B: the same footprint as the fp32 buffers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int KB = 16;                 // k per chunk
constexpr int BROW = 48;               // bytes per LDS row
constexpr int BARR = TILE * BROW;      // bytes per [panel][split] array

// Staging map: thread -> (row, k-quad) of a 128 x 16 chunk.  Four lanes share a row (64 contiguous bytes of the
// panel); the 16 contiguous lanes that a ds_write_b64 services together take rows r, r+2, r+4, r+6, whose 32-byte
// pieces at a 48-byte pitch fall on disjoint banks (bank = word mod 32 for stores): rows r..r+3 would be 2-way
// (29 % of the kernel's LDS cycles were conflict cycles with the plain map; worth 0.5 % of its time).
__device__ __forceinline__ int stage_row(int tid) {
    return (tid >> 6) * 16 + ((tid >> 5) & 1) * 8 + ((tid >> 4) & 1) + 2 * ((tid >> 2) & 3);
}

__device__ __forceinline__ void request_chunk_b(const DdCtx& c, int2 t, int chunk, f32x4& gx, f32x4& gy) {
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.X), (short)0, c.ld * c.pitch * 4, 0x00020000);
    const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.Y), (short)0, c.ld * c.pitch * 4, 0x00020000);
    const int tid = threadIdx.x;
    const int voff = (stage_row(tid) * c.pitch + 4 * (tid & 3)) * 4;
    if (c.dbg & 8) t = make_int2(0, 0);        // experiment (DBG build only sets dbg): every tile reads the panels of tile (0,0)
    gx = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (t.x * TILE * c.pitch + chunk * KB) * 4, 0));
    gy = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, voff, (t.y * TILE * c.pitch + chunk * KB) * 4, 0));
}

// two fp32 -> packed (h, m, l) bf16 pairs; the bf16 values are widened back with bit operations
__device__ __forceinline__ void split2(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {v0, v1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r1 = {v0 - __uint_as_float(h << 16), v1 - __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = {r1.x - __uint_as_float(m << 16), r1.y - __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

template <bool DBG>
__device__ __forceinline__ void fill_lds_b(const DdCtx& c, char* sm, int buf, const f32x4& gx, const f32x4& gy) {
    const int tid = threadIdx.x;
    char* base = sm + buf * (6 * BARR) + stage_row(tid) * BROW + (tid & 3) * 8;
#pragma unroll
    for (int pnl = 0; pnl < 2; ++pnl) {
        const f32x4 g = pnl ? gy : gx;
        unsigned h0, m0, l0, h1, m1, l1;
        if (DBG && (c.dbg & 16)) {             // experiment: no split arithmetic (wrong numbers)
            h0 = m0 = l0 = __float_as_uint(g.x) ^ __float_as_uint(g.y);
            h1 = m1 = l1 = __float_as_uint(g.z) ^ __float_as_uint(g.w);
        } else {
        split2(g.x, g.y, h0, m0, l0);
        split2(g.z, g.w, h1, m1, l1);
        }
        *reinterpret_cast<u32x2*>(base + (3 * pnl + 0) * BARR) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(base + (3 * pnl + 1) * BARR) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(base + (3 * pnl + 2) * BARR) = u32x2{l0, l1};
    }
}

// one chunk = one k16 step: 3 + 2 x 3 fragments, 2 x 6 MFMAs, smallest terms first
template <bool DBG>
__device__ __forceinline__ void mfma_chunk_b(const DdCtx& c, const char* sm, int buf, f32x16 (&acc)[2]) {
    if (DBG && (c.dbg & 2)) return;
    const char* base = sm + buf * (6 * BARR) + c.l31 * BROW + c.lh * 16;
    bf16x8 a[3];                       // Y: the wave's 32 columns (the MFMA's A operand, as in read_frag)
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) a[sp] = *reinterpret_cast<const bf16x8*>(base + (3 + sp) * BARR + (32 * c.wc) * BROW);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        bf16x8 b[3];                   // X: 32 of the wave's 64 rows
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) b[sp] = *reinterpret_cast<const bf16x8*>(base + sp * BARR + (64 * c.wr + 32 * rb) * BROW);
        if (!(c.xflags & 2)) __builtin_amdgcn_s_setprio(3);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[rb], 0, 0, 0);      // (m, m)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[rb], 0, 0, 0);      // (h, l)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[rb], 0, 0, 0);      // (l, h)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[rb], 0, 0, 0);      // (h, m)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[rb], 0, 0, 0);      // (m, h)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[rb], 0, 0, 0);      // (h, h)
        if (!(c.xflags & 2)) __builtin_amdgcn_s_setprio(0);
    }
}

// ---- the same path fed from a PRE-SPLIT panel -------------------------------------------------------------------
// In the reference form (X = Y = W1) the W1 kernel also writes the panel already split into bf16 (h, m, l) and laid
// out as this path's LDS image: [row block of 128][chunk of 16 columns][h|m|l][128 rows][32 bytes], the two 16-byte
// halves of a row exchanged when (row >> 3) is odd -- with a 32-byte pitch that swizzle makes the fragment's
// ds_read_b128 conflict-free (the four 16-lane groups of the instruction each cover all 64 banks once).  A chunk
// of a panel is then 12,288 contiguous bytes: the fill is three 16-byte copies per thread, no arithmetic, and the
// two buffers take 48 KB instead of 72.
constexpr int IMG_ARR = TILE * 32;              // bytes per [split] array of a chunk
constexpr int IMG_CHUNK = 3 * IMG_ARR;          // bytes per (row block, chunk)
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void request_chunk_p(const DdCtx& c, int2 t, int chunk, u32x4b (&g)[3]) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.img), (short)0, 0x7fffffff, 0x00020000);
    const int tid = threadIdx.x;
#ifdef SLAMHIP_EXPERIMENTS
    // experiment (SLAMHIP_DEBUG bit 65536, round 4): EVERY tile reads the image of tile (0, 0) -- 192 KB that never leave the
    // L2s: what the panels' fabric traffic (every XCD fetches every column band's image once: 0.12 GB of the 0.2 GB the launch
    // moves beyond P itself) costs in time.  Wrong numbers.
    if (c.xflags & 256) t = make_int2(0, 0);
#endif
    const int sx = (t.x * c.img_nch + chunk) * IMG_CHUNK, sy = (t.y * c.img_nch + chunk) * IMG_CHUNK;
    // pieces tid, tid + 512, tid + 1024 of the 1536 sixteen-byte pieces [X image | Y image]
    g[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, sx, 0);
#ifdef SLAMHIP_EXPERIMENTS
    if (c.dbg & 64) {          // experiment: the Y image is NOT loaded (what a column panel resident in LDS would save; wrong numbers)
        if (tid < 256) g[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 512) * 16, sx, 0);
        return;
    }
#endif
    g[1] = tid < 256 ? __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 512) * 16, sx, 0)
                     : __builtin_amdgcn_raw_buffer_load_b128(rs, (tid - 256) * 16, sy, 0);
    g[2] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 256) * 16, sy, 0);
}

__device__ __forceinline__ void fill_lds_p(char* sm, int buf, const u32x4b (&g)[3]) {
    char* base = sm + buf * (2 * IMG_CHUNK) + threadIdx.x * 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4b*>(base + j * 8192) = g[j];
}

template <bool DBG>
__device__ __forceinline__ void mfma_chunk_p(const DdCtx& c, const char* sm, int buf, f32x16 (&acc)[2], bool negate = false) {
    if (DBG && (c.dbg & 2)) return;
    // row r of an array sits at r * 32, its k-half h at ((h ^ (r >> 3)) & 1) * 16; all rows here are l31 + multiples of 32
    const char* base = sm + buf * (2 * IMG_CHUNK) + c.l31 * 32 + ((c.lh ^ (c.l31 >> 3)) & 1) * 16;
    bf16x8 a[3];
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) a[sp] = *reinterpret_cast<const bf16x8*>(base + IMG_CHUNK + sp * IMG_ARR + (32 * c.wc) * 32);
    if (DBG && negate) {       // WRAP experiment: the second walk over the image takes back what the first one subtracted
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
            u32x4b t = __builtin_bit_cast(u32x4b, a[sp]);
            t ^= 0x80008000u;
            a[sp] = __builtin_bit_cast(bf16x8, t);
        }
    }
    bf16x8 b[3];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        // (experiment, bit 8: the second row block reuses the first one's fragments -- a third fewer LDS reads, wrong numbers:
        //  what a 64 x 64 register tile per wave would save on the fragment side)
        if (!(DBG && (c.dbg & 8) && rb == 1)) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) b[sp] = *reinterpret_cast<const bf16x8*>(base + sp * IMG_ARR + (64 * c.wr + 32 * rb) * 32);
        }
        // (experiments build, SLAMHIP_X bit 128: the priorities the other way round -- a wave is favoured while it ISSUES
        //  memory and LDS operations and steps back during its MFMAs)
        if (c.xflags & 128) __builtin_amdgcn_s_setprio(0);
        else if (!(c.xflags & 2)) __builtin_amdgcn_s_setprio(3);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[rb], 0, 0, 0);      // (m, m)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[rb], 0, 0, 0);      // (h, l)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[rb], 0, 0, 0);      // (l, h)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[rb], 0, 0, 0);      // (h, m)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[rb], 0, 0, 0);      // (m, h)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[rb], 0, 0, 0);      // (h, h)
        if (c.xflags & 128) __builtin_amdgcn_s_setprio(3);
        else if (!(c.xflags & 2)) __builtin_amdgcn_s_setprio(0);
    }
}

// WRAP (timing experiment, experiments build only): the image's 8 chunks are walked twice -- the LDS fills, fragment
// reads and MFMAs of a panel of 256 columns, i.e. of TWO steps' down-dates applied in one pass over P (wrong numbers).
// DYN (round 3): a PERSISTENT grid that CLAIMS its tiles.  `ctr` (one counter per XCD list, zeroed before the launch)
// hands out the list positions nper, nper + 1, ... in order: the band-major walk stays what it was, and a workgroup
// that was given slow tiles simply claims fewer.  One workgroup per tile from the hardware dispatcher (round 2) balances
// as well, but every tile then pays a workgroup launch, an exposed first panel chunk and the drain of its 64 KB of
// stores before the CU slot is free again; a persistent workgroup requests the next tile's first chunks during this
// tile's last ones and lets its stores drain behind the next tile's MFMAs (dd_stream's point), which a STATIC split
// could not turn into time because it ends with its slowest workgroup.  The claim costs nothing: lane 0 of wave 0
// issues the returning atomic right after chunk 0's barrier, BEFORE the request of chunk 2 (vector-memory results
// return in order: anything issued behind the P tile would wait for the P tile; this sits in front of a panel chunk
// that is waited for anyway), stores the position to an LDS word before chunk 1's barrier, and every wave reads it
// after that barrier -- wave-uniform, so the list entry comes through the SCALAR cache, outside the in-order queue.
constexpr int CLAIM_OFF = 2 * 2 * IMG_CHUNK;        // the LDS word behind the two image buffers
// The claim word goes through these two: a VOLATILE access through a generic pointer is not rewritten to the LDS address
// space by the compiler and comes out as flat_store / flat_load followed by s_waitcnt vmcnt(0) -- which, in the middle of a
// tile, waited for the whole P tile just requested and for the previous tile's stores (found in round 4 in the ISA; the
// kernel's phases "adding instead of overlapping" was largely this wait).  With the address space spelled out they are
// ds_write_b32 / ds_read_b32 and touch lgkmcnt only.
typedef __attribute__((address_space(3))) unsigned lds_u32_t;
__device__ __forceinline__ void lds_store_u32(char* p, unsigned v) { *(volatile lds_u32_t*)p = v; }
__device__ __forceinline__ unsigned lds_load_u32(const char* p) { return *(const volatile lds_u32_t*)p; }
#ifndef DD_PCH0
#define DD_PCH0 0
#endif
#ifndef DD_DMA_PCH8       // (build-time A/B at eight chunks per tile: the LDS-DMA pipeline requests the P tile at the start of this step)
#define DD_DMA_PCH8 2
#endif
#ifndef DD_POFF8          // (build-time A/B at eight chunks per tile: the P tile is requested at chunk 8 - DD_POFF8)
#define DD_POFF8 (7 + DD_PCH0)
#endif

template <bool DBG, int NCH, int POFF, bool WRAP = false, bool DYN = false>
__device__ __forceinline__ void dd_stream_p(DdCtx& c, const int2* __restrict__ list, int L, int nper, int& slot, char* sm,
                                            unsigned* __restrict__ ctr = nullptr) {
    auto fetch = [&](int sl) { return sl < L ? list[sl] : make_int2(-1, -1); };
    auto request_chunk_p = [&](const DdCtx& cc, int2 t, int chunk, u32x4b (&gg)[3]) { ::request_chunk_p(cc, t, WRAP ? (chunk & 7) : chunk, gg); };
    constexpr int PCH = NCH - POFF > 0 ? NCH - POFF : 0;
    static_assert(!DYN || NCH >= 4, "the claimed position is read after chunk 1 and used at chunk NCH - 2");
    int2 tile = fetch(slot);
    int2 next = DYN ? make_int2(-1, -1) : fetch(slot + nper);
    int next_slot = slot + nper;
    unsigned claimed = 0;
    f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    float pold[2][16];
    u32x4b g[3];
    request_chunk_p(c, tile, 0, g);
    fill_lds_p(sm, 0, g);
    __syncthreads();
    request_chunk_p(c, tile, 1, g);
    int base = 0;
    for (;;) {
        bool next_off = next.x >= 0 && next.x != next.y;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int pb = (base + ch) & 1;
            if (ch == PCH && !(DBG && (c.dbg & 4))) load_p_mfma(c, tile.x * TILE, tile.y * TILE, pold);
            mfma_chunk_p<DBG>(c, sm, pb, acc, WRAP && ch >= 8);
            if (DYN && ch == 1 && threadIdx.x == 0) lds_store_u32(sm + CLAIM_OFF, claimed);
            if (ch < NCH - 1) {
                fill_lds_p(sm, pb ^ 1, g);
                __syncthreads();
                if (DYN && ch == 0 && threadIdx.x == 0)
                    claimed = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (DYN && ch == 1) {
                    next_slot = nper + __builtin_amdgcn_readfirstlane((int)lds_load_u32(sm + CLAIM_OFF));
                    next = fetch(next_slot);
                    next_off = next.x >= 0 && next.x != next.y;
                }
                if (ch + 2 < NCH) request_chunk_p(c, tile, ch + 2, g);
                else if (next_off) request_chunk_p(c, next, 0, g);
            } else {
                if (next_off) fill_lds_p(sm, pb ^ 1, g);
                __syncthreads();
                if (next_off) request_chunk_p(c, next, 1, g);
                if (!(DBG && (c.dbg & 1))) store_p_mfma(c, tile.x * TILE, tile.y * TILE, pold, acc);
            }
        }
        slot = DYN ? next_slot : slot + nper;
        if (!next_off) return;
        tile = next;
        if (!DYN) next = fetch(slot + nper);
        base = (base + NCH) & 1;
    }
}


// ---- the image path with the panel chunks moved by LDS-DMA, two chunks ahead (round 4) ------------------------------------------
// dd_stream_p gives a chunk ONE step to come from the L2 (requested after step c's barrier, written to LDS at the end of step
// c + 1) and pays three ds_write_b128 plus their wait per wave and chunk.  Here the chunks go global -> LDS directly
// (buffer_load_dwordx4 ... lds: the image IS the LDS layout, one wave instruction = 1 KB of it) into THREE buffers of 24 KB --
// the 72 KB this kernel's LDS array has anyway --, requested TWO steps ahead: after step c's barrier buffer c % 3 is free and
// receives chunk c + 3, chunk c + 1 has landed (waited for before that barrier), chunk c + 2 is on its way.  No staging
// registers, no LDS stores, and what sits between two chunks in a wave's in-order memory queue -- the P tile's 32 loads,
// its 32 stores -- has two steps to get out of the way instead of one.
//   The waits are COUNTED by hand (inline s_waitcnt; the barrier is a bare s_barrier: __syncthreads() would drain the queue):
// at the end of step c everything up to chunk c + 1's three pieces must be back, i.e. at most what was issued after them may
// be outstanding: chunk c + 2's three pieces, plus the P tile's 32 loads where they were issued in between (PCH = c - 1 or c),
// plus the previous tile's 32 stores (c = 0, 1).  The count has to be a LOWER bound of what was issued behind the chunk (a
// larger one lets the wait pass while the chunk is still on its way): it drops to what is really there on a workgroup's first
// tile (no stores yet) and last one (no chunks of a next tile).  Operations the count does not know -- wave 0's claim atomic,
// the one side-array store -- only make a wave wait for one operation more than it has to.
constexpr int DMA_BUF = 2 * IMG_CHUNK;          // [X image chunk | Y image chunk]
typedef __attribute__((address_space(3))) void lds_void_t;

// The DMA itself is the compiler's builtin -- so that its own counted waits (the P tile's loads before the stores, the claim
// atomic) include these operations --, but every LDS READ of this path is inline assembly: the compiler puts s_waitcnt vmcnt(0)
// in front of any LDS access it can see while an LDS-DMA is in flight (it cannot tell the buffers apart), which would be the
// pipeline gone.
__device__ __forceinline__ void dma_chunk(const DdCtx& c, int2 t, int chunk, char* sm, int buf, int wave) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.img), (short)0, 0x7fffffff, 0x00020000);
    const int sx = (t.x * c.img_nch + chunk) * IMG_CHUNK + wave * 1024, sy = (t.y * c.img_nch + chunk) * IMG_CHUNK + wave * 1024;
    const int lane16 = (threadIdx.x & 63) * 16;
    char* dst = sm + buf * DMA_BUF + wave * 1024;
    // the 1536 sixteen-byte pieces [X image | Y image] of a chunk: wave w moves pieces 64 w .. 64 w + 63 of each third
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, lane16, sx, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + 8192), 16, lane16, wave < 4 ? sx + 8192 : sy - 4096, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + 16384), 16, lane16, sy + 4096, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(unsigned long long)(const lds_void_t*)p; }
__device__ __forceinline__ void asm_lds_store_u32(char* p, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned asm_lds_load_u32(const char* p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {          // f(integral_constant<int, I>) for I = I .. N - 1, unrolled at compile time
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bare_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct DdFrags { bf16x8 a[3], b[3], b1[3]; };
#ifdef SLAMHIP_EXPERIMENTS
// EXPERIMENT (SLAMHIP_X bit 1048576, timing only, WRONG numbers): the step's matrix work as 24 v_mfma_f32_16x16x32_bf16 on the
// same fragment registers instead of 12 v_mfma_f32_32x32x16_bf16 -- the same pipe cycles per FLOP; MI355X_MICROARCH.md ('DVFS
// give-back' item 7) reports a higher clock held on the 16x16x32 shape.  Run with the stores off.
typedef float f32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_frags_16(const DdFrags& f, f32x16 (&acc)[2]) {
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        f32x4m c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = f32x4m{acc[rb][4 * q], acc[rb][4 * q + 1], acc[rb][4 * q + 2], acc[rb][4 * q + 3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bf16x8 bb = rb ? f.b1[q % 3] : f.b[q % 3];
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[0], bb, c[q], 0, 0, 0);
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[1], bb, c[q], 0, 0, 0);
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[2], bb, c[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][4 * q + r] = c[q][r];
    }
    __builtin_amdgcn_s_setprio(0);
}
#endif
__device__ __forceinline__ void read_frags_d(const DdCtx& c, const char* sm, int buf, DdFrags& f) {
    const unsigned base = lds_addr(sm) + buf * DMA_BUF + c.l31 * 32 + ((c.lh ^ (c.l31 >> 3)) & 1) * 16;
    const unsigned ya = base + IMG_CHUNK + (32 * c.wc) * 32, xa = base + (64 * c.wr) * 32;
    asm volatile(
        "ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:4096\n\tds_read_b128 %2, %9 offset:8192\n\t"
        "ds_read_b128 %3, %10\n\tds_read_b128 %4, %10 offset:4096\n\tds_read_b128 %5, %10 offset:8192\n\t"
        "ds_read_b128 %6, %10 offset:1024\n\tds_read_b128 %7, %10 offset:5120\n\tds_read_b128 %8, %10 offset:9216\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(f.a[0]), "=&v"(f.a[1]), "=&v"(f.a[2]), "=&v"(f.b[0]), "=&v"(f.b[1]), "=&v"(f.b[2]), "=&v"(f.b1[0]), "=&v"(f.b1[1]), "=&v"(f.b1[2])
        : "v"(ya), "v"(xa)
        : "memory");
}
__device__ __forceinline__ void mfma_frags_d(const DdFrags& f, f32x16 (&acc)[2]) {
#ifdef DD_MFMA_INTERLEAVE     // (build-time A/B: the two row blocks' chains interleaved, no MFMA directly behind the one it depends on)
    __builtin_amdgcn_s_setprio(3);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b[1], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b1[1], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[2], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b1[2], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[2], f.b[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[2], f.b1[0], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[1], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b1[1], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b1[0], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b1[0], acc[1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    return;
#endif
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        __builtin_amdgcn_s_setprio(3);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], rb ? f.b1[1] : f.b[1], acc[rb], 0, 0, 0);      // (m, m)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], rb ? f.b1[2] : f.b[2], acc[rb], 0, 0, 0);      // (h, l)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[2], rb ? f.b1[0] : f.b[0], acc[rb], 0, 0, 0);      // (l, h)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], rb ? f.b1[1] : f.b[1], acc[rb], 0, 0, 0);      // (h, m)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], rb ? f.b1[0] : f.b[0], acc[rb], 0, 0, 0);      // (m, h)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], rb ? f.b1[0] : f.b[0], acc[rb], 0, 0, 0);      // (h, h)
        __builtin_amdgcn_s_setprio(0);
    }
}

// PCH (2 <= PCH <= NCH - 2): the step at whose start the P tile is requested
// Measured and dropped (round 4, one box each, profiles/r04_downdate_dma_experiments.txt): the barrier between the fragment reads
// and the MFMAs; the second row block's fragment reads behind the first one's MFMAs; a tile's stores behind the next tile's first
// MFMAs instead of one burst -- all within 1 % of this form.
template <bool DBG, int NCH, int PCH>
__device__ __forceinline__ void dd_stream_dma(DdCtx& c, const int2* __restrict__ list, int L, int nper, int& slot, char* sm,
                                              unsigned* __restrict__ ctr, int wave, char* cw, unsigned long long* prof = nullptr) {
    static_assert(NCH >= 5 && PCH >= 2 && PCH <= NCH - 2, "");
#ifdef SLAMHIP_EXPERIMENTS
    // (SLAMHIP_STAMPS=1: where a wave's cycles go -- shader-clock sums over all its steps: fragment reads, MFMA issue, the wait
    //  for the chunk per step of the tile, barrier, DMA issue, stores)
    unsigned long long ph_lds = 0, ph_mfma = 0, ph_bar = 0, ph_issue = 0, ph_store = 0, ph_vm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_steps = 0;
    // (the in-kernel clock: shader-clock ticks over 100 MHz ticks around the whole stream, MI355X_MICROARCH.md 'DVFS give-back' item 6)
    const unsigned long long clk0 = prof ? __builtin_amdgcn_s_memtime() : 0, rt0 = prof ? __builtin_amdgcn_s_memrealtime() : 0;
#define STAMP(var) do { if (prof) { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STAMP(var) do { } while (0)
#endif
    constexpr int RD = NCH - 3;        // the step at whose end the next tile's first chunk is requested
    auto fetch = [&](int sl) { return sl < L ? list[sl] : make_int2(-1, -1); };
    int2 tile = fetch(slot);
    int2 next = make_int2(-1, -1);
    int next_slot = slot + nper;
    unsigned claimed = 0;
    f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    float pold[2][16];
    dma_chunk(c, tile, 0, sm, 0, wave);
    dma_chunk(c, tile, 1, sm, 1, wave);
    dma_chunk(c, tile, 2, sm, 2, wave);
    wait_vm<6>();
    bare_barrier();
    int base = 0;                      // the buffer that holds chunk 0 of the current tile
    bool first = true;
    for (;;) {
        bool next_off = false;
        static_for<0, NCH>([&](auto CH) {
            constexpr int ch = decltype(CH)::value;
            const int buf = (base + ch) % 3;
            [[maybe_unused]] unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
            STAMP(t0);
#ifdef SLAMHIP_EXPERIMENTS
            // switch-off experiments on THIS pipeline (SLAMHIP_X bits, wrong results except where P stays what it was):
            // 1024 no MFMAs, 2048 no stores, 4096 no P loads (with 2048 only), 8192 no fragment reads, 16384 no chunk requests
            const int xo = c.xflags;
#else
            constexpr int xo = 0;
#endif
#ifdef SLAMHIP_EXPERIMENTS
            if (ch == PCH && (xo & 524288)) load_p_mfma<0>(c, tile.x * TILE, tile.y * TILE, pold);                  // (cache policy of the P loads: default)
            else
#endif
            // (the hand-counted waits below assume the queue order [chunk request][P loads] and [chunk request][P stores]: nothing but
            //  these scheduling fences keeps the compiler from swapping two builtins that do not depend on each other)
            __builtin_amdgcn_sched_barrier(0);
            if (ch == PCH && !(DBG && (c.dbg & 4)) && !(xo & 4096)) load_p_mfma(c, tile.x * TILE, tile.y * TILE, pold);
            DdFrags fr;
            if (!(xo & 8192)) read_frags_d(c, sm, buf, fr);
            else     // (the registers are taken as they are: no instruction stands in for the reads)
                asm volatile("" : "=v"(fr.a[0]), "=v"(fr.a[1]), "=v"(fr.a[2]), "=v"(fr.b[0]), "=v"(fr.b[1]), "=v"(fr.b[2]), "=v"(fr.b1[0]), "=v"(fr.b1[1]), "=v"(fr.b1[2]));
            STAMP(t1);
#ifdef SLAMHIP_EXPERIMENTS
            if (xo & 1048576) mfma_frags_16(fr, acc);
            else
#endif
            if (!(xo & 1024)) mfma_frags_d(fr, acc);
            else asm volatile("" ::"v"(fr.a[0]), "v"(fr.a[1]), "v"(fr.a[2]), "v"(fr.b[0]), "v"(fr.b[1]), "v"(fr.b[2]), "v"(fr.b1[0]), "v"(fr.b1[1]), "v"(fr.b1[2]));
            STAMP(t2);
            if (ch == RD && threadIdx.x == 0) asm_lds_store_u32(cw, claimed);
            // chunk ch + 1 (this tile's, or the next tile's first) has landed when at most these remain outstanding
            constexpr int PL = (ch == PCH || ch == PCH + 1) ? 32 : 0;           // the P tile's loads sit behind it
            // (the timing experiments that switch the P stores / loads off do not issue those 32 operations: the count drops)
            const bool no_st = (DBG && (c.dbg & 1)) || (xo & 2048), no_ld = (DBG && (c.dbg & 4)) || (xo & 4096);
            if (ch <= 1) {                                                      // ... the previous tile's stores and chunk ch + 2
                if (first || no_st) wait_vm<3>();
                else wait_vm<35>();
            } else if (ch + 2 < NCH || next_off) {                              // (chunk ch + 2 is this tile's or the next tile's)
                if (no_ld) wait_vm<3>();
                else wait_vm<3 + PL>();
            } else if (no_ld) wait_vm<0>();
            else wait_vm<PL>();
            STAMP(t3);
            bare_barrier();
            STAMP(t4);
            if (ch == 0 && threadIdx.x == 0) claimed = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ch == RD) {          // (the claim of step 0 is read as late as the next tile's first chunk allows: its return
                                     //  comes behind the previous tile's stores in wave 0's queue)
                next_slot = nper + __builtin_amdgcn_readfirstlane((int)asm_lds_load_u32(cw));
                next = fetch(next_slot);
                next_off = next.x >= 0 && next.x != next.y;
            }
            if (!(xo & 16384)) {
                if (ch + 3 < NCH) dma_chunk(c, tile, ch + 3, sm, buf, wave);
                else if (next_off) dma_chunk(c, next, ch + 3 - NCH, sm, buf, wave);
            }
            __builtin_amdgcn_sched_barrier(0);                                  // (the chunk request stays in front of the tile's stores)
            STAMP(t5);
#ifdef DD_TIMING_BASE
            constexpr bool tbase = true;
#else
            constexpr bool tbase = false;
#endif
            if (ch == NCH - 1 && ((xo & 4194304) || tbase)) {          // (P stored back UNCHANGED, the accumulators consumed: the baseline of the 16x16x32 experiment)
                f32x16 zero[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { asm volatile("" ::"v"(acc[rb][r])); zero[rb][r] = 0.0f; acc[rb][r] = 0.0f; }
                store_p_mfma(c, tile.x * TILE, tile.y * TILE, pold, zero);
            } else
            if (ch == NCH - 1 && (xo & 2048)) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (!(xo & 4096)) asm volatile("" ::"v"(pold[rb][r] - acc[rb][r]));
                        else asm volatile("" ::"v"(acc[rb][r]));
                        acc[rb][r] = 0.0f;
                    }
            } else
#ifdef SLAMHIP_EXPERIMENTS
            if (ch == NCH - 1 && (xo & 65536)) store_p_mfma<0>(c, tile.x * TILE, tile.y * TILE, pold, acc);          // (cache policy of the P stores: default)
            else if (ch == NCH - 1 && (xo & 131072)) store_p_mfma<1>(c, tile.x * TILE, tile.y * TILE, pold, acc);   // (sc0)
            else if (ch == NCH - 1 && (xo & 262144)) store_p_mfma<3>(c, tile.x * TILE, tile.y * TILE, pold, acc);   // (sc0 nt)
            else
#endif
            if (ch == NCH - 1 && !(DBG && (c.dbg & 1))) store_p_mfma(c, tile.x * TILE, tile.y * TILE, pold, acc);
            STAMP(t6);
#ifdef SLAMHIP_EXPERIMENTS
            if (prof) {
                ph_lds += t1 - t0; ph_mfma += t2 - t1; ph_vm[ch < 8 ? ch : 7] += t3 - t2; ph_bar += t4 - t3; ph_issue += t5 - t4; ph_store += t6 - t5;
                ++ph_steps;
            }
#endif
        });
        slot = next_slot;
        if (!next_off) break;
        tile = next;
        base = (base + NCH) % 3;
        first = false;
    }
    wait_vm<0>();                      // (nothing of this path is in flight when the diagonal tiles take the LDS array over)
#ifdef SLAMHIP_EXPERIMENTS
    if (prof && (threadIdx.x & 63) == 0) {
        unsigned long long* o = prof + ((size_t)blockIdx.x * NWAVE + wave) * 16;
        o[0] = ph_steps; o[1] = ph_lds; o[2] = ph_mfma; o[3] = ph_bar; o[4] = ph_issue; o[5] = ph_store;
        for (int i = 0; i < 8; ++i) o[6 + i] = ph_vm[i];
        o[14] = __builtin_amdgcn_s_memtime() - clk0; o[15] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
#undef STAMP
}


#if defined(SLAMHIP_EXPERIMENTS) || defined(DD_TIMING_16)
#include "ekf_syrk_exp_streams.inc"
#endif

// As dd_stream: processes list[slot], list[slot + nper], ... while they are off-diagonal; NCH = kp / 16 (>= 2).  On
// entry gx/gy hold the request for chunk 0 of the first tile.  On return `slot` is the first unprocessed position;
// nothing is in flight for it (the diagonal tiles that follow use the fp32 pipeline and request their own panels).
// The P tile is streamed: read once, written once per launch.  Its loads and stores carry the non-temporal policy
// (aux = 2), which leaves the XCD's L2 to the panels (one-box A/B: loads -1.5 %, stores -2 %, both -3.5 % of the
// kernel's time, and the kernels that follow it gain as much again).
template <bool DBG, int NCH, int POFF, int AUXL = 2, int AUXS = 2>
__device__ __forceinline__ void dd_stream_b(DdCtx& c, const int2* __restrict__ list, int L, int nper, int& slot, char* sm,
                                            f32x4& gx, f32x4& gy) {
    auto fetch = [&](int sl) { return sl < L ? list[sl] : make_int2(-1, -1); };
    constexpr int PCH = NCH - POFF > 0 ? NCH - POFF : 0;  // the chunk at whose start the P tile is requested
    int2 tile = fetch(slot);
    int2 next = fetch(slot + nper);
    f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    float pold[2][16];
    fill_lds_b<DBG>(c, sm, 0, gx, gy);
    __syncthreads();
    request_chunk_b(c, tile, 1, gx, gy);
    int base = 0;
    for (;;) {
        const bool next_off = next.x >= 0 && next.x != next.y;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int pb = (base + ch) & 1;
            if (ch == PCH && !(DBG && (c.dbg & 4))) load_p_mfma<AUXL>(c, tile.x * TILE, tile.y * TILE, pold);
            mfma_chunk_b<DBG>(c, sm, pb, acc);
            if (ch < NCH - 1) {
                fill_lds_b<DBG>(c, sm, pb ^ 1, gx, gy);                  // chunk ch + 1 of this tile
                __syncthreads();
                if (ch + 2 < NCH) request_chunk_b(c, tile, ch + 2, gx, gy);
                else if (next_off) request_chunk_b(c, next, 0, gx, gy);
            } else {
                if (next_off) fill_lds_b<DBG>(c, sm, pb ^ 1, gx, gy);    // chunk 0 of the next tile
                __syncthreads();
                if (next_off) request_chunk_b(c, next, 1, gx, gy);
                if (!(DBG && (c.dbg & 1))) store_p_mfma<AUXS>(c, tile.x * TILE, tile.y * TILE, pold, acc);
            }
        }
        slot += nper;
        if (!next_off) return;
        tile = next;
        next = fetch(slot + nper);
        base = (base + NCH) & 1;
    }
}

// PERSISTENT kernel: gridDim.x = 8 * nper workgroups (two per CU: four waves per SIMD); workgroup b
// walks the tile list of XCD b % 8 with stride nper.  Memory operations of a wave are asynchronous: a
// wave that moves on to the next tile's MFMAs lets its stores drain behind them.  The first tile is
// peeled so that the loop header sees the same load/store history on both of its incoming edges and the
// compiler can emit counted vmcnt waits for the panel chunk instead of vmcnt(0).
template <bool DBG, int STREAM = 0, int POFF = 2, bool BF = false>   // STREAM = chunks per tile (2..4) for the streaming path, 0: dd_tile only; BF: split-bf16 streaming path where kp allows
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void downdate_f32_mfma(float* __restrict__ P, int ld, int n,
                                                                 const float* __restrict__ X, const float* __restrict__ Y,
                                                                 int pitch, int kp, const int2* __restrict__ tiles, int L,
                                                                 const int32_t* __restrict__ status, int dbg,
                                                                 unsigned long long* __restrict__ prof,
                                                                 const int32_t* __restrict__ dcount, int joseph,
                                                                 const char* __restrict__ img, int img_nch,
                                                                 unsigned* __restrict__ claim,        // non-null: the grid claims its tiles (dd_stream_p<DYN>)
                                                                 float* __restrict__ side, int side_n) {
    if (status[0] != 0) return;
    if (dcount) {                     // observe(): the host's kp is an upper bound
        const int k = 2 * dcount[0];
        kp = joseph ? 2 * ((k + SLAM_KPAD - 1) / SLAM_KPAD * SLAM_KPAD) : (k + 15) / 16 * 16;
        if (kp == 0) return;
    }
    __shared__ __attribute__((aligned(16))) float smem[2][2][TILE][LDSP];   // [buffer][X|Y][row][k]  73,728 B
    __shared__ unsigned dma_claim_word[4];                                   // (dd_stream_dma's claim word: its three buffers fill smem)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // provably wave-uniform
    DdCtx c;
    c.side = side; c.side_n = side_n;
    c.P = P; c.X = X; c.Y = Y; c.ld = ld; c.pitch = pitch; c.kp = kp; c.dbg = dbg & 0xff; c.xflags = dbg >> 8;      // (experiments build: bit 64 = no Y image loads)
    c.img = img; c.img_nch = img_nch;
    c.nchunks = (kp + KC - 1) / KC;                           // kp is a multiple of 16: the last chunk may be half
    c.wr = wave & 1;                  // row half of the tile
    c.wc = wave >> 1;                 // column quarter
    c.l31 = lane & 31;
    c.lh = lane >> 5;
    c.q = lane & 7;
    c.cl = lane >> 3;
    c.srow = tid >> 3;                // staging rows srow, srow + 64
    c.sc4 = tid & 7;
    c.t_head = c.t_wait = c.t_epi = c.t_total = 0;
    const unsigned long long rt0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0, mt0 = DBG ? __builtin_amdgcn_s_memtime() : 0;
    const int xcd = blockIdx.x & 7;
    const int rk = blockIdx.x >> 3;
    const int nper = gridDim.x >> 3;
    // persistent launches only (fewer workgroups than tiles): the second half of the grid starts ~3.4 us late (speed only)
    if (nper < L && rk >= (nper >> 1) && !(c.xflags & 1) && !claim) __builtin_amdgcn_s_sleep(127);
    const int2* list = tiles + (size_t)xcd * L;
    float* sD = &smem[0][0][0][0] + wave * (2 * 32 * SP);                   // per-wave scratches alias the panel buffers
    float* sV = sD + 32 * SP;
    (void)n;

    int slot = rk;
    int2 tile = slot < L ? list[slot] : make_int2(-1, -1);
    if (tile.x < 0) return;
    f32x4 gx[2], gy[2];
    if (BF && claim && img && kp >= 5 * KB && kp <= 8 * KB && !joseph) {
        // the product's launch at 80 <= k <= 128: a persistent grid that CLAIMS its tiles (dd_stream_p<DYN>)
        char* sm = reinterpret_cast<char*>(&smem[0][0][0][0]);
        unsigned* ctr = claim + 16 * xcd;                     // one counter per XCD list, 64 bytes apart
#ifdef SLAMHIP_EXPERIMENTS
        if (tile.x != tile.y && (c.xflags & 2097152) && kp == 8 * KB) {     // the 16x16x32 timing experiment (P unchanged)
            char* cw = reinterpret_cast<char*>(&dma_claim_word[0]);
            dd_stream_dma16<8, 2>(c, list, L, nper, slot, sm, ctr, wave, cw);
            return;                                                          // (diagonal tiles skipped: P stays what it was everywhere)
        } else
        if (tile.x != tile.y && (c.xflags & 32768)) {        // SLAMHIP_X bit 32768 (experiments build): the LDS-DMA pipeline without a barrier per step
            char* cw = reinterpret_cast<char*>(&dma_claim_word[0]);
            switch (kp / KB) {
                case 8: dd_stream_dma2<DBG, 8, 2>(c, list, L, nper, slot, sm, ctr, wave, cw); break;
                case 7: dd_stream_dma2<DBG, 7, 2>(c, list, L, nper, slot, sm, ctr, wave, cw); break;
                case 6: dd_stream_dma2<DBG, 6, 2>(c, list, L, nper, slot, sm, ctr, wave, cw); break;
                default: dd_stream_dma2<DBG, 5, 2>(c, list, L, nper, slot, sm, ctr, wave, cw); break;
            }
        } else
#endif
#ifdef DD_TIMING_16
        if (tile.x != tile.y && kp == 8 * KB) {
            char* cw = reinterpret_cast<char*>(&dma_claim_word[0]);
            dd_stream_dma16<8, 2>(c, list, L, nper, slot, sm, ctr, wave, cw);
            return;                                            // (diagonal tiles skipped: P stays what it was everywhere)
        } else
#endif
        if (tile.x != tile.y && !(c.xflags & 512)) {         // the LDS-DMA pipeline, chunks two steps ahead (SLAMHIP_X bit 512: round 3's register-staged pipeline below)
            static_assert(sizeof(smem) == 3 * DMA_BUF, "three 24 KB chunk buffers");
#ifdef SLAMHIP_EXPERIMENTS
#define DMA_PROF prof
#else
#define DMA_PROF nullptr
#endif
            char* cw = reinterpret_cast<char*>(&dma_claim_word[0]);
            switch (kp / KB) {
                case 8: dd_stream_dma<DBG, 8, DD_DMA_PCH8>(c, list, L, nper, slot, sm, ctr, wave, cw, DMA_PROF); break;
                case 7: dd_stream_dma<DBG, 7, 2>(c, list, L, nper, slot, sm, ctr, wave, cw, DMA_PROF); break;
                case 6: dd_stream_dma<DBG, 6, 2>(c, list, L, nper, slot, sm, ctr, wave, cw, DMA_PROF); break;
                default: dd_stream_dma<DBG, 5, 2>(c, list, L, nper, slot, sm, ctr, wave, cw, DMA_PROF); break;
            }
        } else
        if (tile.x != tile.y) {
            switch (kp / KB) {
                // (DD_PCH0 = 1, build-time A/B: the P tile requested at the tile's FIRST chunk instead of its second)
                case 8: dd_stream_p<DBG, 8, DD_POFF8, false, true>(c, list, L, nper, slot, sm, ctr); break;
                case 7: dd_stream_p<DBG, 7, 6 + DD_PCH0, false, true>(c, list, L, nper, slot, sm, ctr); break;
                case 6: dd_stream_p<DBG, 6, 5 + DD_PCH0, false, true>(c, list, L, nper, slot, sm, ctr); break;
                default: dd_stream_p<DBG, 5, 4 + DD_PCH0, false, true>(c, list, L, nper, slot, sm, ctr); break;
            }
        }
        // what is left for this workgroup: the list's diagonal tiles (fp32 pipeline, 1.3 % of the tiles), claimed one at a time
#ifdef DD_TIMING_BASE
        return;
#endif
#ifdef SLAMHIP_EXPERIMENTS
        if (c.xflags & (1024 | 2048 | 4194304)) return;        // (switch-off experiments: P stays what it was, diagonal tiles included)
#endif
        while (slot < L) {
            tile = list[slot];
            if (tile.x < 0) break;
            __syncthreads();                                   // every wave is done with the image buffers and the claim word
            request_chunk(c, tile, 0, gx, gy);
            if (tile.x != tile.y) dd_tile<false, DBG>(c, tile, make_int2(-1, -1), smem, sD, sV, gx, gy);
            else dd_tile<true, DBG>(c, tile, make_int2(-1, -1), smem, sD, sV, gx, gy);
            if (tid == 0)
                lds_store_u32(sm + CLAIM_OFF, __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __syncthreads();
            slot = nper + __builtin_amdgcn_readfirstlane((int)lds_load_u32(sm + CLAIM_OFF));
        }
        return;
    }
    if (BF && tile.x != tile.y && kp >= 5 * KB && kp <= 8 * KB && !joseph) {
        // off-diagonal tiles on the bf16 matrix cores (kp = 80, 96, 112 or 128); the diagonal ones follow below
        char* sm = reinterpret_cast<char*>(&smem[0][0][0][0]);
        if (img) {                     // the W1 kernel left the panel pre-split, as this path's LDS image
            switch (kp / KB) {
                case 8:
#ifdef SLAMHIP_EXPERIMENTS
                    if (DBG && (c.dbg & 128)) { dd_stream_p<DBG, 16, 15, true>(c, list, L, nper, slot, sm); break; }
#endif
                    dd_stream_p<DBG, 8, 7>(c, list, L, nper, slot, sm); break;
                case 7: dd_stream_p<DBG, 7, 6>(c, list, L, nper, slot, sm); break;
                case 6: dd_stream_p<DBG, 6, 5>(c, list, L, nper, slot, sm); break;
                default: dd_stream_p<DBG, 5, 4>(c, list, L, nper, slot, sm); break;
            }
        } else {
        request_chunk_b(c, tile, 0, gx[0], gy[0]);
        // (the P tile is requested at the start of the tile's second chunk: one-box A/B of offsets NCH-4 / NCH-2 /
        //  NCH-1 gave 0.411 / 0.408 / 0.402 ms)
        switch (kp / KB) {
            case 8: dd_stream_b<DBG, 8, 7>(c, list, L, nper, slot, sm, gx[0], gy[0]); break;
            case 7: dd_stream_b<DBG, 7, 6>(c, list, L, nper, slot, sm, gx[0], gy[0]); break;
            case 6: dd_stream_b<DBG, 6, 5>(c, list, L, nper, slot, sm, gx[0], gy[0]); break;
            default: dd_stream_b<DBG, 5, 4>(c, list, L, nper, slot, sm, gx[0], gy[0]); break;
        }
        }
        tile = slot < L ? list[slot] : make_int2(-1, -1);
        if (tile.x < 0) return;
    }
    request_chunk(c, tile, 0, gx, gy);
    slot += nper;
    int2 next = slot < L ? list[slot] : make_int2(-1, -1);
    // off-diagonal tiles first (the lists keep the diagonal ones at their end)
    if (STREAM >= 2 && tile.x != tile.y && c.nchunks == STREAM) {
        slot = rk;
        dd_stream<DBG, (STREAM >= 2 ? STREAM : 2), POFF>(c, list, L, nper, slot, smem, gx, gy);
        tile = slot < L ? list[slot] : make_int2(-1, -1);
        slot += nper;
        next = slot < L ? list[slot] : make_int2(-1, -1);
    } else if (tile.x != tile.y) {
        dd_tile<false, DBG>(c, tile, next, smem, sD, sV, gx, gy);               // peeled first tile
        while (next.x >= 0 && next.x != next.y) {
            tile = next;
            slot += nper;
            next = slot < L ? list[slot] : make_int2(-1, -1);
            dd_tile<false, DBG>(c, tile, next, smem, sD, sV, gx, gy);
        }
        tile = next;
        slot += nper;
        next = slot < L ? list[slot] : make_int2(-1, -1);
    }
    // diagonal tiles: same pipeline, element-wise masks where a 4-group straddles the diagonal
    while (tile.x >= 0) {
        dd_tile<true, DBG>(c, tile, next, smem, sD, sV, gx, gy);
        tile = next;
        slot += nper;
        next = slot < L ? list[slot] : make_int2(-1, -1);
    }
    if (DBG && prof && lane == 0) {                   // per wave: panel waits, lifetime (10 ns), epilogue, lifetime (clocks)
        unsigned long long* o = prof + ((size_t)blockIdx.x * NWAVE + wave) * 4;
        o[0] = c.t_head + c.t_wait; o[1] = __builtin_amdgcn_s_memrealtime() - rt0; o[2] = c.t_epi;
        o[3] = __builtin_amdgcn_s_memtime() - mt0;
    }
}


#ifdef SLAMHIP_EXPERIMENTS
#include "ekf_syrk_exp_half.inc"
#endif

// ---- fp64 down-date on the fp64 matrix cores ------------------------------------------
// 64 x 64 tile per 256-thread workgroup, wave (wr, wc) owns rows 32 wr.., columns 32 wc.. as 2 x 2 blocks of
// v_mfma_f64_16x16x4_f64 (exact fp64 FMAs).  Orientation as in the fp32 kernel: D[i][j] with i = COLUMN of P and
// j = ROW of P, so the 16 lanes of a quarter-wave hold 16 consecutive rows of one column = one 128-byte line.
// The P tile is requested before the k-loop; at the k of BASELINE.json's fp64 configuration (Joseph form, two live
// k-ranges of 16 columns) a tile needs 1 us of matrix-core time against 13 us of its HBM share, so several
// resident workgroups per CU are all the overlap this kernel needs.  (The VALU version this replaces was LDS-bound: 12.5 TFLOP/s, 1.7 TB/s.)
constexpr int DT = 64;     // tile edge
constexpr int DK = 16;     // k-chunk
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void downdate_f64_mfma(double* __restrict__ P, int ld, int n, const double* __restrict__ X,
                                                         const double* __restrict__ Y, int pitch, int kp,
                                                         const int2* __restrict__ tiles, int L,
                                                         const int32_t* __restrict__ status,
                                                         const int32_t* __restrict__ dcount, int joseph, int k16,
                                                         double* __restrict__ side, int side_n) {
    if (status[0] != 0) return;
    // Joseph form: X = [K | T], Y = [T | K], each half padded to SLAM_KPAD columns of which only the first
    // k16 = round_up(k, 16) are non-zero: the k-loop walks the two live ranges and skips the zero padding.
    int khalf = joseph ? kp / 2 : 0;
    if (dcount) {                     // observe(): the host's kp is an upper bound
        const int k = 2 * dcount[0];
        if (k == 0) return;
        k16 = (k + 15) / 16 * 16;
        khalf = joseph ? (k + SLAM_KPAD - 1) / SLAM_KPAD * SLAM_KPAD : 0;
    }
    const int2 tile = tiles[(size_t)(blockIdx.x & 7) * L + (blockIdx.x >> 3)];     // workgroup b -> list b % 8, slot b / 8
    if (tile.x < 0) return;
    __shared__ double sX[DT][DK + 1];
    __shared__ double sY[DT][DK + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int wr = wave & 1, wc = wave >> 1;
    const int R0 = tile.x * DT;
    const int C0 = tile.y * DT;
    const bool diag = tile.x == tile.y;
    double* Pt = P + tile_base(tile.x, tile.y, ld >> 6, 6);        // the tile: one contiguous 64 x 64 column-major block
    // P tile -> registers (rows/columns >= n are padding inside the allocation: P is allocated in whole tiles)
    f64x4 pold[2][2], acc[2][2];
    auto load_p = [&]() {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cl = 32 * wc + 16 * cb + 4 * r + kk, rl = 32 * wr + 16 * rb + li;
                    pold[cb][rb][r] = Pt[cl * DT + rl];             // (non-temporal loads/stores: no change here, 15.8 ms either way)
                }
    };
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) acc[cb][rb] = f64x4{0.0, 0.0, 0.0, 0.0};
    load_p();          // (requested AFTER the k-loop instead, at four workgroups per CU: 15.0 / 15.5 against 15.7 / 15.0 ms -- noise)
    for (int seg = 0; seg <= joseph; ++seg)
    for (int kc = seg * khalf; kc < seg * khalf + k16; kc += DK) {
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
        for (int s = 0; s < 4; ++s) {
            double ay[2], bx[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) ay[cb] = sY[32 * wc + 16 * cb + li][4 * s + kk];     // A[i = column][k]
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) bx[rb] = sX[32 * wr + 16 * rb + li][4 * s + kk];     // B[k][j = row]
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[cb][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay[cb], bx[rb], acc[cb][rb], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cl = 32 * wc + 16 * cb + 4 * r + kk, rl = 32 * wr + 16 * rb + li;
                const double val = pold[cb][rb][r] - acc[cb][rb][r];
                if (!diag || rl >= cl) Pt[cl * DT + rl] = val;
                // the packed 2 x 2 diagonal blocks (device_math.h: side_note): diagonal tiles, and the tile below one for the
                // landmark that straddles the tile boundary
                if ((diag && rl >= cl) || tile.x == tile.y + 1) side_note(side, side_n, R0 + rl, C0 + cl, val);
                // in-tile mirror of a diagonal tile: element (row rl, column cl) of the lower triangle also goes to (row cl,
                // column rl) -- a strided store straight from the registers (1 tile in 390 at C5; a staging buffer for it
                // cost 33 KB of LDS in EVERY workgroup and held the kernel at three workgroups per CU)
                if (diag && rl > cl) Pt[rl * DT + cl] = val;
            }
    (void)n;
}

// Tile lists: eight lists of equal length L (padded with -1), laid out [xcd][slot].  XCD x
// (= workgroup id % 8 under round-robin dispatch) walks super-rows of SR tile rows, largest
// first, column by column; its workgroups take slots rk, rk + nper, ...  Diagonal tiles come last.
// order 0: XCD x walks super-rows of SR tile rows column by column (row panels stay in its L2, every column panel
//          fetched is used SR times): the order of the persistent kernels (fp32 matrix cores, fp64).
// order 2: band-major -- column band J outer, tile rows inner -- with tile row I owned by XCD I % 8 (snake order over
//          16 rows, so the row lengths balance).  All workgroups of the chip then work in the same few column bands
//          (a band is a nearly contiguous 10 MB of the column-major matrix), every XCD keeps ITS 1/8 of the row
//          panels (1.9 MB) in its L2 for the whole launch and streams the band's column panel.  The order of the
//          split-bf16 path (one-box A/B tools/gpu_order.sh: -3 %; with one workgroup per tile: -11 %).
void build_tile_order(int T, std::vector<int2>& out, int order) {
    constexpr int NX = 8;
    const int SR = slam_exp_env("SLAMHIP_SR", 4);      // tile rows per super-row (experiments build only)
    const int nsr = (T + SR - 1) / SR;
    std::vector<std::vector<int2>> lists(NX);
    if (order == 2) {
        for (int J = 0; J < T; ++J)
            for (int I = J + 1; I < T; ++I) {
                const int r = I % (2 * NX);
                lists[r < NX ? r : 2 * NX - 1 - r].push_back(make_int2(I, J));
            }
    } else {
        std::vector<long> load(NX, 0);
        std::vector<std::vector<int>> mine(NX);
        for (int s = nsr - 1; s >= 0; --s) {
            const int I0 = s * SR, I1 = std::min(T, I0 + SR);
            long cnt = 0;
            for (int I = I0; I < I1; ++I) cnt += I;            // off-diagonal tiles of these rows
            int best = 0;
            for (int xcd = 1; xcd < NX; ++xcd)
                if (load[xcd] < load[best]) best = xcd;
            load[best] += cnt;
            mine[best].push_back(s);
        }
        for (int xcd = 0; xcd < NX; ++xcd)
            for (int s : mine[xcd]) {
                const int I0 = s * SR, I1 = std::min(T, I0 + SR);
                for (int J = 0; J < I1; ++J)
                    for (int I = std::max(I0, J + 1); I < I1; ++I) lists[xcd].push_back(make_int2(I, J));
            }
    }
    // the T diagonal tiles go to the END of the lists, shortest list first: they fill the ragged last round
    for (int I = 0; I < T; ++I) {
        int best = 0;
        for (int xcd = 1; xcd < NX; ++xcd)
            if (lists[xcd].size() < lists[best].size()) best = xcd;
        lists[best].push_back(make_int2(I, I));
    }
    size_t L = 0;
    for (int xcd = 0; xcd < NX; ++xcd) L = std::max(L, lists[xcd].size());
    out.assign(L * NX, make_int2(-1, -1));
    for (int xcd = 0; xcd < NX; ++xcd)
        for (size_t slot = 0; slot < lists[xcd].size(); ++slot) out[xcd * L + slot] = lists[xcd][slot];
}

int ensure_tile_order(slam_ekf* h, int T) {
    if (h->tiles && h->tiles_T == T) return SLAM_OK;
    std::vector<int2> order, orderB, orderH;
    build_tile_order(T, order, 0);
    build_tile_order(T, orderB, 2);
#ifdef SLAMHIP_EXPERIMENTS
    {   // the half-tile experiment's lists: the band-major order with every off-diagonal tile as its two 64-row halves
        std::vector<std::vector<int2>> lists(8);
        for (int J = 0; J < T; ++J)
            for (int I = J + 1; I < T; ++I) {
                const int r = I % 16;
                auto& l = lists[r < 8 ? r : 15 - r];
                l.push_back(make_int2(I, 2 * J));
                l.push_back(make_int2(I, 2 * J + 1));
            }
        size_t LH = 1;
        for (auto& l : lists) LH = std::max(LH, l.size());
        orderH.assign(LH * 8, make_int2(-1, -1));
        for (int x = 0; x < 8; ++x)
            for (size_t i = 0; i < lists[x].size(); ++i) orderH[x * LH + i] = lists[x][i];
    }
#endif
    const size_t total = order.size() + orderB.size() + orderH.size();
    HIP_TRY(hipStreamSynchronize(h->stream));          // earlier down-dates may still read the old list
    if ((int)total > h->tiles_cap) {
        if (h->tiles) (void)hipFree(h->tiles);
        h->tiles = nullptr;
        h->tiles_cap = 0;
        HIP_TRY(hipMalloc((void**)&h->tiles, sizeof(int2) * total));
        h->tiles_cap = (int)total;
    }
    HIP_TRY(hipMemcpy(h->tiles, order.data(), sizeof(int2) * order.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->tiles + order.size(), orderB.data(), sizeof(int2) * orderB.size(), hipMemcpyHostToDevice));
    if (!orderH.empty())
        HIP_TRY(hipMemcpy(h->tiles + order.size() + orderB.size(), orderH.data(), sizeof(int2) * orderH.size(), hipMemcpyHostToDevice));
    h->tilesH_off = (int)(order.size() + orderB.size());
    h->tilesH_len = (int)orderH.size() / 8;
    h->tiles_T = T;
    h->tiles_len = (int)order.size() / 8;          // L: entries per XCD list
    h->tilesB_off = (int)order.size();
    h->tilesB_len = (int)orderB.size() / 8;
    return SLAM_OK;
}

// ---- the copy floor (bench.py: roofline.copy_floor_ms) -------------------------------------------------------------------
// What the memory system alone asks for the down-date's P traffic ON THIS BOX, IN THIS RUN, on the handle's own buffer: every
// stored tile read once and written back unchanged (x * one, `one` = 1.0 from the kernel arguments so that the store is
// not elided; bit-exact for every value), 16 bytes per lane, non-temporal like the kernel's own P accesses, in the order
// the split-bf16 down-date walks them (band-major = one linear stream through the tile-major matrix).  No panels, no
// MFMAs, no LDS.  Two launch forms, as tools/micro_tilewalk.hip found them to differ: one workgroup per tile from the
// dispatcher, and the persistent grid of two workgroups per CU.  The boxes of the pool differ by +-6 %: the down-date's
// time divided by this floor does not.
// (walks 64 KiB units -- one fp32 tile, two fp64 tiles -- so that both dtypes move the same bytes per workgroup turn)
template <typename T>
__global__ __launch_bounds__(512) void tile_copy_floor_kernel(T* __restrict__ P, long long total_bytes, T one) {
    typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
    const long long nunits = (total_bytes + 65535) >> 16;
    for (long long t = blockIdx.x; t < nunits; t += gridDim.x) {
        char* base = reinterpret_cast<char*>(P) + (t << 16);
        const bool full = (t << 16) + 65536 <= total_bytes;          // (the last unit of an odd number of fp64 tiles is half)
        vec_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (full || u < 4) v[u] = __builtin_nontemporal_load(reinterpret_cast<vec_t*>(base) + u * 512 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (full || u < 4) __builtin_nontemporal_store(v[u] * one, reinterpret_cast<vec_t*>(base) + u * 512 + threadIdx.x);
    }
}

}  // namespace

// out = {milliseconds of the fastest pass, its launch form: 0 = one workgroup per tile / 1 = persistent}.  Synchronises.
#ifdef SLAMHIP_EXPERIMENTS
#include "ekf_syrk_exp_copy.inc"
#endif

int launch_copy_floor(slam_ekf* h, int reps, double out[2]) {
    const int n = 3 + 2 * h->N;
    const int tlog = h->dtype == SLAM_F32 ? 7 : 6, E = 1 << tlog;
    const long long T = (n + E - 1) / E;                       // tile rows in use
    const long long Tall = h->ld >> tlog;                      // tile rows of the allocation
    // Column band J stores its tiles from the diagonal one down: T - J of its Tall - J tiles are in use.  The walk covers the
    // bands 0 .. T-1 as ONE contiguous run -- what the down-date touches plus, where the allocation is taller than the map, the
    // bands' unused last tiles (zero padding that stays zero: one tile per band at N = 50k fp64, 0.06 % of the bytes; one launch
    // per band instead was 1563 launches of 10 us and no floor at all).
    const long long ntiles = T * Tall - T * (T - 1) / 2;
    const long long bytes = ntiles * (long long)E * E * (long long)h->esz, units = (bytes + 65535) >> 16;
    auto one_pass = [&](int form) {
        const long long grid = form == 0 ? units : std::min<long long>(units, 2 * h->num_cus);
#ifdef SLAMHIP_EXPERIMENTS
        if (h->dtype == SLAM_F32 && slam_exp_env("SLAMHIP_COPY_LAG", 0) == 2) {
            if (ensure_tile_order(h, (int)T) == SLAM_OK)
                hipLaunchKernelGGL(tile_copy_mfma_kernel, dim3(8 * (2 * h->num_cus / 8)), dim3(NTHREADS), 0, h->stream, (float*)h->P, h->ld,
                                   (const int2*)h->tiles + h->tilesB_off, h->tilesB_len, (float*)h->Pside, h->npad / 2);
            return;
        }
        if (h->dtype == SLAM_F32 && slam_exp_env("SLAMHIP_COPY_LAG", 0)) {
            hipLaunchKernelGGL(tile_copy_lag_kernel<float>, dim3((unsigned)std::min<long long>(units, 2 * h->num_cus)), dim3(512), 0, h->stream, (float*)h->P, bytes, 1.0f);
            return;
        }
#endif
        if (h->dtype == SLAM_F32)
            hipLaunchKernelGGL(tile_copy_floor_kernel<float>, dim3((unsigned)grid), dim3(512), 0, h->stream, (float*)h->P, bytes, 1.0f);
        else
            hipLaunchKernelGGL(tile_copy_floor_kernel<double>, dim3((unsigned)grid), dim3(512), 0, h->stream, (double*)h->P, bytes, 1.0);
    };
    // every pass between its own pair of events; the floor is the FASTEST pass (the boxes' memory clocks wander: the mean of ten
    // passes moved by 7 % between two calls on one box, the minimum by 0.3 %)
    std::vector<hipEvent_t> ev(reps + 1, nullptr);
    hipError_t e = hipSuccess;
    for (int r = 0; r <= reps && e == hipSuccess; ++r) e = hipEventCreate(&ev[r]);
    double best = 1e30;
    int best_form = 0;
    for (int form = 0; form < 2 && e == hipSuccess; ++form) {
        one_pass(form);                                                     // warm-up
        one_pass(form);
        e = hipEventRecord(ev[0], h->stream);
        for (int r = 0; r < reps && e == hipSuccess; ++r) {
            one_pass(form);
            e = hipEventRecord(ev[r + 1], h->stream);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventSynchronize(ev[reps]);
        for (int r = 0; r < reps && e == hipSuccess; ++r) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, ev[r], ev[r + 1]);
            if (e == hipSuccess && ms < best) { best = ms; best_form = form; }
        }
    }
    for (hipEvent_t x : ev)
        if (x) (void)hipEventDestroy(x);
    if (e != hipSuccess) {
        slam_set_error("HIP error in the copy-floor measurement: %s", hipGetErrorString(e));
        return SLAM_E_HIP;
    }
    out[0] = best;
    out[1] = (double)best_form;
    return SLAM_OK;
}

int launch_downdate(slam_ekf* h, int kp_total, const void* X, const void* Y, int pitch, const int32_t* dcount, int joseph, int k16,
                    const void* img) {
#define IMGARGS (const char*)img, h->kcap / 16, (unsigned*)nullptr, (float*)h->Pside, h->npad / 2
#define IMGARGS_CLAIM (const char*)img, h->kcap / 16, h->dd_claim, (float*)h->Pside, h->npad / 2
    const int n = 3 + 2 * h->N;
    const int edge = h->dtype == SLAM_F32 ? TILE : DT;
    const int rc = ensure_tile_order(h, (n + edge - 1) / edge);
    if (rc) return rc;
    KTimer t(h, SLAM_K_SYRK);
    if (h->dtype == SLAM_F32) {
        // persistent: two workgroups per CU (VGPR- and LDS-limited residency), never more than there are tiles
        // persistent: two workgroups per CU (VGPR- and LDS-limited residency), never more than there are tiles
        int per_xcd = slam_exp_env("SLAMHIP_PER_CU", 2) * h->num_cus / 8;      // (experiments build: 1 = one workgroup per CU)
        if (per_xcd > h->tiles_len) per_xcd = h->tiles_len;
        if (per_xcd < 1) per_xcd = 1;
#ifdef SLAMHIP_EXPERIMENTS
        if (h->debug_flags & 32) {  // timing experiments on the split-bf16 path (1 no stores, 2 no MFMAs, 4 no P loads, 16 no split), launched like the product's
            const bool bandB = slam_exp_env("SLAMHIP_ORDER", 2) != 0;
            const int2* lst = bandB ? (const int2*)h->tiles + h->tilesB_off : (const int2*)h->tiles;
            const int L = bandB ? h->tilesB_len : h->tiles_len;
            int wgs = slam_exp_env("SLAMHIP_WGS", L);
            if (wgs > L) wgs = L;
            hipLaunchKernelGGL((downdate_f32_mfma<true, 4, 3, true>), dim3(8 * wgs), dim3(NTHREADS), 0, h->stream, (float*)h->P, h->ld, n,
                               (const float*)X, (const float*)Y, pitch, kp_total, lst, L,
                               h->d_status, h->debug_flags & ~32, (unsigned long long*)nullptr, dcount, joseph, IMGARGS);
        }
        else if (h->debug_flags)      // timing experiments only (SLAMHIP_DEBUG): parts of the kernel switched off
            hipLaunchKernelGGL(downdate_f32_mfma<true>, dim3(8 * per_xcd), dim3(NTHREADS), 0, h->stream, (float*)h->P, h->ld, n,
                               (const float*)X, (const float*)Y, pitch, kp_total, (const int2*)h->tiles, h->tiles_len,
                               h->d_status, h->debug_flags, (unsigned long long*)h->dd_prof, dcount, joseph, IMGARGS);
        else
#endif
        if (!(h->xflags & 4) && kp_total > 32 && kp_total <= 128) {     // (observe(): kp_total is an upper bound; the kernel falls back to dd_tile if the real chunk count differs)
            // streaming (tile-boundary-free) path for the off-diagonal tiles, one instantiation per chunk count
            const int nch = (kp_total + KC - 1) / KC;
// (P tile requested three chunks before the epilogue at four chunks per tile, two otherwise: one-box A/B, tools/gpu_abx.sh:
            //  offsets 1 / 2 / 3 / 4 gave 0.449 / 0.440 / 0.434 / 0.453 ms)
#define DD_LAUNCH_STREAM4()                                                                                            \
    hipLaunchKernelGGL((downdate_f32_mfma<false, 4, 3>), dim3(8 * per_xcd), dim3(NTHREADS), 0, h->stream, (float*)h->P, h->ld, \
                       n, (const float*)X, (const float*)Y, pitch, kp_total, (const int2*)h->tiles, h->tiles_len,       \
                       h->d_status, h->xflags << 8, (unsigned long long*)nullptr, dcount, joseph, IMGARGS)
#define DD_LAUNCH_STREAM(NCH)                                                                                          \
    hipLaunchKernelGGL((downdate_f32_mfma<false, NCH>), dim3(8 * per_xcd), dim3(NTHREADS), 0, h->stream, (float*)h->P, h->ld, \
                       n, (const float*)X, (const float*)Y, pitch, kp_total, (const int2*)h->tiles, h->tiles_len,       \
                       h->d_status, h->xflags << 8, (unsigned long long*)nullptr, dcount, joseph, IMGARGS)
            if (nch >= 3 && !joseph && !(h->xflags & 8)) {         // split-bf16 path (SLAMHIP_X bit 8 switches it off)
                // HBM-bound: ONE workgroup per tile, handed out by the hardware dispatcher in the band-major order.
                // Measured against the persistent grid of 2 workgroups per CU walking the super-row lists (one-box
                // A/B, tools/gpu_r2b.sh): 0.341 against 0.386 ms.  A static split ends with its slowest workgroup (the
                // CUs do not get equal shares of the memory system; tools/micro_tilewalk.hip shows the same 17 % on a
                // bare read + rewrite of the tiles); the dispatcher keeps every CU busy to the end.  SLAMHIP_WGS =
                // workgroups per XCD list (64 = two per CU) and SLAMHIP_ORDER=0 restore the old launch for A/B runs.
                const bool bandB = slam_exp_env("SLAMHIP_ORDER", 2) != 0;
                const int2* lst = bandB ? (const int2*)h->tiles + h->tilesB_off : (const int2*)h->tiles;
                const int L = bandB ? h->tilesB_len : h->tiles_len;
                // Round 3: a persistent grid (two workgroups per CU) that CLAIMS its tiles from per-XCD counters in list
                // order (dd_stream_p<DYN>): the dispatcher's load balance without a workgroup launch, an exposed first
                // panel chunk and a store drain per tile.  SLAMHIP_WGS (experiments build): n > 0 = the static persistent
                // grid with n workgroups per list, 0 = one workgroup per tile from the dispatcher (round 2's launch).
                const int wgs_env = slam_exp_env("SLAMHIP_WGS", -1);
                const bool dyn = wgs_env < 0 && img != nullptr;
                int wgs = dyn ? per_xcd : (wgs_env > 0 ? wgs_env : L);
                if (wgs > L) wgs = L;
                if (wgs < 1) wgs = 1;
#ifdef SLAMHIP_EXPERIMENTS
                if (dyn && slam_exp_env("SLAMHIP_SP", 0)) {        // the software-pipelined one-workgroup-per-CU probe (timing only: P unchanged)
                    const int per = h->num_cus / 8;
                    hipLaunchKernelGGL(downdate_f32_sp, dim3(8 * per), dim3(NTHREADS), 0, h->stream, (float*)h->P, h->ld, lst, L, h->d_status,
                                       dcount, (const char*)img, h->kcap / 16, (float*)h->Pside, h->npad / 2);
                } else
                if (dyn && slam_exp_env("SLAMHIP_HALF", 0)) {      // the half-tile experiment (off-diagonal tiles only: WRONG results)
                    // SLAMHIP_HALF: 1 = as written (spills 6 registers), 2 = without stores, 3 = the P tile's second row block late
                    // (SPLITP), 4 = three workgroups per CU (170 registers)
                    const int hv = slam_exp_env("SLAMHIP_HALF", 0);
                    const int per = (hv == 4 ? 3 : 4) * h->num_cus / 8;
                    auto kern = hv == 2 ? downdate_f32_half<true, false, 4> : hv == 3 ? downdate_f32_half<false, true, 4>
                              : hv == 4 ? downdate_f32_half<false, false, 3> : downdate_f32_half<false, false, 4>;
                    hipLaunchKernelGGL(kern, dim3(8 * per), dim3(256), 0, h->stream, (float*)h->P, h->ld,
                                       (const int2*)h->tiles + h->tilesH_off, h->tilesH_len, h->d_status, dcount, kp_total, (const char*)img,
                                       h->kcap / 16, h->dd_claim, (float*)h->Pside, h->npad / 2);
                } else
#endif
                if (dyn) {                 // (the counters were zeroed by the W1 kernel that wrote the image: ekf_update.hip)
                    hipLaunchKernelGGL((downdate_f32_mfma<false, 4, 3, true>), dim3(8 * wgs), dim3(NTHREADS), 0, h->stream,
                                       (float*)h->P, h->ld, n, (const float*)X, (const float*)Y, pitch, kp_total, lst, L,
                                       h->d_status, h->xflags << 8,
#ifdef SLAMHIP_EXPERIMENTS
                                       (unsigned long long*)h->dd_prof,
#else
                                       (unsigned long long*)nullptr,
#endif
                                       dcount, joseph, IMGARGS_CLAIM);
                } else
                hipLaunchKernelGGL((downdate_f32_mfma<false, 4, 3, true>), dim3(8 * wgs), dim3(NTHREADS), 0, h->stream,
                                   (float*)h->P, h->ld, n, (const float*)X, (const float*)Y, pitch, kp_total, lst, L,
                                   h->d_status, h->xflags << 8, (unsigned long long*)nullptr, dcount, joseph, IMGARGS);
            }
            else if (nch == 4) DD_LAUNCH_STREAM4();
            else if (nch == 3) DD_LAUNCH_STREAM(3);
            else DD_LAUNCH_STREAM(2);
#undef DD_LAUNCH_STREAM
#undef DD_LAUNCH_STREAM4
        }
        else
            hipLaunchKernelGGL(downdate_f32_mfma<false>, dim3(8 * per_xcd), dim3(NTHREADS), 0, h->stream, (float*)h->P, h->ld, n,
                               (const float*)X, (const float*)Y, pitch, kp_total, (const int2*)h->tiles, h->tiles_len,
                               h->d_status, h->xflags << 8, (unsigned long long*)nullptr, dcount, joseph, IMGARGS);
    } else {
        const bool bandB = slam_exp_env("SLAMHIP_ORDER64", 0) == 2;      // experiments build only (speed only)
        const int2* lst = bandB ? (const int2*)h->tiles + h->tilesB_off : (const int2*)h->tiles;
        const int L = bandB ? h->tilesB_len : h->tiles_len;
        hipLaunchKernelGGL(downdate_f64_mfma, dim3(8 * L), dim3(256), 0, h->stream, (double*)h->P, h->ld, n,
                           (const double*)X, (const double*)Y, pitch, kp_total, lst, L,
                           h->d_status, dcount, joseph, joseph ? k16 : kp_total, (double*)h->Pside, h->npad / 2);
    }
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
#undef IMGARGS
#undef IMGARGS_CLAIM
}
