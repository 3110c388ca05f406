// device_math.h -- scalar geometry of the range-bearing observation model.
// Always evaluated in double, whatever the storage type of x and P.
#pragma once
#include <hip/hip_runtime.h>

#define SLAM_PI_D 3.14159265358979323846

// mpi_to_pi, src/common.jl:102-110: ONE conditional wrap, not a modulo.
__host__ __device__ inline double mpi_to_pi_d(double phi) {
    if (phi > SLAM_PI_D) return phi - 2.0 * SLAM_PI_D;
    if (phi < -SLAM_PI_D) return phi + 2.0 * SLAM_PI_D;
    return phi;
}

// The non-zero part of predict_observation (src/common.jl:146-162) for the
// landmark at (lx, ly) seen from pose (xv, yv, phi).
//   zp = [d; atan2(dy,dx) - phi]        (bearing NOT wrapped, :152)
//   Hv = [-dx/d -dy/d 0; dy/d2 -dx/d2 -1]   (row-major here: Hv[row*3+col])
//   Hf = [ dx/d  dy/d;  -dy/d2  dx/d2]      (row-major: Hf[row*2+col])
struct ObsModel {
    double zp[2];
    double Hv[6];
    double Hf[4];
};

__host__ __device__ inline ObsModel obs_model(double xv, double yv, double phi, double lx, double ly) {
    ObsModel o;
    const double dx = lx - xv;
    const double dy = ly - yv;
    const double d2 = dx * dx + dy * dy;
    const double d = sqrt(d2);
    o.zp[0] = d;
    o.zp[1] = atan2(dy, dx) - phi;
    const double xd = dx / d, yd = dy / d, xd2 = dx / d2, yd2 = dy / d2;
    o.Hv[0] = -xd;  o.Hv[1] = -yd;  o.Hv[2] = 0.0;
    o.Hv[3] = yd2;  o.Hv[4] = -xd2; o.Hv[5] = -1.0;
    o.Hf[0] = xd;   o.Hf[1] = yd;
    o.Hf[2] = -yd2; o.Hf[3] = xd2;
    return o;
}

// ---- storage of the covariance: TILE-MAJOR, block lower -----------------------------------------------------------------
// Only the square tiles (edge E = 2^L: 128 for fp32, 64 for fp64) ON and BELOW the diagonal exist (the rank-k down-date
// updates one triangle, like BLAS syrk).  Each tile is one contiguous E x E column-major block; the tiles of column band J
// follow each other, I = J (the diagonal tile, stored complete and symmetric), J + 1, ..., T - 1, band after band:
//     tile (I, J) is block number  J*T - J*(J-1)/2 + (I - J),      T = ld >> L  tile rows of the ALLOCATION.
// The down-date walks the bands in this order, so its P traffic is ONE linear stream through memory -- a 128 x 128 tile of a
// column-major matrix is 128 runs of 512 bytes 80 KB apart, which the memory system serves 7-10 % slower
// (tools/micro_tilewalk.hip) -- and half of the square matrix is never allocated.  `ld` (= npad, a multiple of 128) only
// carries T; every access to P goes through p_off / sym_at / p_store_sym.
__host__ __device__ inline size_t tile_base(int I, int J, int T, int L) {
    return ((size_t)J * (size_t)T - (size_t)J * (size_t)(J - 1) / 2 + (size_t)(I - J)) << (2 * L);
}

// offset of element (r, c); requires (r >> L) >= (c >> L)
__host__ __device__ inline size_t p_off(int ld, int L, int r, int c) {
    const int m = (1 << L) - 1;
    return tile_base(r >> L, c >> L, ld >> L, L) + ((size_t)(c & m) << L) + (size_t)(r & m);
}

// element (r, c) of the symmetric matrix, from whichever of (r, c) / (c, r) lies in a stored tile
template <typename T>
__device__ inline T sym_at(const T* __restrict__ P, int ld, int tile_log2, int r, int c) {
    return ((r >> tile_log2) >= (c >> tile_log2)) ? P[p_off(ld, tile_log2, r, c)] : P[p_off(ld, tile_log2, c, r)];
}

// store P[r, c] = P[c, r] = v wherever those positions exist (both inside a diagonal tile, one otherwise)
template <typename T>
__device__ inline void p_store_sym(T* __restrict__ P, int ld, int tile_log2, int r, int c, T v) {
    if ((r >> tile_log2) >= (c >> tile_log2)) P[p_off(ld, tile_log2, r, c)] = v;
    if ((c >> tile_log2) >= (r >> tile_log2) && r != c) P[p_off(ld, tile_log2, c, r)] = v;
}

// store P[r, c] = v if that position exists (its mirror is the caller's business)
template <typename T>
__device__ inline void p_store(T* __restrict__ P, int ld, int tile_log2, int r, int c, T v) {
    if ((r >> tile_log2) >= (c >> tile_log2)) P[p_off(ld, tile_log2, r, c)] = v;
}

// ---- the landmarks' 2 x 2 diagonal blocks, packed (SURVEY 7 "hard parts") -------------------------------------------------
// side[0][j] = P[f, f], side[1][j] = P[f+1, f], side[2][j] = P[f+1, f+1] with f = 3 + 2 j, three rows of side_n values each
// in the state's dtype.  In the tile-major matrix these entries sit on the diagonals of the diagonal tiles, 129 elements
// apart: every landmark of the gating sweep touched two 128-byte lines of its own for 12 useful bytes (6.5x the sweep's
// algorithmic bytes).  The side array is kept by EVERY writer of those entries -- the state upload (pack_kernel),
// add_features (augment_kernel) and the diagonal-tile epilogues of the down-dates -- through side_note(), and read by the
// sweep (ekf_gate.hip) as three coalesced rows.  The matrix itself stays complete: every other reader uses P.
template <typename T>
__device__ __forceinline__ void side_note(T* __restrict__ side, int side_n, int r, int c, T v) {
    if (c < 3) return;
    const int o = c - 3, j = o >> 1;
    if (r == c) side[(size_t)((o & 1) ? 2 : 0) * side_n + j] = v;
    else if (r == c + 1 && !(o & 1)) side[(size_t)side_n + j] = v;
}
