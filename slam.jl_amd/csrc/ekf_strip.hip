// ekf_strip.hip -- K7 predict and K8 add_features: O(n) strip kernels.
//
// predict()       src/ekf.jl:8-43
// add_features()  src/ekf.jl:84-122
//
// Both only touch the three pose rows/columns of P (plus the new rows/columns
// for add_features).  The COLUMN strip P[:, 0:3] lives in the tiles of the first
// column band (tile-major storage, device_math.h): 128 consecutive rows of a
// column are contiguous, so it is read and written coalesced; the row strip
// P[0:3, :] is its mirror and exists only inside the first (diagonal) tile.  The
// reference reads the row strip (:34) and re-allocates all of P per new feature
// (:108-109); here the capacity was allocated once at create.
#include "common.h"
#include "device_math.h"

namespace {

// add_features: thread t owns old state index c = t (0 .. n0-1) and writes, for
// every new feature a, the 2 x 1 cross block P[fa:fa+1, c] = Gv_a * P[0:3, c] and its
// mirror.  P[0:3, c] is read as P[c, 0:3] (symmetric).  Block 0 additionally writes
// the new-new blocks and the new entries of x.
template <typename T>
__global__ __launch_bounds__(256) void augment_kernel(T* __restrict__ x, T* __restrict__ P, int ld, int n0,
                                                       const double* __restrict__ zn, int nn, double R0, double R1,
                                                       double R2, double R3, unsigned long long* __restrict__ pmax, int tlog,
                                                       T* __restrict__ side, int side_n) {
    const double xv = (double)x[0], yv = (double)x[1], phi = (double)x[2];   // ekf.jl:88 (phi fixed for the call)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n0) {
        const double p0 = (double)P[p_off(ld, tlog, c, 0)];
        const double p1 = (double)P[p_off(ld, tlog, c, 1)];
        const double p2 = (double)P[p_off(ld, tlog, c, 2)];
        for (int a = 0; a < nn; ++a) {
            const double r = zn[2 * a], b = zn[2 * a + 1];
            const double s = sin(phi + b), co = cos(phi + b);
            // Gv = [1 0 -r*s; 0 1 r*c]                               (:102)
            const T v0 = (T)(p0 - r * s * p2);
            const T v1 = (T)(p1 + r * co * p2);
            const int fa = n0 + 2 * a;
            p_store_sym(P, ld, tlog, fa, c, v0);        // P[fa, c] (row of the new feature, :113,:117) and its mirror (:114,:118)
            p_store_sym(P, ld, tlog, fa + 1, c, v1);
        }
    }
    if (blockIdx.x != 0) return;
    double Pvv[3][3];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) Pvv[r][cc] = (double)P[p_off(ld, tlog, r, cc)];
    const double R[2][2] = {{R0, R2}, {R1, R3}};
    // pairs (a, b) with b <= a
    const int npairs = nn * (nn + 1) / 2;
    for (int pidx = threadIdx.x; pidx < npairs; pidx += blockDim.x) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= pidx) ++a;
        const int b = pidx - a * (a + 1) / 2;
        const double ra = zn[2 * a], ba = zn[2 * a + 1];
        const double sa = sin(phi + ba), ca = cos(phi + ba);
        const double Ga[2][3] = {{1.0, 0.0, -ra * sa}, {0.0, 1.0, ra * ca}};
        const int fa = n0 + 2 * a;
        if (a == b) {
            // P[rng,rng] = Gv*Pvv*Gv' + Gz*R*Gz'                      (:112)
            const double Gz[2][2] = {{ca, -ra * sa}, {sa, ra * ca}};
            double GP[2][3], GR[2][2];
            for (int r = 0; r < 2; ++r)
                for (int cc = 0; cc < 3; ++cc)
                    GP[r][cc] = Ga[r][0] * Pvv[0][cc] + Ga[r][1] * Pvv[1][cc] + Ga[r][2] * Pvv[2][cc];
            for (int r = 0; r < 2; ++r)
                for (int cc = 0; cc < 2; ++cc) GR[r][cc] = Gz[r][0] * R[0][cc] + Gz[r][1] * R[1][cc];
            for (int r = 0; r < 2; ++r)
                for (int cc = 0; cc < 2; ++cc) {
                    const double val = (GP[r][0] * Ga[cc][0] + GP[r][1] * Ga[cc][1] + GP[r][2] * Ga[cc][2]) +
                                       (GR[r][0] * Gz[cc][0] + GR[r][1] * Gz[cc][1]);
                    p_store(P, ld, tlog, fa + r, fa + cc, (T)val);      // (the entry above the diagonal exists inside a diagonal tile only)
                    if (r >= cc) side_note(side, side_n, fa + r, fa + cc, (T)val);
                    // the new landmark's variances enter the pre-gate's bound (ekf_gate.hip): bit pattern of a
                    // non-negative double orders like the integer; anything else disables the pre-gate (+inf)
                    if (r == cc) {
                        const double sv = (double)(T)val;
                        atomicMax(pmax, (unsigned long long)__double_as_longlong(sv >= 0.0 ? sv : __builtin_inf()));
                    }
                }
            x[fa] = (T)(xv + ra * ca);                                  // (:99)
            x[fa + 1] = (T)(yv + ra * sa);
        } else {
            // P[rng_a, rng_b] = Gv_a * P[0:3, rng_b],  P[0:3, rng_b] = (Gv_b*Pvv)'   (:114,:117 with rnm grown)
            const double rb = zn[2 * b], bb = zn[2 * b + 1];
            const double sb = sin(phi + bb), cb = cos(phi + bb);
            const double Gb[2][3] = {{1.0, 0.0, -rb * sb}, {0.0, 1.0, rb * cb}};
            const int fb = n0 + 2 * b;
            double GbP[2][3];   // Gv_b * Pvv
            for (int r = 0; r < 2; ++r)
                for (int cc = 0; cc < 3; ++cc)
                    GbP[r][cc] = Gb[r][0] * Pvv[0][cc] + Gb[r][1] * Pvv[1][cc] + Gb[r][2] * Pvv[2][cc];
            for (int r = 0; r < 2; ++r)
                for (int cc = 0; cc < 2; ++cc) {
                    // (Gv_a * (Gv_b Pvv)')[r][cc] = sum_t Ga[r][t] * GbP[cc][t]
                    const T val = (T)(Ga[r][0] * GbP[cc][0] + Ga[r][1] * GbP[cc][1] + Ga[r][2] * GbP[cc][2]);
                    p_store_sym(P, ld, tlog, fa + r, fb + cc, val);     // P[fa+r, fb+cc] and its mirror (:118)
                }
        }
    }
}

// Column-major <-> tile-major.  pack: every element of every stored tile (rows / columns >= n: the zero padding) from a
// column-major n x n source; a workgroup owns 32 x 32 elements of one tile (consecutive threads walk rows: coalesced on
// both sides).  unpack: the full symmetric n x n matrix, each element from whichever tile stores it.
// Both work on a BAND of columns [cf, cf + 32 gridDim.y): src / dst hold that band only (column cf first), so the state of
// a large map is packed / unpacked through a bounded staging buffer (slam_ekf_set_state / get_state).
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(T* __restrict__ P, int ld, int tlog, const T* __restrict__ src, int lds, int n,
                                                    int cf, T* __restrict__ side, int side_n) {
    const int r0 = 32 * blockIdx.x, c0 = cf + 32 * blockIdx.y;
    if ((r0 >> tlog) < (c0 >> tlog)) return;                           // (32 divides the tile edge)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + tx, c = c0 + j;
        const T v = (r < n && c < n) ? src[(size_t)(c - cf) * lds + r] : (T)0;
        P[p_off(ld, tlog, r, c)] = v;
        side_note(side, side_n, r, c, v);                              // the packed 2 x 2 diagonal blocks follow the upload
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_kernel(const T* __restrict__ P, int ld, int tlog, T* __restrict__ dst_band, int ldd,
                                                      int n, int cf) {
    __shared__ T sh[32][33];
    T* __restrict__ dst = dst_band;
    const int r0 = 32 * blockIdx.x, c0 = cf + 32 * blockIdx.y;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool lower = (r0 >> tlog) >= (c0 >> tlog);
    if (lower) {
        for (int j = ty; j < 32; j += 8) {
            const int r = r0 + tx, c = c0 + j;
            if (r < n && c < n) dst[(size_t)(c - cf) * ldd + r] = P[p_off(ld, tlog, r, c)];
        }
        return;
    }
    // a block above the diagonal tiles: read its mirror (rows c0.., columns r0..) coalesced, transpose through LDS
    for (int j = ty; j < 32; j += 8) {
        const int rr = c0 + tx, cc = r0 + j;                            // element (rr, cc) of the stored triangle
        sh[j][tx] = (rr < n && cc < n) ? P[p_off(ld, tlog, rr, cc)] : (T)0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + tx, c = c0 + j;
        if (r < n && c < n) dst[(size_t)(c - cf) * ldd + r] = sh[tx][j];      // P[r, c] = P[c, r]
    }
}

// Telemetry (sim/browser/wsserver.jl:60-65,72-85): the covariance ellipse of the vehicle position (index 0) and of
// every landmark j >= 1 straight from the 2 x 2 diagonal blocks -- 5 values per landmark instead of shipping P.
//   l, u = eig(P_jj) (ascending);  out = [cx, cy, sqrt(l1), sqrt(l2), atan2(u[2,1], u[1,1])]
// The sign of an eigenvector is arbitrary (LAPACK's choice in the reference); here u[1,1] >= 0, so phi lies in
// [-pi/2, pi/2] -- the same ellipse.  Closed form for the symmetric 2 x 2 block, in double.
template <typename T>
__global__ __launch_bounds__(256) void ellipse_kernel(const T* __restrict__ x, const T* __restrict__ P, int ld, int N,
                                                       double* __restrict__ out, int tlog) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;          // 0: vehicle, 1..N: landmarks
    if (j > N) return;
    const int f = j == 0 ? 0 : 3 + 2 * (j - 1);
    const double a = (double)P[p_off(ld, tlog, f, f)], b = (double)P[p_off(ld, tlog, f + 1, f)],
                 d = (double)P[p_off(ld, tlog, f + 1, f + 1)];
    const double tr = a + d, df = a - d;
    const double disc = sqrt(df * df + 4.0 * b * b);
    const double l2 = 0.5 * (tr + disc);
    const double l1 = l2 > 0.0 ? (a * d - b * b) / l2 : 0.5 * (tr - disc);       // det / l2: no cancellation
    // eigenvector of l1: (b, l1 - a) or (l1 - d, b), whichever is longer
    double u0 = b, u1 = l1 - a;
    const double w0 = l1 - d, w1 = b;
    if (w0 * w0 + w1 * w1 > u0 * u0 + u1 * u1) { u0 = w0; u1 = w1; }
    if (u0 == 0.0 && u1 == 0.0) { u0 = a <= d ? 1.0 : 0.0; u1 = a <= d ? 0.0 : 1.0; }     // isotropic / diagonal block
    if (u0 < 0.0 || (u0 == 0.0 && u1 < 0.0)) { u0 = -u0; u1 = -u1; }
    double* o = out + (size_t)5 * j;
    o[0] = (double)x[f];
    o[1] = (double)x[f + 1];
    o[2] = sqrt(l1 > 0.0 ? l1 : 0.0);
    o[3] = sqrt(l2 > 0.0 ? l2 : 0.0);
    o[4] = atan2(u1, u0);
}

// slam_ekf_get_block / slam_ekf_get_diag: element (r, c) of the SYMMETRIC matrix, wherever it is maintained (sym_at),
// gathered into a dense column-major nr x nc buffer (nc = 1, diag = 1: the diagonal).  Consecutive threads walk rows.
template <typename T>
__global__ __launch_bounds__(256) void block_gather_kernel(const T* __restrict__ P, int ld, int tile_log2, int r0, int c0, int nr,
                                                            int nc, int diag, T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= nr || j >= nc) return;
    const int r = r0 + i, c = diag ? r : c0 + j;
    out[(size_t)j * nr + i] = sym_at(P, ld, tile_log2, r, c);
}

}  // namespace

int launch_block_gather(slam_ekf* h, int r0, int c0, int nr, int nc, int diag, void* d_out) {
    const dim3 grid((nr + 255) / 256, nc);
    if (h->dtype == SLAM_F32)
        hipLaunchKernelGGL(block_gather_kernel<float>, grid, dim3(256), 0, h->stream, (const float*)h->P, h->ld, 7, r0, c0, nr, nc,
                           diag, (float*)d_out);
    else
        hipLaunchKernelGGL(block_gather_kernel<double>, grid, dim3(256), 0, h->stream, (const double*)h->P, h->ld, 6, r0, c0, nr,
                           nc, diag, (double*)d_out);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

int launch_ellipses(slam_ekf* h, double* d_out) {
    const int cnt = h->N + 1;
    if (h->dtype == SLAM_F32)
        hipLaunchKernelGGL(ellipse_kernel<float>, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, (const float*)h->x,
                           (const float*)h->P, h->ld, h->N, d_out, 7);
    else
        hipLaunchKernelGGL(ellipse_kernel<double>, dim3((cnt + 255) / 256), dim3(256), 0, h->stream, (const double*)h->x,
                           (const double*)h->P, h->ld, h->N, d_out, 6);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

// The packed 2 x 2 diagonal blocks rebuilt FROM the matrix (slam_ekf_state_written: a caller has written landmark rows of
// P through the raw views): one thread per landmark, three reads of the diagonal tile that holds them (the landmark whose
// two rows straddle a tile boundary takes P[f+1, f] from the tile below).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void side_rebuild_kernel(const T* __restrict__ P, int ld, int tlog, int N, T* __restrict__ side, int side_n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const int f = 3 + 2 * j;
    side[j] = P[p_off(ld, tlog, f, f)];
    side[(size_t)side_n + j] = P[p_off(ld, tlog, f + 1, f)];
    side[(size_t)2 * side_n + j] = P[p_off(ld, tlog, f + 1, f + 1)];
}
}  // namespace

int launch_side_rebuild(slam_ekf* h) {
    if (h->N == 0) return SLAM_OK;
    const dim3 grid((h->N + 255) / 256);
    if (h->dtype == SLAM_F32)
        hipLaunchKernelGGL(side_rebuild_kernel<float>, grid, dim3(256), 0, h->stream, (const float*)h->P, h->ld, 7, h->N, (float*)h->Pside, h->npad / 2);
    else
        hipLaunchKernelGGL(side_rebuild_kernel<double>, grid, dim3(256), 0, h->stream, (const double*)h->P, h->ld, 6, h->N, (double*)h->Pside, h->npad / 2);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

// columns [cf, cf + ncols) of the tile-major state (cf, ncols multiples of 32; up to npad: the zero padding is written too)
// from d_src, which holds that band column-major with leading dimension lds
int launch_pack(slam_ekf* h, const void* d_src, int lds, int n, int cf, int ncols) {
    const dim3 grid(h->npad / 32, (ncols + 31) / 32);
    if (h->dtype == SLAM_F32)
        hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(256), 0, h->stream, (float*)h->P, h->ld, 7, (const float*)d_src, lds, n, cf,
                           (float*)h->Pside, h->npad / 2);
    else
        hipLaunchKernelGGL(pack_kernel<double>, grid, dim3(256), 0, h->stream, (double*)h->P, h->ld, 6, (const double*)d_src, lds, n, cf,
                           (double*)h->Pside, h->npad / 2);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

// columns [cf, cf + ncols) of the full symmetric matrix into d_dst (that band, column-major, leading dimension ldd)
int launch_unpack(slam_ekf* h, void* d_dst, int ldd, int n, int cf, int ncols) {
    const dim3 grid((n + 31) / 32, (ncols + 31) / 32);
    if (h->dtype == SLAM_F32)
        hipLaunchKernelGGL(unpack_kernel<float>, grid, dim3(256), 0, h->stream, (const float*)h->P, h->ld, 7, (float*)d_dst, ldd, n, cf);
    else
        hipLaunchKernelGGL(unpack_kernel<double>, grid, dim3(256), 0, h->stream, (const double*)h->P, h->ld, 6, (double*)d_dst, ldd, n, cf);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

namespace {

// E1 as ONE launch (the reference's sim! calls predict nine times per observation step: launches are what it costs).
// Every workgroup needs the PRE-update heading for the strip; the pose and P_vv are rewritten by whichever
// workgroup arrives LAST at the counter, i.e. after every workgroup has read x[2].  Nothing is published between
// workgroups (the strip and the pose block are disjoint), so the arrival is a relaxed atomic without fences.
template <typename T>
__global__ __launch_bounds__(256) void predict_kernel(T* __restrict__ x, T* __restrict__ P, int ld, int n, double v, double g,
                                                       double w, double Q0, double Q1, double Q2, double Q3, double dt,
                                                       int32_t* __restrict__ arrive, int tlog) {
    const double phi = (double)x[2];
    const double sn = sin(g + phi), cs = cos(g + phi);
    const double vts = v * dt * sn, vtc = v * dt * cs;
    const int f = 3 + blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) {
        const double p0 = (double)P[p_off(ld, tlog, f, 0)];
        const double p1 = (double)P[p_off(ld, tlog, f, 1)];
        const double p2 = (double)P[p_off(ld, tlog, f, 2)];
        const T n0 = (T)(p0 - vts * p2);     // Gv = [1 0 -vts; 0 1 vtc; 0 0 1]
        const T n1 = (T)(p1 + vtc * p2);
        const T n2 = (T)p2;
        p_store_sym(P, ld, tlog, f, 0, n0);  // column strip P[f, 0:3]; its mirror P[0:3, f] exists inside the first tile only
        p_store_sym(P, ld, tlog, f, 1, n1);
        p_store_sym(P, ld, tlog, f, 2, n2);
    }
    __shared__ int last;
    __syncthreads();                         // every thread of the workgroup has its heading
    if (threadIdx.x == 0)
        last = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // re-armed
    const double Gv[3][3] = {{1.0, 0.0, -vts}, {0.0, 1.0, vtc}, {0.0, 0.0, 1.0}};
    const double Gu[3][2] = {{dt * cs, -vts}, {dt * sn, vtc}, {dt * sin(g) / w, v * dt * cos(g) / w}};
    const double Q[2][2] = {{Q0, Q2}, {Q1, Q3}};   // column-major input
    double Pvv[3][3], GP[3][3], GQ[3][2], out[3][3];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) Pvv[r][cc] = (double)P[p_off(ld, tlog, r, cc)];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc)
            GP[r][cc] = Gv[r][0] * Pvv[0][cc] + Gv[r][1] * Pvv[1][cc] + Gv[r][2] * Pvv[2][cc];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 2; ++cc) GQ[r][cc] = Gu[r][0] * Q[0][cc] + Gu[r][1] * Q[1][cc];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc)
            out[r][cc] = (GP[r][0] * Gv[cc][0] + GP[r][1] * Gv[cc][1] + GP[r][2] * Gv[cc][2]) +
                         (GQ[r][0] * Gu[cc][0] + GQ[r][1] * Gu[cc][1]);
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) P[p_off(ld, tlog, r, cc)] = (T)out[r][cc];
    const double x0 = (double)x[0], x1 = (double)x[1];
    x[0] = (T)(x0 + vtc);
    x[1] = (T)(x1 + vts);
    x[2] = (T)mpi_to_pi_d(phi + v * dt * sin(g) / w);
}

}  // namespace

int launch_predict(slam_ekf* h, double v, double g, double w, const double Q[4], double dt) {
    const int n = 3 + 2 * h->N;
    {
        KTimer t(h, SLAM_K_PREDICT);
        const int blocks = h->N > 0 ? (2 * h->N + 255) / 256 : 1;
        int32_t* arrive = h->d_count + 3;
        if (h->dtype == SLAM_F32)
            hipLaunchKernelGGL(predict_kernel<float>, dim3(blocks), dim3(256), 0, h->stream, (float*)h->x, (float*)h->P, h->ld, n,
                               v, g, w, Q[0], Q[1], Q[2], Q[3], dt, arrive, 7);
        else
            hipLaunchKernelGGL(predict_kernel<double>, dim3(blocks), dim3(256), 0, h->stream, (double*)h->x, (double*)h->P, h->ld,
                               n, v, g, w, Q[0], Q[1], Q[2], Q[3], dt, arrive, 6);
    }
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

int launch_augment(slam_ekf* h, int nn, const double R[4], const double* zn_dev) {
    const int n0 = 3 + 2 * h->N;
    const int blocks = (n0 + 255) / 256;
    {
        KTimer t(h, SLAM_K_AUGMENT);
        if (h->dtype == SLAM_F32)
            hipLaunchKernelGGL(augment_kernel<float>, dim3(blocks), dim3(256), 0, h->stream, (float*)h->x, (float*)h->P,
                               h->ld, n0, zn_dev, nn, R[0], R[1], R[2], R[3], (unsigned long long*)h->d_pmax, 7, (float*)h->Pside,
                               h->npad / 2);
        else
            hipLaunchKernelGGL(augment_kernel<double>, dim3(blocks), dim3(256), 0, h->stream, (double*)h->x,
                               (double*)h->P, h->ld, n0, zn_dev, nn, R[0], R[1], R[2], R[3], (unsigned long long*)h->d_pmax, 6,
                               (double*)h->Pside, h->npad / 2);
    }
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}
