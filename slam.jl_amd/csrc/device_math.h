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

// P is stored "block lower": the square tiles (edge 2^tile_log2: 128 for fp32, 64 for fp64) on and
// below the diagonal are maintained, the tiles above it are not (the rank-k down-date updates one
// triangle, like BLAS syrk).  Every read of P goes through this: element (r, c) of the symmetric matrix.
template <typename T>
__device__ inline T sym_at(const T* __restrict__ P, int ld, int tile_log2, int r, int c) {
    return ((r >> tile_log2) >= (c >> tile_log2)) ? P[(size_t)c * ld + r] : P[(size_t)r * ld + c];
}
