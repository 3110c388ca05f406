// Experiment: how much of the down-date's memory time is the ACCESS PATTERN?
// Reads and rewrites (x -> x * 1.0001f) the lower-triangle 128 x 128 tiles of an n x n fp32 matrix,
//   mode 0: column-major with leading dimension ld (a tile = 128 runs of 512 B, 80 KB apart), the pattern of
//           downdate_f32_mfma: 16-byte accesses, 8 lanes per 128-byte line;
//   mode 1: the same tiles stored tile-major (a tile = one contiguous 64 KiB block).
// Persistent grid, 512 workgroups x 512 threads, tiles strided like the real kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void tile_rw(float* P, int ld, int T, const int2* tiles, int ntiles, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 7, cl = lane >> 3, wr = wave & 1, wc = wave >> 1;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int2 tl = tiles[t];
        f32x4 v[2][4];
        if (mode == 0) {
            float* base = P + (size_t)(tl.y * 128) * ld + tl.x * 128;
            for (int rb = 0; rb < 2; ++rb)
                for (int s = 0; s < 4; ++s)
                    v[rb][s] = *(f32x4*)(base + (size_t)(32 * wc + 8 * s + cl) * ld + 64 * wr + 32 * rb + 4 * q);
            for (int rb = 0; rb < 2; ++rb)
                for (int s = 0; s < 4; ++s) {
                    v[rb][s] *= 1.0001f;
                    *(f32x4*)(base + (size_t)(32 * wc + 8 * s + cl) * ld + 64 * wr + 32 * rb + 4 * q) = v[rb][s];
                }
        } else {
            float* base = P + ((size_t)tl.x * (tl.x + 1) / 2 + tl.y) * 16384;
            for (int rb = 0; rb < 2; ++rb)
                for (int s = 0; s < 4; ++s)
                    v[rb][s] = *(f32x4*)(base + (32 * wc + 8 * s + cl) * 128 + 64 * wr + 32 * rb + 4 * q);
            for (int rb = 0; rb < 2; ++rb)
                for (int s = 0; s < 4; ++s) {
                    v[rb][s] *= 1.0001f;
                    *(f32x4*)(base + (32 * wc + 8 * s + cl) * 128 + 64 * wr + 32 * rb + 4 * q) = v[rb][s];
                }
        }
    }
}
int main() {
    const int n = 20003, T = (n + 127) / 128, ld = T * 128;
    float* P;
    hipMalloc(&P, (size_t)ld * ld * 4);
    hipMemset(P, 0, (size_t)ld * ld * 4);
    std::vector<int2> tiles;
    for (int I = 0; I < T; ++I)
        for (int J = 0; J <= I; ++J) tiles.push_back(make_int2(I, J));
    int2* d;
    hipMalloc(&d, tiles.size() * sizeof(int2));
    hipMemcpy(d, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) tile_rw<<<512, 512>>>(P, ld, T, d, (int)tiles.size(), mode);
        hipEventRecord(a);
        for (int rep = 0; rep < 10; ++rep) tile_rw<<<512, 512>>>(P, ld, T, d, (int)tiles.size(), mode);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = 2.0 * tiles.size() * 65536.0;
        printf("mode %d (%s): %.3f ms per pass, %.2f TB/s\n", mode, mode ? "tile-major" : "column-major", ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
