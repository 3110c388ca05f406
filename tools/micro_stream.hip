// Calibration: plain streaming rates on this chip for the byte counts of the N = 10k down-date (0.8 GB read + 0.8 GB
// written): in-place scale of a contiguous buffer, copy A -> B, read-only, write-only; plain and non-temporal; several
// grid shapes.  Tells how much of the tile kernel's 4.6-5.0 TB/s is the tile structure and how much is the chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NT, int U>
__global__ __launch_bounds__(512) void k(f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (MODE != 3 && j < n4) v[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j >= n4) continue;
            if (MODE == 0) { v[u] *= 1.0001f; if (NT) __builtin_nontemporal_store(v[u], a + j); else a[j] = v[u]; }
            if (MODE == 1) { if (NT) __builtin_nontemporal_store(v[u], b + j); else b[j] = v[u]; }
            if (MODE == 2) acc += v[u];
            if (MODE == 3) { const f32x4 c = {1.f, 2.f, 3.f, (float)j}; if (NT) __builtin_nontemporal_store(c, b + j); else b[j] = c; }
        }
    }
    if (MODE == 2 && acc.x == 12345.678f) b[0] = acc;
}
template <int MODE, int NT, int U>
void run(const char* name, f32x4* a, f32x4* b, size_t n4, int grid, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k<MODE, NT, U><<<grid, 512>>>(a, b, n4);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<MODE, NT, U><<<grid, 512>>>(a, b, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %s U=%d grid %5d: %.4f ms  %.2f TB/s\n", name, NT ? "nt   " : "plain", U, grid, ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
}
int main() {
    const size_t n4 = (size_t)200 * 1024 * 1024 / 4;    // 0.8 GiB of float4 = 50 Mi float4
    const size_t bytes1 = n4 * 16;
    f32x4 *a, *b;
    hipMalloc(&a, bytes1); hipMalloc(&b, bytes1);
    hipMemset(a, 0, bytes1); hipMemset(b, 0, bytes1);
    for (int grid : {512, 1024, 2048, 8192}) {
        run<0, 0, 8>("in-place scale", a, b, n4, grid, 2.0 * bytes1);
        run<0, 1, 8>("in-place scale", a, b, n4, grid, 2.0 * bytes1);
        run<1, 0, 8>("copy a->b", a, b, n4, grid, 2.0 * bytes1);
        run<1, 1, 8>("copy a->b", a, b, n4, grid, 2.0 * bytes1);
        run<2, 0, 8>("read only", a, b, n4, grid, 1.0 * bytes1);
        run<2, 1, 8>("read only", a, b, n4, grid, 1.0 * bytes1);
        run<3, 0, 8>("write only", a, b, n4, grid, 1.0 * bytes1);
        run<3, 1, 8>("write only", a, b, n4, grid, 1.0 * bytes1);
    }
    run<0, 1, 4>("in-place scale", a, b, n4, 2048, 2.0 * bytes1);
    run<0, 1, 16>("in-place scale", a, b, n4, 1024, 2.0 * bytes1);
    return 0;
}
