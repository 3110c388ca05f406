// Experiment (round 2): the memory side of the FastSLAM sweep at C4 -- 262 144 particles, 16 of 512 landmark records
// (5 floats each) read and rewritten per particle, nothing else (x -> x * 1.0001f) -- for the product's record layout
// and for candidates.  What does the ACCESS PATTERN alone cost, before any arithmetic?
//
//   layout 0  [landmark][field][particle]            (the product: five 1 MB-strided streams per landmark)
//   layout 1  [landmark][particle / 64][field][64]   (a wave's five fields contiguous: 1280 B)
//   layout 2  [landmark][particle / 256][field][256] (a workgroup's five fields contiguous: 5 KB)
// each with: all 16 records requested before the first use (deep) or one at a time (serial); plain or non-temporal;
// read-only; workgroups of 64 / 128 / 256 / 512.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int LAYOUT>
__device__ __forceinline__ size_t off(int l, int f, long p, long n) {
    if (LAYOUT == 0) return ((size_t)l * 5 + f) * n + p;
    if (LAYOUT == 1) return (size_t)l * 5 * n + (size_t)(p >> 6) * 320 + f * 64 + (p & 63);
    return (size_t)l * 5 * n + (size_t)(p >> 8) * 1280 + f * 256 + (p & 255);
}

template <int LAYOUT, int DEEP, int NT, int RO>
__global__ void sweep(float* lm, long n, const int* ids, int m, float* sink) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float acc = 0.f;
    if (DEEP) {
        for (int i0 = 0; i0 < m; i0 += DEEP) {
            float v[DEEP][5];
#pragma unroll
            for (int u = 0; u < DEEP; ++u)
#pragma unroll
                for (int f = 0; f < 5; ++f) {
                    const float* a = lm + off<LAYOUT>(ids[i0 + u], f, p, n);
                    v[u][f] = NT ? __builtin_nontemporal_load(a) : *a;
                }
#pragma unroll
            for (int u = 0; u < DEEP; ++u)
#pragma unroll
                for (int f = 0; f < 5; ++f) {
                    float* a = lm + off<LAYOUT>(ids[i0 + u], f, p, n);
                    const float w = v[u][f] * 1.0001f;
                    if (RO) acc += w;
                    else if (NT) __builtin_nontemporal_store(w, a);
                    else *a = w;
                }
        }
    }
    if (RO) sink[p] = acc;
}

template <int LAYOUT, int DEEP, int NT, int RO>
void run(const char* name, float* lm, long n, const int* ids, int m, float* sink, int wg) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = (int)((n + wg - 1) / wg);
    for (int r = 0; r < 5; ++r) sweep<LAYOUT, DEEP, NT, RO><<<grid, wg>>>(lm, n, ids + 16 * (r % 32), m, sink);
    hipEventRecord(a);
    for (int r = 0; r < 64; ++r) sweep<LAYOUT, DEEP, NT, RO><<<grid, wg>>>(lm, n, ids + 16 * (r % 32), m, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms / 64 * 1e3;
    const double bytes = (double)n * m * 5 * 4 * (RO ? 1 : 2);
    printf("%-44s wg %3d: %6.1f us  %5.2f TB/s\n", name, wg, us, bytes / (us * 1e-6) / 1e12);
}

int main() {
    const long n = 262144;
    const int NL = 512, m = 16;
    float* lm; int* ids; float* sink;
    hipMalloc(&lm, (size_t)NL * 5 * n * 4);
    hipMemset(lm, 0, (size_t)NL * 5 * n * 4);
    hipMalloc(&sink, n * 4);
    int h[512];
    for (int i = 0; i < 512; ++i) h[i] = i;            // step t observes landmarks 16 t .. 16 t + 15 (the bench's pattern)
    hipMalloc(&ids, sizeof(h));
    hipMemcpy(ids, h, sizeof(h), hipMemcpyHostToDevice);
    for (int wg : {64, 128, 256, 512}) {
        run<0, 1, 0, 0>("layout 0 serial", lm, n, ids, m, sink, wg);
        run<0, 2, 0, 0>("layout 0 two in flight", lm, n, ids, m, sink, wg);
        run<0, 4, 0, 0>("layout 0 four in flight", lm, n, ids, m, sink, wg);
        run<0, 16, 0, 0>("layout 0 all sixteen in flight", lm, n, ids, m, sink, wg);
    }
    run<0, 4, 1, 0>("layout 0 four in flight, non-temporal", lm, n, ids, m, sink, 256);
    run<0, 16, 1, 0>("layout 0 sixteen in flight, non-temporal", lm, n, ids, m, sink, 256);
    run<0, 4, 0, 1>("layout 0 four in flight, read only", lm, n, ids, m, sink, 256);
    run<0, 16, 0, 1>("layout 0 sixteen in flight, read only", lm, n, ids, m, sink, 256);
    for (int wg : {128, 256}) {
        run<1, 2, 0, 0>("layout 1 (wave-contiguous) two in flight", lm, n, ids, m, sink, wg);
        run<1, 4, 0, 0>("layout 1 (wave-contiguous) four in flight", lm, n, ids, m, sink, wg);
        run<1, 16, 0, 0>("layout 1 (wave-contiguous) sixteen", lm, n, ids, m, sink, wg);
    }
    run<2, 2, 0, 0>("layout 2 (wg-contiguous) two in flight", lm, n, ids, m, sink, 256);
    run<2, 4, 0, 0>("layout 2 (wg-contiguous) four in flight", lm, n, ids, m, sink, 256);
    run<2, 16, 0, 0>("layout 2 (wg-contiguous) sixteen", lm, n, ids, m, sink, 256);
    run<1, 4, 1, 0>("layout 1 four in flight, non-temporal", lm, n, ids, m, sink, 256);
    return 0;
}
