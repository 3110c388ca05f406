// Micro-benchmark: what does one step of the register-resident elimination cost on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_factor tools/micro_factor.hip && /tmp/micro_factor
// Variants (all: one workgroup, k = 110 steps on a 128 x 128 double matrix):
//   0  barrier only                      (nthreads threads)
//   1  barrier + publish + pivot read    (LDS round trip per step)
//   2  + reciprocal
//   3  full step, 256 workers of `nthreads`
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NB, int VAR>
__global__ void elim(double* out, int k, int nthreads_work) {
    __shared__ double rowbuf[2][128], colbuf[2][128];
    const int tid = threadIdx.x;
    const bool worker = tid < nthreads_work;
    const int tx = tid & 15, ty = (tid >> 4) & 15;
    double a[NB][NB];
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int v = 0; v < NB; ++v) a[u][v] = (ty + 16 * u == tx + 16 * v) ? 200.0 + tid : 1.0 / (1 + ty + 16 * u + tx + 16 * v);
#pragma unroll
    for (int ub = 0; ub < NB; ++ub) {
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * ub + jj;
            if (j >= k) break;
            double* rb = rowbuf[j & 1];
            double* cb = colbuf[j & 1];
            if (VAR >= 1) {
                if (worker && ty == jj) {
#pragma unroll
                    for (int v = 0; v < NB; ++v) rb[tx + 16 * v] = a[ub][v];
                }
                if (worker && tx == jj) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) cb[ty + 16 * u] = a[u][ub];
                }
            }
            __syncthreads();
            if (VAR >= 1 && worker) {
                const double piv = rb[j];
                double rp = piv;
                if (VAR >= 2) {
                    rp = __builtin_amdgcn_rcp(piv);
                    rp = __builtin_fma(rp, __builtin_fma(-piv, rp, 1.0), rp);
                    rp = __builtin_fma(rp, __builtin_fma(-piv, rp, 1.0), rp);
                }
                if (VAR >= 3) {
                    double rowv[NB], colv[NB];
#pragma unroll
                    for (int v = 0; v < NB; ++v) rowv[v] = rb[tx + 16 * v];
#pragma unroll
                    for (int u = 0; u < NB; ++u) colv[u] = cb[ty + 16 * u];
                    const bool fix = tx == jj;
#pragma unroll
                    for (int u = ub; u < NB; ++u) {
                        const bool act = (u > ub) || (ty > jj);
                        const double mu = act ? colv[u] * rp : 0.0;
#pragma unroll
                        for (int v = 0; v < NB; ++v) a[u][v] = __builtin_fma(-mu, rowv[v], a[u][v]);
                        a[u][ub] = (act && fix) ? -mu : a[u][ub];
                    }
                } else {
                    a[0][0] += rp;
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int v = 0; v < NB; ++v) s += a[u][v];
    out[tid] = s;
}

template <int VAR>
void run(const char* name, int nthreads, int workers) {
    double* d;
    hipMalloc(&d, 1024 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((elim<8, VAR>), dim3(1), dim3(nthreads), 0, 0, d, 110, workers);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((elim<8, VAR>), dim3(1), dim3(nthreads), 0, 0, d, 110, workers);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s threads %4d workers %4d : %7.2f us per kernel, %6.1f ns per step\n", name, nthreads, workers,
           ms * 1000 / reps, ms * 1e6 / reps / 110);
    hipFree(d);
}

int main() {
    run<0>("barrier only", 256, 256);
    run<0>("barrier only", 512, 256);
    run<0>("barrier only", 1024, 256);
    run<1>("barrier + publish + pivot read", 256, 256);
    run<1>("barrier + publish + pivot read", 512, 256);
    run<2>("  + reciprocal", 256, 256);
    run<3>("full step", 256, 256);
    run<3>("full step", 512, 256);
    run<3>("full step", 1024, 256);
    return 0;
}
