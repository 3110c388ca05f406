// Prints the lane/register layout of v_mfma_f64_16x16x4_f64 (used to write the W1 panel kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    // A[i][k] = 1000*i + k (assume lane = i + 16k), B[k][j] = (k == kk) ? 1 : 0 for j... use B = identity-like probes
    // Run 4 probes: B[k][j] = (k == p) * (j+1): then D[i][j] = A[i][p] * (j+1)
    for (int p = 0; p < 4; ++p) {
        const double a = 1000.0 * (l % 16) + (l / 16);
        const double b = ((l / 16) == p) ? (double)((l % 16) + 1) : 0.0;
        d4 c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(p * 64 + l) * 4 + r] = c[r];
    }
}
int main() {
    double* d; hipMalloc(&d, 4 * 64 * 4 * 8);
    probe<<<1, 64>>>(d);
    static double h[4 * 64 * 4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // decode: value = (1000*i + p) * (j+1) if assumptions on A/B hold
    int bad = 0;
    for (int p = 0; p < 4; ++p)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const double v = h[(p * 64 + l) * 4 + r];
                // hypothesis H1: i = 4*(l/16) + r, j = l%16
                const int i1 = 4 * (l / 16) + r, j1 = l % 16;
                const double e1 = (1000.0 * i1 + p) * (j1 + 1);
                if (v != e1) ++bad;
            }
    printf("H1 (i = 4*(lane/16)+r, j = lane%%16) mismatches: %d\n", bad);
    for (int l = 0; l < 64; l += 13) printf("lane %d p0: %.0f %.0f %.0f %.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
