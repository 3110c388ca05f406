// Experiment: accuracy of v_rcp_f64 and of one / two Newton steps on top of it (factor_kernel's pivot reciprocal).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = __builtin_fma(r0, __builtin_fma(-v, r0, 1.0), r0);
    double r2 = __builtin_fma(r1, __builtin_fma(-v, r1, 1.0), r1);
    o[3 * i] = r0; o[3 * i + 1] = r1; o[3 * i + 2] = r2;
}
int main() {
    const int n = 1 << 20;
    double* hx = new double[n]; double* ho = new double[3 * n];
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX - 0.5) * 40.0) * (1.0 + (double)rand() / RAND_MAX);
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(ho, dout, 3 * n * 8, hipMemcpyDeviceToHost);
    double e[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) { double err = fabs(ho[3 * i + j] * hx[i] - 1.0); if (err > e[j]) e[j] = err; }
    printf("max |r*x - 1|: v_rcp_f64 %.3e, + 1 Newton %.3e, + 2 Newton %.3e\n", e[0], e[1], e[2]);
    return 0;
}
