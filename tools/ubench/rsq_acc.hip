// Accuracy of v_rsq_f64 and of refinements of it (relative error against a long-double host value).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double *d, double *raw, double *nr2, double *hal, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = d[i];
    double y = __builtin_amdgcn_rsq(a);
    raw[i] = y;
    const double h = 0.5 * a;
    double z = y * fma(-h * y, y, 1.5);
    z = z * fma(-h * z, z, 1.5);
    nr2[i] = z;
    // one third-order step: e = 1 - a y^2, y' = y + y e (1/2 + 3/8 e)
    const double e = fma(-a * y, y, 1.0);
    hal[i] = fma(y * e, fma(0.375, e, 0.5), y);
}
int main()
{
    const int n = 1 << 22;
    double *h = (double *)malloc(n * 8), *r = (double *)malloc(n * 8), *z = (double *)malloc(n * 8), *q = (double *)malloc(n * 8);
    srand(1);
    for (int i = 0; i < n; i++) h[i] = ldexp(1.0 + rand() / (double)RAND_MAX, (rand() % 80) - 40);
    double *d, *dr, *dz, *dq;
    hipMalloc(&d, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dz, n * 8); hipMalloc(&dq, n * 8);
    hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, dr, dz, dq, n);
    hipMemcpy(r, dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(z, dz, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q, dq, n * 8, hipMemcpyDeviceToHost);
    long double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / sqrtl((long double)h[i]);
        e0 = fmaxl(e0, fabsl(r[i] - t) / t); e1 = fmaxl(e1, fabsl(z[i] - t) / t); e2 = fmaxl(e2, fabsl(q[i] - t) / t);
    }
    printf("max relative error: v_rsq_f64 %.3Le (2^%.1Lf)   two Newton steps %.3Le   one third-order step %.3Le  (2^-53 = 1.11e-16)\n",
           e0, log2l(e0), e1, e2);
    return 0;
}
