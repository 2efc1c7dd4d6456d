// Does the number of vector-register operands of an FP64 instruction change its issue rate?  Whole chip, four
// waves per SIMD, eight independent chains per lane, wall-clock time.  Variants of v = fma(v, A, B):
//   0: A, B scalar   1: A vector, B scalar   2: A, B vectors (distinct per chain)   3: v = fma(A, B, v) (accumulate)
//   4: v = v * A (vector)   5: v = v + A (vector)
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void chains(double *out, int iters, double a, double b)
{
    constexpr int CH = 8;
    double v[CH], w[CH], u[CH];
    for (int c = 0; c < CH; c++) {
        v[c] = a + c + threadIdx.x * 1e-3;
        w[c] = a + 1e-9 * (c + threadIdx.x);
        u[c] = b * (1 + c + threadIdx.x);
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (MODE == 0) v[c] = fma(v[c], a, b);
                else if (MODE == 1) v[c] = fma(v[c], w[c], b);
                else if (MODE == 2) v[c] = fma(v[c], w[c], u[c]);
                else if (MODE == 3) v[c] = fma(w[c], u[(c + r) & 7], v[c]);
                else if (MODE == 4) v[c] = v[c] * w[c];
                else v[c] = v[c] + u[c];
            }
        }
    }
    double s = 0;
    for (int c = 0; c < CH; c++) s += v[c] + w[c] + u[c];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *what, double *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int threads = 1024, blocks = 256 * 4, iters = 2000;
    chains<MODE><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    chains<MODE><<<blocks, threads>>>(out, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = 16.0 * 8 * iters * threads * blocks;           // lane-instructions
    printf("%-44s %7.3f ms  %6.2f T lane-instructions/s (%.0f %% of 39.3)\n", what, ms, instr / ms * 1e-9,
           instr / ms * 1e-9 / 39.3 * 100);
}

int main()
{
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 1024 * 4096);
    run<0>("fma(v, s, s)", out);
    run<1>("fma(v, w, s)", out);
    run<2>("fma(v, w, u)", out);
    run<3>("fma(w, u', v)  (accumulate)", out);
    run<4>("v * w", out);
    run<5>("v + u", out);
    return 0;
}
