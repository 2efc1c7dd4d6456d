// Sustained FP64 vector rate of the whole chip in wall-clock terms: every SIMD runs `waves` waves of
// independent fused multiply-add chains for several milliseconds; prints TFLOP/s and the clock the
// shader counter saw.  (78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz.)
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CH>
__global__ void fma_chains(double *out, long long *clk, int iters, double a, double b)
{
    double v[CH];
    for (int c = 0; c < CH; c++) v[c] = a + c + threadIdx.x * 1e-3;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) v[c] = fma(v[c], a, b);
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; c++) s += v[c];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main()
{
    const int CH = 8;
    double *out; long long *clk;
    hipMalloc(&out, sizeof(double) * 1024 * 4096);
    hipMalloc(&clk, sizeof(long long) * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {1, 2, 4}) {                 // waves per SIMD
        for (int rounds : {1, 4, 16}) {           // workgroups per CU, one after the other
            const int threads = 256 * waves, blocks = 256 * rounds, iters = 4000;
            fma_chains<CH><<<blocks, threads>>>(out, clk, iters, 1.0000001, 1e-9);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            fma_chains<CH><<<blocks, threads>>>(out, clk, iters, 1.0000001, 1e-9);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[4096]; hipMemcpy(h, clk, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
            double cyc = 0; for (int b = 0; b < blocks; b++) cyc += h[b]; cyc /= blocks;
            const double flop = 2.0 * 16 * CH * (double)iters * threads * blocks;
            const double winstr_per_simd = 16.0 * CH * iters * waves;   // per workgroup residency
            printf("waves/SIMD=%d workgroups/CU=%2d: %7.3f ms  %6.2f TFLOP/s (%.0f %% of 78.6)  %.2f counter ticks per wave-instruction and SIMD\n",
                   waves, rounds, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 78.6 * 100, cyc / winstr_per_simd);
        }
    }
    return 0;
}
