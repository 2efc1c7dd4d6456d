// Issue rate of FP64 vector instructions on one CU: cycles per wave-instruction per SIMD
// for 1, 2 and 4 waves per SIMD and 1..8 independent dependency chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

enum { FMA, MUL, ADD, RSQ, MINOP, MOV64, FMA32 };

template <int OP, int CH>
__global__ void k(double *out, long long *clk, int iters, double a, double b)
{
    double v[CH];
    float f[CH];
    for (int c = 0; c < CH; c++) { v[c] = a + c + threadIdx.x * 1e-3; f[c] = (float)v[c]; }
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (OP == FMA) v[c] = fma(v[c], a, b);
                else if (OP == MUL) v[c] = v[c] * a;
                else if (OP == ADD) v[c] = v[c] + b;
                else if (OP == RSQ) v[c] = __builtin_amdgcn_rsq(v[c]);
                else if (OP == MINOP) v[c] = __builtin_fmin(v[c], b + r);
                else if (OP == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(v[c]) : "v"(v[c]));
                else f[c] = fmaf(f[c], (float)a, (float)b);
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; c++) s += v[c] + f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int OP, int CH>
void run(const char *name, int threads, int blocks)
{
    double *out; long long *clk;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&clk, sizeof(long long) * blocks);
    const int iters = 2000;
    k<OP, CH><<<blocks, threads>>>(out, clk, iters, 1.0000001, 1e-9);
    k<OP, CH><<<blocks, threads>>>(out, clk, iters, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, clk, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < blocks; b++) avg += h[b];
    avg /= blocks;
    const double per_simd_instr = (double)iters * 8 * CH * (threads / 256.0);   // wave-instructions per SIMD
    printf("%-6s chains=%d waves/SIMD=%d blocks=%d : %.2f cyc per wave-instruction per SIMD\n", name, CH, threads / 256, blocks,
           avg / per_simd_instr);
    hipFree(out); hipFree(clk);
}

template <int OP>
void sweep(const char *name)
{
    for (int threads : {256, 512, 1024}) {
        run<OP, 1>(name, threads, 256);
        run<OP, 2>(name, threads, 256);
        run<OP, 4>(name, threads, 256);
        run<OP, 8>(name, threads, 256);
    }
}

int main()
{
    sweep<FMA>("fma64");
    sweep<MUL>("mul64");
    sweep<ADD>("add64");
    sweep<RSQ>("rsq64");
    sweep<MINOP>("min64");
    sweep<MOV64>("mov64");
    sweep<FMA32>("fma32");
    return 0;
}
