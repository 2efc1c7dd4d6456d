// Where the waves of a workgroup land: for workgroups of four waves (a launch shaped like the fit kernels: two
// workgroups to a CU by LDS), the SIMD of every wave from the hardware id register, and how often wave w sits on
// SIMD w.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256, 2) void where(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    double v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = fma(v, 1.0000001, 1e-9);
    lds[threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
    if (v == 123.456) out[0] = 0;
}

int main()
{
    const int blocks = 4096;
    unsigned *out;
    (void)hipMalloc(&out, sizeof(unsigned) * blocks * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(where), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    where<<<blocks, 256, 72 * 1024>>>(out, 20000);
    (void)hipDeviceSynchronize();
    static unsigned h[4096 * 4];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0, hist[4][4] = {{0}};
    int distinct4 = 0;
    for (int b = 0; b < blocks; b++) {
        unsigned seen = 0;
        for (int w = 0; w < 4; w++) {
            const unsigned simd = (h[b * 4 + w] >> 4) & 3;        // HW_ID: wave_id[3:0], simd_id[5:4], pipe, cu_id[11:8] ...
            hist[w][simd]++;
            same += simd == (unsigned)w;
            seen |= 1u << simd;
        }
        distinct4 += seen == 15u;
    }
    printf("workgroups whose four waves sit on four different SIMDs: %d of %d\n", distinct4, blocks);
    for (int w = 0; w < 4; w++)
        printf("wave %d on SIMD 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("first workgroups (hw id words): ");
    for (int k = 0; k < 8; k++) printf("%08x ", h[k]);
    printf("\n");
    return 0;
}
