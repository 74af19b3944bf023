// Which SIMD does wave i of a 512-thread workgroup run on?  (s_getreg_b32 HW_REG_HW_ID: gfx9 layout wave_id[3:0] simd_id[5:4]
// pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...).  Prints the SIMD id of the 8 waves of a few workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out, int big_lds) {
    extern __shared__ char lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    if (big_lds < 0) lds[threadIdx.x] = 0;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4096 * 8 * 4);
    for (int lds : {0, 150000}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
        hipLaunchKernelGGL(k, dim3(512), dim3(512), lds, 0, d, lds);
        unsigned h[512 * 8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("dynamic LDS %d\n", lds);
        for (int b : {0, 1, 2, 3, 100, 255, 256, 511}) {
            printf("  wg %3d: simd of waves 0..7 =", b);
            for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
            printf("   wave slots =");
            for (int w = 0; w < 8; ++w) printf(" %u", h[b * 8 + w] & 15);
            printf("   cu %u\n", (h[b * 8] >> 8) & 15);
        }
    }
    return 0;
}
