// Dev micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate of ONE wave per SIMD vs TWO (no memory traffic), 4 or 8
// independent accumulators per wave.  hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_peak.hip -o tools/exp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NACC, int WPE>
__global__ __launch_bounds__(256 * (WPE > 1 ? 2 : 1)) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(float* out, int iters, float a0, float b0) {
    f32x4v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// 16 MFMAs (4 accumulators) + ONE memory instruction per group, as in the strip GEMM's steady state: kind 0 none,
// 1 global_load_dwordx4 (L2-resident), 2 global_load_dword, 3 ds_read_b128, 4 global_store_dwordx4,
// 5 global_load_lds_dwordx4 (direct to LDS), 6 = 5 + a ds_read_b128 (the LDS-staged operand stream)
template <int KIND, int WPE, int VG>
__global__ __launch_bounds__(256 * WPE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void km(float* buf, float* out, int iters, float a0, float b0) {
    __shared__ f32x4v lds[1024];
    f32x4v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    lds[threadIdx.x & 1023] = (f32x4v){a0, b0, a0, b0};
    __syncthreads();
    float a = a0 + threadIdx.x, b = b0;
    const f32x4v* p = reinterpret_cast<const f32x4v*>(buf) + (blockIdx.x & 63) * 4096 + (threadIdx.x & 255);
    f32x4v* q = reinterpret_cast<f32x4v*>(buf) + (64 + blockIdx.x) * 4096 + (threadIdx.x & 255);
    f32x4v r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int v = 0; v < VG; ++v)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + r[(u + 8) & 15][v], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (KIND == 1) r[u] = p[u * 256];
            if (KIND == 2) r[u][0] = reinterpret_cast<const float*>(p)[u * 256];
            if (KIND == 3) r[u] = lds[(threadIdx.x + u * 64) & 1023];
            if (KIND == 4) q[u * 256] = acc[u & 3];
            if (KIND == 5 || KIND == 6)      // direct-to-LDS load: no VGPR return
                __builtin_amdgcn_global_load_lds((gptr_t)(p + u * 256), (lptr_t)(lds + (((threadIdx.x & 0x3c0) + (u & 1) * 512) & 1023)), 16, 0, 0);
            if (KIND == 6) r[u] = lds[(threadIdx.x + u * 64) & 1023];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND, int WPE = 1, int VG = 4>
void runm(const char* name) {
    float *out, *buf;
    hipMalloc(&out, 4096);
    hipMalloc(&buf, (64 + 256) * 4096 * 16);
    hipMemset(buf, 0, (64 + 256) * 4096 * 16);
    const int iters = 500;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((km<KIND, WPE, VG>), dim3(256), dim3(256 * WPE), 0, 0, buf, out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = 256.0 * 4 * WPE * iters * 16 * 4 * VG;
        if (rep == 2) printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per MFMA; %.0f extra cycles per memory instruction)\n", name, ms, mfmas * 2048 / ms / 1e9,
                             ms * 1e-3 * 2.4e9 / (mfmas / 1024), (ms * 1e-3 * 2.4e9 / (mfmas / 1024) - 32.3) * 4 * VG);
    }
    hipFree(out); hipFree(buf);
}

template <int NACC, int WPE>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 2000, threads = 256 * (WPE > 1 ? 2 : 1);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, WPE>), dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = 256.0 * (threads / 64) * iters * 16 * NACC;
        if (rep == 2) printf("%-28s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, ms, mfmas * 2048 / ms / 1e9,
                             ms * 1e-3 * 2.4e9 / (mfmas / 1024));
    }
    hipFree(out);
}

int main() {
    run<4, 1>("1 wave/SIMD, 4 acc");
    run<8, 1>("1 wave/SIMD, 8 acc");
    run<16, 1>("1 wave/SIMD, 16 acc");
    run<4, 2>("2 waves/SIMD, 4 acc");
    run<8, 2>("2 waves/SIMD, 8 acc");
    runm<0>("16 MFMA groups, no memory op");
    runm<1>("16 MFMA + 1 global_load_dwordx4");
    runm<2>("16 MFMA + 1 global_load_dword");
    runm<3>("16 MFMA + 1 ds_read_b128");
    runm<4>("16 MFMA + 1 global_store_dwordx4");
    runm<1, 2, 2>("2 waves/SIMD: 8 MFMA + 1 global_load_dwordx4");
    runm<1, 2, 4>("2 waves/SIMD: 16 MFMA + 1 global_load_dwordx4");
    runm<3, 2, 2>("2 waves/SIMD: 8 MFMA + 1 ds_read_b128");
    runm<1, 1, 2>("1 wave/SIMD: 8 MFMA + 1 global_load_dwordx4");
    runm<5>("16 MFMA + 1 global_load_lds_dwordx4");
    runm<6>("16 MFMA + 1 load_lds + 1 ds_read_b128");
    runm<5, 2, 2>("2 waves/SIMD: 8 MFMA + 1 load_lds");
    runm<6, 2, 2>("2 waves/SIMD: 8 MFMA + 1 load_lds + 1 ds_read");
    runm<3, 2, 4>("2 waves/SIMD: 16 MFMA + 1 ds_read_b128");
    return 0;
}
