// Experiment (round 3): variants of the NCHW -> NHWC relayout (csrc/layout.hip: transpose_tiles_kernel) on the level-0 shape of
// config 3 (384 images x 256 channels x 64*176 pixels, fp32) next to a plain copy of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/relayout_bench.hip -o tools/exp/relayout_bench && tools/exp/relayout_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct TrArgs { const float* in; float* out; int R, S; };

// V0: the product kernel (64 x 64 tile, grid x = pixel tiles)      V1: same, grid x = channel tiles (YFAST)
template <bool YFAST, bool NT>
__global__ __launch_bounds__(256) void tr64(const TrArgs a) {
    constexpr int TS = 64, TLD = 65;
    __shared__ float tile[TS * TLD];
    const int tid = threadIdx.x;
    const int s0 = (YFAST ? blockIdx.y : blockIdx.x) * TS, r0 = (YFAST ? blockIdx.x : blockIdx.y) * TS;
    const long long img = blockIdx.z;
    const float* in = a.in + img * a.R * a.S;
    float* out = a.out + img * a.R * a.S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + (tid >> 4) + 16 * i, s = s0 + (tid & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < a.R && s < a.S) {
            const float4* p = reinterpret_cast<const float4*>(in + (long long)r * a.S + s);
            if (NT) { v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w); }
            else v = *p;
        }
        const int lr = (tid >> 4) + 16 * i, ls = (tid & 15) * 4;
        tile[(ls + 0) * TLD + lr] = v.x; tile[(ls + 1) * TLD + lr] = v.y; tile[(ls + 2) * TLD + lr] = v.z; tile[(ls + 3) * TLD + lr] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ls = (tid >> 4) + 16 * i, lr = (tid & 15) * 4;
        const int s = s0 + ls, r = r0 + lr;
        if (s < a.S && r < a.R) {
            float* p = out + (long long)s * a.R + r;
            if (NT) {
                __builtin_nontemporal_store(tile[ls * TLD + lr], p); __builtin_nontemporal_store(tile[ls * TLD + lr + 1], p + 1);
                __builtin_nontemporal_store(tile[ls * TLD + lr + 2], p + 2); __builtin_nontemporal_store(tile[ls * TLD + lr + 3], p + 3);
            } else {
                *reinterpret_cast<float4*>(p) = make_float4(tile[ls * TLD + lr], tile[ls * TLD + lr + 1], tile[ls * TLD + lr + 2], tile[ls * TLD + lr + 3]);
            }
        }
    }
}

// V3: PX pixels x CH channels per workgroup (CH = 256: whole 1 KB output rows), PX * 4 B input runs per channel plane.
template <int PX, int CH>
__global__ __launch_bounds__(256) void tr_rows(const TrArgs a) {
    constexpr int LD = CH + 1;
    __shared__ float tile[PX * LD];       // tile[pixel][channel]
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * CH, s0 = blockIdx.y * PX;
    const long long img = blockIdx.z;
    const float* in = a.in + img * a.R * a.S;
    float* out = a.out + img * a.R * a.S;
    constexpr int LPC = PX / 4;                    // lanes per channel (float4 each)
    constexpr int CPP = 256 / LPC;                 // channels per pass
    constexpr int NP = CH / CPP;
    float4 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = r0 + i * CPP + tid / LPC, s = s0 + (tid % LPC) * 4;
        v[i] = s < a.S ? *reinterpret_cast<const float4*>(in + (long long)c * a.S + s) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = i * CPP + tid / LPC, ls = (tid % LPC) * 4;
        tile[(ls + 0) * LD + c] = v[i].x; tile[(ls + 1) * LD + c] = v[i].y; tile[(ls + 2) * LD + c] = v[i].z; tile[(ls + 3) * LD + c] = v[i].w;
    }
    __syncthreads();
    constexpr int LPP = CH / 4;                    // lanes per pixel
    constexpr int PPP = 256 / LPP;                 // pixels per pass
#pragma unroll
    for (int i = 0; i < PX / PPP; ++i) {
        const int ls = i * PPP + tid / LPP, c = (tid % LPP) * 4;
        const int s = s0 + ls;
        if (s < a.S)
            *reinterpret_cast<float4*>(out + (long long)s * a.R + r0 + c) = make_float4(tile[ls * LD + c], tile[ls * LD + c + 1], tile[ls * LD + c + 2], tile[ls * LD + c + 3]);
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long n4) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n4; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy4_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long n4) {   // 4 float4 per thread, one-shot
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x);
    const long long q = n4 / 4;
    if (i < q) {
        const float4 a = in[i], b = in[i + q], c = in[i + 2 * q], d = in[i + 3 * q];
        out[i] = a; out[i + q] = b; out[i + 2 * q] = c; out[i + 3 * q] = d;
    }
}

template <typename F>
static float time_ms(F launch, int reps = 10) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 384, R = 256, S = argc > 2 ? atoi(argv[2]) : 64 * 176;
    const long long n = (long long)N * R * S;
    float *in, *out, *ref;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ref, n * 4));
    std::vector<float> h((size_t)R * S);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.f / 65536.f);
    for (int i = 0; i < N; ++i) CK(hipMemcpy(in + (long long)i * R * S, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    TrArgs a{in, ref, R, S};
    hipLaunchKernelGGL((tr64<false, false>), dim3((S + 63) / 64, R / 64, N), dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    a.out = out;
    const double gb = 2.0 * n * 4 / 1e9;
    auto check = [&](const char* name) {
        std::vector<float> x((size_t)R * S), y((size_t)R * S);
        CK(hipMemcpy(x.data(), out + (long long)(N - 1) * R * S, x.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), ref + (long long)(N - 1) * R * S, y.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < x.size(); ++i) if (x[i] != y[i]) { printf("  %s: MISMATCH at %zu\n", name, i); return; }
    };
    auto run = [&](const char* name, auto launch) {
        CK(hipMemset(out, 0, n * 4));
        const float ms = time_ms(launch);
        check(name);
        printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3);
    };
    printf("relayout of %d images x %d channels x %d pixels fp32: %.2f GB moved per launch\n", N, R, S, gb);
    run("copy (grid-stride, 256 CUs x 8)", [&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, (const float4*)ref, (float4*)out, n / 4); });
    run("copy (one-shot, 4 float4/thread)", [&] { hipLaunchKernelGGL(copy4_kernel, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, (const float4*)ref, (float4*)out, n / 4); });
    CK(hipMemcpy(out, ref, 16, hipMemcpyDeviceToDevice));
    run("V0 tile 64x64 (product)", [&] { hipLaunchKernelGGL((tr64<false, false>), dim3((S + 63) / 64, R / 64, N), dim3(256), 0, 0, a); });
    run("V1 tile 64x64, channel-tile fastest", [&] { hipLaunchKernelGGL((tr64<true, false>), dim3(R / 64, (S + 63) / 64, N), dim3(256), 0, 0, a); });
    run("V2 tile 64x64, nontemporal", [&] { hipLaunchKernelGGL((tr64<false, true>), dim3((S + 63) / 64, R / 64, N), dim3(256), 0, 0, a); });
    run("V2b V1 + nontemporal", [&] { hipLaunchKernelGGL((tr64<true, true>), dim3(R / 64, (S + 63) / 64, N), dim3(256), 0, 0, a); });
    run("V3 32 px x 256 ch", [&] { hipLaunchKernelGGL((tr_rows<32, 256>), dim3(1, (S + 31) / 32, N), dim3(256), 0, 0, a); });
    run("V3 16 px x 256 ch", [&] { hipLaunchKernelGGL((tr_rows<16, 256>), dim3(1, (S + 15) / 16, N), dim3(256), 0, 0, a); });
    run("V3 64 px x 128 ch", [&] { hipLaunchKernelGGL((tr_rows<64, 128>), dim3(2, (S + 63) / 64, N), dim3(256), 0, 0, a); });
    run("V3 128 px x 64 ch", [&] { hipLaunchKernelGGL((tr_rows<128, 64>), dim3(4, (S + 127) / 128, N), dim3(256), 0, 0, a); });
    run("V3 128 px x 32 ch", [&] { hipLaunchKernelGGL((tr_rows<128, 32>), dim3(8, (S + 127) / 128, N), dim3(256), 0, 0, a); });
    run("V3 256 px x 32 ch", [&] { hipLaunchKernelGGL((tr_rows<256, 32>), dim3(8, (S + 255) / 256, N), dim3(256), 0, 0, a); });
    run("V3 64 px x 64 ch", [&] { hipLaunchKernelGGL((tr_rows<64, 64>), dim3(4, (S + 63) / 64, N), dim3(256), 0, 0, a); });
    return 0;
}
