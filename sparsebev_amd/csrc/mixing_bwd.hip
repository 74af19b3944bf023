// Adaptive mixing core, BACKWARD (gfx950).  SURVEY.md section 8f rank 4.
//
// Forward (mixing.hip; models/sparsebev_transformer.py:362-374), per (query, group):
//     y1 = x[Pin,C] @ M[C,C]         h1 = LN_all(y1)      n1 = relu(h1)
//     y2 = S[Pout,Pin] @ n1[Pin,C]   h2 = LN_all(y2)      out = relu(h2)        (LN over the whole block, no affine)
// Backward, given g_out[Pout,C] (the forward is recomputed here -- nothing but x and the dynamic parameters is kept from
// the forward pass, the same policy as the reference's activation checkpoint around inner_forward, :383-387):
//     g2  = g_out * (h2 > 0)         dy2 = rstd2 * (g2 - mean(g2) - h2 * mean(g2 * h2))
//     dS  = dy2 @ n1^T [Pout,Pin]    dn1 = S^T @ dy2 [Pin,C]
//     g1  = dn1 * (h1 > 0)           dy1 = rstd1 * (g1 - mean(g1) - h1 * mean(g1 * h1))
//     dM  = x^T @ dy1 [C,C]          dx  = dy1 @ M^T [Pin,C]
// One 256-thread workgroup per (b*Q + q, g); x, h1, y2 -> dy2, dn1 -> dy1 live in LDS (rows padded to 65 floats so both
// row- and column-walking reads are conflict-free), M's column / row and the S column of a thread live in registers or come
// through the scalar/L1 path.  Plain fp32 FMAs: six small products (1.2 M MAC per item) -- the backward is not the
// benchmarked path; the GEMMs around it (gemm_any.hip) dominate a training step.
#include "sbev_common.hpp"

namespace {

constexpr int C = 64, POUT = 128, LD = 65;

struct MixBwdArgs {
    const float* x;        // [BQ, G, Pin, C]
    const float* params;   // [BQ, G, C*C + Pout*Pin]
    const float* gout;     // [BQ, G, Pout, C]
    float* gx;             // [BQ, G, Pin, C]
    float* gparams;        // [BQ, G, C*C + Pout*Pin]
    long long n_items;
    int Pin;
    float eps;
};

__device__ __forceinline__ float block_sum(float v, float* red, int wave, int lane) {
    v = sbev::wave_sum_dpp(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void mixing_bwd_kernel(const MixBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Pin = a.Pin;
    float* xs = smem;                    // [Pin][LD]
    float* h1 = xs + Pin * LD;           // [Pin][LD]   y1 -> h1
    float* d1 = h1 + Pin * LD;           // [Pin][LD]   dn1 -> g1 -> dy1
    float* y2 = d1 + Pin * LD;           // [POUT][LD]  y2 -> h2 -> dy2
    float* red = y2 + POUT * LD;         // [8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long item = blockIdx.x;
    const int NP = C * C + POUT * Pin;
    const float* __restrict__ x = a.x + item * Pin * C;
    const float* __restrict__ Mg = a.params + item * NP;
    const float* __restrict__ Sg = Mg + C * C;
    const float* __restrict__ go = a.gout + item * POUT * C;
    float* __restrict__ gM = a.gparams + item * NP;
    float* __restrict__ gS = gM + C * C;
    float* __restrict__ gx = a.gx + item * Pin * C;
    const float n1cnt = (float)(Pin * C), n2cnt = (float)(POUT * C);

    for (int i = tid; i < Pin * C; i += 256) xs[(i >> 6) * LD + (i & 63)] = x[i];
    // ---- y1 = x M : thread = column co (this lane) x rows p = wave, wave + 4, ...; its M column sits in registers
    float mcol[C];
#pragma unroll
    for (int ci = 0; ci < C; ++ci) mcol[ci] = Mg[ci * C + lane];
    __syncthreads();
    float s = 0.f;
    for (int p = wave; p < Pin; p += 4) {
        float acc = 0.f;
#pragma unroll
        for (int ci = 0; ci < C; ++ci) acc += xs[p * LD + ci] * mcol[ci];
        h1[p * LD + lane] = acc;
        s += acc;
    }
    const float mean1 = block_sum(s, red, wave, lane) / n1cnt;
    s = 0.f;
    for (int p = wave; p < Pin; p += 4) {
        const float d = h1[p * LD + lane] - mean1;
        s += d * d;
    }
    const float rstd1 = rsqrtf(block_sum(s, red, wave, lane) / n1cnt + a.eps);
    for (int p = wave; p < Pin; p += 4) h1[p * LD + lane] = (h1[p * LD + lane] - mean1) * rstd1;
    __syncthreads();
    // ---- y2 = S n1 : thread = channel c (lane) x rows o = wave, wave + 4, ...   (S[o][p] is wave-uniform)
    s = 0.f;
    for (int o = wave; o < POUT; o += 4) {
        float acc = 0.f;
        for (int p = 0; p < Pin; ++p) acc += Sg[o * Pin + p] * fmaxf(h1[p * LD + lane], 0.f);
        y2[o * LD + lane] = acc;
        s += acc;
    }
    const float mean2 = block_sum(s, red, wave, lane) / n2cnt;
    s = 0.f;
    for (int o = wave; o < POUT; o += 4) {
        const float d = y2[o * LD + lane] - mean2;
        s += d * d;
    }
    const float rstd2 = rsqrtf(block_sum(s, red, wave, lane) / n2cnt + a.eps);
    // ---- dy2 (in place): g2 = g_out * (h2 > 0); dy2 = rstd2 (g2 - mean(g2) - h2 mean(g2 h2))
    float sg = 0.f, sgh = 0.f;
    for (int o = wave; o < POUT; o += 4) {
        const float h = (y2[o * LD + lane] - mean2) * rstd2;
        y2[o * LD + lane] = h;
        const float g = h > 0.f ? go[o * C + lane] : 0.f;
        sg += g;
        sgh += g * h;
    }
    const float mg2 = block_sum(sg, red, wave, lane) / n2cnt;
    const float mgh2 = block_sum(sgh, red + 4, wave, lane) / n2cnt;
    for (int o = wave; o < POUT; o += 4) {
        const float h = y2[o * LD + lane];
        const float g = h > 0.f ? go[o * C + lane] : 0.f;
        y2[o * LD + lane] = rstd2 * (g - mg2 - h * mgh2);
    }
    __syncthreads();
    // ---- dS[o][p] = sum_c dy2[o][c] n1[p][c]  (consecutive threads = consecutive p: stride-65 rows, conflict-free)
    for (int i = tid; i < POUT * Pin; i += 256) {
        const int o = i / Pin, p = i - o * Pin;
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) acc += y2[o * LD + c] * fmaxf(h1[p * LD + c], 0.f);
        gS[i] = acc;
    }
    // ---- dn1[p][c] = sum_o S[o][p] dy2[o][c];  g1 = dn1 * (h1 > 0)
    sg = 0.f; sgh = 0.f;
    for (int p = wave; p < Pin; p += 4) {
        float acc = 0.f;
        for (int o = 0; o < POUT; ++o) acc += Sg[o * Pin + p] * y2[o * LD + lane];
        const float h = h1[p * LD + lane];
        const float g = h > 0.f ? acc : 0.f;
        d1[p * LD + lane] = g;
        sg += g;
        sgh += g * h;
    }
    const float mg1 = block_sum(sg, red, wave, lane) / n1cnt;
    const float mgh1 = block_sum(sgh, red + 4, wave, lane) / n1cnt;
    for (int p = wave; p < Pin; p += 4) d1[p * LD + lane] = rstd1 * (d1[p * LD + lane] - mg1 - h1[p * LD + lane] * mgh1);
    __syncthreads();
    // ---- dM[ci][co] = sum_p x[p][ci] dy1[p][co] : thread = co (lane) x ci = wave, wave + 4, ...
    for (int ci = wave; ci < C; ci += 4) {
        float acc = 0.f;
        for (int p = 0; p < Pin; ++p) acc += xs[p * LD + ci] * d1[p * LD + lane];
        gM[ci * C + lane] = acc;
    }
    // ---- dx[p][ci] = sum_co dy1[p][co] M[ci][co] : thread = ci (lane), its M row in registers
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
        const float4 v = *reinterpret_cast<const float4*>(Mg + lane * C + c4 * 4);
        mcol[c4 * 4] = v.x; mcol[c4 * 4 + 1] = v.y; mcol[c4 * 4 + 2] = v.z; mcol[c4 * 4 + 3] = v.w;
    }
    for (int p = wave; p < Pin; p += 4) {
        float acc = 0.f;
#pragma unroll
        for (int co = 0; co < C; ++co) acc += d1[p * LD + co] * mcol[co];
        gx[p * C + lane] = acc;
    }
}

}  // namespace

extern "C" int sbev_adaptive_mixing_bwd_f32(const float* x, const float* params, const float* grad_y, float* grad_x, float* grad_params,
                                            int64_t BQ, int G, int Pin, int Cg, int Pout, float eps, sbev_stream_t stream) {
    SBEV_REQUIRE(BQ >= 0 && G >= 1, "sbev_adaptive_mixing_bwd_f32: bad sizes");
    SBEV_REQUIRE(Cg == C && Pout == POUT, "sbev_adaptive_mixing_bwd_f32: built for C=64 channels per group and 128 out points (got %d, %d)", Cg, Pout);
    SBEV_REQUIRE(Pin >= 1 && Pin <= 120, "sbev_adaptive_mixing_bwd_f32: in_points=%d must be in 1..120 (LDS budget)", Pin);
    if (BQ == 0) return SBEV_OK;
    SBEV_REQUIRE(x && params && grad_y && grad_x && grad_params, "sbev_adaptive_mixing_bwd_f32: null pointer");
    SBEV_REQUIRE(BQ * G <= 0x7fffffffLL, "sbev_adaptive_mixing_bwd_f32: too many items");
    SBEV_REQUIRE((((uintptr_t)params) & 15) == 0 && (Pin * Pout) % 4 == 0, "sbev_adaptive_mixing_bwd_f32: params must be 16-byte aligned");
    MixBwdArgs a{x, params, grad_y, grad_x, grad_params, BQ * G, Pin, eps};
    const size_t lds = (size_t)((3 * Pin + POUT) * LD + 8) * sizeof(float);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mixing_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            sbev::set_error("sbev_adaptive_mixing_bwd_f32: cannot reserve %zu bytes of LDS", lds);
            return SBEV_ELAUNCH;
        }
    }
    hipLaunchKernelGGL(mixing_bwd_kernel, dim3((unsigned)a.n_items), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_adaptive_mixing_bwd_f32");
}
