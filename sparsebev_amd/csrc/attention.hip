// Scale-adaptive self attention core (gfx950): flash-style, the [B,8,Q,Q] bias tensor never exists.
//
// Replaces SparseBEVSelfAttention.inner_forward minus its Linear layers (models/sparsebev_transformer.py:
// 210-228,236-248) and the softmax(QK^T/sqrt(d) + mask)V core of torch.nn.MultiheadAttention that mmcv's
// MultiheadAttention wraps:
//     logits[h,i,j] = (q_i . k_j)/sqrt(d) - ||c_i - c_j||_2 * tau[i,h]      (+ -inf where the DN mask is set)
//     out[i, h*d:(h+1)*d] = softmax_j(logits) @ v
// c = decoded (x, y) box centres in metres.  The in-projection (with gen_tau's 8 rows appended: one GEMM,
// N = 3D + H) runs before this kernel and the out-projection + residual + norm1 after it (sbev_linear_f32).
//
// One workgroup = NW waves x 16 query rows of one (batch, head).  K/V tiles of 64 keys are staged in LDS
// (16-B coalesced loads: a key's 32-float head slice is one 128-B line), S = QK^T and O += PV run on
// v_mfma_f32_16x16x4_f32 (exact fp32), the distance bias is recomputed from the centres on the fly, the
// softmax is the online (running max / running sum) form, and P goes from the MFMA C layout to the A layout
// through a per-wave 4-KiB LDS patch.  1.3 GFLOP per layer-sample: latency-, not throughput-critical.
#include "sbev_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 32;        // head dim (embed 256 / 8 heads)
constexpr int KT = 64;        // keys per tile
constexpr int NW = 2;         // waves per workgroup -> 32 query rows
constexpr int LDK = HD + 4;   // K tile row stride (B operand of QK^T is read along d: rows = keys)
constexpr int LDV = HD + 16;  // V tile row stride (B operand of PV is read along keys: stride = 16 banks)
constexpr int LDP = KT + 4;   // P patch row stride

struct AttnArgs {
    const float* qkvt;          // [B, Q, ld]: q | k | v | tau
    const float* centers;       // [B, Q, 2] metres
    const unsigned char* mask;  // [Q, Q] (1 = masked) or null
    float* out;                 // [B, Q, H*HD]
    int B, Q, H, ld;
    float scale;                // 1/sqrt(HD)
};

__device__ __forceinline__ float row16_max(float v) {   // reduce over the 16 lanes that share (lane >> 4)
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(64 * NW) void sasa_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[KT * LDV];
    __shared__ __attribute__((aligned(16))) float Cs[KT * 2];
    __shared__ __attribute__((aligned(16))) float Ps[NW * 16 * LDP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fk = lane >> 4;
    const int qtiles = (a.Q + 16 * NW - 1) / (16 * NW);
    const int qt = blockIdx.x % qtiles;
    const int h = (blockIdx.x / qtiles) % a.H;
    const int b = blockIdx.x / (qtiles * a.H);
    const int D = a.H * HD;
    const float* base = a.qkvt + (long long)b * a.Q * a.ld;
    const int q0 = qt * 16 * NW + wave * 16;                 // this wave's first query row

    // Q fragments (A operand: row = fi, k = 4s + fk), pre-scaled like torch's MHA (q * head_dim^-0.5)
    float qf[HD / 4];
    {
        const int qi = min(q0 + fi, a.Q - 1);
#pragma unroll
        for (int s = 0; s < HD / 4; ++s) qf[s] = base[(long long)qi * a.ld + h * HD + 4 * s + fk] * a.scale;
    }
    // per-lane rows of the C layout: row r = fk*4 + e  ->  query q0 + r
    float cx[4], cy[4], tau[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qi = min(q0 + fk * 4 + e, a.Q - 1);
        cx[e] = a.centers[((long long)b * a.Q + qi) * 2 + 0];
        cy[e] = a.centers[((long long)b * a.Q + qi) * 2 + 1];
        tau[e] = base[(long long)qi * a.ld + 3 * D + h];
    }
    float m_run[4], l_run[4];
    f32x4 o_acc[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m_run[e] = -INFINITY; l_run[e] = 0.f; }
    o_acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o_acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* Pw = Ps + wave * 16 * LDP;

    for (int k0 = 0; k0 < a.Q; k0 += KT) {
        __syncthreads();                                       // previous tile fully consumed
        for (int i = tid; i < KT * (HD / 4); i += 64 * NW) {   // 8 x float4 per key
            const int r = i / (HD / 4), c4 = (i % (HD / 4)) * 4;
            const int kj = min(k0 + r, a.Q - 1);
            const float* row = base + (long long)kj * a.ld + h * HD + c4;
            *reinterpret_cast<float4*>(&Ks[r * LDK + c4]) = *reinterpret_cast<const float4*>(row + D);
            *reinterpret_cast<float4*>(&Vs[r * LDV + c4]) = *reinterpret_cast<const float4*>(row + 2 * D);
        }
        for (int i = tid; i < KT; i += 64 * NW) {
            const int kj = min(k0 + i, a.Q - 1);
            Cs[2 * i] = a.centers[((long long)b * a.Q + kj) * 2];
            Cs[2 * i + 1] = a.centers[((long long)b * a.Q + kj) * 2 + 1];
        }
        __syncthreads();

        // S = (Q/sqrt(d)) K^T : 4 key sub-tiles of 16
        f32x4 s_acc[KT / 16];
#pragma unroll
        for (int c = 0; c < KT / 16; ++c) {
            s_acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < HD / 4; ++s)
                s_acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], Ks[(c * 16 + fi) * LDK + 4 * s + fk], s_acc[c], 0, 0, 0);
        }
        // + distance bias, masks; tile row max.  C layout: column (key) = fi, row (query) = fk*4 + e
        float tmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int c = 0; c < KT / 16; ++c) {
            const int kj = k0 + c * 16 + fi;
            const float kx = Cs[2 * (c * 16 + fi)], ky = Cs[2 * (c * 16 + fi) + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx = cx[e] - kx, dy = cy[e] - ky;
                float v = s_acc[c][e] + (-sqrtf(dx * dx + dy * dy)) * tau[e];
                const int qi = q0 + fk * 4 + e;
                bool dead = kj >= a.Q;
                if (a.mask && !dead && qi < a.Q) dead = a.mask[(long long)qi * a.Q + kj] != 0;
                v = dead ? -INFINITY : v;
                s_acc[c][e] = v;
                tmax[e] = fmaxf(tmax[e], v);
            }
        }
        float alpha[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m_new = fmaxf(m_run[e], row16_max(tmax[e]));
            const float m_use = m_new == -INFINITY ? 0.f : m_new;       // fully masked so far: keep everything 0
            alpha[e] = __expf(m_run[e] - m_use);                        // exp(-inf) = 0 on the first tile
            m_run[e] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int c = 0; c < KT / 16; ++c) {
                const float p = __expf(s_acc[c][e] - m_use);
                s_acc[c][e] = p;
                psum += p;
            }
            l_run[e] = l_run[e] * alpha[e] + row16_sum(psum);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) o_acc[t][e] *= alpha[e];
        // P: C layout -> LDS -> A layout (row = fi, k = key)
#pragma unroll
        for (int c = 0; c < KT / 16; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) Pw[(fk * 4 + e) * LDP + c * 16 + fi] = s_acc[c][e];
        // the patch is private to this wave and a wave's DS operations execute in issue order, so the reads
        // below see the writes above without a workgroup barrier; only keep the compiler from reordering them
        __builtin_amdgcn_wave_barrier();
        // O += P V : 2 column tiles of 16 dims, K = 64 keys
#pragma unroll
        for (int s = 0; s < KT / 4; ++s) {
            const float pa = Pw[fi * LDP + 4 * s + fk];
            o_acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, Vs[(4 * s + fk) * LDV + fi], o_acc[0], 0, 0, 0);
            o_acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, Vs[(4 * s + fk) * LDV + 16 + fi], o_acc[1], 0, 0, 0);
        }
    }
    // normalise and store: C layout column = dim (fi), row = query (fk*4 + e)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qi = q0 + fk * 4 + e;
        if (qi < a.Q) {
            const float inv = 1.f / l_run[e];
            float* o = a.out + ((long long)b * a.Q + qi) * D + h * HD;
            o[fi] = o_acc[0][e] * inv;
            o[16 + fi] = o_acc[1][e] * inv;
        }
    }
}

struct MiscArgs {
    const float* bbox;      // [BQ,10]
    const float* reg;       // [BQ,code]
    const float* vel_div;   // [B] or null
    float* out;             // [BQ,code]
    float* centers;         // [BQ,2]
    float lo[3], span[3];
    long long BQ;
    int Q, code;
};

// box centres in metres for the distance bias (decode_bbox xy, models/bbox/utils.py:63-71)
__global__ void centers_kernel(const MiscArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.BQ) return;
    a.centers[2 * i] = a.bbox[i * 10] * a.span[0] + a.lo[0];
    a.centers[2 * i + 1] = a.bbox[i * 10 + 1] * a.span[1] + a.lo[1];
}

// refine_bbox + velocity / time_diff (models/sparsebev_transformer.py:155-160,179-183; inverse_sigmoid
// models/utils.py:87-102)
__global__ void refine_kernel(const MiscArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.BQ) return;
    const float* r = a.reg + i * a.code;
    float* o = a.out + i * a.code;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float p = a.bbox[i * 10 + d];
        p = fminf(fmaxf(p, 0.f), 1.f);
        const float logit = logf(fmaxf(p, 1e-5f) / fmaxf(1.f - p, 1e-5f));
        const float z = r[d] + logit;
        o[d] = 1.f / (1.f + expf(-z));
    }
    for (int d = 3; d < a.code; ++d) {
        float v = r[d];
        if (d >= 8 && a.vel_div) v = v / a.vel_div[i / a.Q];
        o[d] = v;
    }
}

}  // namespace

extern "C" int sbev_sasa_f32(const float* qkvt, int64_t ld, const float* centers, const uint8_t* mask, float* out,
                             int B, int Q, int H, int head_dim, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && H >= 1, "sbev_sasa_f32: bad sizes");
    SBEV_REQUIRE(head_dim == HD, "sbev_sasa_f32: built for head_dim 32 (got %d)", head_dim);
    SBEV_REQUIRE(ld >= 3 * H * HD + H && ld % 4 == 0, "sbev_sasa_f32: row stride %lld must be >= 3*H*32 + H and a multiple of 4", (long long)ld);
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(qkvt && centers && out, "sbev_sasa_f32: null pointer");
    SBEV_REQUIRE((((uintptr_t)qkvt) & 15) == 0, "sbev_sasa_f32: qkvt must be 16-byte aligned");
    AttnArgs a{qkvt, centers, mask, out, B, Q, H, (int)ld, 1.0f / sqrtf((float)HD)};
    const long long blocks = (long long)B * H * ((Q + 16 * NW - 1) / (16 * NW));
    SBEV_REQUIRE(blocks <= 0x7fffffffLL, "sbev_sasa_f32: too many blocks");
    hipLaunchKernelGGL(sasa_kernel, dim3((unsigned)blocks), dim3(64 * NW), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_sasa_f32");
}

extern "C" int sbev_box_centers(const float* query_bbox, const double* pc_range, float* centers, int64_t BQ,
                                sbev_stream_t stream) {
    if (BQ <= 0) return SBEV_OK;
    SBEV_REQUIRE(query_bbox && pc_range && centers, "sbev_box_centers: null pointer");
    MiscArgs a{};
    a.bbox = query_bbox; a.centers = centers; a.BQ = BQ;
    for (int i = 0; i < 3; ++i) {
        a.lo[i] = (float)pc_range[i];
        a.span[i] = (float)(pc_range[3 + i] - pc_range[i]);
    }
    hipLaunchKernelGGL(centers_kernel, dim3((unsigned)((BQ + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_box_centers");
}

extern "C" int sbev_refine_bbox(const float* query_bbox, const float* reg, const float* vel_div, float* out,
                                int B, int Q, int code_size, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && code_size >= 3, "sbev_refine_bbox: bad sizes");
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(query_bbox && reg && out, "sbev_refine_bbox: null pointer");
    MiscArgs a{};
    a.bbox = query_bbox; a.reg = reg; a.vel_div = vel_div; a.out = out;
    a.BQ = (long long)B * Q; a.Q = Q; a.code = code_size;
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)((a.BQ + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_refine_bbox");
}
