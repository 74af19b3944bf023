// Scale-adaptive self attention, BACKWARD (and the training forward with attention dropout) -- gfx950.
//
// SURVEY.md section 8f rank 4.  The reference differentiates SparseBEVSelfAttention.inner_forward through autograd and
// recomputes it under torch.utils.checkpoint (models/sparsebev_transformer.py:210-234); torch.nn.MultiheadAttention
// materialises the [B*8, Q, Q] probabilities for that.  Here nothing of size Q x Q exists in either direction:
//     S_ij = (q_i . k_j) / sqrt(d) - dist_ij * tau_ih   (-inf under the DN mask),   P = softmax_j(S),   O_i = sum_j Pd_ij v_j
//     Pd = dropout(P)  (training, attn_drop = 0.1: mmcv MultiheadAttention, models/sparsebev_transformer.py:202)
//     dV_j = sum_i Pd_ij dO_i        dPd_ij = dO_i . v_j        dS_ij = P_ij (keep_ij dPd_ij / (1-p) - D_i),  D_i = dO_i . O_i
//     dq_i = sum_j dS_ij k_j / sqrt(d)     dk_j = sum_i dS_ij q_i / sqrt(d)     dtau_ih = -sum_j dS_ij dist_ij
// (dist carries no gradient: calc_bbox_dists is @torch.no_grad, :236-248.)
// The backward runs on the matrix cores (attention_bwd_mfma.hip: a ROW kernel for dq / dtau / log-sum-exp / D_i and a COLUMN
// kernel for dk / dv, flash-attention style, no atomics).  This file keeps the C entry points and the TRAINING FORWARD with
// attention dropout (only taken when attn_drop > 0: plain VALU math, lane = one query of one head, four waves sharing the
// keys; the inference forward is attention.hip).  The dropout keep decision is a hash of (seed, b, h, i, j), so the forward
// and both backward kernels regenerate the same mask.
#include "sbev_common.hpp"

namespace {

constexpr int HD = 32;
constexpr int TILE = 64;

struct SasaBwdArgs {
    const float* qkvt;          // [B, Q, ld]: q | k | v | tau
    const float* bbox;          // [B, Q, 10]
    float lo[2], span[2];
    const unsigned char* mask;  // [Q, Q] or null
    float* out;                 // [B, Q, D]
    int B, Q, H, ld;
    float scale;
    float p_drop, inv_keep;
    unsigned long long seed;
    const unsigned long long* seed_dev;     // null, or a word of device memory ADDED to `seed` when the kernel runs (captured training steps)
};

__device__ __forceinline__ unsigned mix32(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ bool keep_of(const SasaBwdArgs& a, unsigned thr, int b, int h, int i, int j) {
    const unsigned long long idx = (((unsigned long long)b * a.H + h) * a.Q + i) * a.Q + j;
    return mix32((a.seed + (a.seed_dev ? *a.seed_dev : 0ull)) * 0x100000001b3ull + idx) >= thr;
}

constexpr int NW = 4;              // waves per workgroup: wave w walks the key tiles w, w + 4, ...
constexpr int LDT = HD + 4;        // staged rows: 36 floats = 16-byte aligned, so a row is read as 8 broadcast ds_read_b128
constexpr int ROWF = 2 * TILE * LDT + 2 * TILE;      // floats of one wave's private tile area: K | V | centres

// Training forward with attention dropout.  grid = (ceil(Q / 64), H, B), 256 threads = 4 waves x 64 query rows: lane = query
// row, wave w owns the key tiles w, w + 4, ... (its own LDS tile area: no workgroup barrier inside the loops -- a wave's DS
// operations execute in order), and the four partial results of a row are merged through LDS: (max, sum) after pass 1, the
// partial outputs after pass 2.
__global__ __launch_bounds__(64 * NW) void sasa_dropout_fwd_kernel(const SasaBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[NW * ROWF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float (*Ks)[LDT] = reinterpret_cast<float (*)[LDT]>(smem + wave * ROWF);
    float (*Vs)[LDT] = reinterpret_cast<float (*)[LDT]>(smem + wave * ROWF + TILE * LDT);
    float (*Cs)[2] = reinterpret_cast<float (*)[2]>(smem + wave * ROWF + 2 * TILE * LDT);
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = a.H * HD;
    const int i = blockIdx.x * 64 + lane;
    const bool live = i < a.Q;
    const int ic = live ? i : a.Q - 1;
    const float* base = a.qkvt + (long long)b * a.Q * a.ld;
    const unsigned thr = (unsigned)((double)a.p_drop * 4294967296.0);
    float q[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        q[d] = base[(long long)ic * a.ld + h * HD + d] * a.scale;
        acc[d] = 0.f;
    }
    const float cx = a.bbox[((long long)b * a.Q + ic) * 10] * a.span[0] + a.lo[0];
    const float cy = a.bbox[((long long)b * a.Q + ic) * 10 + 1] * a.span[1] + a.lo[1];
    const float tau = base[(long long)ic * a.ld + 3 * D + h];
    float m = -INFINITY, l = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        for (int k0 = wave * TILE; k0 < a.Q; k0 += NW * TILE) {
            __builtin_amdgcn_wave_barrier();                 // the wave is done reading its previous tile
            {
                const int kj = min(k0 + lane, a.Q - 1);
                const float* row = base + (long long)kj * a.ld + h * HD;
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4) {
                    *reinterpret_cast<float4*>(&Ks[lane][4 * d4]) = *reinterpret_cast<const float4*>(row + D + 4 * d4);
                    *reinterpret_cast<float4*>(&Vs[lane][4 * d4]) = *reinterpret_cast<const float4*>(row + 2 * D + 4 * d4);
                }
                Cs[lane][0] = a.bbox[((long long)b * a.Q + kj) * 10] * a.span[0] + a.lo[0];
                Cs[lane][1] = a.bbox[((long long)b * a.Q + kj) * 10 + 1] * a.span[1] + a.lo[1];
            }
            __builtin_amdgcn_wave_barrier();
            const int nk = min(TILE, a.Q - k0);
            for (int jj = 0; jj < nk; ++jj) {
                const int j = k0 + jj;
                if (a.mask && a.mask[(long long)ic * a.Q + j]) continue;
                float s = 0.f;
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4) {
                    const float4 k4 = *reinterpret_cast<const float4*>(&Ks[jj][4 * d4]);
                    s += q[4 * d4] * k4.x; s += q[4 * d4 + 1] * k4.y; s += q[4 * d4 + 2] * k4.z; s += q[4 * d4 + 3] * k4.w;
                }
                const float dx = cx - Cs[jj][0], dy = cy - Cs[jj][1];
                s -= sqrtf(dx * dx + dy * dy) * tau;
                if (pass == 0) {
                    const float mn = fmaxf(m, s);
                    l = l * __expf(m - mn) + __expf(s - mn);
                    m = mn;
                } else if (a.p_drop <= 0.f || keep_of(a, thr, b, h, ic, j)) {
                    const float pd = __expf(s - m) / l * a.inv_keep;
#pragma unroll
                    for (int d4 = 0; d4 < HD / 4; ++d4) {
                        const float4 v4 = *reinterpret_cast<const float4*>(&Vs[jj][4 * d4]);
                        acc[4 * d4] += pd * v4.x; acc[4 * d4 + 1] += pd * v4.y; acc[4 * d4 + 2] += pd * v4.z; acc[4 * d4 + 3] += pd * v4.w;
                    }
                }
            }
        }
        if (pass == 0) {
            // merge the four waves' (max, sum) of every row: all waves end up with the row's global softmax statistics
            __syncthreads();
            float* mg = smem;                                // [NW][2][64]
            mg[(wave * 2 + 0) * 64 + lane] = m;
            mg[(wave * 2 + 1) * 64 + lane] = l;
            __syncthreads();
            float mm = -INFINITY;
#pragma unroll
            for (int w = 0; w < NW; ++w) mm = fmaxf(mm, mg[(w * 2) * 64 + lane]);
            float ll = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float mw = mg[(w * 2) * 64 + lane];
                ll += mw == -INFINITY ? 0.f : mg[(w * 2 + 1) * 64 + lane] * __expf(mw - mm);
            }
            m = mm;
            l = ll;
            __syncthreads();                                 // the merge area aliases the tile areas
        }
    }
    // merge the partial outputs (fixed order: deterministic); [NW][HD][64] floats alias the tile areas
    __syncthreads();
    float* pr = smem;
#pragma unroll
    for (int d = 0; d < HD; ++d) pr[(wave * HD + d) * 64 + lane] = acc[d];
    __syncthreads();
    if (wave != 0 || !live) return;
#pragma unroll
    for (int d = 0; d < HD; ++d)
        a.out[((long long)b * a.Q + i) * D + h * HD + d] =
            (pr[d * 64 + lane] + pr[(HD + d) * 64 + lane]) + (pr[(2 * HD + d) * 64 + lane] + pr[(3 * HD + d) * 64 + lane]);
}

int fill(SasaBwdArgs& a, const float* qkvt, int64_t ld, const float* bbox, const double* pc_range, const uint8_t* mask,
         int B, int Q, int H, int head_dim, float p_drop, uint64_t seed, const uint64_t* seed_dev, const char* who) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && H >= 1, "%s: bad sizes", who);
    SBEV_REQUIRE(head_dim == HD, "%s: built for head_dim 32 (got %d)", who, head_dim);
    SBEV_REQUIRE(ld >= 3 * H * HD + H && ld % 4 == 0 && (((uintptr_t)qkvt) & 15) == 0, "%s: row stride %lld must be >= 3*H*32 + H and a multiple of 4, qkvt 16-byte aligned", who, (long long)ld);
    SBEV_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "%s: need 0 <= attn_drop < 1", who);
    SBEV_REQUIRE((int64_t)Q * Q * H * (B > 0 ? B : 1) >= 0, "%s: overflow", who);
    a.qkvt = qkvt; a.bbox = bbox; a.mask = mask;
    a.B = B; a.Q = Q; a.H = H; a.ld = (int)ld; a.scale = 1.0f / sqrtf((float)HD);
    a.p_drop = p_drop; a.inv_keep = 1.f / (1.f - p_drop); a.seed = seed;
    a.seed_dev = reinterpret_cast<const unsigned long long*>(seed_dev);
    for (int i = 0; i < 2; ++i) {
        a.lo[i] = (float)pc_range[i];
        a.span[i] = (float)(pc_range[3 + i] - pc_range[i]);
    }
    return SBEV_OK;
}

}  // namespace

extern "C" int sbev_sasa_train_fwd_f32(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                                       const uint8_t* mask, float* out, int B, int Q, int H, int head_dim,
                                       float attn_drop, uint64_t seed, sbev_stream_t stream) {
    return sbev_sasa_train_fwd_f32_ds(qkvt, ld, query_bbox, pc_range, mask, out, B, Q, H, head_dim, attn_drop, seed, nullptr, stream);
}

extern "C" int sbev_sasa_train_fwd_f32_ds(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                                          const uint8_t* mask, float* out, int B, int Q, int H, int head_dim,
                                          float attn_drop, uint64_t seed, const uint64_t* seed_dev, sbev_stream_t stream) {
    SasaBwdArgs a{};
    int st = fill(a, qkvt, ld, query_bbox, pc_range, mask, B, Q, H, head_dim, attn_drop, seed, seed_dev, "sbev_sasa_train_fwd_f32");
    if (st != SBEV_OK) return st;
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(qkvt && query_bbox && pc_range && out, "sbev_sasa_train_fwd_f32: null pointer");
    a.out = out;
    hipLaunchKernelGGL(sasa_dropout_fwd_kernel, dim3((unsigned)((Q + 63) / 64), (unsigned)H, (unsigned)B), dim3(64 * NW), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_sasa_train_fwd_f32");
}

extern "C" int sbev_sasa_bwd_f32(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                                 const uint8_t* mask, const float* out, const float* grad_out, float* grad_qkvt,
                                 float* workspace, int B, int Q, int H, int head_dim, float attn_drop, uint64_t seed,
                                 sbev_stream_t stream) {
    return sbev_sasa_bwd_f32_ds(qkvt, ld, query_bbox, pc_range, mask, out, grad_out, grad_qkvt, workspace, B, Q, H, head_dim, attn_drop, seed,
                                nullptr, stream);
}

extern "C" int sbev_sasa_bwd_f32_ds(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                                    const uint8_t* mask, const float* out, const float* grad_out, float* grad_qkvt,
                                    float* workspace, int B, int Q, int H, int head_dim, float attn_drop, uint64_t seed,
                                    const uint64_t* seed_dev, sbev_stream_t stream) {
    SasaBwdArgs a{};
    int st = fill(a, qkvt, ld, query_bbox, pc_range, mask, B, Q, H, head_dim, attn_drop, seed, seed_dev, "sbev_sasa_bwd_f32");
    if (st != SBEV_OK) return st;
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(qkvt && query_bbox && pc_range && out && grad_out && grad_qkvt && workspace, "sbev_sasa_bwd_f32: null pointer");
    float* lse = workspace;                              // [B, H, Q]
    float* dvec = workspace + (long long)B * H * Q;      // [B, H, Q]
    return sbev::launch_sasa_bwd_mfma(qkvt, ld, query_bbox, a.lo, a.span, mask, out, grad_out, grad_qkvt, lse, dvec, B, Q, H,
                                      a.scale, attn_drop, seed, seed_dev, reinterpret_cast<hipStream_t>(stream));
}
