// Row-wise helper ops of the decoder layer as DEVICE functions (one definition, used by their own stand-alone kernels in
// gemm.hip / attention.hip / layout.hip and by the co-launch kernel in gemm.hip that runs two independent ones side by
// side).  Every function handles the rows of ONE 256-thread workgroup `block`; out-of-range rows return.
#pragma once
#include "sbev_common.hpp"

namespace sbev_ops {

struct ReduceArgs {
    const float* slabs;  // [splits, M, N]
    const float* bias;   // [N] or null
    const float* res;    // [M, N] or null
    const float* ln_w;   // [N] or null -> no LayerNorm
    const float* ln_b;
    const float* post;   // [M, N] or null: added after LayerNorm / ReLU  (x = query_feat + pos-encoding form)
    float* Y;            // [M, N]
    long long M;
    int N, splits, relu;
    float eps;
};

// one wave per output row (N <= 1024, N % 4 == 0): sum the split-K slabs, + bias, (+ residual), optional
// LayerNorm over the row (two-pass in registers), optional ReLU.
__device__ __forceinline__ void reduce_rows(const ReduceArgs& a, unsigned block) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)block * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    constexpr int MAXV = 4;                              // float4 chunks per lane: N <= 64*4*4 = 1024
    float4 v[MAXV];
    const int nvec = a.N / 4;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int i4 = lane + 64 * c;
        v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < nvec) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int z = 0; z < a.splits; ++z) {
                const float4 p = *reinterpret_cast<const float4*>(a.slabs + ((long long)z * a.M + row) * a.N + i4 * 4);
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
            if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + i4 * 4);
                acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
            }
            if (a.relu && !a.ln_w) {
                acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
            }
            if (a.res) {
                const float4 r = *reinterpret_cast<const float4*>(a.res + row * a.N + i4 * 4);
                acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
            }
            v[c] = acc;
            s += (acc.x + acc.y) + (acc.z + acc.w);
        }
    }
    if (a.ln_w) {
        s = sbev::wave_sum_dpp(s);
        const float mean = s / (float)a.N;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXV; ++c)
            if (lane + 64 * c < nvec) {
                const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        q = sbev::wave_sum_dpp(q);
        const float rstd = rsqrtf(q / (float)a.N + a.eps);
#pragma unroll
        for (int c = 0; c < MAXV; ++c) {
            const int i4 = lane + 64 * c;
            if (i4 < nvec) {
                const float4 g = *reinterpret_cast<const float4*>(a.ln_w + i4 * 4);
                const float4 b = *reinterpret_cast<const float4*>(a.ln_b + i4 * 4);
                float4 o;
                o.x = (v[c].x - mean) * rstd * g.x + b.x;
                o.y = (v[c].y - mean) * rstd * g.y + b.y;
                o.z = (v[c].z - mean) * rstd * g.z + b.z;
                o.w = (v[c].w - mean) * rstd * g.w + b.w;
                if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                v[c] = o;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int i4 = lane + 64 * c;
        if (i4 < nvec) {
            if (a.post) {
                const float4 r = *reinterpret_cast<const float4*>(a.post + row * a.N + i4 * 4);
                v[c].x += r.x; v[c].y += r.y; v[c].z += r.z; v[c].w += r.w;
            }
            *reinterpret_cast<float4*>(a.Y + row * a.N + i4 * 4) = v[c];
        }
    }
}


struct MiscArgs {
    const float* bbox;      // [BQ,10]
    const float* reg;       // [BQ,code]
    const float* vel_div;   // [B] or null
    float* out;             // [BQ,code]
    long long BQ;
    int Q, code;
};

// refine_bbox + velocity / time_diff (models/sparsebev_transformer.py:155-160,179-183; inverse_sigmoid
// models/utils.py:87-102)
__device__ __forceinline__ void refine_rows(const MiscArgs& a, unsigned block) {
    const long long i = (long long)block * 256 + threadIdx.x;
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
        if (d >= 8 && a.vel_div) v = v / a.vel_div[(unsigned)i / (unsigned)a.Q];     // BQ < 2^31 (host-checked)
        o[d] = v;
    }
}


struct PosArgs {
    const float* x;     // [M, ldx] (first 3 columns used)
    const float* w;     // [N, 3]
    const float* b;     // [N]
    const float* ln_w;  // [N]
    const float* ln_b;
    float* y;           // [M, N]
    long long M;
    int N, ldx;
    float eps;
    float* pre = nullptr;   // [M, N] or null: the Linear's output BEFORE LayerNorm (kept for the backward pass in training)
};

// Linear(3 -> N) + LayerNorm(N) + ReLU, one wave per row (N <= 1024, N % 4 == 0): 3 FMAs per output are not
// a GEMM.  (position_encoder[0..2], models/sparsebev_transformer.py:116-119)
__device__ __forceinline__ void lin3_rows(const PosArgs& a, unsigned block) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)block * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const float x0 = a.x[row * a.ldx], x1 = a.x[row * a.ldx + 1], x2 = a.x[row * a.ldx + 2];
    constexpr int MAXV = 4;
    float v[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int n0 = (lane + 64 * c) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = 0.f;
            if (n0 + e < a.N) {
                const float* w = a.w + (long long)(n0 + e) * 3;
                t = ((x0 * w[0] + x1 * w[1]) + x2 * w[2]) + a.b[n0 + e];
                s += t;
            }
            v[c][e] = t;
        }
        if (a.pre && n0 < a.N) *reinterpret_cast<float4*>(a.pre + row * a.N + n0) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)a.N;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((lane + 64 * c) * 4 + e < a.N) {
                const float d = v[c][e] - mean;
                q += d * d;
            }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)a.N + a.eps);
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int n0 = (lane + 64 * c) * 4;
        if (n0 < a.N) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf((v[c][e] - mean) * rstd * a.ln_w[n0 + e] + a.ln_b[n0 + e], 0.f);
            *reinterpret_cast<float4*>(a.y + row * a.N + n0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}


}  // namespace sbev_ops
