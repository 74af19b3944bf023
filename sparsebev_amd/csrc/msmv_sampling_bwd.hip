// Multi-scale multi-view bilinear sampling, BACKWARD (gfx950).  SURVEY.md section 8f rank 1.
//
// Replaces ms_deformable_col2im_gpu_kernel_gm_c2345 / _c23456 (models/csrc/msmv_sampling/msmv_sampling_backward.cu:
// 108-361, helper :29-105) behind sbev_msmv_bwd.  The reference maps one thread to one (b', q, channel, point) and
// issues, per tap, 4 atomics into grad_value PLUS 3 same-address atomics (grad_attn_weight, grad_sampling_loc x/y)
// on which all 64 channel threads of a point collide.  Here one WAVE owns one (b', q), as in the forward:
//   * grad_sampling_loc and grad_attn_weight are reduced across the wave's 64 lanes in registers and written
//     with plain stores -- the wave is their only writer, so they need NO atomics at all;
//   * only grad_value is scattered with atomics (different queries do hit the same pixels): lane = corner k x
//     channel j, the 4 channels of a lane are j, j+16, j+32, j+48, so each global_atomic_add_f32 instruction
//     covers four 64-byte runs (one per corner) instead of 64 scattered dwords.
// grad wrt the view coordinate is zero by definition (the reference never writes it, :102-104).
#include "sbev_common.hpp"

namespace {

struct BwdArgs {
    const float* feat[SBEV_MAX_LEVELS];
    float* gfeat[SBEV_MAX_LEVELS];
    int H[SBEV_MAX_LEVELS];
    int W[SBEV_MAX_LEVELS];
    long long stride_bo[SBEV_MAX_LEVELS];
    long long stride_v[SBEV_MAX_LEVELS];
    long long stride_g, stride_px;
    const float* loc;
    const float* w;
    const float* gout;     // [B',Q,C,P] (reference layout) or [B,Q,G,T*P,C] (mixing layout, b' = (b*T + t)*G + g)
    int gout_mix, T, G;    // gout_mix != 0: mixing layout
    int want_gfeat;        // 0: grad_value is not needed (frozen features): no atomics at all
    float* gloc;           // [B',Q,P,3]
    float* gw;             // [B',Q,P,L]
    long long n_waves;
    int N, C, Q, P, gdiv;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int L>
__global__ __launch_bounds__(256) void msmv_bwd_kernel(const BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wave = (long long)blockIdx.x * 4 + wv;
    if (wave >= a.n_waves) return;
    const long long bp = wave / a.Q;
    const int k = lane >> 4, kh = k >> 1, kw = k & 1, j = lane & 15;
    const long long bo = bp / a.gdiv, gi = bp - bo * a.gdiv;
    const int P = a.P, C = a.C;
    const float* __restrict__ locq = a.loc + wave * P * 3;
    const float* __restrict__ wq = a.w + wave * P * L;
    // element (c, p) of this item's grad_out: gq[c * gs_c + p * gs_p]
    const float* __restrict__ gq = a.gout + wave * C * P;
    long long gs_c = P, gs_p = 1;
    if (a.gout_mix) {
        const long long q = wave - bp * a.Q, bt = bp / a.G, g = bp - bt * a.G, b = bt / a.T, t = bt - b * a.T;
        gq = a.gout + ((((b * a.Q + q) * a.G + g) * a.T + t) * (long long)P) * C;
        gs_c = 1; gs_p = C;
    }
    const float nm1 = (float)(a.N - 1);

    for (int p = 0; p < P; ++p) {
        const float x = locq[p * 3 + 0], y = locq[p * 3 + 1];
        int view = (int)roundf(locq[p * 3 + 2] * nm1);
        view = min(max(view, 0), a.N - 1);
        float gx = 0.f, gy = 0.f, gwl[L];
#pragma unroll
        for (int l = 0; l < L; ++l) gwl[l] = 0.f;
        for (int c0 = 0; c0 < C; c0 += 64) {
            float g[4];
            bool cok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + j + 16 * i;
                cok[i] = c < C;
                g[i] = cok[i] ? gq[c * gs_c + p * gs_p] : 0.f;
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int H = a.H[l], W = a.W[l];
                const float h_im = y * (float)(H - 1), w_im = x * (float)(W - 1);
                const bool lvl_ok = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
                if (!lvl_ok) continue;                                   // wave-uniform
                const float hf = floorf(h_im), wf = floorf(w_im);
                const float lh = h_im - hf, lw = w_im - wf;
                const int hc = (int)hf + kh, wc = (int)wf + kw;
                const bool inb = hc >= 0 && hc <= H - 1 && wc >= 0 && wc <= W - 1;
                const float ch = kh ? lh : 1.f - lh, cwid = kw ? lw : 1.f - lw;
                const float cw = ch * cwid;
                const float wl = wq[p * L + l];
                const long long off = bo * a.stride_bo[l] + gi * a.stride_g + view * a.stride_v[l] +
                                      ((long long)min(max(hc, 0), H - 1) * W + min(max(wc, 0), W - 1)) * a.stride_px;
                const float* f = a.feat[l] + off;
                float* gf = a.gfeat[l] + off;
                float dot = 0.f;                                         // sum_i g_i * v_i  (this corner, this lane)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + j + 16 * i;
                    if (inb && cok[i]) {
                        dot += g[i] * f[c];
                        if (a.want_gfeat) atomicAdd(gf + c, cw * wl * g[i]);   // global_atomic_add_f32 (-munsafe-fp-atomics)
                    }
                }
                gwl[l] += cw * dot;                                      // d/d weight_l = sum_c g_c * bilinear_c
                // d bilinear / d w_im = hh (v2 - v1) + lh (v4 - v3);  d / d h_im = hw (v3 - v1) + lw (v4 - v2)
                gx += (kw ? 1.f : -1.f) * ch * dot * wl * (float)(W - 1);
                gy += (kh ? 1.f : -1.f) * cwid * dot * wl * (float)(H - 1);
            }
        }
        gx = wave_sum(gx);
        gy = wave_sum(gy);
#pragma unroll
        for (int l = 0; l < L; ++l) gwl[l] = wave_sum(gwl[l]);
        if (lane == 0) {
            float* o = a.gloc + (wave * P + p) * 3;
            o[0] = gx; o[1] = gy; o[2] = 0.f;
            float* ow = a.gw + (wave * P + p) * L;
#pragma unroll
            for (int l = 0; l < L; ++l) ow[l] = gwl[l];
        }
    }
}


// ---- C == 64 fast path (the decoder's 64 channels per group) ------------------------------------------------------
// Same ownership (one wave per (b', q)) but lane = CHANNEL and the 4 bilinear corners live in registers: every load
// and every atomic instruction covers one corner's 64 contiguous channels = two full 128-byte lines, and the channel
// dot products need no cross-lane work until the per-point DPP sums.  One point at a time: all 4*L corner taps of the
// point are requested up front as unconditional (clamped) loads, then consumed.
// Bound: the L2 atomic units.  Config 2 issues 118 M float atomics per call and runs at 370 us = 318 G atomics/s
// = ~1.2 per L2-channel clock (128 channels), with clustered (projected) and with uniformly random sample locations
// alike -- so it is the atomic ALU rate, not same-address contention, and only fewer atomic dwords would go faster.
// Occupancy is what keeps that pipe full, so the kernel is kept small (32-bit offsets, no multi-point staging): a
// 4-point-chunk version needed 312 registers (1 wave per SIMD) and took 848 us; 16-byte-per-lane loads with
// stride-16-byte atomics 1635 us (4x more cache lines per instruction); the generic corner x 16-channel kernel 395 us.
template <int L>
__global__ __launch_bounds__(256) void msmv_bwd_c64_kernel(const BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wave = (long long)blockIdx.x * 4 + wv;
    if (wave >= a.n_waves) return;
    const long long bp = wave / a.Q;
    const long long bo = bp / a.gdiv, gi = bp - bo * a.gdiv;
    const int P = a.P;
    const float* __restrict__ locq = a.loc + wave * P * 3;
    const float* __restrict__ wq = a.w + wave * P * L;
    const float* __restrict__ gq = a.gout + wave * 64 * P + lane * P;        // this lane's channel row of [C, P]
    int gs_p = 1;
    if (a.gout_mix) {                                                        // [B,Q,G,T*P,C]: point rows of 64 channels
        const long long q = wave - bp * a.Q, bt = bp / a.G, g = bp - bt * a.G, b = bt / a.T, t = bt - b * a.T;
        gq = a.gout + ((((b * a.Q + q) * a.G + g) * a.T + t) * (long long)P) * 64 + lane;
        gs_p = 64;
    }
    const float nm1 = (float)(a.N - 1);
    const float* fb[L];
    float* gb[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {                     // (b', group) base of every level: the rest fits 32 bits (checked by the host)
        const long long o = bo * a.stride_bo[l] + gi * a.stride_g + lane;
        fb[l] = a.feat[l] + o;
        gb[l] = a.gfeat[l] + o;
    }

    for (int p = 0; p < P; ++p) {
        const float x = locq[p * 3 + 0], y = locq[p * 3 + 1];
        int view = (int)roundf(locq[p * 3 + 2] * nm1);
        view = min(max(view, 0), a.N - 1);
        const float g = gq[p * gs_p];
        float v[L][4], lhs[L], lws[L];
        int off[L][4], msk[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int H = a.H[l], W = a.W[l];
            const float h_im = y * (float)(H - 1), w_im = x * (float)(W - 1);
            const bool lvl_ok = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            lhs[l] = h_im - hf;
            lws[l] = w_im - wf;
            const int vb = view * (int)a.stride_v[l];
            int mk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int hc = (int)hf + (k >> 1), wc = (int)wf + (k & 1);
                const bool inb = lvl_ok && hc >= 0 && hc <= H - 1 && wc >= 0 && wc <= W - 1;
                mk |= inb ? (1 << k) : 0;
                off[l][k] = vb + (min(max(hc, 0), H - 1) * W + min(max(wc, 0), W - 1)) * (int)a.stride_px;
                v[l][k] = fb[l][off[l][k]];
            }
            msk[l] = mk;
        }
        __builtin_amdgcn_sched_barrier(0);
        float gx = 0.f, gy = 0.f, gwl[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float lh = lhs[l], lw = lws[l], hh = 1.f - lh, hw = 1.f - lw;
            const int mk = msk[l];
            // out-of-map corners contribute 0 to every sum: mask the VALUE, as the reference does (:54-67)
            const float m0 = mk & 1 ? v[l][0] : 0.f, m1 = mk & 2 ? v[l][1] : 0.f, m2 = mk & 4 ? v[l][2] : 0.f, m3 = mk & 8 ? v[l][3] : 0.f;
            const float wl = wq[p * L + l];
            gwl[l] = g * ((hh * hw * m0 + hh * lw * m1) + (lh * hw * m2 + lh * lw * m3));
            gx += g * (hh * (m1 - m0) + lh * (m3 - m2)) * (wl * (float)(a.W[l] - 1));
            gy += g * (hw * (m2 - m0) + lw * (m3 - m1)) * (wl * (float)(a.H[l] - 1));
            const float gv = wl * g;
            if (!a.want_gfeat) continue;
            if (mk & 1) atomicAdd(gb[l] + off[l][0], hh * hw * gv);      // global_atomic_add_f32, no return; wave-uniform guards
            if (mk & 2) atomicAdd(gb[l] + off[l][1], hh * lw * gv);
            if (mk & 4) atomicAdd(gb[l] + off[l][2], lh * hw * gv);
            if (mk & 8) atomicAdd(gb[l] + off[l][3], lh * lw * gv);
        }
        gx = sbev::wave_sum_dpp(gx);
        gy = sbev::wave_sum_dpp(gy);
#pragma unroll
        for (int l = 0; l < L; ++l) gwl[l] = sbev::wave_sum_dpp(gwl[l]);
        if (lane == 0) {
            float* o = a.gloc + (wave * P + p) * 3;
            o[0] = gx; o[1] = gy; o[2] = 0.f;
            float* ow = a.gw + (wave * P + p) * L;
#pragma unroll
            for (int l = 0; l < L; ++l) ow[l] = gwl[l];
        }
    }
}

template <int L>
int launch_bwd(const BwdArgs& a, hipStream_t s) {
    const long long blocks = (a.n_waves + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        sbev::set_error("sbev_msmv_bwd: B'*Q too large");
        return SBEV_EINVAL;
    }
    bool fast = a.C == 64 && a.stride_px < (1 << 20);
    for (int l = 0; l < L; ++l)      // everything below the (b', group) base must fit a 32-bit offset
        fast = fast && a.stride_v[l] * a.N + (long long)a.H[l] * a.W[l] * a.stride_px < 0x7fffffffLL;
    if (fast)
        hipLaunchKernelGGL(msmv_bwd_c64_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(msmv_bwd_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return sbev::check_launch("sbev_msmv_bwd");
}

}  // namespace

static int msmv_bwd_impl(const void* const* feats, void* const* grad_feats, const int32_t* hw, int L,
                         int64_t Bp, int N, int C, int Q, int P,
                         int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                         const float* loc, const float* weights, const float* grad_out, int grad_out_layout, int T, int G,
                         float* grad_loc, float* grad_weights, sbev_stream_t stream) {
    SBEV_REQUIRE(feats && hw && stride_bo && stride_v, "sbev_msmv_bwd: null descriptor array");
    SBEV_REQUIRE(L >= 1 && L <= SBEV_MAX_LEVELS, "sbev_msmv_bwd: L=%d", L);
    SBEV_REQUIRE(P >= 1 && P <= SBEV_MAX_POINTS, "sbev_msmv_bwd: num_point exceed limits (P=%d)", P);
    SBEV_REQUIRE(C >= 1 && N >= 1 && Q >= 0 && Bp >= 0 && gdiv >= 1, "sbev_msmv_bwd: bad sizes");
    SBEV_REQUIRE(grad_out_layout == SBEV_OUT_REF || grad_out_layout == SBEV_OUT_MIX, "sbev_msmv_bwd: grad_out_layout %d", grad_out_layout);
    if (grad_out_layout == SBEV_OUT_MIX)
        SBEV_REQUIRE(T >= 1 && G >= 1 && Bp % ((int64_t)T * G) == 0, "sbev_msmv_bwd: B'=%lld is not B*T*G (T=%d, G=%d)", (long long)Bp, T, G);
    if (Bp == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(loc && weights && grad_out && grad_loc && grad_weights, "sbev_msmv_bwd: null pointer");
    BwdArgs a{};
    a.want_gfeat = grad_feats != nullptr;
    for (int l = 0; l < L; ++l) {
        SBEV_REQUIRE(feats[l] && (!grad_feats || grad_feats[l]), "sbev_msmv_bwd: level %d pointer is null", l);
        a.feat[l] = static_cast<const float*>(feats[l]);
        a.gfeat[l] = grad_feats ? static_cast<float*>(grad_feats[l]) : const_cast<float*>(static_cast<const float*>(feats[l]));   // never written when !want_gfeat
        a.H[l] = hw[2 * l];
        a.W[l] = hw[2 * l + 1];
        a.stride_bo[l] = stride_bo[l];
        a.stride_v[l] = stride_v[l];
    }
    a.stride_g = stride_g; a.stride_px = stride_px;
    a.loc = loc; a.w = weights; a.gout = grad_out; a.gloc = grad_loc; a.gw = grad_weights;
    a.gout_mix = grad_out_layout == SBEV_OUT_MIX; a.T = T; a.G = G;
    a.n_waves = Bp * Q;
    a.N = N; a.C = C; a.Q = Q; a.P = P; a.gdiv = gdiv;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (L) {
        case 1: return launch_bwd<1>(a, s);
        case 2: return launch_bwd<2>(a, s);
        case 3: return launch_bwd<3>(a, s);
        case 4: return launch_bwd<4>(a, s);
        default: return launch_bwd<5>(a, s);
    }
}

extern "C" int sbev_msmv_bwd(const void* const* feats, void* const* grad_feats, const int32_t* hw, int L,
                             int64_t Bp, int N, int C, int Q, int P,
                             int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                             const float* loc, const float* weights, const float* grad_out,
                             float* grad_loc, float* grad_weights, sbev_stream_t stream) {
    SBEV_REQUIRE(grad_feats != nullptr, "sbev_msmv_bwd: null descriptor array");
    return msmv_bwd_impl(feats, grad_feats, hw, L, Bp, N, C, Q, P, gdiv, stride_bo, stride_g, stride_v, stride_px, loc, weights,
                         grad_out, SBEV_OUT_REF, 1, 1, grad_loc, grad_weights, stream);
}

extern "C" int sbev_msmv_bwd_ex(const void* const* feats, void* const* grad_feats, const int32_t* hw, int L,
                                int64_t Bp, int N, int C, int Q, int P,
                                int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                                const float* loc, const float* weights, const float* grad_out, int grad_out_layout, int T, int G,
                                float* grad_loc, float* grad_weights, sbev_stream_t stream) {
    return msmv_bwd_impl(feats, grad_feats, hw, L, Bp, N, C, Q, P, gdiv, stride_bo, stride_g, stride_v, stride_px, loc, weights,
                         grad_out, grad_out_layout, T, G, grad_loc, grad_weights, stream);
}
