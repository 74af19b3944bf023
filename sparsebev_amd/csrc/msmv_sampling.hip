// Multi-scale multi-view bilinear sampling, forward -- hand-written for gfx950 (CDNA4, wave64).
//
// Replaces the reference's CUDA op ms_deformable_im2col_gpu_kernel_c2345 / _c23456
// (models/csrc/msmv_sampling/msmv_sampling_forward.cu:75-267) behind sbev_msmv_fwd (include/sbev_hip.h).
// Written from the op's mathematical definition, not from the CUDA source: the reference maps one
// THREAD to one (b', q, channel) and issues 4*L*P scalar 4-byte loads per thread, re-reading the
// query's coordinates in all 64 channel threads.  Here one WAVE owns one (b', q):
//
//   lane = corner k (0..3) * 16 + channel quad j (0..15)
//   one wave-wide 16-byte load = the 4 bilinear corners x 64 fp32 channels of ONE (point, level) tap
//       = 4 x 256 B contiguous segments, i.e. every HBM/L2 request is a full 128-B line pair;
//   the chunk's coordinates and level weights arrive in ONE coalesced load; the tap geometry is computed once per
//   wave with lanes = (tap, corner) pairs and handed to the gathering lanes with ds_bpermute, so the 4*L taps of a
//   4-point chunk are independent back-to-back loads -> up to 20 KiB in flight per wave, ~15 VALU ops per tap;
//   each lane FMAs its float4 with one coefficient = bilinear-corner weight x level weight;
//   the 4 corner partials are combined once per 4-point chunk with a reduce-scatter built from
//   v_permlane16_swap / v_permlane32_swap (no LDS, no ds_bpermute), which leaves each 16-lane row
//   holding exactly the float4 it has to store, so the wave writes ONE fully coalesced 1-KiB row.
//
// Bound: HBM/L2 gather bandwidth (about 2 flop per loaded float).  Algorithmic bytes per sampled point
// (SURVEY.md section 8d): L*4*C*sizeof(feat) + 12 + 4L + 4C  (4124 B at L=4, C=64, fp32).
#include "sbev_common.hpp"
#include <atomic>
#include <cstdlib>

namespace {

#include "msmv_common.hpp"

template <int L, typename FT, int OUT, int QPW, bool BUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(msmv_min_waves<L, FT>()))) void msmv_fwd_kernel(const MsmvArgs a) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long item0 = ((long long)blockIdx.x * 4 + wv) * QPW;
    if (item0 >= a.n_waves) return;
    float lv_pref = 0.f;
    if (QPW > 1) {
        if (lane < a.P * 3) lv_pref = a.loc[item0 * a.P * 3 + lane];
        if (lane >= 16 && lane < 16 + a.P * L) lv_pref = a.w[item0 * a.P * L + (lane - 16)];
    }
#pragma unroll 1
    for (int qi = 0; qi < QPW; ++qi) {
    const long long wave = item0 + qi;                       // = b' * Q + q  (wave-uniform)
    if (wave >= a.n_waves) break;
    // index arithmetic in 32 bits (n_waves < 2^31, host-checked): every 64-bit division here is ~100 scalar instructions,
    // and five of them per item were the majority of the kernel's instruction count (the kernel is issue-bound)
    const unsigned uw = (unsigned)wave;
    const unsigned bp = uw / (unsigned)a.Q;
    const int q = (int)(uw - bp * (unsigned)a.Q);
    const int k = lane >> 4;
    const int j4 = (lane & 15) * 4;
    unsigned ubo = bp / (unsigned)a.gdiv;
    const long long gi = bp - ubo * (unsigned)a.gdiv;
    if (a.ring_T) {                                          // (b, t) -> (b, slot[t]) in the per-frame feature ring
        const unsigned b = ubo / (unsigned)a.ring_T;
        ubo = b * (unsigned)a.n_slots + (unsigned)a.slots[(int)(ubo - b * (unsigned)a.ring_T)];
    }
    const long long bo = ubo;
    const int P = a.P, C = a.C;
    const float* __restrict__ locq = a.loc + wave * P * 3;
    const float* __restrict__ wq = a.w + wave * P * L;
    const float nm1 = (float)(a.N - 1);

    for (int c0 = 0; c0 < C; c0 += 64) {
        const bool chan_ok = (c0 + j4) < C;
        const int cj = chan_ok ? c0 + j4 : 0;
        long long slab[L];
#pragma unroll
        for (int l = 0; l < L; ++l) slab[l] = bo * a.stride_bo[l] + gi * a.stride_g;       // wave-uniform
        TapSrc<L, FT, BUF> src;
        src.init(a, slab, cj);

        for (int p0 = 0; p0 < P; p0 += 4) {
            // ONE coalesced request for this chunk's 12 coordinates (lanes 0..11) and 4*L level weights (lanes
            // 16..16+4L).  Nothing else is loaded between here and the feature taps, so all 4*L tap loads can be issued
            // back to back.  (Loading a weight inside the tap made hipcc guard it with the in-bounds branch and wait
            // vmcnt(0) -- draining every earlier tap: 99 us instead of 64.)
            const int npts = min(4, P - p0);
            float lv = 0.f;
            if (QPW > 1) {
                lv = lv_pref;
                if (qi + 1 < QPW) {                              // unconditional (clamped) request for the next item
                    const long long wn = wave + 1 < a.n_waves ? wave + 1 : wave;
                    float nx = 0.f;
                    if (lane < P * 3) nx = a.loc[wn * P * 3 + lane];
                    if (lane >= 16 && lane < 16 + P * L) nx = a.w[wn * P * L + (lane - 16)];
                    lv_pref = nx;
                }
            } else {
                if (lane < npts * 3) lv = locq[p0 * 3 + lane];
                if (lane >= 16 && lane < 16 + npts * L) lv = wq[p0 * L + (lane - 16)];
            }
#include "msmv_chunk.inc"
            if (OUT == SBEV_OUT_REF) {
                // out[b', q, c, p]: scatter by CHANNEL -- row k ends up with channel c0 + 4j + k of all 4 points
                float r[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
                    r[pp] = corner_reduce_scatter(acc[pp].x, acc[pp].y, acc[pp].z, acc[pp].w);
                float* o = a.out + (wave * C + (c0 + j4 + k)) * P + p0;
                if (chan_ok) {
                    if (P == 4) {
                        *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);   // 64 lanes -> 1 KiB row
                    } else {
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp)
                            if (p0 + pp < P) o[pp] = r[pp];
                    }
                }
            } else {
                // out[b, q, g, t*P + p, c] with b' = (b*T + t)*G + g: scatter by POINT -- row k ends up with
                // the 4 channels 4j..4j+3 of point p0 + k, i.e. the wave stores 4 contiguous 256-B point rows
                float4 s;
                s.x = corner_reduce_scatter(acc[0].x, acc[1].x, acc[2].x, acc[3].x);
                s.y = corner_reduce_scatter(acc[0].y, acc[1].y, acc[2].y, acc[3].y);
                s.z = corner_reduce_scatter(acc[0].z, acc[1].z, acc[2].z, acc[3].z);
                s.w = corner_reduce_scatter(acc[0].w, acc[1].w, acc[2].w, acc[3].w);
                const unsigned bt = bp / (unsigned)a.G;
                const int g = (int)(bp - bt * (unsigned)a.G);
                const long long b = bt / (unsigned)a.T;
                const int t = (int)(bt - (unsigned)b * (unsigned)a.T);
                if (chan_ok && p0 + k < P) {
                    float* o = a.out + ((((b * a.Q + q) * a.G + g) * a.T + t) * (long long)P + p0 + k) * C + c0 + j4;
                    *reinterpret_cast<float4*>(o) = s;
                }
            }
        }
    }
    }   // item loop
}

template <int L, typename FT, bool BUF>
int launch_l(const MsmvArgs& a, int out_layout, hipStream_t s) {
    // pipelined items per wave when a query is a single chunk and there are enough items to keep every SIMD fed
    const bool pipe = a.P <= 4 && a.C <= 64 && a.n_waves >= 4LL * 1024 * SBEV_MSMV_QPW;
    const long long waves = pipe ? (a.n_waves + SBEV_MSMV_QPW - 1) / SBEV_MSMV_QPW : a.n_waves;
    const long long blocks = (waves + 3) / 4;
    if (blocks <= 0) return SBEV_OK;
    if (blocks > 0x7fffffffLL || a.n_waves > 0x7fffffffLL) {
        sbev::set_error("sbev_msmv_fwd: B'*Q = %lld too large for one launch", a.n_waves);
        return SBEV_EINVAL;
    }
    hipEvent_t e0, e1;
    const bool prof = sbev::profile_begin(s, &e0, &e1);
    if (out_layout == SBEV_OUT_REF) {
        if (pipe) hipLaunchKernelGGL((msmv_fwd_kernel<L, FT, SBEV_OUT_REF, SBEV_MSMV_QPW, BUF>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((msmv_fwd_kernel<L, FT, SBEV_OUT_REF, 1, BUF>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    } else {
        if (pipe) hipLaunchKernelGGL((msmv_fwd_kernel<L, FT, SBEV_OUT_MIX, SBEV_MSMV_QPW, BUF>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((msmv_fwd_kernel<L, FT, SBEV_OUT_MIX, 1, BUF>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    if (prof) sbev::profile_end(s, e0, e1);
    return sbev::check_launch("sbev_msmv_fwd");
}

template <typename FT, bool BUF>
int launch_b(const MsmvArgs& a, int L, int out_layout, hipStream_t s) {
    switch (L) {
        case 1: return launch_l<1, FT, BUF>(a, out_layout, s);
        case 2: return launch_l<2, FT, BUF>(a, out_layout, s);
        case 3: return launch_l<3, FT, BUF>(a, out_layout, s);
        case 4: return launch_l<4, FT, BUF>(a, out_layout, s);
        default: return launch_l<5, FT, BUF>(a, out_layout, s);
    }
}
// buffer-load taps (hardware zeros for an out-of-map corner) when every slab of every level is below 2 GiB;
// sbev_msmv_buffer_taps(0) / SBEV_MSMV_NO_BUF=1 force the 64-bit global-load path (tests, A/B; same results)
std::atomic<int> g_buffer_taps{getenv("SBEV_MSMV_NO_BUF") ? 0 : 1};
template <typename FT>
int launch_t(const MsmvArgs& a, int L, int out_layout, bool slabs_fit_buffer, hipStream_t s) {
    return slabs_fit_buffer && g_buffer_taps.load(std::memory_order_relaxed) != 0 ? launch_b<FT, true>(a, L, out_layout, s)
                                                                                   : launch_b<FT, false>(a, L, out_layout, s);
}

}  // namespace

static int msmv_fwd_impl(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                         int64_t Bp, int N, int C, int Q, int P,
                         int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v,
                         int64_t stride_px, const float* loc, const float* weights, float* out,
                         int out_layout, int T, int G, const int32_t* frame_slots, int n_slots, sbev_stream_t stream) {
    SBEV_REQUIRE(feats && hw && stride_bo && stride_v, "sbev_msmv_fwd: null descriptor array");
    SBEV_REQUIRE(L >= 1 && L <= SBEV_MAX_LEVELS, "sbev_msmv_fwd: L=%d not in 1..%d", L, SBEV_MAX_LEVELS);
    SBEV_REQUIRE(P >= 1 && P <= SBEV_MAX_POINTS, "sbev_msmv_fwd: num_point exceed limits (P=%d > %d)", P, SBEV_MAX_POINTS);
    SBEV_REQUIRE(C >= 4 && C % 4 == 0, "sbev_msmv_fwd: C=%d must be a positive multiple of 4", C);
    SBEV_REQUIRE(N >= 1 && Q >= 0 && Bp >= 0 && gdiv >= 1, "sbev_msmv_fwd: bad sizes");
    SBEV_REQUIRE(feat_dtype == SBEV_F32 || feat_dtype == SBEV_BF16 || feat_dtype == SBEV_F16, "sbev_msmv_fwd: feat_dtype %d", feat_dtype);
    SBEV_REQUIRE(out_layout == SBEV_OUT_REF || out_layout == SBEV_OUT_MIX, "sbev_msmv_fwd: out_layout %d", out_layout);
    SBEV_REQUIRE(stride_px % 4 == 0 && stride_g % 4 == 0, "sbev_msmv_fwd: pixel/group strides must be multiples of 4 elements");
    if (out_layout == SBEV_OUT_MIX)
        SBEV_REQUIRE(T >= 1 && G >= 1 && Bp % ((int64_t)T * G) == 0, "sbev_msmv_fwd: B'=%lld is not B*T*G (T=%d, G=%d)", (long long)Bp, T, G);
    const bool empty = Bp == 0 || Q == 0;          // an empty call still validates its level descriptors
    SBEV_REQUIRE(empty || (loc && weights && out), "sbev_msmv_fwd: null loc/weights/out");
    MsmvArgs a{};
    bool fit = true;        // every (sample-batch, group) slab + a lane's channel offset addressable by a 31-bit BYTE offset
    const int64_t esize = feat_dtype == SBEV_F32 ? 4 : 2;
    for (int l = 0; l < L; ++l) {
        SBEV_REQUIRE(empty || feats[l] != nullptr, "sbev_msmv_fwd: feats[%d] is null", l);
        SBEV_REQUIRE(hw[2 * l] >= 1 && hw[2 * l + 1] >= 1, "sbev_msmv_fwd: level %d has empty map", l);
        SBEV_REQUIRE(stride_bo[l] % 4 == 0 && stride_v[l] % 4 == 0, "sbev_msmv_fwd: level %d strides must be multiples of 4 elements", l);
        // the kernel keeps a tap's offset INSIDE one sample-batch slab (view * stride_v + pixel * stride_px) in 32 bits with
        // bit 31 as its "outside the map" flag; the slab base itself is 64-bit.  Refuse maps one slab of which does not fit.
        SBEV_REQUIRE(stride_v[l] >= 0 && stride_px >= 0 &&
                         (int64_t)(N - 1) * stride_v[l] + ((int64_t)hw[2 * l] * hw[2 * l + 1] - 1) * stride_px + C <= 0x7fffffffLL,
                     "sbev_msmv_fwd: level %d: one (sample-batch) slab spans %lld elements, the in-slab tap offset is 32-bit (limit 2^31 - 1)",
                     l, (long long)((int64_t)(N - 1) * stride_v[l] + ((int64_t)hw[2 * l] * hw[2 * l + 1] - 1) * stride_px + C));
        fit = fit && msmv_slab_fits_buffer(N, hw[2 * l], hw[2 * l + 1], stride_v[l], stride_px, C, esize);
        a.feat[l] = feats[l];
        a.H[l] = hw[2 * l];
        a.W[l] = hw[2 * l + 1];
        a.stride_bo[l] = stride_bo[l];
        a.stride_v[l] = stride_v[l];
    }
    if (empty) return SBEV_OK;
    a.stride_g = stride_g;
    a.stride_px = stride_px;
    a.loc = loc;
    a.w = weights;
    a.out = out;
    a.n_waves = Bp * Q;
    a.N = N; a.C = C; a.Q = Q; a.P = P; a.gdiv = gdiv; a.T = T; a.G = G;
    if (frame_slots) {
        SBEV_REQUIRE(T >= 1 && T <= SBEV_MAX_FRAMES && n_slots >= T && gdiv == G && Bp % ((int64_t)T * G) == 0,
                     "sbev_msmv_fwd_ring: need 1 <= T <= %d, n_slots >= T, gdiv == G, B' = B*T*G", SBEV_MAX_FRAMES);
        a.ring_T = T;
        a.n_slots = n_slots;
        for (int t = 0; t < T; ++t) {
            SBEV_REQUIRE(frame_slots[t] >= 0 && frame_slots[t] < n_slots, "sbev_msmv_fwd_ring: frame_slots[%d] = %d out of range", t, frame_slots[t]);
            a.slots[t] = frame_slots[t];
        }
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return feat_dtype == SBEV_F32 ? launch_t<float>(a, L, out_layout, fit, s)
           : feat_dtype == SBEV_F16 ? launch_t<_Float16>(a, L, out_layout, fit, s)
                                    : launch_t<unsigned short>(a, L, out_layout, fit, s);
}

extern "C" int sbev_msmv_buffer_taps(int enable) {
    return g_buffer_taps.exchange(enable ? 1 : 0, std::memory_order_relaxed);
}

extern "C" int sbev_msmv_fwd(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                             int64_t Bp, int N, int C, int Q, int P,
                             int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v,
                             int64_t stride_px, const float* loc, const float* weights, float* out,
                             int out_layout, int T, int G, sbev_stream_t stream) {
    return msmv_fwd_impl(feats, hw, L, feat_dtype, Bp, N, C, Q, P, gdiv, stride_bo, stride_g, stride_v, stride_px, loc, weights,
                         out, out_layout, T, G, nullptr, 0, stream);
}

extern "C" int sbev_msmv_fwd_ring(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                                  int64_t Bp, int N, int C, int Q, int P,
                                  int gdiv, const int64_t* stride_slot, int64_t stride_g, const int64_t* stride_v,
                                  int64_t stride_px, const float* loc, const float* weights, float* out,
                                  int out_layout, int T, int G, const int32_t* frame_slots, int n_slots,
                                  sbev_stream_t stream) {
    SBEV_REQUIRE(frame_slots != nullptr, "sbev_msmv_fwd_ring: frame_slots is null");
    return msmv_fwd_impl(feats, hw, L, feat_dtype, Bp, N, C, Q, P, gdiv, stride_slot, stride_g, stride_v, stride_px, loc, weights,
                         out, out_layout, T, G, frame_slots, n_slots, stream);
}
