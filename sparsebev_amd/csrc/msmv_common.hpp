// Shared declarations of the multi-scale multi-view sampler (msmv_sampling.hip) and of the fused sampler + adaptive
// mixing kernel (mixing.hip): argument block, tap load / widen helpers, the LDS-free corner reduce-scatter.  Included
// INSIDE each translation unit's anonymous namespace.  Design notes: msmv_sampling.hip.
#pragma once

struct MsmvArgs {
    const void* feat[SBEV_MAX_LEVELS];
    int H[SBEV_MAX_LEVELS];
    int W[SBEV_MAX_LEVELS];
    long long stride_bo[SBEV_MAX_LEVELS];
    long long stride_v[SBEV_MAX_LEVELS];
    long long stride_g;
    long long stride_px;
    const float* loc;
    const float* w;
    float* out;
    long long n_waves;  // B' * Q
    int N, C, Q, P, gdiv, T, G;
    // online frame ring (sbev_msmv_fwd_ring): logical frame t of a sample lives in physical slot slots[t] of n_slots
    int ring_T, n_slots;
    int slots[SBEV_MAX_FRAMES];
};

// A tap is kept in its storage form until it is consumed: 4 bf16 channels stay two registers while the 4 * L loads of a
// chunk are in flight (converted tap by tap in phase 3), which is what decides the waves per SIMD of this latency-bound
// kernel (c5, L = 5: 71.5 -> 59 us; L = 4 bf16: 39.9 -> 34.5 us).
__device__ __forceinline__ float4 load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 load_raw(const unsigned short* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 widen(const float4 r) { return r; }
__device__ __forceinline__ float4 widen(const uint2 r) {  // 4 x bf16 -> fp32 (exact)
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}

// Reduce-scatter over the 4 corner groups (16-lane rows r = 0..3 of the wave) without LDS:
// given one value per item i = 0..3 in every lane, returns in row r the sum over all 4 rows of item r.
// permlane16_swap(x, y) exchanges the odd rows of x with the even rows of y, so x + y afterwards holds
// [i0(r0+r1), i1(r0+r1), i0(r2+r3), i1(r2+r3)]; permlane32_swap(x, y) exchanges the upper half of x with
// the lower half of y and finishes the sum.  3 swaps + 3 adds for 4 items (an all-reduce needs 8 + 8).
__device__ __forceinline__ float pair16(float a, float b) {
    auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
__device__ __forceinline__ float corner_reduce_scatter(float i0, float i1, float i2, float i3) {
    const float u = pair16(i0, i1), v = pair16(i2, i3);
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// QPW > 1 (only for P <= 4, C <= 64: one chunk per query): a wave walks QPW consecutive (b', q) items and requests the
// NEXT item's coordinates / level weights while it works on the current one, so each item costs one exposed memory round
// trip (its feature taps) instead of two (coordinates, then taps).  PMC on the QPW = 1 kernel: 45 % of the wave cycles sit
// in s_waitcnt, and bf16 features (half the bytes) ran no faster than fp32 -- latency-, not bandwidth-bound.
#ifndef SBEV_MSMV_QPW
#define SBEV_MSMV_QPW 2
#endif
// Waves per SIMD asked of the register allocator: with bf16 taps the 5-level kernel lands 2 registers above the 3-wave
// budget (170 vs 168), which the allocator closes when told to (4 spilled registers; 59.2 -> 58.0 us at c5).  Not for
// L = 4: 134 -> 128 registers for a 4th wave costs 8 spills and measured 40.0 vs 34.5 us.  fp32 is left alone.
template <int L, typename FT>
constexpr int msmv_min_waves() {
#ifdef SBEV_MSMV_NO_BF16_WAVES
    return 1;
#else
    return (sizeof(FT) == 2 && L >= 5) ? 3 : 1;
#endif
}

