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

typedef float msmv_f2 __attribute__((ext_vector_type(2)));      // a channel pair: the operand type of the packed fmas in msmv_chunk.inc

// lane i of every 16-lane row <- lane n of that row (v_mov_b32_dpp row_newbcast:n, gfx90a+; n is a constant after unrolling: the
// switch folds).  All 64 lanes must be active.
// bound_ctrl = true: every lane of a row_newbcast reads a valid lane (all 64 active), so `old` is never taken -- saying so lets the
// compiler drop the `v_mov_b32 dst, 0` it otherwise emits in front of EVERY broadcast (32 per 4-level chunk, 11 % of the gather loop's
// VALU instructions; round 5).  Same values bit for bit.
template <int N>
__device__ __forceinline__ int msmv_row_bcast_c(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, true); }
__device__ __forceinline__ int msmv_row_bcast(int n, int v) {
    switch (n) {
        case 0: return msmv_row_bcast_c<0>(v);
        case 1: return msmv_row_bcast_c<1>(v);
        case 2: return msmv_row_bcast_c<2>(v);
        case 3: return msmv_row_bcast_c<3>(v);
        case 4: return msmv_row_bcast_c<4>(v);
        case 5: return msmv_row_bcast_c<5>(v);
        case 6: return msmv_row_bcast_c<6>(v);
        case 7: return msmv_row_bcast_c<7>(v);
        case 8: return msmv_row_bcast_c<8>(v);
        case 9: return msmv_row_bcast_c<9>(v);
        case 10: return msmv_row_bcast_c<10>(v);
        case 11: return msmv_row_bcast_c<11>(v);
        case 12: return msmv_row_bcast_c<12>(v);
        case 13: return msmv_row_bcast_c<13>(v);
        case 14: return msmv_row_bcast_c<14>(v);
        default: return msmv_row_bcast_c<15>(v);
    }
}

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
// fp16 storage (FT = _Float16, round 4): the same two registers per tap, widened with v_cvt_f32_f16 (exact) -- the features an fp16
// backbone emits (the reference casts them to fp32 first: @auto_fp16(out_fp32=True), models/sparsebev.py:46) sampled in place
struct MsmvH4 { uint2 v; };
typedef _Float16 msmv_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ MsmvH4 load_raw(const _Float16* p) { return MsmvH4{*reinterpret_cast<const uint2*>(p)}; }
__device__ __forceinline__ float4 widen(const MsmvH4 r) {
    const msmv_h2 a = __builtin_bit_cast(msmv_h2, r.v.x), b = __builtin_bit_cast(msmv_h2, r.v.y);
    return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}
template <typename FT>
struct MsmvIsF16 { static constexpr bool value = false; };
template <>
struct MsmvIsF16<_Float16> { static constexpr bool value = true; };

// ---- where a lane's taps come from ------------------------------------------------------------------------------------------
// BUF = true (round 4, every realistic map): raw BUFFER loads.  The (sample-batch, group) slab of a level is a wave-uniform buffer
// resource (base in SGPRs, num_records 2^31 - 1), a tap is a 32-bit BYTE offset in one VGPR, and a corner OUTSIDE its map carries bit
// 31 in that offset: the hardware's range check answers such a load with zeros and touches no memory -- exactly the reference's
// semantics (msmv_sampling_forward.cu:47-66 never reads an out-of-map corner, so an Inf / NaN stored in a border pixel does not
// reach a tap whose footprint only straddles it), at no VALU cost: round 3's lean form multiplied the CLAMPED pixel by a zero
// coefficient (NaN for a non-finite pixel), the form before it paid 80 v_cndmask per 5-level chunk.  Also per tap: one v_add_u32
// instead of a sign extension + v_lshl_add_u64, and one address register instead of two.
// BUF = false: 64-bit global loads + a select on the loaded channels, for slabs of 2 GiB and more (the in-slab element offset is
// still 32-bit; host-checked).  Same results.
constexpr unsigned MSMV_OUTSIDE = 0x80000000u;
constexpr unsigned MSMV_BUF_RECORDS = 0x7fffffffu;
constexpr int MSMV_RSRC_DW3 = 0x00020000;           // gfx9 raw buffer, 32-bit data format (composable_kernel's constant for gfx9)

// host: the largest in-slab byte offset a tap of this level can carry (last pixel of the last view + the lanes' channel offset + one
// 16-byte load) stays below the buffer's 2^31 - 1 records
inline bool msmv_slab_fits_buffer(long long N, long long H, long long W, long long stride_v, long long stride_px, long long C, long long esize) {
    const long long last = (N - 1) * stride_v + (H * W - 1) * stride_px + (C > 64 ? C : 64) + 4;
    return last * esize < (long long)MSMV_BUF_RECORDS;
}

template <int L, typename FT, bool BUF>
struct TapSrc;
template <int L, typename FT>
struct TapSrc<L, FT, true> {
    __amdgpu_buffer_rsrc_t rs[L];
    unsigned lane_off;                                // bytes: this lane's channel quad inside a pixel
    // slab_elems[l]: wave-uniform element offset of the (sample-batch, group) slab; lane_elems: this lane's channel offset
    __device__ __forceinline__ void init(const MsmvArgs& a, const long long* slab_elems, int lane_elems) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const unsigned long long p = (unsigned long long)(reinterpret_cast<const FT*>(a.feat[l]) + slab_elems[l]);
            // wave-uniform by construction (blockIdx, readfirstlane'd wave index): say so, or the resource costs a waterfall loop
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
            rs[l] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)MSMV_BUF_RECORDS, MSMV_RSRC_DW3);
        }
        lane_off = (unsigned)lane_elems * (unsigned)sizeof(FT);
    }
    // setup lane: element offset inside the slab + "inside the map" -> what the gathering lanes are handed
    __device__ __forceinline__ static int encode(int off_elems, bool inb) {
        const unsigned b = (unsigned)off_elems * (unsigned)sizeof(FT);          // < 2^31 (host-checked)
        return (int)(inb ? b : (b | MSMV_OUTSIDE));
    }
    __device__ __forceinline__ auto load(int l, int toff) const {
        const int vo = (int)((unsigned)toff + lane_off);                        // bit 31 survives: lane_off < 2^16
        if constexpr (sizeof(FT) == 4) return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs[l], vo, 0, 0));
        else if constexpr (MsmvIsF16<FT>::value) return MsmvH4{__builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs[l], vo, 0, 0))};
        else return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs[l], vo, 0, 0));
    }
};
template <int L, typename FT>
struct TapSrc<L, FT, false> {
    const FT* base[L];
    __device__ __forceinline__ void init(const MsmvArgs& a, const long long* slab_elems, int lane_elems) {
#pragma unroll
        for (int l = 0; l < L; ++l) base[l] = reinterpret_cast<const FT*>(a.feat[l]) + slab_elems[l] + lane_elems;
    }
    __device__ __forceinline__ static int encode(int off_elems, bool inb) { return inb ? off_elems : (int)((unsigned)off_elems | MSMV_OUTSIDE); }
    __device__ __forceinline__ auto load(int l, int toff) const { return load_raw(base[l] + (toff & 0x7fffffff)); }      // always a valid address
};

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

