// Feature relayout NCHW -> NHWC (gfx950) and the position encoder's first layer.
//
// sbev_nchw_to_nhwc: the FPN hands the decoder [B*T*N, G*C, H, W] maps (models/sparsebev.py:126-131); the
// sampler wants a pixel's channels contiguous.  The reference does this (and a per-group split) with
// permute + contiguous on every call (models/sparsebev_transformer.py:73-85).  Here it is one batched tiled
// transpose -- 64 channels x 64 pixels per workgroup through LDS, 16-byte global accesses on both sides --
// and it is skipped entirely when the neck already emits channels-last memory.  Pure HBM traffic: 2 x bytes.
#include "sbev_common.hpp"
#include "small_ops.hpp"

namespace {

#include "lazy_relayout.hpp"

using sbev_ops::PosArgs;
using sbev_ops::lin3_rows;

struct TrArgs {
    const float* in;   // [N, R, S]   (R = channels, S = H*W pixels)
    float* out;        // [N, S, R]
    int R, S;
    // indirect source (sbev_nchw_to_nhwc_f32_indirect): `in` is read from table[index] when the kernel starts -- a captured graph
    // then follows a caller that passes newly allocated feature tensors every step by refreshing one device word, not a node
    const void* const* table;
    int index;
};

constexpr int TS = 64, TLD = 65;

template <bool VEC>
__global__ __launch_bounds__(256) void transpose_tiles_kernel(const TrArgs a) {
    __shared__ float tile[TS * TLD];                    // tile[pixel][channel]
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * TS, r0 = blockIdx.y * TS;
    const long long img = blockIdx.z;
    const float* in = (a.table ? static_cast<const float*>(a.table[a.index]) : a.in) + img * a.R * a.S;
    float* out = a.out + img * a.R * a.S;
    if (VEC) {   // S % 4 == 0 and R % 4 == 0
        // read: thread -> (channel = tid/16 + 16*i, 4 pixels at (tid%16)*4)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + (tid >> 4) + 16 * i, s = s0 + (tid & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < a.R && s < a.S) v = *reinterpret_cast<const float4*>(in + (long long)r * a.S + s);
            const int lr = (tid >> 4) + 16 * i, ls = (tid & 15) * 4;
            tile[(ls + 0) * TLD + lr] = v.x;
            tile[(ls + 1) * TLD + lr] = v.y;
            tile[(ls + 2) * TLD + lr] = v.z;
            tile[(ls + 3) * TLD + lr] = v.w;
        }
        __syncthreads();
        // write: thread -> (pixel = tid/16 + 16*i, 4 channels at (tid%16)*4)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ls = (tid >> 4) + 16 * i, lr = (tid & 15) * 4;
            const int s = s0 + ls, r = r0 + lr;
            if (s < a.S && r < a.R) {
                const float4 v = make_float4(tile[ls * TLD + lr], tile[ls * TLD + lr + 1], tile[ls * TLD + lr + 2], tile[ls * TLD + lr + 3]);
                *reinterpret_cast<float4*>(out + (long long)s * a.R + r) = v;
            }
        }
    } else {
        for (int i = tid; i < TS * TS; i += 256) {
            const int lr = i / TS, ls = i % TS;
            const int r = r0 + lr, s = s0 + ls;
            tile[ls * TLD + lr] = (r < a.R && s < a.S) ? in[(long long)r * a.S + s] : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < TS * TS; i += 256) {
            const int ls = i / TS, lr = i % TS;
            const int r = r0 + lr, s = s0 + ls;
            if (r < a.R && s < a.S) out[(long long)s * a.R + r] = tile[ls * TLD + lr];
        }
    }
}

// All levels of a pyramid in ONE launch (round 5): the per-level launches of a step (4 at r50, 5 at r101 / eva02) spend a launch boundary each
// and the small levels cannot fill the chip (8 x 22 pixels x 48 images: one short round); here block b belongs to the level whose block range
// holds it -- levels in the caller's order, finest first, so that the coarse levels' few blocks run in the tail of the big one.  Same tile
// code (transpose_tiles_kernel<true>: every level S % 4 == 0, R % 4 == 0), sources from the pointer table.
struct TrMultiArgs {
    const void* const* table;
    int index[SBEV_MAX_LEVELS];
    float* out[SBEV_MAX_LEVELS];
    int S[SBEV_MAX_LEVELS];
    unsigned tiles_s[SBEV_MAX_LEVELS];           // ceil(S / TS)
    unsigned first_block[SBEV_MAX_LEVELS + 1];   // level l owns blocks [first_block[l], first_block[l + 1])
    int n_levels, R;
    unsigned tiles_r;                            // ceil(R / TS)
};
__global__ __launch_bounds__(256) void transpose_tiles_multi_kernel(const TrMultiArgs a) {
    __shared__ float tile[TS * TLD];
    const int tid = threadIdx.x;
    int l = 0;
#pragma unroll
    for (int j = 1; j < SBEV_MAX_LEVELS; ++j)
        if (j < a.n_levels && blockIdx.x >= a.first_block[j]) l = j;
    unsigned rel = blockIdx.x - a.first_block[l];
    const unsigned ts = a.tiles_s[l];
    const unsigned per_img = ts * a.tiles_r;
    const long long img = rel / per_img;
    rel -= (unsigned)img * per_img;
    const int r0 = (int)(rel / ts) * TS, s0 = (int)(rel % ts) * TS;
    const int R = a.R, S = a.S[l];
    const float* in = static_cast<const float*>(a.table[a.index[l]]) + img * R * S;
    float* out = a.out[l] + img * R * S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + (tid >> 4) + 16 * i, sx = s0 + (tid & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && sx < S) v = *reinterpret_cast<const float4*>(in + (long long)r * S + sx);
        const int lr = (tid >> 4) + 16 * i, ls = (tid & 15) * 4;
        tile[(ls + 0) * TLD + lr] = v.x;
        tile[(ls + 1) * TLD + lr] = v.y;
        tile[(ls + 2) * TLD + lr] = v.z;
        tile[(ls + 3) * TLD + lr] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ls = (tid >> 4) + 16 * i, lr = (tid & 15) * 4;
        const int sx = s0 + ls, r = r0 + lr;
        if (sx < S && r < R) {
            const float4 v = make_float4(tile[ls * TLD + lr], tile[ls * TLD + lr + 1], tile[ls * TLD + lr + 2], tile[ls * TLD + lr + 3]);
            *reinterpret_cast<float4*>(out + (long long)sx * R + r) = v;
        }
    }
}

// ---- the same relayout for 2-byte elements (bf16 / fp16 STORAGE: pure byte movement) ------------------------------------------------
// An fp16 backbone (the reference's eval mode, val.py:115) or a bf16 neck emits [N, R, S] maps of 2-byte channels.  A workgroup moves
// 128 channels x 64 pixels: a thread reads 4 consecutive pixels (8 bytes) of TWO neighbouring channels, interleaves them into four
// (channel pair) words -- in NHWC the pair is adjacent, so the transposition itself is a plain 32-bit one through LDS -- and writes 16
// bytes = 8 channels of one pixel: 128-byte runs on the read side, 256-byte runs on the write side.
struct Tr16Args {
    const unsigned short* in;   // [N, R, S]
    unsigned short* out;        // [N, S, R]
    int R, S;
    const void* const* table;   // indirect source (see TrArgs)
    int index;
};
constexpr int T16_CP = 64, T16_PX = 64, T16_LD = T16_CP + 1;      // channel PAIRS x pixels per tile

template <bool VEC>
__global__ __launch_bounds__(256) void transpose_tiles16_kernel(const Tr16Args a) {
    __shared__ unsigned tile[T16_PX * T16_LD];                      // tile[pixel][channel pair]
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * T16_PX, r0 = blockIdx.y * (2 * T16_CP);
    const long long img = blockIdx.z;
    const unsigned short* in = (a.table ? static_cast<const unsigned short*>(a.table[a.index]) : a.in) + img * a.R * a.S;
    unsigned short* out = a.out + img * a.R * a.S;
    if (VEC) {   // S % 4 == 0, R % 8 == 0, 8-byte aligned planes, 16-byte aligned output
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cp = (tid >> 4) + 16 * i, q = tid & 15;       // channel pair of the tile, pixel quad
            const int r = r0 + 2 * cp, s = s0 + 4 * q;
            uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
            if (r < a.R && s < a.S) {                               // (R is even: r + 1 < R too)
                lo = *reinterpret_cast<const uint2*>(in + (long long)r * a.S + s);
                hi = *reinterpret_cast<const uint2*>(in + (long long)(r + 1) * a.S + s);
            }
            // lo = channel r at pixels s .. s+3 (two per word), hi = channel r + 1: word (pixel) = lo16 | hi16 << 16
            tile[(4 * q + 0) * T16_LD + cp] = __builtin_amdgcn_perm(hi.x, lo.x, 0x05040100u);
            tile[(4 * q + 1) * T16_LD + cp] = __builtin_amdgcn_perm(hi.x, lo.x, 0x07060302u);
            tile[(4 * q + 2) * T16_LD + cp] = __builtin_amdgcn_perm(hi.y, lo.y, 0x05040100u);
            tile[(4 * q + 3) * T16_LD + cp] = __builtin_amdgcn_perm(hi.y, lo.y, 0x07060302u);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = (tid >> 4) + 16 * i, k = tid & 15;       // pixel of the tile, 16-byte piece (4 pairs = 8 channels)
            const int s = s0 + px, r = r0 + 8 * k;
            if (s < a.S && r < a.R) {
                const unsigned* t = &tile[px * T16_LD + 4 * k];
                *reinterpret_cast<uint4*>(out + (long long)s * a.R + r) = make_uint4(t[0], t[1], t[2], t[3]);
            }
        }
    } else {
        unsigned short* t16 = reinterpret_cast<unsigned short*>(tile);      // [pixel][2 * T16_LD] halves
        for (int i = tid; i < 2 * T16_CP * T16_PX; i += 256) {
            const int lr = i / T16_PX, ls = i % T16_PX;
            const int r = r0 + lr, s = s0 + ls;
            t16[ls * (2 * T16_LD) + lr] = (r < a.R && s < a.S) ? in[(long long)r * a.S + s] : (unsigned short)0;
        }
        __syncthreads();
        for (int i = tid; i < 2 * T16_CP * T16_PX; i += 256) {
            const int ls = i / (2 * T16_CP), lr = i % (2 * T16_CP);
            const int r = r0 + lr, s = s0 + ls;
            if (r < a.R && s < a.S) out[(long long)s * a.R + r] = t16[ls * (2 * T16_LD) + lr];
        }
    }
}

// ---- on-demand ("lazy") relayout: only the units a sample point will read (round 6) -----------------------------------------------------
// The reference regroups every pixel of every level on every call (models/sparsebev_transformer.py:73-85) and the dense kernels above
// move 2 x the feature bytes per step -- 14 % of the step at config 2, 29 % at config 3, 41 % at config 4 -- although the gather reads
// only the 4 bilinear corners of each sample point per level (msmv_sampling_forward.cu:41-66): under half of the pyramid
// (tools/relayout_footprint.py: 46 % of the units after layer 0, 49 % after all six layers at config 2).  Here the kernel that selects a
// point's camera (sample_point.hpp::touch_units, inside the attention row chain / sample_project_kernel) marks the units the point
// reads -- unit = 64 consecutive pixels of one image of one level x the 64 channels of ONE group, a byte each, word = the tile's 4 groups
// -- and these kernels move exactly the marked units that have not been moved in this step:
//   layer 0  : lazy_tiles_kernel<first> -- one workgroup per TILE (the dense grid: the pyramid is ~half marked), done = need;
//   layer >= 1: lazy_scan_kernel -- one thread per tile looks for need & ~done (a few hundred units at layer 1, a handful later), the
//               workgroup moves what its 256 tiles found; the last layer's launch clears `need` for the next step.
// Plain loads / stores only: `need` bytes are written by idempotent stores, `done` words by the one thread that owns the tile.  A stale or
// uninitialised `need` (first step on a new workspace, an aborted step) only moves MORE units; `done` is rebuilt by every step's first launch.
// Untouched units of the NHWC buffers keep whatever an earlier step left there: no tap ever reads them.
// one unit: channels [64 g, 64 g + 64) x pixels [64 ts, 64 ts + 64) of image `img` of level l.  fp32: the dense tile code.
__device__ __forceinline__ void lazy_move_unit(const LazyArgs& a, int l, long long img, int ts, int g, float* tile, const float*) {
    const int tid = threadIdx.x;
    const int R = a.R, S = a.S[l];
    const int r0 = g * TS, s0 = ts * TS;
    const float* in = static_cast<const float*>(a.table ? a.table[a.index[l]] : a.src[l]) + img * R * S;
    float* out = static_cast<float*>(a.out[l]) + img * R * S;
    if ((S & 3) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + (tid >> 4) + 16 * i, sx = s0 + (tid & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sx < S) v = *reinterpret_cast<const float4*>(in + (long long)r * S + sx);
            const int lr = (tid >> 4) + 16 * i, ls = (tid & 15) * 4;
            tile[(ls + 0) * TLD + lr] = v.x;
            tile[(ls + 1) * TLD + lr] = v.y;
            tile[(ls + 2) * TLD + lr] = v.z;
            tile[(ls + 3) * TLD + lr] = v.w;
        }
    } else {                                   // planes whose rows are not 16-byte aligned (e.g. a 10 x 25 level): scalar reads
        for (int i = tid; i < TS * TS; i += 256) {
            const int lr = i >> 6, ls = i & 63, sx = s0 + ls;
            tile[ls * TLD + lr] = sx < S ? in[(long long)(r0 + lr) * S + sx] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ls = (tid >> 4) + 16 * i, lr = (tid & 15) * 4;
        const int sx = s0 + ls;
        if (sx < S) {
            const float4 v = make_float4(tile[ls * TLD + lr], tile[ls * TLD + lr + 1], tile[ls * TLD + lr + 2], tile[ls * TLD + lr + 3]);
            *reinterpret_cast<float4*>(out + (long long)sx * R + r0 + lr) = v;
        }
    }
    __syncthreads();
}
// 2-byte channels (bf16 / fp16 storage: bytes are moved, never interpreted): 32 channel PAIRS x 64 pixels, the pair interleave of
// transpose_tiles16_kernel; 128-byte runs in, 128-byte runs out
__device__ __forceinline__ void lazy_move_unit(const LazyArgs& a, int l, long long img, int ts, int g, float* tilef, const unsigned short*) {
    unsigned* tile = reinterpret_cast<unsigned*>(tilef);          // tile[pixel][pair], row stride 33 words
    constexpr int PLD = 33;
    const int tid = threadIdx.x;
    const int R = a.R, S = a.S[l];
    const int r0 = g * 64, s0 = ts * 64;
    const unsigned short* in = static_cast<const unsigned short*>(a.table ? a.table[a.index[l]] : a.src[l]) + img * R * S;
    unsigned short* out = static_cast<unsigned short*>(a.out[l]) + img * R * S;
    if ((S & 3) == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cp = (tid >> 4) + 16 * i, q = tid & 15;     // channel pair of the unit, pixel quad
            const int r = r0 + 2 * cp, sx = s0 + 4 * q;
            uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
            if (sx < S) {
                lo = *reinterpret_cast<const uint2*>(in + (long long)r * S + sx);
                hi = *reinterpret_cast<const uint2*>(in + (long long)(r + 1) * S + sx);
            }
            tile[(4 * q + 0) * PLD + cp] = __builtin_amdgcn_perm(hi.x, lo.x, 0x05040100u);
            tile[(4 * q + 1) * PLD + cp] = __builtin_amdgcn_perm(hi.x, lo.x, 0x07060302u);
            tile[(4 * q + 2) * PLD + cp] = __builtin_amdgcn_perm(hi.y, lo.y, 0x05040100u);
            tile[(4 * q + 3) * PLD + cp] = __builtin_amdgcn_perm(hi.y, lo.y, 0x07060302u);
        }
    } else {
        for (int i = tid; i < 32 * 64; i += 256) {
            const int cp = i >> 6, ls = i & 63, sx = s0 + ls;
            unsigned w = 0u;
            if (sx < S) w = (unsigned)in[(long long)(r0 + 2 * cp) * S + sx] | ((unsigned)in[(long long)(r0 + 2 * cp + 1) * S + sx] << 16);
            tile[ls * PLD + cp] = w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = (tid >> 3) + 32 * i, k = tid & 7;          // pixel of the unit, 16-byte piece (4 pairs = 8 channels)
        const int sx = s0 + px;
        if (sx < S) {
            const unsigned* t = &tile[px * PLD + 4 * k];
            *reinterpret_cast<uint4*>(out + (long long)sx * R + r0 + 8 * k) = make_uint4(t[0], t[1], t[2], t[3]);
        }
    }
    __syncthreads();
}


// the step's FIRST lazy launch: one workgroup per tile, `done` is rebuilt from `need`
template <typename ET>
__global__ __launch_bounds__(256) void lazy_tiles_kernel(const LazyArgs a) {
    __shared__ float tile[TS * TLD];
    const unsigned t = blockIdx.x;
    const unsigned pend = (unsigned)__builtin_amdgcn_readfirstlane((int)lazy_bytes_nonzero(a.need[t]));
    if (threadIdx.x == 0) {
        a.done[t] = lazy_bits_to_bytes(pend);
        if (a.last) a.need[t] = 0u;
    }
    if (pend == 0u) return;
    int l, ts;
    long long img;
    lazy_locate(a, t, l, img, ts);
    for (int g = 0; g < 4; ++g)
        if ((pend >> g) & 1u) lazy_move_unit(a, l, img, ts, g, tile, static_cast<const ET*>(nullptr));
}

// later launches of the step: a thread per tile finds what the layer's sample points marked and no earlier launch moved.  The units a
// layer adds come in CLUSTERS of neighbouring tiles (boxes move a little: a band of new rows in one image, all 4 groups of a tile), and
// every move is a chain of two memory round trips (~3 us), so (1) workgroup w looks at tiles w, w + n, w + 2 n, ... (n workgroups,
// LAZY_SCAN tiles each): a run of consecutive new tiles lands on as many different workgroups; (2) a unit is moved by ONE wave
// (lazy_move_unit_wave: no workgroup barrier), wave k takes entries k, k + 4, ... of the workgroup's list -- the 4 groups of a new tile
// move side by side.  (First version: 64 consecutive tiles per workgroup, one unit at a time by the whole workgroup: 30 us per launch at
// config 2 for ~600 units -- a few workgroups with a dozen serial moves each.)
template <typename ET>
__global__ __launch_bounds__(256) void lazy_scan_kernel(const LazyArgs a) {
    __shared__ float wtile[4][TS * WLD];
    __shared__ unsigned list[LAZY_SCAN * 4];
    __shared__ unsigned n_list;
    lazy_scan_share<ET>(a, blockIdx.x, gridDim.x, LAZY_SCAN, &wtile[0][0], list, &n_list);
}

__global__ __launch_bounds__(256) void linear3_ln_relu_kernel(const PosArgs a) { lin3_rows(a, blockIdx.x); }

// contiguous copy with widening to fp32 (channels-last frames handed to the online ring: fp32 / fp16 / bf16 storage), 4 elements per thread
template <typename ST>
__device__ __forceinline__ float widen1(ST v);
template <>
__device__ __forceinline__ float widen1<float>(float v) { return v; }
template <>
__device__ __forceinline__ float widen1<_Float16>(_Float16 v) { return (float)v; }
template <>
__device__ __forceinline__ float widen1<unsigned short>(unsigned short v) { return __uint_as_float((unsigned)v << 16); }   // bf16

template <typename ST>
__global__ __launch_bounds__(256) void copy_widen_kernel(const ST* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 o;
        o.x = widen1<ST>(src[i]); o.y = widen1<ST>(src[i + 1]); o.z = widen1<ST>(src[i + 2]); o.w = widen1<ST>(src[i + 3]);
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
            *reinterpret_cast<float4*>(dst + i) = o;
        } else {
            dst[i] = o.x; dst[i + 1] = o.y; dst[i + 2] = o.z; dst[i + 3] = o.w;
        }
    } else {
        for (long long j = i; j < n; ++j) dst[j] = widen1<ST>(src[j]);
    }
}

// contiguous byte copies (up to 4 segments in ONE launch) whose source pointers are read from table[index[k]]; 16-byte aligned
// sources and destinations
struct CopySegs {
    const void* const* table;
    int nseg;
    int index[4];
    unsigned char* dst[4];
    long long nbytes[4];
    unsigned first_block[5];     // segment k owns blocks [first_block[k], first_block[k + 1])
};
__global__ __launch_bounds__(256) void copy_indirect_kernel(const CopySegs a) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (j < a.nseg && blockIdx.x >= a.first_block[j]) k = j;
    const unsigned char* __restrict__ src = static_cast<const unsigned char*>(a.table[a.index[k]]);
    unsigned char* __restrict__ dst = a.dst[k];
    const long long nbytes = a.nbytes[k];
    const long long i = ((long long)(blockIdx.x - a.first_block[k]) * 256 + threadIdx.x) * 16;
    if (i + 16 <= nbytes) {
        *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
    } else {
        for (long long j = i; j < nbytes; ++j) dst[j] = src[j];
    }
}


// ---- launch order of the gather items (sbev_query_order) -------------------------------------------------------------------------
// One workgroup per sample sorts its Q rows by the direction of the box centre around the ego origin (a bitonic network over
// (key << 32 | row) words in LDS: a total order, hence always a permutation, NaN centres included).  The key is the "diamond angle" of
// (x, y) -- monotone in atan2(y, x), no transcendental: queries that are neighbours in the order look into the same camera at nearby
// columns, which is what the gather's L2 footprint follows (tools/sampler_footprint.py).
constexpr int ORDER_MAX_Q = 4096, ORDER_THREADS = 1024;
struct OrderArgs {
    const float* bbox;     // [B, Q, ld]: columns 0, 1 = normalised centre
    int* order;            // [B * Q]: sample b's rows b*Q + q in its slots [b*Q, (b+1)*Q)
    int Q, ld, n2;         // n2 = power of two >= Q
    float ox, oy, sx, sy;  // metres = c * s + o  (decode_bbox, models/bbox/utils.py:69-70)
};
__global__ __launch_bounds__(ORDER_THREADS) void query_order_kernel(const OrderArgs a) {
    __shared__ unsigned long long keys[ORDER_MAX_Q];
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    for (int i = tid; i < a.n2; i += ORDER_THREADS) {
        unsigned long long k = ~0ull;                   // padding sorts last
        if (i < a.Q) {
            const float* r = a.bbox + (b * a.Q + i) * a.ld;
            const float x = r[0] * a.sx + a.ox, y = r[1] * a.sy + a.oy;
            const float n = fabsf(x) + fabsf(y);
            const float d = n > 0.f ? x / n : 1.f;       // cos-like, in [-1, 1]
            const float ang = y >= 0.f ? 1.f - d : 3.f + d;      // [0, 4): 0 = +x, 1 = +y, 2 = -x, 3 = -y
            k = ((unsigned long long)__float_as_uint(ang) << 32) | (unsigned)i;     // ang >= 0: its bit pattern orders like the value
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= a.n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < a.n2; i += ORDER_THREADS) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long u = keys[i], v = keys[p];
                    const bool up = (i & k) == 0;
                    if ((u > v) == up) { keys[i] = v; keys[p] = u; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < a.Q; i += ORDER_THREADS) a.order[b * a.Q + i] = (int)(b * a.Q) + (int)(unsigned)(keys[i] & 0xffffffffull);
}

}  // namespace

// The two staging launches of a replayable step (runtime.StepGraphs): their SOURCE pointer is table[index], read on the device when
// the kernel starts -- table is a device array the host refreshes before every graph launch, so a caller may hand in newly allocated
// tensors of the same shape each step (the reference's timing.py / val.py loops do) and still replay ONE captured graph.
// Sources must be 16-byte aligned (the caller checks the tensors it writes into the table).
extern "C" int sbev_nchw_to_nhwc_f32_indirect(const void* const* table, int index, float* out, int64_t n_images, int channels, int hw,
                                              sbev_stream_t stream) {
    SBEV_REQUIRE(n_images >= 0 && channels >= 1 && hw >= 1 && index >= 0, "sbev_nchw_to_nhwc_f32_indirect: bad sizes");
    if (n_images == 0) return SBEV_OK;
    SBEV_REQUIRE(table && out && (((uintptr_t)table) & 7) == 0, "sbev_nchw_to_nhwc_f32_indirect: null / unaligned pointer");
    SBEV_REQUIRE(n_images <= 65535, "sbev_nchw_to_nhwc_f32_indirect: at most 65535 images per call");
    TrArgs a{nullptr, out, channels, hw, table, index};
    dim3 grid((hw + TS - 1) / TS, (channels + TS - 1) / TS, (unsigned)n_images);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool vec = (hw % 4 == 0) && (channels % 4 == 0) && ((((uintptr_t)out) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(transpose_tiles_kernel<true>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(transpose_tiles_kernel<false>, grid, dim3(256), 0, s, a);
    return sbev::check_launch("sbev_nchw_to_nhwc_f32_indirect");
}

// every level of one pyramid in one launch (the levels share n_images and channels: [n_images, channels, hw[l]] each).  Returns
// SBEV_EINVAL for shapes the vector tile code does not take (hw[l] % 4, channels % 4, 16-byte alignment): the caller then launches
// sbev_nchw_to_nhwc_f32_indirect per level.
extern "C" int sbev_nchw_to_nhwc_f32_multi_indirect(const void* const* table, int n_levels, const int32_t* index, float* const* out,
                                                    int64_t n_images, int channels, const int32_t* hw, sbev_stream_t stream) {
    SBEV_REQUIRE(n_levels >= 1 && n_levels <= SBEV_MAX_LEVELS && n_images >= 0 && channels >= 4 && channels % 4 == 0, "sbev_nchw_to_nhwc_f32_multi_indirect: bad sizes");
    if (n_images == 0) return SBEV_OK;
    SBEV_REQUIRE(table && index && out && hw && (((uintptr_t)table) & 7) == 0, "sbev_nchw_to_nhwc_f32_multi_indirect: null / unaligned pointer");
    TrMultiArgs a{};
    a.table = table; a.n_levels = n_levels; a.R = channels; a.tiles_r = (unsigned)((channels + TS - 1) / TS);
    long long blocks = 0;
    for (int l = 0; l < n_levels; ++l) {
        SBEV_REQUIRE(index[l] >= 0 && out[l] && hw[l] >= 4 && hw[l] % 4 == 0 && (((uintptr_t)out[l]) & 15) == 0,
                     "sbev_nchw_to_nhwc_f32_multi_indirect: level %d (hw %% 4 == 0, 16-byte aligned destination)", l);
        a.index[l] = index[l]; a.out[l] = out[l]; a.S[l] = hw[l];
        a.tiles_s[l] = (unsigned)((hw[l] + TS - 1) / TS);
        a.first_block[l] = (unsigned)blocks;
        blocks += (long long)a.tiles_s[l] * a.tiles_r * n_images;
        SBEV_REQUIRE(blocks <= 0x7fffffffLL, "sbev_nchw_to_nhwc_f32_multi_indirect: too many tiles for one launch");
    }
    a.first_block[n_levels] = (unsigned)blocks;
    hipLaunchKernelGGL(transpose_tiles_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_nchw_to_nhwc_f32_multi_indirect");
}

extern "C" int sbev_copy_indirect(const void* const* table, int nseg, const int32_t* index, void* const* dst, const int64_t* nbytes,
                                  sbev_stream_t stream) {
    SBEV_REQUIRE(nseg >= 0 && nseg <= 4, "sbev_copy_indirect: 0 .. 4 segments");
    if (nseg == 0) return SBEV_OK;
    SBEV_REQUIRE(table && index && dst && nbytes && (((uintptr_t)table) & 7) == 0, "sbev_copy_indirect: null / unaligned pointer");
    CopySegs a{};
    a.table = table;
    a.nseg = nseg;
    long long blocks = 0;
    for (int k = 0; k < nseg; ++k) {
        SBEV_REQUIRE(index[k] >= 0 && nbytes[k] > 0 && dst[k] && (((uintptr_t)dst[k]) & 15) == 0, "sbev_copy_indirect: segment %d (index >= 0, nbytes > 0, 16-byte aligned dst)", k);
        a.index[k] = index[k];
        a.dst[k] = static_cast<unsigned char*>(dst[k]);
        a.nbytes[k] = nbytes[k];
        a.first_block[k] = (unsigned)blocks;
        blocks += (nbytes[k] + 4095) / 4096;
        SBEV_REQUIRE(blocks <= 0x7fffffffLL, "sbev_copy_indirect: too many bytes for one launch");
    }
    a.first_block[nseg] = (unsigned)blocks;
    hipLaunchKernelGGL(copy_indirect_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_copy_indirect");
}

// ---- the decoder step's last launch: stacked cls / bbox -> the caller's tensors, nan_to_num'ed ----------------------------------------
// torch.nan_to_num(x) of SparseBEVTransformer.forward (models/sparsebev_transformer.py:35-36): NaN -> 0, +Inf -> FLT_MAX, -Inf -> -FLT_MAX,
// everything else bit for bit.  Both outputs in ONE launch, out of place (it is also the copy out of a replayed graph's own buffers),
// destinations either direct or table[idx] read on the device (a captured step writes into tensors allocated per call).
namespace {
struct FinishArgs {
    const float* src[2];
    float* dst[2];
    const void* const* table;   // null: dst[] direct
    int index[2];
    long long n[2];
    unsigned first_block1;      // output 1 owns blocks [first_block1, ...)
};
__device__ __forceinline__ float nan_to_num1(float v) {
    const unsigned u = __float_as_uint(v), e = u & 0x7f800000u;
    if (e != 0x7f800000u) return v;
    if (u & 0x007fffffu) return 0.f;                                  // NaN
    return __uint_as_float((u & 0x80000000u) | 0x7f7fffffu);          // +-Inf -> +-FLT_MAX
}
__global__ __launch_bounds__(256) void finish_outputs_kernel(const FinishArgs a) {
    const int k = blockIdx.x >= a.first_block1 ? 1 : 0;
    const float* __restrict__ src = a.src[k];
    float* __restrict__ dst = a.table ? static_cast<float*>(const_cast<void*>(a.table[a.index[k]])) : a.dst[k];
    const long long i = ((long long)(blockIdx.x - (k ? a.first_block1 : 0u)) * 256 + threadIdx.x) * 4;
    if (i + 4 <= a.n[k] && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0)) {
        float4 v = *reinterpret_cast<const float4*>(src + i);
        v.x = nan_to_num1(v.x); v.y = nan_to_num1(v.y); v.z = nan_to_num1(v.z); v.w = nan_to_num1(v.w);
        *reinterpret_cast<float4*>(dst + i) = v;
    } else {
        for (long long j = i; j < a.n[k] && j < i + 4; ++j) dst[j] = nan_to_num1(src[j]);
    }
}
int finish_outputs(const void* const* table, int idx_cls, int idx_box, const float* cls_src, const float* box_src, float* cls_dst,
                   float* box_dst, int64_t n_cls, int64_t n_box, sbev_stream_t stream, const char* who) {
    SBEV_REQUIRE(n_cls >= 0 && n_box >= 0 && n_cls + n_box <= (1LL << 40), "%s: bad sizes", who);
    if (n_cls + n_box == 0) return SBEV_OK;
    SBEV_REQUIRE((n_cls == 0 || cls_src) && (n_box == 0 || box_src), "%s: null source", who);
    FinishArgs a{};
    a.src[0] = cls_src; a.src[1] = box_src; a.dst[0] = cls_dst; a.dst[1] = box_dst;
    a.table = table; a.index[0] = idx_cls; a.index[1] = idx_box; a.n[0] = n_cls; a.n[1] = n_box;
    const long long b0 = (n_cls + 1023) / 1024, b1 = (n_box + 1023) / 1024;
    SBEV_REQUIRE(b0 + b1 <= 0x7fffffffLL, "%s: too many elements for one launch", who);
    a.first_block1 = (unsigned)b0;
    hipLaunchKernelGGL(finish_outputs_kernel, dim3((unsigned)(b0 + b1)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch(who);
}
}  // namespace

extern "C" int sbev_finish_outputs(const float* cls_src, const float* box_src, float* cls_dst, float* box_dst, int64_t n_cls, int64_t n_box,
                                   sbev_stream_t stream) {
    SBEV_REQUIRE((n_cls == 0 || cls_dst) && (n_box == 0 || box_dst), "sbev_finish_outputs: null destination");
    return finish_outputs(nullptr, 0, 0, cls_src, box_src, cls_dst, box_dst, n_cls, n_box, stream, "sbev_finish_outputs");
}

extern "C" int sbev_finish_outputs_indirect(const void* const* table, int idx_cls, int idx_box, const float* cls_src, const float* box_src,
                                            int64_t n_cls, int64_t n_box, sbev_stream_t stream) {
    SBEV_REQUIRE(table && (((uintptr_t)table) & 7) == 0 && idx_cls >= 0 && idx_box >= 0, "sbev_finish_outputs_indirect: null / unaligned table or negative index");
    return finish_outputs(table, idx_cls, idx_box, cls_src, box_src, nullptr, nullptr, n_cls, n_box, stream, "sbev_finish_outputs_indirect");
}

extern "C" int sbev_nchw_to_nhwc_f32(const float* in, float* out, int64_t n_images, int channels, int hw,
                                     sbev_stream_t stream) {
    SBEV_REQUIRE(n_images >= 0 && channels >= 1 && hw >= 1, "sbev_nchw_to_nhwc_f32: bad sizes");
    if (n_images == 0) return SBEV_OK;
    SBEV_REQUIRE(in && out && in != out, "sbev_nchw_to_nhwc_f32: null or aliased pointers");
    SBEV_REQUIRE(n_images <= 65535, "sbev_nchw_to_nhwc_f32: at most 65535 images per call");
    TrArgs a{in, out, channels, hw, nullptr, 0};
    dim3 grid((hw + TS - 1) / TS, (channels + TS - 1) / TS, (unsigned)n_images);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool vec = (hw % 4 == 0) && (channels % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(transpose_tiles_kernel<true>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(transpose_tiles_kernel<false>, grid, dim3(256), 0, s, a);
    return sbev::check_launch("sbev_nchw_to_nhwc_f32");
}

extern "C" int sbev_linear3_ln_relu_f32(const float* x, int64_t ldx, const float* w, const float* b,
                                        const float* ln_w, const float* ln_b, float eps, float* y,
                                        int64_t M, int N, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024 && ldx >= 3, "sbev_linear3_ln_relu_f32: need N %% 4 == 0, N <= 1024, ldx >= 3");
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(x && w && b && ln_w && ln_b && y, "sbev_linear3_ln_relu_f32: null pointer");
    PosArgs a{x, w, b, ln_w, ln_b, y, M, N, (int)ldx, eps};
    hipLaunchKernelGGL(linear3_ln_relu_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_linear3_ln_relu_f32");
}

// Same launch, additionally storing the Linear's pre-LayerNorm output (training forward: the backward pass needs it).
extern "C" int sbev_linear3_ln_relu_ex_f32(const float* x, int64_t ldx, const float* w, const float* b,
                                           const float* ln_w, const float* ln_b, float eps, float* y, float* pre,
                                           int64_t M, int N, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024 && ldx >= 3, "sbev_linear3_ln_relu_ex_f32: need N %% 4 == 0, N <= 1024, ldx >= 3");
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(x && w && b && ln_w && ln_b && y, "sbev_linear3_ln_relu_ex_f32: null pointer");
    PosArgs a{x, w, b, ln_w, ln_b, y, M, N, (int)ldx, eps, pre};
    hipLaunchKernelGGL(linear3_ln_relu_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_linear3_ln_relu_ex_f32");
}

// dst[i] = (float)src[i] for a contiguous run: src_dtype 0 = fp32, 1 = bf16 (enum sbev_dtype), 2 = fp16.  The online frame
// ring's path for channels-last frames (cache.FrameFeatureCache.push): replaces a torch copy_ kernel on the product path.
extern "C" int sbev_copy_widen_f32(const void* src, int src_dtype, float* dst, int64_t n, sbev_stream_t stream) {
    SBEV_REQUIRE(n >= 0 && src_dtype >= 0 && src_dtype <= 2, "sbev_copy_widen_f32: bad arguments");
    if (n == 0) return SBEV_OK;
    SBEV_REQUIRE(src && dst, "sbev_copy_widen_f32: null pointer");
    const long long blocks = (n / 4 + 255) / 256 + 1;
    SBEV_REQUIRE(blocks <= 0x7fffffffLL, "sbev_copy_widen_f32: too many elements for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (src_dtype == 0) hipLaunchKernelGGL(copy_widen_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float*>(src), dst, (long long)n);
    else if (src_dtype == 1) hipLaunchKernelGGL(copy_widen_kernel<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const unsigned short*>(src), dst, (long long)n);
    else hipLaunchKernelGGL(copy_widen_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const _Float16*>(src), dst, (long long)n);
    return sbev::check_launch("sbev_copy_widen_f32");
}

// order[b*Q + i] = the row (b*Q + q) that comes i-th when sample b's queries are sorted by the direction of their box centre around the
// ego origin: the launch order sbev_sample_mix_*_ordered walks (one contiguous arc of the camera ring per XCD).  Purely a placement
// hint -- any permutation gives bit-identical results.  Q <= sbev_query_order_max() (one workgroup sorts a sample in LDS).
extern "C" int sbev_query_order_max(void) { return ORDER_MAX_Q; }

extern "C" int sbev_query_order(const float* query_bbox, int64_t ld, const double* pc_range, int B, int Q, int32_t* order, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && ld >= 2, "sbev_query_order: bad sizes");
    SBEV_REQUIRE(Q <= ORDER_MAX_Q, "sbev_query_order: at most %d queries per sample (got %d)", ORDER_MAX_Q, Q);
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(query_bbox && pc_range && order, "sbev_query_order: null pointer");
    SBEV_REQUIRE((long long)B * Q <= 0x7fffffffLL, "sbev_query_order: too many rows");
    OrderArgs a{};
    a.bbox = query_bbox; a.order = order; a.Q = Q; a.ld = (int)ld;
    a.n2 = 1;
    while (a.n2 < Q) a.n2 <<= 1;
    a.ox = (float)pc_range[0]; a.oy = (float)pc_range[1];
    a.sx = (float)(pc_range[3] - pc_range[0]); a.sy = (float)(pc_range[4] - pc_range[1]);
    hipLaunchKernelGGL(query_order_kernel, dim3((unsigned)B), dim3(ORDER_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_query_order");
}

// NCHW -> NHWC for 2-byte channels (bf16 or fp16 storage: bytes are moved, never interpreted).  `in` NULL: the source address is
// table[index], read on the device (the staged form of a replayable step, like sbev_nchw_to_nhwc_f32_indirect).
static int nchw_to_nhwc_b16(const void* in, const void* const* table, int index, void* out, int64_t n_images, int channels, int hw,
                            sbev_stream_t stream, const char* who) {
    SBEV_REQUIRE(n_images >= 0 && channels >= 1 && hw >= 1 && index >= 0, "%s: bad sizes", who);
    if (n_images == 0) return SBEV_OK;
    SBEV_REQUIRE((in || table) && out && in != out, "%s: null or aliased pointers", who);
    SBEV_REQUIRE(!table || (((uintptr_t)table) & 7) == 0, "%s: unaligned pointer table", who);
    SBEV_REQUIRE(n_images <= 65535, "%s: at most 65535 images per call", who);
    Tr16Args a{static_cast<const unsigned short*>(in), static_cast<unsigned short*>(out), channels, hw, in ? nullptr : table, index};
    dim3 grid((hw + T16_PX - 1) / T16_PX, (channels + 2 * T16_CP - 1) / (2 * T16_CP), (unsigned)n_images);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (an indirect source is required to be 16-byte aligned by its caller, like the fp32 form)
    const bool vec = (hw % 4 == 0) && (channels % 8 == 0) && ((((uintptr_t)out) & 15) == 0) && (!in || (((uintptr_t)in) & 7) == 0);
    if (vec)
        hipLaunchKernelGGL(transpose_tiles16_kernel<true>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(transpose_tiles16_kernel<false>, grid, dim3(256), 0, s, a);
    return sbev::check_launch(who);
}

extern "C" int sbev_nchw_to_nhwc_b16(const void* in, void* out, int64_t n_images, int channels, int hw, sbev_stream_t stream) {
    SBEV_REQUIRE(n_images == 0 || in, "sbev_nchw_to_nhwc_b16: null input");
    return nchw_to_nhwc_b16(in, nullptr, 0, out, n_images, channels, hw, stream, "sbev_nchw_to_nhwc_b16");
}

extern "C" int sbev_nchw_to_nhwc_b16_indirect(const void* const* table, int index, void* out, int64_t n_images, int channels, int hw,
                                              sbev_stream_t stream) {
    SBEV_REQUIRE(n_images == 0 || table, "sbev_nchw_to_nhwc_b16_indirect: null table");
    return nchw_to_nhwc_b16(nullptr, table, index, out, n_images, channels, hw, stream, "sbev_nchw_to_nhwc_b16_indirect");
}

// ---- on-demand relayout: host side ------------------------------------------------------------------------------------------------
namespace sbev {
// tiles of a pyramid [n_images, 256, hw[l]] per level; false when the lazy kernels do not take the shape
bool lazy_plan(int n_levels, const int32_t* hw, long long n_images, int channels, LazyPlan* p) {
    if (n_levels < 1 || n_levels > SBEV_MAX_LEVELS || channels != 256 || n_images < 1) return false;
    long long tiles = 0;
    p->n_levels = n_levels; p->n_images = n_images; p->R = channels;
    for (int l = 0; l < n_levels; ++l) {
        if (hw[l] < 1) return false;
        p->S[l] = hw[l];
        p->tiles[l] = (unsigned)((hw[l] + TS - 1) / TS);
        p->base[l] = (unsigned)tiles;
        tiles += (long long)p->tiles[l] * n_images;
        if (tiles > 0x3fffffffLL) return false;
    }
    p->base[n_levels] = (unsigned)tiles;
    return true;
}

int launch_lazy_relayout(const LazyPlan& p, const void* const* table, const int32_t* index, const void* const* src, void* const* out,
                         int esize, uint32_t* need, uint32_t* done, bool first, bool last, hipStream_t s) {
    LazyArgs a{};
    a.table = table; a.n_levels = p.n_levels; a.R = p.R; a.need = need; a.done = done; a.first = first; a.last = last;
    for (int l = 0; l < p.n_levels; ++l) {
        a.index[l] = table ? index[l] : 0;
        a.src[l] = table ? nullptr : src[l];
        a.out[l] = out[l];
        a.S[l] = p.S[l]; a.tiles[l] = p.tiles[l]; a.base[l] = p.base[l];
    }
    a.base[p.n_levels] = p.base[p.n_levels];
    const unsigned total = p.base[p.n_levels];
    if (first) {
        if (esize == 4) hipLaunchKernelGGL(lazy_tiles_kernel<float>, dim3(total), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(lazy_tiles_kernel<unsigned short>, dim3(total), dim3(256), 0, s, a);
    } else {
        if (esize == 4) hipLaunchKernelGGL(lazy_scan_kernel<float>, dim3((total + LAZY_SCAN - 1) / LAZY_SCAN), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(lazy_scan_kernel<unsigned short>, dim3((total + LAZY_SCAN - 1) / LAZY_SCAN), dim3(256), 0, s, a);
    }
    return check_launch("lazy relayout");
}
}  // namespace sbev

extern "C" int64_t sbev_lazy_relayout_tiles(int n_levels, const int32_t* hw, int64_t n_images, int channels) {
    sbev::LazyPlan p;
    if (!hw || !sbev::lazy_plan(n_levels, hw, n_images, channels, &p)) return -1;
    return (int64_t)p.base[n_levels];
}

extern "C" int sbev_nchw_to_nhwc_lazy(const void* const* table, const int32_t* index, const void* const* src, void* const* out, int n_levels,
                                      const int32_t* hw, int64_t n_images, int channels, int dtype, uint32_t* need, uint32_t* done, int first,
                                      int last, sbev_stream_t stream) {
    SBEV_REQUIRE(hw && out && need && done && (table ? index != nullptr : src != nullptr), "sbev_nchw_to_nhwc_lazy: null pointer");
    SBEV_REQUIRE(dtype == SBEV_F32 || dtype == SBEV_BF16 || dtype == SBEV_F16, "sbev_nchw_to_nhwc_lazy: dtype %d", dtype);
    SBEV_REQUIRE(!table || (((uintptr_t)table) & 7) == 0, "sbev_nchw_to_nhwc_lazy: unaligned pointer table");
    sbev::LazyPlan p;
    SBEV_REQUIRE(sbev::lazy_plan(n_levels, hw, n_images, channels, &p),
                 "sbev_nchw_to_nhwc_lazy: needs 1..%d levels of 256 channels (got %d levels, %d channels)", SBEV_MAX_LEVELS, n_levels, channels);
    for (int l = 0; l < n_levels; ++l)
        SBEV_REQUIRE(out[l] && (((uintptr_t)out[l]) & 15) == 0 && (table ? index[l] >= 0 : (src[l] && (((uintptr_t)src[l]) & 15) == 0)),
                     "sbev_nchw_to_nhwc_lazy: level %d (16-byte aligned source and destination)", l);
    return sbev::launch_lazy_relayout(p, table, index, src, out, dtype == SBEV_F32 ? 4 : 2, need, done, first != 0, last != 0,
                                      reinterpret_cast<hipStream_t>(stream));
}
