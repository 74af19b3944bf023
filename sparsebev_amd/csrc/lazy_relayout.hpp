// On-demand relayout, device side shared by layout.hip (the stand-alone move launches) and gemm_bf16s.hip (the generator GEMM runs the
// layers-1..5 scan in its own prologue).  Included INSIDE each translation unit's anonymous namespace, like msmv_common.hpp.
// Design notes: layout.hip ("on-demand relayout").
#pragma once

constexpr int LZ_TS = 64;                    // pixels (and channels) of a unit
struct LazyArgs {
    const void* const* table;              // sources: table[index[l]] (replayable step) or src[l]
    int index[SBEV_MAX_LEVELS];
    const void* src[SBEV_MAX_LEVELS];      // [n_images, R, S_l]
    void* out[SBEV_MAX_LEVELS];            // [n_images, S_l, R]
    int S[SBEV_MAX_LEVELS];
    unsigned tiles[SBEV_MAX_LEVELS];       // ceil(S_l / 64)
    unsigned base[SBEV_MAX_LEVELS + 1];    // level l owns tiles [base[l], base[l + 1])
    int n_levels, R;                       // R = 4 groups x 64 channels
    unsigned* need;                        // [total tiles] 4 bytes each: group g of the tile is read by some sample point
    unsigned* done;                        // [total tiles] 4 bytes each: moved in this step
    int first, last;
};

constexpr int LAZY_SCAN = 16;        // tiles per workgroup of the stand-alone scan launch
__device__ __forceinline__ unsigned lazy_bytes_nonzero(unsigned w) {      // byte k != 0 -> bit k
    return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
}
__device__ __forceinline__ unsigned lazy_bits_to_bytes(unsigned m) {      // bit k -> byte k = 1
    return (m & 1u) | ((m & 2u) << 7) | ((m & 4u) << 14) | ((m & 8u) << 21);
}

__device__ __forceinline__ void lazy_locate(const LazyArgs& a, unsigned t, int& l, long long& img, int& ts) {
    l = 0;
#pragma unroll
    for (int j = 1; j < SBEV_MAX_LEVELS; ++j)
        if (j < a.n_levels && t >= a.base[j]) l = j;
    const unsigned rel = t - a.base[l];
    const unsigned i = rel / a.tiles[l];
    img = i;
    ts = (int)(rel - i * a.tiles[l]);
}

// ---- one unit moved by ONE wave (the scan launches): no workgroup barrier, so the 4 waves of a workgroup move 4 units side by side and a
// lane has all its loads of a pass in flight at once.  wt: the wave's own [64 pixels][33 words] of LDS.
constexpr int WLD = 33;
__device__ __forceinline__ void lazy_wave_sync() {          // LDS traffic of one wave executes in issue order; keep the compiler from reordering it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void lazy_move_unit_wave(const LazyArgs& a, int l, long long img, int ts, int g, float* wt, const float*) {
    const int lane = threadIdx.x & 63;
    const int R = a.R, S = a.S[l];
    const int s0 = ts * LZ_TS;
    const float* in = static_cast<const float*>(a.table ? a.table[a.index[l]] : a.src[l]) + img * R * S;
    float* out = static_cast<float*>(a.out[l]) + img * R * S;
    const int rg = g * LZ_TS;
    const bool vec = (S & 3) == 0;
    float4 v[2][8];                                         // both 32-channel halves requested up front: one exposed round trip per unit
    if (vec) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int lr = 32 * half + (lane >> 4) + 4 * i, sx = s0 + (lane & 15) * 4;
                v[half][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (sx < S) v[half][i] = *reinterpret_cast<const float4*>(in + (long long)(rg + lr) * S + sx);
            }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {                  // 32 channels x 64 pixels per pass through the wave's LDS tile
        const int r0 = rg + 32 * half;
        if (vec) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int lr = (lane >> 4) + 4 * i, ls = (lane & 15) * 4;
                wt[(ls + 0) * WLD + lr] = v[half][i].x;
                wt[(ls + 1) * WLD + lr] = v[half][i].y;
                wt[(ls + 2) * WLD + lr] = v[half][i].z;
                wt[(ls + 3) * WLD + lr] = v[half][i].w;
            }
        } else {
            for (int i = lane; i < 32 * LZ_TS; i += 64) {
                const int lr = i >> 6, ls = i & 63, sx = s0 + ls;
                wt[ls * WLD + lr] = sx < S ? in[(long long)(r0 + lr) * S + sx] : 0.f;
            }
        }
        lazy_wave_sync();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ls = (lane >> 3) + 8 * i, lr = (lane & 7) * 4;
            const int sx = s0 + ls;
            if (sx < S) {
                const float4 o = make_float4(wt[ls * WLD + lr], wt[ls * WLD + lr + 1], wt[ls * WLD + lr + 2], wt[ls * WLD + lr + 3]);
                *reinterpret_cast<float4*>(out + (long long)sx * R + r0 + lr) = o;
            }
        }
        lazy_wave_sync();
    }
}
__device__ __forceinline__ void lazy_move_unit_wave(const LazyArgs& a, int l, long long img, int ts, int g, float* wtf, const unsigned short*) {
    unsigned* wt = reinterpret_cast<unsigned*>(wtf);          // wt[pixel][channel pair]
    const int lane = threadIdx.x & 63;
    const int R = a.R, S = a.S[l];
    const int r0 = g * 64, s0 = ts * 64;
    const unsigned short* in = static_cast<const unsigned short*>(a.table ? a.table[a.index[l]] : a.src[l]) + img * R * S;
    unsigned short* out = static_cast<unsigned short*>(a.out[l]) + img * R * S;
    if ((S & 3) == 0) {
        uint2 lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cp = (lane >> 4) + 4 * i, sx = s0 + 4 * (lane & 15);
            lo[i] = make_uint2(0u, 0u); hi[i] = make_uint2(0u, 0u);
            if (sx < S) {
                lo[i] = *reinterpret_cast<const uint2*>(in + (long long)(r0 + 2 * cp) * S + sx);
                hi[i] = *reinterpret_cast<const uint2*>(in + (long long)(r0 + 2 * cp + 1) * S + sx);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cp = (lane >> 4) + 4 * i, q = lane & 15;
            wt[(4 * q + 0) * WLD + cp] = __builtin_amdgcn_perm(hi[i].x, lo[i].x, 0x05040100u);
            wt[(4 * q + 1) * WLD + cp] = __builtin_amdgcn_perm(hi[i].x, lo[i].x, 0x07060302u);
            wt[(4 * q + 2) * WLD + cp] = __builtin_amdgcn_perm(hi[i].y, lo[i].y, 0x05040100u);
            wt[(4 * q + 3) * WLD + cp] = __builtin_amdgcn_perm(hi[i].y, lo[i].y, 0x07060302u);
        }
    } else {
        for (int i = lane; i < 32 * 64; i += 64) {
            const int cp = i >> 6, ls = i & 63, sx = s0 + ls;
            unsigned w = 0u;
            if (sx < S) w = (unsigned)in[(long long)(r0 + 2 * cp) * S + sx] | ((unsigned)in[(long long)(r0 + 2 * cp + 1) * S + sx] << 16);
            wt[ls * WLD + cp] = w;
        }
    }
    lazy_wave_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int px = (lane >> 3) + 8 * i, k = lane & 7;
        const int sx = s0 + px;
        if (sx < S) {
            const unsigned* t = &wt[px * WLD + 4 * k];
            *reinterpret_cast<uint4*>(out + (long long)sx * R + r0 + 8 * k) = make_uint4(t[0], t[1], t[2], t[3]);
        }
    }
    lazy_wave_sync();
}


// A workgroup's share of a scan: workgroup `wg` of `nwg` looks at tiles wg, wg + nwg, wg + 2 nwg, ... (a run of consecutive new tiles
// lands on as many different workgroups), `per_batch` of them at a time (one per thread); what it finds is moved unit by unit, one unit per
// WAVE through that wave's [64][33]-word LDS tile (wtiles: one per wave), entries dealt round-robin over the waves.  Ends with a workgroup
// barrier; the caller drains its stores (s_waitcnt / kernel end).  Any number of waves.
template <typename ET>
__device__ __forceinline__ void lazy_scan_share(const LazyArgs& a, unsigned wg, unsigned nwg, int per_batch, float* wtiles, unsigned* list,
                                                unsigned* n_list) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6, nwave = (int)blockDim.x >> 6;
    const unsigned total = a.base[a.n_levels];
    const unsigned per_wg = (total + nwg - 1) / nwg;
    for (unsigned b0 = 0; b0 < per_wg; b0 += (unsigned)per_batch) {
        if (tid == 0) *n_list = 0u;
        __syncthreads();
        const unsigned slot = b0 + (unsigned)tid;
        const unsigned t = slot * nwg + wg;
        if (tid < per_batch && slot < per_wg && t < total) {
            const unsigned nw = a.need[t], dw = a.done[t];    // (both requests in flight together: the scan is a chain of round trips)
            if (nw != 0u) {
                const unsigned pend = lazy_bytes_nonzero(nw) & ~lazy_bytes_nonzero(dw);
                if (pend) {
                    a.done[t] = dw | lazy_bits_to_bytes(pend);
                    const unsigned at = atomicAdd(n_list, (unsigned)__builtin_popcount(pend));
                    unsigned k = 0;
                    for (int g = 0; g < 4; ++g)
                        if ((pend >> g) & 1u) list[at + k++] = (unsigned)tid | ((unsigned)g << 16);
                }
                if (a.last) a.need[t] = 0u;
            }
        }
        __syncthreads();
        const unsigned n = *n_list;
        for (unsigned i = (unsigned)wave; i < n; i += (unsigned)nwave) {
            const unsigned e = list[i];
            int l, ts;
            long long img;
            lazy_locate(a, (b0 + (e & 0xffffu)) * nwg + wg, l, img, ts);
            lazy_move_unit_wave(a, l, img, ts, (int)(e >> 16), wtiles + wave * (LZ_TS * WLD), static_cast<const ET*>(nullptr));
        }
        __syncthreads();
    }
}
