// fp32 "Linear" GEMM on the gfx950 matrix cores:  Y[M,N] = act(X[M,K] . W[N,K]^T + bias) (+ residual)
//
// Used for every nn.Linear of the decoder layer (models/sparsebev_transformer.py:116-153,203,262-263,
// 343-344 and the mmcv MultiheadAttention / FFN projections); the two that matter are the adaptive-mixing
// parameter generator [B*Q,256]x[256,32768] and its out-projection [B*Q,32768]x[32768,256] (15.1 GFLOP
// each per layer-sample at config 2).
//
// Exact-fp32 path: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bitwise an fmaf chain, so parity with the
// reference's fp32 math holds to rounding-order noise).  Peak 157 TF (=the f32 vector peak), reached with
// one wave per SIMD; both operands are K-contiguous ("NT" GEMM), which is exactly the nn.Linear layout.
//
// Tile: 128x128x32 per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA tiles of 32x32 (64 accumulator
// VGPRs).  Global -> registers (16-B loads, issued one K-step ahead) -> LDS (double buffered, ONE barrier per
// K-step) -> ds_read_b64 fragments (two k values per read feed two consecutive MFMAs).  LDS rows are padded
// to 36 floats: keeps ds_write_b128 16-B aligned and leaves a 2-way conflict on the b64 reads (32 reads per
// 64 MFMAs of 64 cycles each: invisible).  Split-K (grid.z) writes fp32 partial slabs for the K=32768
// out-projection; sbev_splitk_reduce_* combines them in the next kernel's prologue (a launch boundary is
// cheaper than an in-launch cross-XCD hand-off at this size, cdna_hip_programming.md section 5).
#include <cstdlib>
#include "sbev_common.hpp"
#include "small_ops.hpp"

namespace {

using sbev_ops::ReduceArgs;
using sbev_ops::reduce_rows;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, LDT = 36;  // LDT: padded LDS row stride (floats)
// tile = (64*WM) x (64*WN): 2x2 waves, each WM x WN MFMA tiles of 32x32.  <2,2> = 128x128 for the big GEMMs,
// <1,1> = 64x64 for the many small [B*Q,256]x[256,256..768] linears, which otherwise launch < 32 workgroups.

struct GemmArgs {
    const float* X;      // [M, ldx]
    const float* W;      // [N, ldw]
    const float* bias;   // [N] or null
    const float* res;    // [M, ldy] or null
    float* Y;            // [M, ldy]   (split-K: [splits, M, N] slabs, ldy = N)
    long long M;
    int N, K;            // K = full reduction length
    long long ldx, ldw, ldy;
    int k_per_split;     // multiple of BK
    int relu;
    // LayerNorm prologue (small-tile kernel only, K == 256 == the LayerNorm width): X rows are normalised on the way in
    const float* ln_g = nullptr;    // [K] or null -> plain linear
    const float* ln_b = nullptr;    // [K]
    const float* ln_add = nullptr;  // [M, ldx] or null: added after LayerNorm (+ ReLU)
    float* xn_out = nullptr;        // [M, ldx] or null: the normalised rows, written by the workgroups of column tile 0
    float ln_eps = 0.f;
    int ln_relu = 0;
};

// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs (MI355X_MICROARCH.md), so give
// each XCD a contiguous run of tiles -> neighbouring tiles (which share a W panel) hit the same L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned id, unsigned n) {
    const unsigned q = n / 8, r = n % 8, x = id % 8, s = id / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

template <bool SPLIT, int WM, int WN, bool RAGGED>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_kernel(const GemmArgs a) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDT];
    float* As = lds;                   // [2][BM][LDT]
    float* Bs = lds + 2 * BM * LDT;    // [2][BN][LDT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const unsigned tiles_n = (a.N + BN - 1) / BN;
    const unsigned tiles_m = (unsigned)((a.M + BM - 1) / BM);
    // tile order inside an XCD's contiguous chunk: walk the SHORTER tile dimension fastest, so the panel of the
    // longer operand (W for the parameter generator: 33.5 MB) is fetched once per XCD and re-used from its L2 by
    // the few tiles of the other dimension, instead of every XCD streaming the whole of it.
    const unsigned t = xcd_swizzle(blockIdx.x, tiles_m * tiles_n);
    const bool m_fast = tiles_n >= tiles_m;
    const unsigned tm = m_fast ? t % tiles_m : t / tiles_n;
    const unsigned tn = m_fast ? t / tiles_m : t % tiles_n;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;
    const int kbeg = SPLIT ? blockIdx.z * a.k_per_split : 0;
    const int kend = SPLIT ? min(a.K, kbeg + a.k_per_split) : a.K;
    const int nk = (kend - kbeg + BK - 1) / BK;

    // staging map: thread -> (row = tid/8 + 32*i, 4 floats at col = (tid%8)*4)
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    constexpr int PA = BM / 32, PB = BN / 32;
    const float* xp[PA];
    const float* wp[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        long long r = m0 + srow + 32 * i;
        r = r < a.M ? r : a.M - 1;                      // clamp: rows past M are computed and never stored
        xp[i] = a.X + r * a.ldx + scol;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        int c = n0 + srow + 32 * i;
        c = c < a.N ? c : a.N - 1;
        wp[i] = a.W + (long long)c * a.ldw + scol;
    }
    float4 ra[PA], rb[PB];
    auto gload = [&](int kt) {
        const int k = kbeg + kt * BK;
        auto ld = [&](const float* p) {
            // RAGGED is a separate instantiation on purpose: a run-time "vector or element-wise" choice makes
            // hipcc branch around every load and wait vmcnt(0) each time (cdna_hip_programming.md, .s trap (c)).
            if (!RAGGED) return *reinterpret_cast<const float4*>(p + k);
            float v[4];                                  // K % 32 != 0: zero-fill element-wise
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (k + scol + e) < kend ? p[k + e] : 0.f;
            return make_float4(v[0], v[1], v[2], v[3]);
        };
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = ld(xp[i]);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = ld(wp[i]);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<float4*>(&As[(buf * BM + srow + 32 * i) * LDT + scol]) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<float4*>(&Bs[(buf * BN + srow + 32 * i) * LDT + scol]) = rb[i];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;   // fragment row / k-half
    if (nk > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);                 // in flight under this step's 64 MFMAs
        const float* Ab = As + (buf * BM + wr * 32 * WM + fr) * LDT + 2 * fh;
        const float* Bb = Bs + (buf * BN + wc * 32 * WN + fr) * LDT + 2 * fh;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float2 fa[WM], fb[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) fa[i] = *reinterpret_cast<const float2*>(Ab + i * 32 * LDT + kk * 4);
#pragma unroll
            for (int j = 0; j < WN; ++j) fb[j] = *reinterpret_cast<const float2*>(Bb + j * 32 * LDT + kk * 4);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);               // the other buffer: nobody reads it during this step
        __syncthreads();
    }

    // epilogue: accumulators -> LDS (the staging buffers are free after the last barrier) -> row-major float4
    // pass, so bias / residual reads and the Y stores are 512-B contiguous per row instead of 128-B column
    // fragments.  C/D layout of the 32x32 MFMA: column j = lane & 31, row i = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
    constexpr int LDC = BN + 4;
    static_assert(BM * LDC <= 2 * (BM + BN) * LDT, "C tile must fit in the staging LDS");
    float* Cs = lds;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wr * 32 * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                Cs[r * LDC + wc * 32 * WN + j * 32 + fr] = acc[i][j][e];
            }
    __syncthreads();
    float* Y = SPLIT ? a.Y + (long long)blockIdx.z * a.M * a.ldy : a.Y;
    constexpr int TPR = BN / 4;                 // threads per row (float4 each)
    constexpr int RPP = 256 / TPR;              // rows per pass
    const int er = tid / TPR, ec = (tid % TPR) * 4;
    const int n = n0 + ec;
    const bool vec = (a.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) && (n + 3 < a.N) &&
                     (SPLIT || !a.res || (reinterpret_cast<uintptr_t>(a.res) & 15) == 0);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (!SPLIT && a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (n + e) < a.N ? a.bias[n + e] : 0.f;
    }
#pragma unroll 4
    for (int it = 0; it < BM / RPP; ++it) {
        const int r = er + it * RPP;
        const long long m = m0 + r;
        if (m >= a.M) break;
        const float4 c4 = *reinterpret_cast<const float4*>(&Cs[r * LDC + ec]);
        float v[4] = {c4.x, c4.y, c4.z, c4.w};
        if (!SPLIT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += bv[e];
                if (a.relu) v[e] = fmaxf(v[e], 0.f);
            }
        }
        float* yp = Y + m * a.ldy + n;
        if (vec) {
            if (!SPLIT && a.res) {
                const float4 r4 = *reinterpret_cast<const float4*>(a.res + m * a.ldy + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < a.N) {
                    if (!SPLIT && a.res) v[e] += a.res[m * a.ldy + n + e];
                    yp[e] = v[e];
                }
        }
    }
}

// ---- latency-oriented variant for the many small linears ([B*Q,256] x [256,256..776], 0.1-0.35 GFLOP) -------
// With 128x128 or 64x64 tiles these launch only 16-60 workgroups whose waves each grind through the whole K:
// ~10 us for 118 MFLOP.  Here a workgroup owns one 32x32 output tile and its 4 waves SPLIT K four ways; every
// wave pulls its A / B fragments straight from global memory (L2-resident operands, each element used once per
// wave, so LDS staging would only add a barrier), runs K/4/2 MFMAs, and the four partial tiles are combined
// through 16 KiB of LDS into a float4 row-major epilogue.  232 workgroups / 928 waves for [900,256]x[256,256].
constexpr int SMALL_LDS_FLOATS = 4 * 16 * 64 + 256 + 512;      // split-K exchange area + the LayerNorm prologue's row statistics + gamma | beta
template <int KC>   // KC = K / 4 / 8: float4 k-blocks per wave (8 for K = 256, 16 for K = 512)
__device__ __forceinline__ void small_tile(const GemmArgs& a, unsigned tile, float* red) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned tiles_n = (a.N + 31) / 32;
    const unsigned tm = tile / tiles_n, tn = tile % tiles_n;
    const long long m0 = (long long)tm * 32;
    const int n0 = tn * 32;
    long long ra = m0 + fr;
    ra = ra < a.M ? ra : a.M - 1;
    int rb = n0 + fr;
    rb = rb < a.N ? rb : a.N - 1;
    const float* xp = a.X + ra * a.ldx + wave * (KC * 8) + 4 * fh;
    const float* wp = a.W + (long long)rb * a.ldw + wave * (KC * 8) + 4 * fh;
    float4 fa[KC], fb[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) {       // all loads in flight before the first MFMA
        fa[i] = *reinterpret_cast<const float4*>(xp + i * 8);
        fb[i] = *reinterpret_cast<const float4*>(wp + i * 8);
    }
    if constexpr (KC == 8) {
        // LayerNorm prologue: the 4 waves of the workgroup hold its 32 rows x 256 k between them (wave = a 64-k slice,
        // lane (fr, fh) = row fr, 32 of those k), so the producer's stand-alone LayerNorm launch (~6 us for 0.9 MB) becomes
        // two 4-way LDS exchanges of row statistics here.  Two-pass variance, as the stand-alone kernel.
        if (a.ln_g) {
            float* stat = red + 4 * 16 * 64;                  // 2 x [4 waves][32 rows], behind the split-K exchange area
            float* gb = stat + 256;                           // gamma | beta staged once per workgroup: holding them in
            gb[tid] = a.ln_g[tid];                            // registers from the top (64 per lane) halves the resident
            gb[256 + tid] = a.ln_b[tid];                      // workgroups per CU and the 725-tile in-projection needs 2 rounds
            const int kb = wave * 64 + 4 * fh;
            float4 ad[KC];
            if (a.ln_add) {                                   // one uniform branch around all 8 requests
#pragma unroll
                for (int i = 0; i < KC; ++i) ad[i] = *reinterpret_cast<const float4*>(a.ln_add + ra * a.ldx + kb + i * 8);
            } else {
#pragma unroll
                for (int i = 0; i < KC; ++i) ad[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < KC; ++i) sm += (fa[i].x + fa[i].y) + (fa[i].z + fa[i].w);
            sm += __shfl_xor(sm, 32);
            if (fh == 0) stat[wave * 32 + fr] = sm;
            __syncthreads();
            const float mean = ((stat[fr] + stat[32 + fr]) + (stat[64 + fr] + stat[96 + fr])) * (1.f / 256.f);
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                const float dx = fa[i].x - mean, dy = fa[i].y - mean, dz = fa[i].z - mean, dw = fa[i].w - mean;
                qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            qq += __shfl_xor(qq, 32);
            if (fh == 0) stat[128 + wave * 32 + fr] = qq;
            __syncthreads();
            const float var = ((stat[128 + fr] + stat[160 + fr]) + (stat[192 + fr] + stat[224 + fr])) * (1.f / 256.f);
            const float rstd = rsqrtf(var + a.ln_eps);
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                const float4 g4 = *reinterpret_cast<const float4*>(gb + kb + i * 8);
                const float4 b4 = *reinterpret_cast<const float4*>(gb + 256 + kb + i * 8);
                float4 o;
                o.x = (fa[i].x - mean) * rstd * g4.x + b4.x;
                o.y = (fa[i].y - mean) * rstd * g4.y + b4.y;
                o.z = (fa[i].z - mean) * rstd * g4.z + b4.z;
                o.w = (fa[i].w - mean) * rstd * g4.w + b4.w;
                if (a.ln_relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                o.x += ad[i].x; o.y += ad[i].y; o.z += ad[i].z; o.w += ad[i].w;
                fa[i] = o;
            }
            if (a.xn_out && tn == 0 && (m0 + fr) < a.M) {
#pragma unroll
                for (int i = 0; i < KC; ++i) *reinterpret_cast<float4*>(a.xn_out + ra * a.ldx + kb + i * 8) = fa[i];
            }
        }
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // lanes with fh = 0 / 1 hold k = 8i + {0..3} / 8i + {4..7}: any k <-> (step, half) bijection is a valid MFMA order
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[i].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    // thread -> row r = tid / 8, columns c..c+3 = (tid % 8) * 4;  C/D layout: row = (e&3) + 8*(e>>2) + 4*fh, col = lane&31
    const int r = tid >> 3, c = (tid & 7) * 4;
    const int e = (r & 3) + 4 * (r >> 3), h = (r >> 2) & 1;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(&red[(w * 16 + e) * 64 + h * 32 + c]);
        v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    }
    const long long m = m0 + r;
    const int n = n0 + c;
    if (m >= a.M || n >= a.N) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (a.bias && n + q < a.N) v[q] += a.bias[n + q];
        if (a.relu) v[q] = fmaxf(v[q], 0.f);
    }
    float* yp = a.Y + m * a.ldy + n;
    const bool vec = (a.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.Y) & 15) == 0) && (n + 3 < a.N) &&
                     (!a.res || (reinterpret_cast<uintptr_t>(a.res) & 15) == 0);
    if (vec) {
        if (a.res) {
            const float4 r4 = *reinterpret_cast<const float4*>(a.res + m * a.ldy + n);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n + q < a.N) yp[q] = v[q] + (a.res ? a.res[m * a.ldy + n + q] : 0.f);
    }
}

template <int KC>
__global__ __launch_bounds__(256) void gemm_nt_f32_small_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float red[SMALL_LDS_FLOATS];
    small_tile<KC>(a, blockIdx.x, red);
}

// Up to 3 INDEPENDENT small linears in one launch (e.g. the classification and regression branches, which both hang
// off the layer output): each costs ~6 us of launch ramp + two dependent memory round trips on its own and keeps a
// fraction of the CUs busy; side by side they share that latency.  Workgroups are partitioned by tile count.
struct GroupArgs {
    GemmArgs p[3];
    unsigned tile_end[3];   // exclusive prefix sums of the problems' tile counts
    int n;
};
template <int KC>
__global__ __launch_bounds__(256) void gemm_group_small_kernel(const GroupArgs g) {
    __shared__ __attribute__((aligned(16))) float red[SMALL_LDS_FLOATS];
    const unsigned id = blockIdx.x;
    if (id < g.tile_end[0]) small_tile<KC>(g.p[0], id, red);
    else if (id < g.tile_end[1]) small_tile<KC>(g.p[1], id - g.tile_end[0], red);
    else small_tile<KC>(g.p[2], id - g.tile_end[1], red);
}

// ---- W-stationary strip kernel for [M, 256] x [N >= 16384, 256]^T (the adaptive-mixing parameter generator) ---------
// With 128x128 tiles Q = 900 rows are 7.03 row tiles: 12 % of the MFMA work is padding AND the 1792..2048 tiles
// quantise to 4 rounds over the 512 resident workgroups (M = 896 costs the same 168 us as M = 1024).  Here a WAVE
// instead owns a 64-column strip of W for half of all row fragments and keeps that strip (64 x 256 fp32) in its own
// registers -- 256 of the 512 VGPR+AGPR a wave has at one wave per SIMD, which is all the f32 MFMA needs to run at
// peak -- while X streams through registers straight from L2 (16 rows x 256 k per fragment, prefetched one fragment
// ahead).  No LDS, no barrier; the grid is N/128 workgroups (2 strips x 2 row-halves) of equal work = exactly one per
// CU at N = 32768; the M padding shrinks to ceil(M/16)*16.  v_mfma_f32_16x16x4_f32 with W as the row operand: a lane
// ends up with 4 CONSECUTIVE output columns of one row, i.e. float4 stores.  k order: lane group fk owns
// k = 16 j + 4 fk + i, so both operands are 16-byte loads (same trick as mixing.hip).
typedef float f32x4v __attribute__((ext_vector_type(4)));
// Designed for one wave per SIMD: say so to the register allocator.
#define SBEV_ONE_WAVE_PER_EU __attribute__((amdgpu_waves_per_eu(1, 1)))

template <bool RELU>
__global__ __launch_bounds__(256) SBEV_ONE_WAVE_PER_EU void gemm_nt_f32_strip_kernel(const GemmArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fk = lane >> 4;
    const long long n0 = (long long)blockIdx.x * 128 + (wave >> 1) * 64;
    const int half = wave & 1;

    f32x4v w[16][4];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int cf = 0; cf < 4; ++cf)
            w[j][cf] = *reinterpret_cast<const f32x4v*>(a.W + (n0 + cf * 16 + fi) * a.ldw + 16 * j + 4 * fk);
    f32x4v bias4[4];
#pragma unroll
    for (int cf = 0; cf < 4; ++cf)
        bias4[cf] = a.bias ? *reinterpret_cast<const f32x4v*>(a.bias + n0 + cf * 16 + 4 * fk) : (f32x4v){0.f, 0.f, 0.f, 0.f};

    const int M = (int)a.M;
    const int last = ((M + 15) >> 4) - 1;

    f32x4v xa[16], xb[16];
#define SBEV_LOAD_X(dst, f)                                                                         \
    {                                                                                               \
        int row_ = 16 * (f) + fi;                                                                   \
        row_ = row_ < M ? row_ : M - 1;                                                             \
        const float* p_ = a.X + (long long)row_ * a.ldx + 4 * fk;                                   \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) dst[j] = *reinterpret_cast<const f32x4v*>(p_ + 16 * j); \
    }
#define SBEV_STRIP(src)                                                                             \
    {                                                                                               \
        /* the accumulators start from the bias: with one wave per SIMD every epilogue instruction is idle */ \
        /* matrix-core time (measured: the bias add + ReLU branch + AGPR reads cost 9 %) */        \
        _Pragma("unroll") for (int cf = 0; cf < 4; ++cf) acc[cf] = bias4[cf];                       \
        _Pragma("unroll") for (int j = 0; j < 16; ++j)                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
                _Pragma("unroll") for (int cf = 0; cf < 4; ++cf)                                    \
                    acc[cf] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][cf][i], src[j][i], acc[cf], 0, 0, 0); \
    }
#define SBEV_STORE(f)                                                                               \
    {                                                                                               \
        const int row_ = 16 * (f) + fi;                                                             \
        if (row_ < M) {                                                                             \
            float* y_ = a.Y + (long long)row_ * a.ldy + n0 + 4 * fk;                                \
            _Pragma("unroll") for (int cf = 0; cf < 4; ++cf) {                                      \
                f32x4v v_ = acc[cf];                                                                \
                if (RELU) { v_[0] = fmaxf(v_[0], 0.f); v_[1] = fmaxf(v_[1], 0.f); v_[2] = fmaxf(v_[2], 0.f); v_[3] = fmaxf(v_[3], 0.f); } \
                *reinterpret_cast<f32x4v*>(y_ + cf * 16) = v_;                                      \
            }                                                                                       \
        }                                                                                           \
    }

    // row fragments alternate between the two waves of a strip; which of them starts at 0 flips with the strip so
    // that the odd extra fragment does not always land on the same SIMD pair.  With ONE wave per SIMD every non-MFMA
    // instruction is dead matrix-core time, so: two register sets (xa / xb) used alternately (no rotate copies), the
    // next fragment's X rows requested before each MFMA block, and the stores of fragment f issued at the top of
    // block f+1 (the loop-top wait then never waits for store acks).
    f32x4v acc[4];
    int f = half ^ ((wave >> 1) & 1);
    if (f > last) return;
    int fprev = f;
    SBEV_LOAD_X(xa, f);
    { const int fn = f + 2 <= last ? f + 2 : last; SBEV_LOAD_X(xb, fn); }
    __builtin_amdgcn_sched_barrier(0);
    SBEV_STRIP(xa);
    f += 2;
    while (f <= last) {
        __builtin_amdgcn_sched_barrier(0);
        { const int fn = f + 2 <= last ? f + 2 : last; SBEV_LOAD_X(xa, fn); }     // unconditional (clamped) prefetch
        SBEV_STORE(fprev);
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch AHEAD of the MFMA block (hipcc sinks it otherwise)
        SBEV_STRIP(xb);
        fprev = f;
        f += 2;
        if (f > last) break;
        __builtin_amdgcn_sched_barrier(0);
        { const int fn = f + 2 <= last ? f + 2 : last; SBEV_LOAD_X(xb, fn); }
        SBEV_STORE(fprev);
        __builtin_amdgcn_sched_barrier(0);
        SBEV_STRIP(xa);
        fprev = f;
        f += 2;
    }
    SBEV_STORE(fprev);
#undef SBEV_STORE
#undef SBEV_LOAD_X
#undef SBEV_STRIP
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceArgs a) { reduce_rows(a, blockIdx.x); }

// ---- two INDEPENDENT small ops in one launch ----------------------------------------------------------------------
// A tiny kernel costs ~5 us of launch + drain whatever it does (measured: 4.8 us per launch for back-to-back INDEPENDENT
// small GEMMs in one stream), so the branch tail of a layer pairs ops that do not depend on each other: the LayerNorm of
// the classification branch runs beside the next regression Linear, the second one beside refine_bbox, and the final
// classification Linear beside the next layer's Linear(3->D)+LayerNorm.  Workgroups [0, blocks_a) run op A, the rest
// op B; each op is the same device function its stand-alone kernel calls (identical arithmetic).
enum { PAIR_GEMM = 1, PAIR_REDUCE = 2, PAIR_REFINE = 3, PAIR_LIN3 = 4 };
struct PairArgs {
    GemmArgs gemm;
    ReduceArgs red;
    sbev_ops::MiscArgs refine;
    sbev_ops::PosArgs lin3;
    unsigned blocks_a;
    int kind_a, kind_b;
};
template <int KC>
__global__ __launch_bounds__(256) void pair_kernel(const PairArgs p) {
    __shared__ __attribute__((aligned(16))) float red[SMALL_LDS_FLOATS];
    const bool first = blockIdx.x < p.blocks_a;
    const int kind = first ? p.kind_a : p.kind_b;
    const unsigned b = first ? blockIdx.x : blockIdx.x - p.blocks_a;
    if (kind == PAIR_GEMM) small_tile<KC>(p.gemm, b, red);
    else if (kind == PAIR_REDUCE) reduce_rows(p.red, b);
    else if (kind == PAIR_REFINE) sbev_ops::refine_rows(p.refine, b);
    else sbev_ops::lin3_rows(p.lin3, b);
}

}  // namespace


namespace {
unsigned blocks_of(int kind, const PairArgs& p) {
    switch (kind) {
        case PAIR_GEMM: return (unsigned)(((p.gemm.M + 31) / 32) * ((p.gemm.N + 31) / 32));
        case PAIR_REDUCE: return (unsigned)((p.red.M + 3) / 4);
        case PAIR_REFINE: return (unsigned)((p.refine.BQ + 255) / 256);
        default: return (unsigned)((p.lin3.M + 3) / 4);
    }
}
int launch_pair(PairArgs& p, hipStream_t s, const char* what) {
    p.blocks_a = blocks_of(p.kind_a, p);
    const unsigned total = p.blocks_a + blocks_of(p.kind_b, p);
    const int K = (p.kind_a == PAIR_GEMM || p.kind_b == PAIR_GEMM) ? p.gemm.K : 256;
    if (K == 512)
        hipLaunchKernelGGL((pair_kernel<16>), dim3(total), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((pair_kernel<8>), dim3(total), dim3(256), 0, s, p);
    return sbev::check_launch(what);
}
}  // namespace

namespace sbev {
// true when sbev_linear_f32 would run [M, K] x [N, K]^T on the 32 x 32 small-tile kernel (the pairs use the same tiles)
bool small_linear_shape(int64_t M, int N, int K) {
    return (K == 256 || K == 512) && K % 32 == 0 && ((M + 127) / 128) * ((N + 127) / 128) < 256 &&
           !(K == 256 && N % 128 == 0 && N / 128 >= 192);
}
int launch_ln_and_linear(const float* X, const float* ln_w, const float* ln_b, float eps, int ln_relu, float* Yln, int64_t M, int N,
                         const float* Xg, const float* W, const float* bias, float* Yg, int Ng, int K, int relu, hipStream_t s) {
    PairArgs p{};
    p.kind_a = PAIR_GEMM; p.kind_b = PAIR_REDUCE;       // the GEMM tiles first: they are the longer workgroups
    p.gemm = GemmArgs{Xg, W, bias, nullptr, Yg, M, Ng, K, K, K, Ng, K, relu};
    p.red = ReduceArgs{X, nullptr, nullptr, ln_w, ln_b, nullptr, Yln, M, N, 1, ln_relu, eps};
    return launch_pair(p, s, "sbev_decoder (LayerNorm || Linear)");
}
int launch_ln_and_refine(const float* X, const float* ln_w, const float* ln_b, float eps, int ln_relu, float* Yln, int64_t M, int N,
                         const float* bbox, const float* reg, const float* vel_div, float* out, int Q, int code, hipStream_t s) {
    PairArgs p{};
    p.kind_a = PAIR_REDUCE; p.kind_b = PAIR_REFINE;
    p.red = ReduceArgs{X, nullptr, nullptr, ln_w, ln_b, nullptr, Yln, M, N, 1, ln_relu, eps};
    p.refine = sbev_ops::MiscArgs{bbox, reg, vel_div, out, M, Q, code};
    return launch_pair(p, s, "sbev_decoder (LayerNorm || refine_bbox)");
}
int launch_linear_and_lin3(const float* Xg, const float* W, const float* bias, float* Yg, int64_t M, int Ng, int K, int relu,
                           const float* x3, int64_t ldx3, const float* w3, const float* b3, const float* ln_w, const float* ln_b,
                           float eps, float* y3, int N3, hipStream_t s) {
    PairArgs p{};
    p.kind_a = PAIR_GEMM; p.kind_b = PAIR_LIN3;
    p.gemm = GemmArgs{Xg, W, bias, nullptr, Yg, M, Ng, K, K, K, Ng, K, relu};
    p.lin3 = sbev_ops::PosArgs{x3, w3, b3, ln_w, ln_b, y3, M, N3, (int)ldx3, eps};
    return launch_pair(p, s, "sbev_decoder (Linear || Linear3+LayerNorm)");
}
}  // namespace sbev

extern "C" int sbev_linear_f32(const float* X, const float* W, const float* bias, const float* residual, float* Y,
                               int64_t M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldy, int relu,
                               sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 1 && K >= 1, "sbev_linear_f32: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && W && Y, "sbev_linear_f32: null pointer");
    SBEV_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && ldx >= K && ldw >= K && ldy >= N, "sbev_linear_f32: leading dimensions (ldx, ldw multiples of 4)");
    SBEV_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15) == 0, "sbev_linear_f32: X and W must be 16-byte aligned");
    GemmArgs a{X, W, bias, residual, Y, M, N, K, ldx, ldw, ldy, K, relu};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long big = ((M + 127) / 128) * ((N + 127) / 128);
    SBEV_REQUIRE(big <= 0x3fffffffLL, "sbev_linear_f32: too many tiles");
    const long long small = ((M + 63) / 64) * ((N + 63) / 64);
    if (K == 256 && N % 128 == 0 && N / 128 >= 192 && M < (1 << 24) && ldy % 4 == 0 && ((uintptr_t)Y & 15) == 0 &&
        (!bias || ((uintptr_t)bias & 15) == 0) && !residual) {
        // parameter-generator shape: W strips stationary in registers, all rows stream past (see the kernel's header)
        hipEvent_t e0, e1;
        const bool prof = sbev::profile_begin(s, &e0, &e1, 1);
        if (relu)
            hipLaunchKernelGGL(gemm_nt_f32_strip_kernel<true>, dim3((unsigned)(N / 128)), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(gemm_nt_f32_strip_kernel<false>, dim3((unsigned)(N / 128)), dim3(256), 0, s, a);
        if (prof) sbev::profile_end(s, e0, e1, 1);
    } else if (K % BK != 0) {  // slow path (never taken by the decoder: its K are 256 / 512 / 32768)
        hipLaunchKernelGGL((gemm_nt_f32_kernel<false, 1, 1, true>), dim3((unsigned)small), dim3(256), 0, s, a);
    } else if (big >= 256) {   // enough 128x128 tiles to fill the 256 CUs
        hipLaunchKernelGGL((gemm_nt_f32_kernel<false, 2, 2, false>), dim3((unsigned)big), dim3(256), 0, s, a);
    } else if (K == 256 || K == 512) {   // the decoder's small linears: 32x32 tiles, K split over the 4 waves
        const long long t32 = ((M + 31) / 32) * ((N + 31) / 32);
        if (K == 256)
            hipLaunchKernelGGL((gemm_nt_f32_small_kernel<8>), dim3((unsigned)t32), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((gemm_nt_f32_small_kernel<16>), dim3((unsigned)t32), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((gemm_nt_f32_kernel<false, 1, 1, false>), dim3((unsigned)small), dim3(256), 0, s, a);
    }
    return sbev::check_launch("sbev_linear_f32");
}


namespace sbev {
bool ln_linear_fusable(int64_t M, int N, int K) {
    static const bool off = getenv("SBEV_NO_LN_FUSE") != nullptr;      // A/B switch: the two launches the prologue replaces
    // every column tile re-normalises its rows, so the prologue's cost grows with M while the launch it saves does not:
    // measured neutral at 3200 rows and -0.5 % at 3600 (c3 / c4), so only up to 2048 rows
    return !off && K == 256 && M <= 2048 && small_linear_shape(M, N, K);
}
}  // namespace sbev

extern "C" int sbev_linear_group_f32(const sbev_linear_problem* probs, int n, sbev_stream_t stream) {
    return sbev::launch_linear_group(probs, n, nullptr, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sbev_ln_linear_f32(const float* X, const float* ln_w, const float* ln_b, float ln_eps, int ln_relu,
                                  const float* ln_add, float* Xn, const float* W, const float* bias, const float* residual,
                                  float* Y, int64_t M, int N, int K, int64_t ldw, int64_t ldy, int relu, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 1 && K >= 4 && K % 4 == 0 && K <= 1024, "sbev_ln_linear_f32: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && ln_w && ln_b && Xn && W && Y, "sbev_ln_linear_f32: null pointer");
    if (sbev::ln_linear_fusable(M, N, K)) {
        const sbev_linear_problem q{X, W, bias, residual, Y, M, N, K, K, ldw, ldy, relu};
        const sbev::LnPrologue ln{ln_w, ln_b, ln_eps, ln_relu, ln_add, Xn};
        return sbev::launch_linear_group(&q, 1, &ln, reinterpret_cast<hipStream_t>(stream));
    }
    // shapes the small-tile kernel does not take: the two launches it replaces
    int st = sbev_layer_norm_f32(X, ln_w, ln_b, ln_eps, ln_add, Xn, M, K, ln_relu, stream);
    if (st != SBEV_OK) return st;
    return sbev_linear_f32(Xn, W, bias, residual, Y, M, N, K, K, ldw, ldy, relu, stream);
}

int sbev::launch_linear_group(const sbev_linear_problem* probs, int n, const LnPrologue* ln, hipStream_t s) {
    SBEV_REQUIRE(probs && n >= 1 && n <= 3, "sbev_linear_group_f32: 1..3 problems per launch (got %d)", n);
    GroupArgs g{};
    g.n = n;
    unsigned end = 0;
    const int K = probs[0].K;
    SBEV_REQUIRE(K == 256 || K == 512, "sbev_linear_group_f32: built for K = 256 / 512 (got %d)", K);
    for (int i = 0; i < 3; ++i) {
        if (i < n) {
            const sbev_linear_problem& q = probs[i];
            SBEV_REQUIRE(q.K == K, "sbev_linear_group_f32: all problems of a group must share K");
            SBEV_REQUIRE(q.M >= 1 && q.N >= 1 && q.X && q.W && q.Y, "sbev_linear_group_f32: problem %d has null / empty operands", i);
            SBEV_REQUIRE(q.ldx % 4 == 0 && q.ldw % 4 == 0 && q.ldx >= K && q.ldw >= K && q.ldy >= q.N, "sbev_linear_group_f32: problem %d leading dimensions", i);
            SBEV_REQUIRE((((uintptr_t)q.X | (uintptr_t)q.W) & 15) == 0, "sbev_linear_group_f32: problem %d X / W not 16-byte aligned", i);
            g.p[i] = GemmArgs{q.X, q.W, q.bias, q.residual, q.Y, q.M, q.N, q.K, q.ldx, q.ldw, q.ldy, q.K, q.relu};
            if (ln) {
                SBEV_REQUIRE(K == 256 && q.X == probs[0].X && q.M == probs[0].M && q.ldx == K,
                             "sbev_ln_linear_f32: the LayerNorm prologue needs K = 256 and one dense X shared by the group");
                SBEV_REQUIRE(((((uintptr_t)ln->g | (uintptr_t)ln->b | (uintptr_t)ln->add | (uintptr_t)ln->xn)) & 15) == 0,
                             "sbev_ln_linear_f32: gamma / beta / add / Xn must be 16-byte aligned");
                g.p[i].ln_g = ln->g; g.p[i].ln_b = ln->b; g.p[i].ln_add = ln->add;
                g.p[i].xn_out = i == 0 ? ln->xn : nullptr;
                g.p[i].ln_eps = ln->eps; g.p[i].ln_relu = ln->relu;
            }
            end += (unsigned)(((q.M + 31) / 32) * ((q.N + 31) / 32));
        }
        g.tile_end[i] = end;
    }
    if (K == 256)
        hipLaunchKernelGGL((gemm_group_small_kernel<8>), dim3(end), dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_group_small_kernel<16>), dim3(end), dim3(256), 0, s, g);
    return sbev::check_launch("sbev_linear_group_f32");
}

namespace {
inline bool regtile_ok(int64_t M, int N, int K) { return N % 128 == 0 && K % 32 == 0 && K >= 2048 && M < (1 << 24); }
}  // namespace

// How many K splits sbev_linear_splitk_f32 should be given for this shape: enough wave tasks (48 rows x 128 columns
// x K/splits) to fill the 1024 one-wave-per-SIMD slots once; shapes the register-tiled kernel does not take fall back
// to 128 x 128 tiles x splits ~ 512 workgroups.
extern "C" int sbev_linear_splitk_plan(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    if (regtile_ok(M, N, K)) return sbev::regtile_plan(M, N, K);      // gemm_regtile.hip
    const long long tiles = ((M + 127) / 128) * ((N + 127) / 128);
    long long s = (512 + tiles - 1) / tiles;
    if (s > K / 512) s = K / 512;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int64_t sbev_linear_splitk_workspace(int64_t M, int N, int splits) {
    return (int64_t)sizeof(float) * M * N * (splits > 0 ? splits : 0);
}

namespace sbev {
// the GEMM half of sbev_linear_splitk_f32: *used partial slabs [used, M, N] in `workspace`, no reduction (the row-chain
// tail kernel of the decoder sums them in its prologue)
int launch_splitk_slabs(const float* X, const float* W, int64_t M, int N, int K, int64_t ldx, int64_t ldw, int splits,
                        float* workspace, int* used, hipStream_t s) {
    if (regtile_ok(M, N, K) && splits <= K / 32) return launch_splitk_regtile(X, W, workspace, M, N, K, ldx, ldw, splits, used, s);
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    *used = (K + kps - 1) / kps;
    GemmArgs a{X, W, nullptr, nullptr, workspace, M, N, K, ldx, ldw, (long long)N, kps, 0};
    const long long tiles = ((M + 127) / 128) * ((N + 127) / 128);
    if (K % BK != 0)
        hipLaunchKernelGGL((gemm_nt_f32_kernel<true, 2, 2, true>), dim3((unsigned)tiles, 1, (unsigned)*used), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_nt_f32_kernel<true, 2, 2, false>), dim3((unsigned)tiles, 1, (unsigned)*used), dim3(256), 0, s, a);
    return check_launch("sbev_linear_splitk_f32 (gemm)");
}
}  // namespace sbev

static int check_splitk_args(const float* X, const float* W, const void* Y, const float* workspace, int64_t M, int N, int K,
                             int64_t ldx, int64_t ldw, int splits) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024 && K >= 1, "sbev_linear_splitk_f32: need N %% 4 == 0, N <= 1024 (N=%d)", N);
    SBEV_REQUIRE(splits >= 1 && splits <= 1024, "sbev_linear_splitk_f32: splits=%d", splits);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && W && Y && workspace, "sbev_linear_splitk_f32: null pointer");
    SBEV_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && ldx >= K && ldw >= K, "sbev_linear_splitk_f32: leading dimensions");
    SBEV_REQUIRE((((uintptr_t)X | (uintptr_t)W | (uintptr_t)workspace) & 15) == 0, "sbev_linear_splitk_f32: 16-byte alignment");
    return SBEV_OK;
}

extern "C" int sbev_linear_splitk_f32(const float* X, const float* W, const float* bias, const float* residual,
                                      const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                                      int64_t M, int N, int K, int64_t ldx, int64_t ldw, int relu,
                                      int splits, float* workspace, sbev_stream_t stream) {
    int st = check_splitk_args(X, W, Y, workspace, M, N, K, ldx, ldw, splits);
    if (st != SBEV_OK || M == 0) return st;
    SBEV_REQUIRE((ln_w == nullptr) == (ln_b == nullptr), "sbev_linear_splitk_f32: ln_w and ln_b go together");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int used;
    st = sbev::launch_splitk_slabs(X, W, M, N, K, ldx, ldw, splits, workspace, &used, s);
    if (st != SBEV_OK) return st;
    ReduceArgs r{workspace, bias, residual, ln_w, ln_b, nullptr, Y, M, N, used, relu, ln_eps};
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, r);
    return sbev::check_launch("sbev_linear_splitk_f32 (reduce)");
}

// The split-K slab reducer on its own (shared with the bf16x3 GEMM): Y = LN?( relu?(sum_z slabs[z] + bias) + residual )
extern "C" int sbev_splitk_reduce_f32(const float* slabs, int splits, const float* bias, const float* residual,
                                      const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                                      int64_t M, int N, int relu, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024 && splits >= 1, "sbev_splitk_reduce_f32: need N %% 4 == 0, N <= 1024");
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(slabs && Y && ((ln_w == nullptr) == (ln_b == nullptr)), "sbev_splitk_reduce_f32: null pointer");
    ReduceArgs r{slabs, bias, residual, ln_w, ln_b, nullptr, Y, M, N, splits, relu, ln_eps};
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), r);
    return sbev::check_launch("sbev_splitk_reduce_f32");
}

// Row-wise LayerNorm (+ optional ReLU) of [M, N] (N % 4 == 0, N <= 1024): the split-K reducer with one slab.
extern "C" int sbev_layer_norm_f32(const float* X, const float* ln_w, const float* ln_b, float eps,
                                   const float* add_after, float* Y, int64_t M, int N, int relu, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024, "sbev_layer_norm_f32: need N %% 4 == 0, N <= 1024 (N=%d)", N);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && Y && ln_w && ln_b, "sbev_layer_norm_f32: null pointer");
    ReduceArgs r{X, nullptr, nullptr, ln_w, ln_b, add_after, Y, M, N, 1, relu, eps};
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), r);
    return sbev::check_launch("sbev_layer_norm_f32");
}
