// Adaptive mixing core (gfx950): per (query, group) the two dynamic matmuls with their LayerNorm + ReLU.
//
// Replaces the middle of AdaptiveMixing.inner_forward (models/sparsebev_transformer.py:362-374):
//     y = relu(layer_norm_[Pin,C]( x[Pin,C] @ M[C,C] ))          adaptive channel mixing
//     y = relu(layer_norm_[Pout,C]( S[Pout,Pin] @ y[Pin,C] ))    adaptive point mixing
// M and S are the per-query dynamic weights produced by the parameter-generator Linear (sbev_linear_f32).
//
// One 256-thread workgroup per (b*Q + q, g); both matmuls run on v_mfma_f32_16x16x4_f32 (exact fp32), wave w owns the
// 16-channel column slab w of both outputs.  The kernel is HBM-bound (72 KiB moved per ~3k MFMA cycles of work: a
// load-and-store-only version of it runs at 47 us = 5.6 TB/s for config 2), so everything is about keeping bytes in
// flight and the dependent chain of one workgroup short (73 -> 54 us):
//   * Pin % 16 == 0 (the decoder's case): the MFMA fragments of x and of this wave's 16 columns of M are loaded straight
//     from HBM into registers -- matmul 1 waits for no LDS staging and no barrier;
//   * y1 never leaves registers: the B fragment matmul 2 needs from a lane is exactly what that lane's matmul-1
//     accumulators hold (same rows 4 fk + j, same column);
//   * S is requested at kernel start, parked in registers and written to LDS behind matmul 1, in front of the
//     LayerNorm-1 exchange whose barriers publish it; LDS carries only S and the output staging: 34.8 KiB, 4 workgroups
//     per CU (was 44.5 KiB / 3);
//   * LayerNorm statistics: per-wave (sum, centred M2) by DPP sums, ONE LDS exchange per LayerNorm merged with Chan's
//     parallel-variance formula (as stable as two-pass, half the barriers);
//   * the [Pout,C] result is transposed through LDS so the workgroup writes its 32 KiB as contiguous 16-byte stores.
// Other Pin keep the generic path (x, M, S staged in LDS, y1 through LDS).
#include "sbev_common.hpp"
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MixArgs {
    const float* x;       // [BQ, G, Pin, C]
    const float* params;  // [BQ, G, C*C + Pout*Pin]   (M = [C_in, C_out] first, then S = [Pout, Pin])
    float* y;             // [BQ, G, Pout, C]
    long long n_items;    // BQ * G
    int Pin;
    float eps;
    float out_up;         // 0: y is fp32.  2^e: y holds (fp16 hi, fp16 lo) PAIRS of y 2^e in each 32-bit slot -- the operand format of
                          // the fp16 out-projection (gemm_bf16s.hip: hi = RNE(y 2^e), lo = RNE(y 2^e - hi)), split here once per element
};

#include "msmv_common.hpp"

// 1: request the item's M / S parameters BEFORE the gather (their HBM latency hides under it, but 32 more live registers
// push the fused kernel from 3 to 2 waves per SIMD); 0: request them after the gather
#ifndef SBEV_SAMPLE_MIX_PREFETCH
#define SBEV_SAMPLE_MIX_PREFETCH 0
#endif

// Fused sampler + mixing (SL > 0 instantiations of the kernel below): the workgroup of item (b*Q + q, g) first GATHERS its
// own x[T*P, 64] -- wave w runs the sampler's per-chunk code (msmv_chunk.inc, one 4-point chunk per frame) for the frames
// t = w, w + 4, ... of sample batch b' = (b*T + t)*G + g and leaves the rows in LDS -- so the sampled features
// (29.5 MB per layer at config 2) are neither written to nor re-read from HBM, and one launch disappears.
struct SampleMixArgs {
    const float* x;       // unused (kept so that the kernel body reads the same member names)
    const float* params;
    float* y;
    long long n_items;
    int Pin;
    float eps;
    float out_up;
    // launch order (sbev_query_order; null = block b is item b): XCD x = b % 8 walks entries [x * order_per, (x + 1) * order_per) of the
    // GROUP-major list (g, position) and takes row order[position] -- one group and one contiguous arc of the camera ring per L2
    const int* order;
    int order_per;
    MsmvArgs s;           // sampler descriptors: s.loc [B*T*G, Q, P, 3], s.w [B*G*T, Q, P, L]; s.Q, s.T, s.G, s.P (== 4), s.N
};
template <int SL>
struct MixArgsOf { typedef MixArgs type; };
template <>
struct MixArgsOf<4> { typedef SampleMixArgs type; };
template <>
struct MixArgsOf<5> { typedef SampleMixArgs type; };

#ifdef SBEV_MIX_TRACE          // wall-clock phase stamps (s_memrealtime, 10 ns ticks) of thread 0 of every workgroup of the fused launch
                               // (tools/exp/r4_mix_trace.py; never in the product build)
constexpr int MIX_TRACE_SLOTS = 16, MIX_TRACE_WGS = 8192;
__device__ long long g_mix_trace[MIX_TRACE_WGS * MIX_TRACE_SLOTS];
#define MIX_STAMP(i)                                                                                               \
    if (L > 0 && threadIdx.x == 0 && blockIdx.x < MIX_TRACE_WGS) {                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        g_mix_trace[blockIdx.x * MIX_TRACE_SLOTS + (i)] = (long long)__builtin_amdgcn_s_memrealtime();            \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
#else
#define MIX_STAMP(i)
#endif

constexpr int C = 64, POUT = 128;
constexpr int LDA = C + 4;    // A-operand rows (x): 16-B aligned rows, <= 2-way bank conflict on fragment reads
constexpr int LDB = C + 16;   // B-operand rows (M, y1): row stride = 16 banks -> conflict-free fragment reads
constexpr int LDY = C + 4;    // output transpose buffer

__device__ __forceinline__ float wave_sum(float v) { return sbev::wave_sum_dpp(v); }

// block-wide sum of one float per thread (4 waves); `red` is 4 floats of LDS; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red, int wave, int lane) {
    v = wave_sum(v);
    __syncthreads();                      // protect `red` from the previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// LayerNorm statistics of the whole block with ONE LDS exchange: every wave reduces its own slab to (sum, M2 about its
// own mean) with DPP sums (no barrier), the four (sum, M2) pairs are merged with Chan's parallel-variance formula --
// as stable as the two-pass form, half the barriers.  n_w = elements per wave, n = 4 n_w.  `red` = 8 floats of LDS.
__device__ __forceinline__ void block_mean_rstd(float s_w, float m2_w, float n_w, float eps, float* red, int wave, int lane,
                                                float& mean, float& rstd) {
    __syncthreads();                      // protect `red` from the previous use (and fence the LDS reads before it)
    if (lane == 0) {
        red[wave] = s_w;
        red[4 + wave] = m2_w;
    }
    __syncthreads();
    const float n = 4.f * n_w;
    mean = ((red[0] + red[1]) + (red[2] + red[3])) / n;
    float m2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float d = red[w] / n_w - mean;
        m2 += red[4 + w] + n_w * d * d;
    }
    rstd = rsqrtf(m2 / n + eps);
}

// (fp16 hi, fp16 lo) of a scaled value in one 32-bit slot: hi = RNE_fp16(x), lo = RNE_fp16(x - hi) (the subtraction is exact)
__device__ __forceinline__ float f16_pair(float x) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    return __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16));
}

// RT = ceil(Pin / 16) row tiles of matmul 1; WIDE: Pin % 16 == 0 (b128 A reads in matmul 2); L > 0: fused sampler with L
// feature levels of storage type FT (WIDE only)
#ifndef SBEV_SAMPLE_MIX_WAVES
#define SBEV_SAMPLE_MIX_WAVES 3          // waves per SIMD asked of the register allocator for the fused instantiations: 184 -> 151 VGPRs, no
                                         // spills, 318 -> 326 samples/s at config 2 (unfused 320; requesting M / S before the gather: 313..319)
#endif
template <int L, int RT>
constexpr int mix_min_waves() { return L > 0 ? SBEV_SAMPLE_MIX_WAVES : 1; }   // (plain RT = 8: 128 registers of x fragments)

// PAD: Pin % 16 != 0 on the WIDE path (round 3; the last row tile of x and the last k block of S are partly padding) and, fused,
// 4 or 8 points per frame -- a template parameter so that the Pin % 16 == 0, P = 4 instantiations stay exactly the tuned code of rounds 1-2
template <int RT, bool WIDE, int L = 0, typename FT = float, bool PAD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(mix_min_waves<L, RT>()))) void adaptive_mixing_kernel(const typename MixArgsOf<L>::type a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Pin = a.Pin;
    // fused, more than 4 row tiles (Pin > 64): S goes through LDS in TWO k halves ([128][64 + 4] floats each, round 4) -- the whole
    // [128][Pin + 4] block was 63.5 KB at Pin = 120 and held a CU to 2 workgroups; with 34.8 KB the registers decide (3 per CU)
    constexpr bool KSPLIT = L > 0 && RT > 4;
    constexpr int KH = 64;                                      // columns of the first half
    const int lds_s = KSPLIT ? KH + 4 : Pin + 4;                // S row stride
    // WIDE (Pin % 16 == 0, the decoder's case): the x fragments come straight from HBM into registers and y1 never leaves
    // them (the B fragment of matmul 2 is exactly what the lane's own matmul-1 accumulators hold), so LDS only carries
    // S (and the output staging): 34.8 KiB -> 4 workgroups per CU instead of 3, and nothing waits for a staging barrier.
    float* red = smem;                                          // [8] block-reduction scratch (never aliased)
    float* Xs = smem + 8;                                       // [RT*16][LDA]  (generic path only; dead after matmul 1)
    float* Y1 = smem + 8;                                       // [RT*16][LDB]  (generic path only) aliases Xs
    float* Ms = Y1 + (WIDE ? 0 : RT * 16 * LDB);                // [C][LDB]  (generic path only)
    float* Ss = Ms + (WIDE ? 0 : C * LDB);                                   // [POUT][lds_s]
    float* Yo = smem + 8;                                       // [POUT][LDY], aliases everything after matmul 2

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long item = blockIdx.x;
    MIX_STAMP(0)
#ifdef SBEV_MIX_TRACE
    if (L > 0 && threadIdx.x == 0 && blockIdx.x < MIX_TRACE_WGS) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_mix_trace[blockIdx.x * MIX_TRACE_SLOTS + 15] = ((long long)xcc << 32) | hwid;
    }
#endif
    if constexpr (L > 0) {
        if (a.order) {
            const unsigned m = (blockIdx.x & 7u) * (unsigned)a.order_per + (blockIdx.x >> 3);
            if (m >= (unsigned)a.n_items) return;               // (the grid is 8 * order_per blocks; whole workgroups leave, before any barrier)
            const unsigned rows = (unsigned)a.n_items / (unsigned)a.s.G;
            const unsigned g = m / rows;
            item = (long long)a.order[m - g * rows] * a.s.G + g;
        }
    }
    const float* xg = a.x + item * Pin * C;
    const float* pg = a.params + item * (C * C + POUT * Pin);
    const float* sg = pg + C * C;

    // ---- stage x, M, S (coalesced float4) --------------------------------------------------------------
    const int fi = lane & 15, fk = lane >> 4;                   // fragment row/col index, k sub-index
    // fused, more than 4 row tiles: the gathered rows stay in LDS (ON the S buffer, see alias_x) and matmul 1 reads its A fragments
    // from there block by block -- 8 row tiles of resident fragments would be 128 registers next to the 64 of the parked S
    constexpr bool XLDS = L > 0 && RT > 4;
    const float* Xrows = nullptr;
    f32x4 xf[WIDE && !XLDS ? RT : 1][C / 16];                   // WIDE: A fragments of matmul 1, x[r*16 + fi][16 blk + 4 fk ..]
    if (WIDE && L == 0) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int blk = 0; blk < C / 16; ++blk) {
                // rows >= Pin of the last tile (Pin % 16 != 0) are zero padding: a valid address, then a select
                const int row = PAD && r == RT - 1 ? min(r * 16 + fi, Pin - 1) : r * 16 + fi;
                xf[r][blk] = *reinterpret_cast<const f32x4*>(xg + row * C + 16 * blk + 4 * fk);
                if (PAD && r == RT - 1 && r * 16 + fi >= Pin) xf[r][blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    }
    for (int i = tid; i < (WIDE ? 0 : RT * 16 * (C / 4)); i += 256) {
        const int r = i / (C / 4), c4 = (i % (C / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < Pin) v = *reinterpret_cast<const float4*>(xg + r * C + c4);
        *reinterpret_cast<float4*>(&Xs[r * LDA + c4]) = v;      // rows >= Pin are zero padding
    }
    // WIDE: the B fragments of matmul 1 (this wave's 16 columns of M, 4 KiB) also go straight to registers -- matmul 1
    // then depends on no LDS staging and no barrier at all
    float mf[WIDE ? C / 16 : 1][4];
#define SBEV_LOAD_MF()                                                                                   \
    _Pragma("unroll") for (int blk = 0; blk < C / 16; ++blk)                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) mf[blk][j] = pg[(16 * blk + 4 * fk + j) * C + wave * 16 + fi];
    if (WIDE && (L == 0 || SBEV_SAMPLE_MIX_PREFETCH)) { SBEV_LOAD_MF() }
    for (int i = tid; i < (WIDE ? 0 : C * (C / 4)); i += 256) {
        const int r = i / (C / 4), c4 = (i % (C / 4)) * 4;
        *reinterpret_cast<float4*>(&Ms[r * LDB + c4]) = *reinterpret_cast<const float4*>(pg + r * C + c4);
    }
    // S is not needed before matmul 2: WIDE requests it now but parks it in registers (2 RT float4 per thread) and writes
    // it to LDS behind matmul 1, in front of the LayerNorm-1 exchange whose barriers publish it -- its HBM latency hides
    // behind matmul 1 and the staging barrier only waits for M
    f32x4 sreg[WIDE ? (KSPLIT ? RT : 2 * RT) : 1];
    // KSPLIT: the same 8 registers carry columns [0, 64) first (thread -> row e / 16, 16-byte column e % 16 of element e = tid + 256 k)
    // and, requested once those are in LDS, columns [64, Pin) (row e / nb, column 64 + 4 (e % nb), nb = (Pin - 64) / 4 sixteen-byte
    // pieces per row; e past the end: clamped, dropped)
    const int nb2 = KSPLIT ? (Pin - KH) >> 2 : 1;
#define SBEV_LOAD_S() \
    if constexpr (KSPLIT) {                                                                                                        \
        _Pragma("unroll") for (int k = 0; k < RT; ++k) {                                                                           \
            const int e = tid + 256 * k;                                                                                           \
            sreg[k] = *reinterpret_cast<const f32x4*>(sg + (e >> 4) * Pin + (e & 15) * 4);                                         \
        }                                                                                                                          \
    } else {                                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 2 * RT; ++k) sreg[k] = *reinterpret_cast<const f32x4*>(sg + (PAD ? min((tid + 256 * k) * 4, POUT * Pin - 4) : (tid + 256 * k) * 4)); \
    }
    if (WIDE && (L == 0 || SBEV_SAMPLE_MIX_PREFETCH)) { SBEV_LOAD_S() }
    for (int i = tid; i < (WIDE ? 0 : POUT * Pin / 4); i += 256) {
        const int r = (i * 4) / Pin, c4 = (i * 4) % Pin;
        *reinterpret_cast<float4*>(&Ss[r * lds_s + c4]) = *reinterpret_cast<const float4*>(sg + i * 4);
    }
    if (!WIDE) __syncthreads();
    if constexpr (L > 0) {
        // ---- fused gather: x rows t*4 + p of this (query, group) item, frames t = wave, wave + 4, ... -------------------
        // (the M / S requests above are already in flight: their HBM latency hides under the gather)
        // [Pin][LDA]: behind S; dead once xf is loaded.  Pin > 64 (S alone is 36 .. 62 KiB): ON S instead, which is written only after
        // every wave has taken its x fragments (one more barrier) -- 2 workgroups per CU instead of 1
        const bool alias_x = RT > 4;                            // (Pin > 64)
        float* Xg = alias_x ? Ss : Ss + POUT * lds_s;
        {
            const MsmvArgs& m = a.s;
            const int G = m.G, T = m.T, Q = m.Q;
            const unsigned ubq = (unsigned)(item / G);
            const int g = (int)(item - (long long)ubq * G);
            const unsigned b = ubq / (unsigned)Q;
            const int q = (int)(ubq - b * (unsigned)Q);
            const int k = lane >> 4;
            const int j4 = (lane & 15) * 4;
            const float nm1 = (float)(m.N - 1);
            // unit u = 4-point chunk h of frame t: u = t * (P / 4) + h, P = 4 or 8 points per frame and group
            // (PAD = false is the tuned P = 4 instantiation: pcs and Pq fold to constants and this is the round-2 loop over frames)
            const int pcs = PAD ? m.P >> 3 : 0;                 // log2(P / 4)
            const int Pq = PAD ? m.P : 4;
            const int NU = T << pcs;
            auto fetch_lv = [&](int u) {
                const int t = u >> pcs, h = u - (t << pcs);
                const long long it = ((long long)(b * (unsigned)T + t) * G + g) * Q + q;      // (b', q) with b' = (b*T + t)*G + g
                float v = 0.f;
                if (lane < 12) v = m.loc[(it * Pq + h * 4) * 3 + lane];
                if (lane >= 16 && lane < 16 + 4 * L) v = m.w[(it * Pq + h * 4) * L + (lane - 16)];
                return v;
            };
            float lv_next = wave < NU ? fetch_lv(wave) : 0.f;
#pragma unroll 1
            for (int u = wave; u < NU; u += 4) {
                if (u == 4) { MIX_STAMP(10) }                  // wave 0's second unit starts (its loc / weights were requested a unit ago)
                const int t = u >> pcs;
                const float lv = lv_next;
                if (u + 4 < NU) lv_next = fetch_lv(u + 4);
                unsigned ubo = b * (unsigned)T + (unsigned)t;
                if (m.ring_T) ubo = b * (unsigned)m.n_slots + (unsigned)m.slots[t];
                long long slab[L];
#pragma unroll
                for (int l = 0; l < L; ++l) slab[l] = (long long)ubo * m.stride_bo[l] + (long long)g * m.stride_g;     // wave-uniform
                constexpr bool BUF = true;                      // buffer-load taps: every slab below 2 GiB (host-checked)
                TapSrc<L, FT, BUF> src;
                src.init(m, slab, j4);
                const int npts = 4;
                [[maybe_unused]] const bool chan_ok = true;
                {
                    const MsmvArgs& a = m;                      // the included chunk code names its argument block `a`
#include "msmv_chunk.inc"
                    if (u == 4) { MIX_STAMP(12) }              // ... its taps are consumed
                    float4 sv;
                    sv.x = corner_reduce_scatter(acc[0].x, acc[1].x, acc[2].x, acc[3].x);
                    sv.y = corner_reduce_scatter(acc[0].y, acc[1].y, acc[2].y, acc[3].y);
                    sv.z = corner_reduce_scatter(acc[0].z, acc[1].z, acc[2].z, acc[3].z);
                    sv.w = corner_reduce_scatter(acc[0].w, acc[1].w, acc[2].w, acc[3].w);
                    *reinterpret_cast<float4*>(&Xg[(u * 4 + k) * LDA + j4]) = sv;       // row k of the wave = point k of unit u
                }
            }
        }
        MIX_STAMP(1)                                            // this wave's gather is done (its rows are in LDS)
        if (!SBEV_SAMPLE_MIX_PREFETCH) {                        // M / S requested only now: the gather phase keeps its 3 waves per SIMD
            __builtin_amdgcn_sched_barrier(0);
            SBEV_LOAD_MF()
            SBEV_LOAD_S()
        }
        __syncthreads();
        MIX_STAMP(2)                                            // every wave's gather is done
        Xrows = Xg;
        if constexpr (!XLDS) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int blk = 0; blk < C / 16; ++blk) {
                    const int row = PAD && r == RT - 1 ? min(r * 16 + fi, Pin - 1) : r * 16 + fi;
                    xf[r][blk] = *reinterpret_cast<const f32x4*>(&Xg[row * LDA + 16 * blk + 4 * fk]);
                    if (PAD && r == RT - 1 && r * 16 + fi >= Pin) xf[r][blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            if (alias_x) __syncthreads();
        }
    }
#undef SBEV_LOAD_MF
#undef SBEV_LOAD_S

    const int cw = wave * 16;                                   // this wave's channel slab
    MIX_STAMP(3)                                                // x fragments in registers

    // ---- matmul 1: y1[Pin, 64] = x[Pin, 64] @ M[64, 64]; wave w -> columns [16w, 16w+16) ----------------
    f32x4 acc1[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc1[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K is walked in blocks of 16: lane group fk owns k = 16*blk + 4*fk + j (j = 0..3), so an A fragment is ONE
    // 16-byte LDS read per 4 MFMAs (any k <-> (step, lane group) bijection is a valid order for A and B together)
    // consecutive MFMAs go to DIFFERENT accumulators (j outer, r inner): four back-to-back MFMAs on one accumulator are a
    // dependent chain, and PMC showed 56 % of the wave cycles of this kernel as issue stalls
#pragma unroll
    for (int blk = 0; blk < C / 16; ++blk) {
        float bq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = WIDE ? mf[blk][j] : Ms[(16 * blk + 4 * fk + j) * LDB + cw + fi];
        f32x4 a4[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            if constexpr (XLDS) {
                const int row = PAD && r == RT - 1 ? min(r * 16 + fi, Pin - 1) : r * 16 + fi;
                a4[r] = *reinterpret_cast<const f32x4*>(&Xrows[row * LDA + 16 * blk + 4 * fk]);
                if (PAD && r == RT - 1 && r * 16 + fi >= Pin) a4[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            } else if (WIDE) a4[r] = xf[r][blk];
            else a4[r] = *reinterpret_cast<const f32x4*>(&Xs[(r * 16 + fi) * LDA + 16 * blk + 4 * fk]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc1[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r][j], bq[j], acc1[r], 0, 0, 0);
    }
    MIX_STAMP(4)                                                // matmul 1 issued (its M fragments have arrived)
    if constexpr (XLDS) __syncthreads();                        // every wave is done with the gathered rows: S may land on them
    // C/D layout (16x16): column = lane & 15, row = (lane >> 4) * 4 + reg
    // ---- LayerNorm over all Pin*64 elements (no affine, biased variance), ReLU --------------------------
    const float nw1 = (float)(Pin * 16);                       // this wave's 16 columns x Pin rows
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += (r * 16 + fk * 4 + e) < Pin ? acc1[r][e] : 0.f;
    s = wave_sum(s);
    const float mw1 = s / nw1;
    float qv = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = acc1[r][e] - mw1;
            qv += (r * 16 + fk * 4 + e) < Pin ? d * d : 0.f;
        }
    qv = wave_sum(qv);
    if constexpr (KSPLIT) {                                     // the first k half
#pragma unroll
        for (int k = 0; k < RT; ++k) {
            const int e = tid + 256 * k;
            *reinterpret_cast<f32x4*>(&Ss[(e >> 4) * lds_s + (e & 15) * 4]) = sreg[k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < RT; ++k) {                          // ... and the request for the second half (lands under LayerNorm 1 and the first four k blocks)
            const int e = min(tid + 256 * k, POUT * nb2 - 1);
            const int r = e / nb2;
            sreg[k] = *reinterpret_cast<const f32x4*>(sg + r * Pin + KH + (e - r * nb2) * 4);
        }
    } else if (WIDE) {
#pragma unroll
        for (int k = 0; k < 2 * RT; ++k) {
            const int i4 = (tid + 256 * k) * 4;
            if (!PAD || i4 < POUT * Pin) *reinterpret_cast<f32x4*>(&Ss[(i4 / Pin) * lds_s + i4 % Pin]) = sreg[k];
        }
    }
    MIX_STAMP(5)                                                // S parked in LDS (its loads have arrived)
    float mean1, rstd1;
    block_mean_rstd(s, qv, nw1, a.eps, red, wave, lane, mean1, rstd1);
    MIX_STAMP(6)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc1[r][e] = fmaxf((acc1[r][e] - mean1) * rstd1, 0.f);      // y1[r*16 + 4 fk + e][cw + fi]
            if (!WIDE) Y1[(r * 16 + fk * 4 + e) * LDB + cw + fi] = acc1[r][e];
        }
    if (!WIDE) __syncthreads();

    // ---- matmul 2: y2[128, 64] = S[128, Pin] @ y1[Pin, 64]; wave w -> columns [16w, 16w+16), 8 row tiles -
    f32x4 acc2[POUT / 16];
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r) acc2[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (WIDE) {
#pragma unroll
        for (int blk = 0; blk < RT; ++blk) {                    // Pin / 16 == RT
            if constexpr (KSPLIT) {
                if (blk == KH / 16) {                           // the second k half replaces the first (same accumulation order as one block)
                    __syncthreads();                            // every wave has read its fragments of the first half
#pragma unroll
                    for (int k = 0; k < RT; ++k) {
                        const int e = tid + 256 * k;
                        const int r = e / nb2;
                        if (e < POUT * nb2) *reinterpret_cast<f32x4*>(&Ss[r * lds_s + (e - r * nb2) * 4]) = sreg[k];
                    }
                    __syncthreads();
                }
            }
            float bq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = acc1[blk][j];   // y1[16 blk + 4 fk + j][cw + fi]: this lane's own accumulators
            f32x4 a4[POUT / 16];
            // k >= Pin (last block, Pin % 16 != 0): the A value is zeroed -- y1's padding rows are finite but not 0 after LayerNorm
            const bool kpad = PAD && blk == RT - 1 && 16 * blk + 4 * fk >= Pin;
            const int kc = kpad ? 0 : 16 * blk + 4 * fk - (KSPLIT && blk >= KH / 16 ? KH : 0);
#pragma unroll
            for (int r = 0; r < POUT / 16; ++r) {
                a4[r] = *reinterpret_cast<const f32x4*>(&Ss[(r * 16 + fi) * lds_s + kc]);
                if (kpad) a4[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < POUT / 16; ++r) acc2[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r][j], bq[j], acc2[r], 0, 0, 0);
        }
    } else {
        for (int k0 = 0; k0 < Pin; k0 += 4) {
            const float b = Y1[(k0 + fk) * LDB + cw + fi];
#pragma unroll
            for (int r = 0; r < POUT / 16; ++r) {
                const float av = Ss[(r * 16 + fi) * lds_s + k0 + fk];
                acc2[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc2[r], 0, 0, 0);
            }
        }
    }
    MIX_STAMP(7)                                                // matmul 2 issued
    // ---- LayerNorm over 128*64 elements, ReLU ------------------------------------------------------------
    const float nw2 = (float)(POUT * 16);
    s = 0.f;
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc2[r][e];
    s = wave_sum(s);
    const float mw2 = s / nw2;
    qv = 0.f;
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = acc2[r][e] - mw2;
            qv += d * d;
        }
    qv = wave_sum(qv);
    float mean2, rstd2;
    block_mean_rstd(s, qv, nw2, a.eps, red, wave, lane, mean2, rstd2);       // (its barriers also fence the Ss / Y1 reads above)
    // ---- transpose through LDS, then one contiguous 32 KiB store (direct 4-row x 64-B stores from the accumulators and
    // staging in two 64-row halves for a smaller LDS footprint both measured no better) ------------------------------
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            Yo[(r * 16 + fk * 4 + e) * LDY + cw + fi] = fmaxf((acc2[r][e] - mean2) * rstd2, 0.f);
    __syncthreads();
    MIX_STAMP(8)                                                // LayerNorm 2 + the transposed tile in LDS
    float* yg = a.y + item * POUT * C;
    const float up = a.out_up;
#pragma unroll
    for (int i = tid; i < POUT * (C / 4); i += 256) {
        const int r = i / (C / 4), c4 = (i % (C / 4)) * 4;
        float4 v = *reinterpret_cast<const float4*>(&Yo[r * LDY + c4]);
        if (up != 0.f) { v.x = f16_pair(v.x * up); v.y = f16_pair(v.y * up); v.z = f16_pair(v.z * up); v.w = f16_pair(v.w * up); }
        *reinterpret_cast<float4*>(yg + r * C + c4) = v;
    }
    MIX_STAMP(9)                                                // stores issued
}

template <int RT, bool WIDE>
int launch_mix_w(const MixArgs& a, hipStream_t s) {
    const int Pin = a.Pin;
    size_t floats = (size_t)(WIDE ? 0 : RT * 16 * LDB + C * LDB) + POUT * (Pin + 4);
    const size_t out_floats = (size_t)POUT * LDY;
    if (floats < out_floats) floats = out_floats;
    const size_t bytes = (floats + 8) * sizeof(float);
    auto k = a.Pin % 16 == 0 || !WIDE ? adaptive_mixing_kernel<RT, WIDE> : adaptive_mixing_kernel<RT, WIDE, 0, float, true>;
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            sbev::set_error("sbev_adaptive_mixing_f32: cannot reserve %zu B of LDS: %s", bytes, hipGetErrorString(e));
            return SBEV_ELAUNCH;
        }
    }
    hipLaunchKernelGGL(k, dim3((unsigned)a.n_items), dim3(256), bytes, s, a);
    return sbev::check_launch("sbev_adaptive_mixing_f32");
}

template <int RT>
int launch_mix(const MixArgs& a, hipStream_t s) {
    static const bool generic = getenv("SBEV_MIX_GENERIC") != nullptr;      // A/B: the LDS-staged path for Pin % 16 != 0 (rounds 1-2)
    return a.Pin % 16 == 0 || !generic ? launch_mix_w<RT, true>(a, s) : launch_mix_w<RT, false>(a, s);
}

template <int RT, int L, typename FT>
int launch_sample_mix(const SampleMixArgs& a, hipStream_t s) {
    const int Pin = a.Pin;
    // S, then the gathered x behind it; Pin > 64: S in two k halves of 64 + 4 columns, the gathered x ON that buffer (Pin <= 128 rows fit)
    size_t floats = Pin > 64 ? (size_t)POUT * (64 + 4) : (size_t)POUT * (Pin + 4) + (size_t)Pin * LDA;
    const size_t out_floats = (size_t)POUT * LDY;
    if (floats < out_floats) floats = out_floats;
    size_t bytes = (floats + 8) * sizeof(float);
#ifdef SBEV_EXP_MIX_PAD          // A/B (tools/exp/r5_run1.sh): unused LDS per workgroup = fewer workgroups per CU (34.8 KB: 4; + 10 KB: 3; + 25 KB: 2)
    static const int exp_pad = getenv("SBEV_EXP_MIX_PAD") ? atoi(getenv("SBEV_EXP_MIX_PAD")) : 0;
    bytes += (size_t)exp_pad;
#endif
    auto k = adaptive_mixing_kernel<RT, true, L, FT, true>;
    if constexpr (RT <= 4) {
        if (a.Pin % 16 == 0 && a.s.P == 4) k = adaptive_mixing_kernel<RT, true, L, FT>;      // the tuned instantiation (4 points per frame, whole row tiles)
    }
    hipEvent_t e0, e1;
    const bool prof = sbev::profile_begin(s, &e0, &e1, 3);
    hipLaunchKernelGGL(k, dim3(a.order ? 8u * (unsigned)a.order_per : (unsigned)a.n_items), dim3(256), bytes, s, a);
    if (prof) sbev::profile_end(s, e0, e1, 3);
    return sbev::check_launch("sbev_sample_mix_f32");
}

template <int L, typename FT>
int launch_sample_mix_rt(const SampleMixArgs& a, hipStream_t s) {
    switch ((a.Pin + 15) / 16) {
        case 1: return launch_sample_mix<1, L, FT>(a, s);
        case 2: return launch_sample_mix<2, L, FT>(a, s);
        case 3: return launch_sample_mix<3, L, FT>(a, s);
        case 4: return launch_sample_mix<4, L, FT>(a, s);
        default: return launch_sample_mix<8, L, FT>(a, s);
    }
}

}  // namespace

#ifdef SBEV_MIX_TRACE
extern "C" int sbev_debug_mix_trace_read(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mix_trace), sizeof(long long) * (size_t)n);
}
#endif

static int adaptive_mixing_impl(const float* x, const float* params, float* y, int64_t BQ, int G, int Pin, int Cg, int Pout, float eps,
                                float out_up, sbev_stream_t stream) {
    SBEV_REQUIRE(BQ >= 0 && G >= 1, "sbev_adaptive_mixing_f32: bad sizes");
    SBEV_REQUIRE(Cg == C && Pout == POUT, "sbev_adaptive_mixing_f32: built for C=64 channels per group and 128 out points (got %d, %d)", Cg, Pout);
    SBEV_REQUIRE(Pin >= 4 && Pin % 4 == 0 && Pin <= 120, "sbev_adaptive_mixing_f32: in_points=%d must be a multiple of 4 in 4..120 (LDS budget)", Pin);
    if (BQ == 0) return SBEV_OK;
    SBEV_REQUIRE(x && params && y, "sbev_adaptive_mixing_f32: null pointer");
    SBEV_REQUIRE(BQ * G <= 0x7fffffffLL, "sbev_adaptive_mixing_f32: too many items");
    MixArgs a{x, params, y, BQ * G, Pin, eps, out_up};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch ((Pin + 15) / 16) {
        case 1: return launch_mix<1>(a, s);
        case 2: return launch_mix<2>(a, s);
        case 3: return launch_mix<3>(a, s);
        case 4: return launch_mix<4>(a, s);
        case 5: return launch_mix<5>(a, s);
        case 6: return launch_mix<6>(a, s);
        case 7: return launch_mix<7>(a, s);
        default: return launch_mix<8>(a, s);
    }
}

extern "C" int sbev_adaptive_mixing_f32(const float* x, const float* params, float* y,
                                        int64_t BQ, int G, int Pin, int Cg, int Pout, float eps,
                                        sbev_stream_t stream) {
    return adaptive_mixing_impl(x, params, y, BQ, G, Pin, Cg, Pout, eps, 0.f, stream);
}

// the same, y as (fp16 hi, fp16 lo) pairs of y 2^up_log2 in the fp32 slots: the fp16 out-projection's operand (sbev_linear_splitk_f16s
// with x_is_pairs), split here instead of inside the GEMM.  |y| 2^up_log2 must stay below 65504 (sbev_decoder_mixed_up_log2)
extern "C" int sbev_adaptive_mixing_pairs_f16(const float* x, const float* params, void* y,
                                              int64_t BQ, int G, int Pin, int Cg, int Pout, float eps, int up_log2,
                                              sbev_stream_t stream) {
    SBEV_REQUIRE(up_log2 >= -100 && up_log2 <= 100, "sbev_adaptive_mixing_pairs_f16: up_log2=%d", up_log2);
    return adaptive_mixing_impl(x, params, static_cast<float*>(y), BQ, G, Pin, Cg, Pout, eps, ldexpf(1.f, up_log2), stream);
}

extern "C" int sbev_sample_mix_supported(int L, int C, int P, int T, int gdiv, int G) {
    // T * P in 4 .. 64 (row tiles 1 .. 4) or 113 .. 120 (8 row tiles: the 15-frame, 8-point configuration); P = 4 or 8 points per chunked frame
    const int pin = T * P;
    return (L == 4 || L == 5) && C == 64 && (P == 4 || P == 8) && gdiv == G && T >= 1 && (pin <= 64 || (pin > 112 && pin <= 120));
}

// the fused kernel gathers through buffer loads only: every level's (sample-batch) slab must stay below 2 GiB (sbev_msmv_fwd has a
// 64-bit path for larger ones).  hw = {H0, W0, H1, W1, ...}, strides in elements.
extern "C" int sbev_sample_mix_slabs_ok(const int32_t* hw, int L, int feat_dtype, int N, int Cg, const int64_t* stride_v, int64_t stride_px) {
    if (!hw || !stride_v || L < 1 || L > SBEV_MAX_LEVELS) return 0;
    for (int l = 0; l < L; ++l)
        if (!msmv_slab_fits_buffer(N, hw[2 * l], hw[2 * l + 1], stride_v[l], stride_px, Cg, feat_dtype == SBEV_F32 ? 4 : 2)) return 0;
    return 1;
}

static int sample_mix_impl(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                           int64_t B, int N, int Q, int T, int G, int P, int Cg,
                           const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                           const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                           const float* params, float* y, int Pout, float eps, float out_up, const int32_t* order, sbev_stream_t stream) {
    SBEV_REQUIRE(feats && hw && stride_bo && stride_v, "sbev_sample_mix_f32: null descriptor array");
    SBEV_REQUIRE(sbev_sample_mix_supported(L, Cg, P, T, G, G), "sbev_sample_mix_f32: needs L in {4,5}, C = 64, P in {4,8}, T*P in 4..64 or 116..120 (got L=%d C=%d P=%d T=%d)", L, Cg, P, T);
    SBEV_REQUIRE(Pout == POUT, "sbev_sample_mix_f32: built for 128 out points");
    SBEV_REQUIRE(B >= 0 && Q >= 0 && N >= 1 && G >= 1, "sbev_sample_mix_f32: bad sizes");
    SBEV_REQUIRE(feat_dtype == SBEV_F32 || feat_dtype == SBEV_BF16 || feat_dtype == SBEV_F16, "sbev_sample_mix_f32: feat_dtype %d", feat_dtype);
    SBEV_REQUIRE(stride_px % 4 == 0 && stride_g % 4 == 0, "sbev_sample_mix_f32: pixel/group strides must be multiples of 4 elements");
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(loc && weights && params && y, "sbev_sample_mix_f32: null pointer");
    SBEV_REQUIRE(B * Q * G <= 0x7fffffffLL && B * (int64_t)T * G * Q <= 0x7fffffffLL, "sbev_sample_mix_f32: too many items");
    SampleMixArgs a{};
    a.params = params; a.y = y; a.n_items = B * Q * G; a.Pin = T * P; a.eps = eps; a.out_up = out_up;
    a.order = order;
    a.order_per = (int)((a.n_items + 7) / 8);
    SBEV_REQUIRE(!order || (((uintptr_t)order) & 3) == 0, "sbev_sample_mix_f32: order must be 4-byte aligned");
    MsmvArgs& m = a.s;
    for (int l = 0; l < L; ++l) {
        SBEV_REQUIRE(feats[l] != nullptr && hw[2 * l] >= 1 && hw[2 * l + 1] >= 1, "sbev_sample_mix_f32: level %d", l);
        SBEV_REQUIRE(stride_bo[l] % 4 == 0 && stride_v[l] % 4 == 0 && stride_v[l] >= 0 && stride_px >= 0 &&
                         msmv_slab_fits_buffer(N, hw[2 * l], hw[2 * l + 1], stride_v[l], stride_px, Cg, feat_dtype == SBEV_F32 ? 4 : 2),
                     "sbev_sample_mix_f32: level %d strides (multiples of 4; one (sample-batch) slab must stay below 2 GiB: the taps are 31-bit "
                     "buffer offsets -- sbev_sample_mix_slabs_ok; use sbev_msmv_fwd + sbev_adaptive_mixing_f32 otherwise)", l);
        m.feat[l] = feats[l];
        m.H[l] = hw[2 * l]; m.W[l] = hw[2 * l + 1];
        m.stride_bo[l] = stride_bo[l]; m.stride_v[l] = stride_v[l];
    }
    m.stride_g = stride_g; m.stride_px = stride_px;
    m.loc = loc; m.w = weights; m.out = nullptr;
    m.n_waves = B * T * G * Q;
    m.N = N; m.C = Cg; m.Q = Q; m.P = P; m.gdiv = G; m.T = T; m.G = G;
    if (frame_slots) {
        SBEV_REQUIRE(T <= SBEV_MAX_FRAMES && n_slots >= T, "sbev_sample_mix_f32: ring needs T <= %d, n_slots >= T", SBEV_MAX_FRAMES);
        m.ring_T = T; m.n_slots = n_slots;
        for (int t = 0; t < T; ++t) {
            SBEV_REQUIRE(frame_slots[t] >= 0 && frame_slots[t] < n_slots, "sbev_sample_mix_f32: frame_slots[%d] out of range", t);
            m.slots[t] = frame_slots[t];
        }
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (feat_dtype == SBEV_F32) return L == 4 ? launch_sample_mix_rt<4, float>(a, s) : launch_sample_mix_rt<5, float>(a, s);
    if (feat_dtype == SBEV_F16) return L == 4 ? launch_sample_mix_rt<4, _Float16>(a, s) : launch_sample_mix_rt<5, _Float16>(a, s);
    return L == 4 ? launch_sample_mix_rt<4, unsigned short>(a, s) : launch_sample_mix_rt<5, unsigned short>(a, s);
}

extern "C" int sbev_sample_mix_f32(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                                   int64_t B, int N, int Q, int T, int G, int P, int Cg,
                                   const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                                   const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                                   const float* params, float* y, int Pout, float eps, sbev_stream_t stream) {
    return sample_mix_impl(feats, hw, L, feat_dtype, B, N, Q, T, G, P, Cg, stride_bo, stride_g, stride_v, stride_px, loc, weights, frame_slots, n_slots,
                           params, y, Pout, eps, 0.f, nullptr, stream);
}

// the same launch with its workgroups in the order of sbev_query_order (order [B*Q]: a permutation of the rows b*Q + q, each sample's
// rows contiguous): results are bit-identical, only WHERE and WHEN an item runs changes (see SampleMixArgs::order); null = launch order
extern "C" int sbev_sample_mix_f32_ordered(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                                           int64_t B, int N, int Q, int T, int G, int P, int Cg,
                                           const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                                           const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                                           const float* params, float* y, int Pout, float eps, const int32_t* order, sbev_stream_t stream) {
    return sample_mix_impl(feats, hw, L, feat_dtype, B, N, Q, T, G, P, Cg, stride_bo, stride_g, stride_v, stride_px, loc, weights, frame_slots, n_slots,
                           params, y, Pout, eps, 0.f, order, stream);
}

// the same, y as (fp16 hi, fp16 lo) pairs of y 2^up_log2 (see sbev_adaptive_mixing_pairs_f16)
extern "C" int sbev_sample_mix_pairs_f16(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                                         int64_t B, int N, int Q, int T, int G, int P, int Cg,
                                         const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                                         const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                                         const float* params, void* y, int Pout, float eps, int up_log2, sbev_stream_t stream) {
    SBEV_REQUIRE(up_log2 >= -100 && up_log2 <= 100, "sbev_sample_mix_pairs_f16: up_log2=%d", up_log2);
    return sample_mix_impl(feats, hw, L, feat_dtype, B, N, Q, T, G, P, Cg, stride_bo, stride_g, stride_v, stride_px, loc, weights, frame_slots, n_slots,
                           params, static_cast<float*>(y), Pout, eps, ldexpf(1.f, up_log2), nullptr, stream);
}

extern "C" int sbev_sample_mix_pairs_f16_ordered(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                                                 int64_t B, int N, int Q, int T, int G, int P, int Cg,
                                                 const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                                                 const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                                                 const float* params, void* y, int Pout, float eps, int up_log2, const int32_t* order,
                                                 sbev_stream_t stream) {
    SBEV_REQUIRE(up_log2 >= -100 && up_log2 <= 100, "sbev_sample_mix_pairs_f16: up_log2=%d", up_log2);
    return sample_mix_impl(feats, hw, L, feat_dtype, B, N, Q, T, G, P, Cg, stride_bo, stride_g, stride_v, stride_px, loc, weights, frame_slots, n_slots,
                           params, static_cast<float*>(y), Pout, eps, ldexpf(1.f, up_log2), order, stream);
}
