// Split-operand Linears for the two 15-GFLOP GEMMs of adaptive mixing on the 16-bit matrix core (round 3): fp32 operands as two or
// three 16-bit images, the image products that matter, fp32 accumulation.  MODE (template; API code `nimg` / `nprod + 1`):
//   f16x3   (code 4, the decoder's default, DESIGN_HISTORY.md section 9.7): x 2^e = hi + lo, two RNE fp16 images of the operand scaled by a
//           power of two per row / tensor (11 + 11 significand bits + lo's sign: the fp32 value to <= 2^-23), products hl + lh + hh
//           on v_mfma_f32_32x32x16_f16; the dropped lo x lo is <= 2^-24 |a b|.  Max and rms error against fp64 BELOW the exact
//           f32-MFMA kernels' at both shapes, also on 12-binade inputs (tests/test_gpu_bf16s.py).   f16x4 (code 5): + lo x lo.
//   bf16x6  (code 3): x = hi + mid + lo, three RNE bf16 images (8 + 8 + 8 bits = an exact split), the SIX products of weight >= 2^-16
//           (hh, hm, mh, mm, hl, lh) on v_mfma_f32_32x32x16_bf16; dropped ml, lm, ll <= 2^-23 |a b|.  fp32-class too, twice the matrix work.
//   bf16x3s (code 2): hi + lo bf16, three products (2^-16 class), the opt-in fast mode of rounds 1-2 on the same kernels.
// Replaces nothing in the reference (torch.nn.Linear, models/sparsebev_transformer.py:358,378); same semantics as
// sbev_linear_f32 / sbev_linear_splitk_f32.
//
// Why these kernels and not gemm_bf16x3.hip's: a 16-byte-per-lane VGPR write-back (global load or ds_read_b128) costs the issuing
// wave ~85 matrix-pipe cycles when it is alone on its SIMD and 23-38 with a partner wave (DESIGN_HISTORY.md section 4), so the
// one-wave-per-SIMD register-stationary strips that win for 32-cycle-per-16x16x4 f32 MFMAs are load-issue-bound at 16-bit rates
// (72 us for 45 GFLOP = 25 % of peak).  Here every workgroup is 8 waves = two per SIMD running as two PHASE GROUPS one barrier phase
// apart (one fetches while the other multiplies), operands are pre-ordered as whole 1-KiB MFMA fragments that an LDS-DMA
// instruction copies verbatim and a ds_read_b128 at lane x 16 reads back conflict-free.
//
//   generator  Y[M, N] = X[M, K] W[N, K]^T + b  (K = 256, N = 32768 ...): gemm_bf16s_gen3_kernel -- persistent workgroups, one
//              256 x 256 (or 128 x 256) tile at a time, 16-k stages in a 3 / 4-deep LDS-DMA ring; two images: epilogue through
//              per-wave 32 x 32 LDS transpose patches (whole 128-byte row segments per store).
//   out-proj   slabs[S, M, 256] = X[M, K] W[256, K]^T over S K-chunks (K = 32768), summed by the row-chain tail / sbev_splitk_reduce:
//              gemm_bf16s_out3_kernel -- 64 rows x 256 columns x one chunk, X (fp32) split inside the kernel (bf16 modes; fp16 with a
//              caller-supplied power of two);  gemm_bf16s_out4_kernel (fp16 modes, the decoder's path) -- <= 128 rows x 256 columns
//              x one chunk, X handed over as (fp16 hi, fp16 lo) PAIRS by the mixing kernel's epilogue (sbev_*_pairs_f16), W through
//              a wave-private LDS-DMA ring, the two K halves folded through an [m][n] image of the tile in LDS.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "sbev_common.hpp"

// ---- ablation switches (tools/build_variant.sh; never defined in the product build) -------------------------------------------
#ifdef SBEV_EXP_NOSTORE
#define SBEV_EXP_STORE_COND && a.M < 0
#else
#define SBEV_EXP_STORE_COND
#endif
#ifdef SBEV_EXP_NOMFMA          // keep the fragment reads alive, drop the matrix work
#define SBEV_MFMA(A, B, C) ([&]() { asm volatile("" ::"v"(A), "v"(B)); return C; }())
#else
#define SBEV_MFMA(A, B, C) mfma16<PR::F16>(A, B, C)
#endif

#ifdef SBEV_EXP_TRACE            // per-phase shader-clock stamps of waves 0 and 4 of workgroup 0 (tools/exp/trace_bf16s.py)
__device__ unsigned long long g_sbev_trace[2][512][8];
#define SBEV_TRACE(G_, SLOT_)                                                                               \
    if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && (G_) < 512) g_sbev_trace[wave >> 2][G_][SLOT_] = __builtin_readcyclecounter();
extern "C" int sbev_debug_trace_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sbev_trace), sizeof(g_sbev_trace));
}
#else
#define SBEV_TRACE(G_, SLOT_)
#endif
// Per workgroup: wall clock (s_memrealtime, 100 MHz) and shader clock (s_memtime) at its first and last instruction -- two stamps per
// workgroup, nothing inside the loops.  cycles / wall time = the clock the kernel actually RAN at (tools/gemm_clock.py writes
// profiles/r5_gemm_clock.json from it; -DSBEV_EXP_WGTIME alone, or with the phase stamps of -DSBEV_EXP_TRACE).  One table per kernel
// (KIND_: 0 tiled generator, 1 weight-stationary generator, 2 out-projection 256-row tiles, 3 out-projection 128-row tiles), the LAST
// launch of each kind stays.  Never in the product build.
#if defined(SBEV_EXP_TRACE) || defined(SBEV_EXP_WGTIME)
__device__ unsigned long long g_sbev_wgtime[4][1024][4];
#define SBEV_WGTIME(KIND_, I_)                                                                              \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                            \
        g_sbev_wgtime[KIND_][blockIdx.x][2 * (I_)] = __builtin_amdgcn_s_memrealtime();                      \
        g_sbev_wgtime[KIND_][blockIdx.x][2 * (I_) + 1] = __builtin_readcyclecounter();                      \
    }
extern "C" int sbev_debug_wgtime_read(unsigned long long* out, int kind) {      // out: [1024][4]
    if (kind < 0 || kind > 3) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sbev_wgtime), sizeof(g_sbev_wgtime[0]), sizeof(g_sbev_wgtime[0]) * (size_t)kind);
}
extern "C" int sbev_debug_wgtime_clear(void) {
    static const unsigned long long zero[1024][4] = {};
    for (int k = 0; k < 4; ++k)
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sbev_wgtime), zero, sizeof(zero), sizeof(zero) * (size_t)k) != hipSuccess) return -1;
    return 0;
}
#else
#define SBEV_WGTIME(KIND_, I_)
#endif

namespace {

#include "lazy_relayout.hpp"      // the generator runs the on-demand relayout's scan in its prologue (gemm_f16s_gen_ws_kernel)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- the split ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
// 8 floats -> NIMG x (8 bf16): image i = RNE_bf16(remainder), remainder -= image (exact: the difference of an fp32 value and its
// bf16 rounding is representable).  hi + mid + lo reproduces every finite normal fp32 value bit for bit.
template <int NIMG>
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, u32x4* out) {
    float r[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int img = 0; img < NIMG; ++img) {
        unsigned p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = pack_bf16(r[2 * i], r[2 * i + 1]);
        out[img] = (u32x4){p[0], p[1], p[2], p[3]};
        if (img + 1 < NIMG) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[2 * i] -= __uint_as_float(p[i] << 16);
                r[2 * i + 1] -= __uint_as_float(p[i] & 0xffff0000u);
            }
        }
    }
}

// fp16 variant (round 3, second half): x * 2^e = hi + lo with hi = RNE_fp16, lo = RNE_fp16(remainder) -- 11 + 11 significand bits +
// the sign of lo = the fp32 value to <= 2^-23 relative (one bit short of fp32's own 2^-24 half-ulp in the worst case; elements
// within 2^-17 of the scaled maximum -- below that lo is a subnormal with absolute error 2^-25, i.e. <= 2^-40 of the maximum).  2^e (a power of two: exact) brings the maximum of the row / tensor to
// [2^14, 2^15) so that fp16's 5-bit exponent covers 29 binades below it; the output is multiplied by 2^-(ex + ew) exactly.
template <int NIMG>
__device__ __forceinline__ void split8h(const f32x4 a, const f32x4 b, float up, u32x4* out) {
    float r[8] = {a.x * up, a.y * up, a.z * up, a.w * up, b.x * up, b.y * up, b.z * up, b.w * up};
#pragma unroll
    for (int img = 0; img < NIMG; ++img) {
        unsigned p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const _Float16 h0 = (_Float16)r[2 * i], h1 = (_Float16)r[2 * i + 1];
            p[i] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            if (img + 1 < NIMG) { r[2 * i] -= (float)h0; r[2 * i + 1] -= (float)h1; }
        }
        out[img] = (u32x4){p[0], p[1], p[2], p[3]};
    }
}

// MODE: 0 = bf16x3s (2 bf16 images, 3 products), 1 = bf16x6 (3 images, 6 products), 2 = f16x3 (2 fp16 images, 3 products),
// 3 = f16x4 (2 fp16 images, all 4 products).  Products in the order they are accumulated (small terms first inside a k-step; the
// last one is hi x hi): image of X, image of W
//   3 products: (h,l) (l,h) (h,h)     4: (l,l) (h,l) (l,h) (h,h)     6: (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
template <int MODE>
struct Fmt {
    static constexpr bool F16 = MODE >= 2;
    static constexpr int NIMG = MODE == 1 ? 3 : 2;
    static constexpr int N = MODE == 1 ? 6 : MODE == 3 ? 4 : 3;
    __host__ __device__ static constexpr int ia(int p) {
        return MODE == 1 ? (p == 1 ? 2 : (p == 2 || p == 4) ? 1 : 0) : MODE == 3 ? ((p == 0 || p == 2) ? 1 : 0) : (p == 1 ? 1 : 0);
    }
    __host__ __device__ static constexpr int ib(int p) {
        return MODE == 1 ? (p == 0 ? 2 : (p == 2 || p == 3) ? 1 : 0) : MODE == 3 ? ((p == 0 || p == 1) ? 1 : 0) : (p == 0 ? 1 : 0);
    }
};
__host__ __device__ constexpr int mode_of(int nimg) { return nimg == 3 ? 1 : nimg == 2 ? 0 : nimg == 4 ? 2 : 3; }     // API code -> MODE
__host__ __device__ constexpr bool mode_ok(int nimg) { return nimg >= 2 && nimg <= 5; }

template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
#ifdef SBEV_EXP_F16_AS_BF16          // timing experiment only (wrong numbers): does the fp16 multiplier array cost clock / power?
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int MODE>
__device__ __forceinline__ void split8m(const f32x4 a, const f32x4 b, float up, u32x4* out) {
    if constexpr (Fmt<MODE>::F16) split8h<Fmt<MODE>::NIMG>(a, b, up, out);
    else split8<Fmt<MODE>::NIMG>(a, b, out);
}

__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned nb) {
    // workgroup b runs on XCD b % 8: give each XCD a contiguous range of logical ids (bijective for any nb)
    const unsigned full = nb >> 3, rem = nb & 7, x = b & 7;
    return x * full + (x < rem ? x : rem) + (b >> 3);
}

// ---- row-major bf16 planes [NIMG][rows][K] of an fp32 matrix [rows, ldx] --------------------------------------------------------
template <int NIMG>
__global__ void split_rows_kernel(const float* x, long long ldx, unsigned short* out, long long rows, int K) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 8 k
    const int kb = K / 8;
    if (i >= rows * kb) return;
    const long long r = i / kb;
    const int c = (int)(i - r * kb);
    const float* p = x + r * ldx + c * 8;
    u32x4 im[NIMG];
    split8<NIMG>(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), im);
#pragma unroll
    for (int img = 0; img < NIMG; ++img)
        *reinterpret_cast<u32x4*>(out + ((long long)img * rows + r) * K + c * 8) = im[img];
}

// ---- MFMA-ordered fragments [N/32][K/16][NIMG][64 lanes][8 bf16] of W [N, ldw]: lane l holds row 32 nf + (l & 31), k = 16 ks +
// 8 (l >> 5) + 0..7, i.e. exactly its operand of one v_mfma_f32_32x32x16_bf16 -----------------------------------------------------
template <int MODE>
__global__ void pack_frags_kernel(const float* w, long long ldw, unsigned short* out, int N, int K, const float* up, int up_stride) {
    constexpr int NIMG = Fmt<MODE>::NIMG;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KS = K / 16;
    if (i >= (long long)((N + 31) / 32) * KS * 64) return;
    const int lane = (int)(i & 63);
    const long long f = i >> 6;
    const int ks = (int)(f % KS);
    const int nf = (int)(f / KS);
    int row = nf * 32 + (lane & 31);
    row = row < N ? row : N - 1;                     // a ragged last block repeats the last row (its outputs are never stored)
    const float* p = w + (long long)row * ldw + ks * 16 + (lane >> 5) * 8;
    u32x4 im[NIMG];
    split8m<MODE>(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), Fmt<MODE>::F16 ? up[(long long)row * up_stride] : 1.f, im);
#pragma unroll
    for (int img = 0; img < NIMG; ++img)
        *reinterpret_cast<u32x4*>(out + ((f * NIMG + img) * 64 + lane) * 8) = im[img];
}

// ---- fp16 scales: up = 2^e with max |x| * 2^e in [2^14, 2^15), down = 2^-e; per row (stride 1) or for the whole matrix --------------
__device__ __forceinline__ void scale_of(float mx, float* up, float* down) {
    int ex = 0;
    float e = 0.f;
    if (mx > 0.f && mx < __builtin_inff()) { (void)frexpf(mx, &ex); e = (float)(15 - ex); }      // mx = m 2^ex, m in [0.5, 1)
    e = fminf(fmaxf(e, -126.f), 126.f);      // (2^e and 2^-e both normal fp32 numbers)
    *up = exp2f(e);                       // exact: exp2f of an integer
    *down = exp2f(-e);
}
__global__ __launch_bounds__(256) void row_scale_kernel(const float* w, long long ldw, int N, int K, float* up, float* down) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float mx = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + (long long)row * ldw + k);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) scale_of(mx, up + row, down + row);
}
// one power of two for a whole matrix: up to 64 blocks each reduce a slice to a partial maximum, the block that draws the last
// ticket reduces the partials (a maximum: order-free) and writes {2^e, 2^-e}.  (One 1024-thread block for 900 x 256 values: 26 us.)
// The partials and the ticket of a launch live in one of SCALE_SLOTS device slots, handed out round-robin by the host: launches in
// flight on different streams (training beside inference, a multi-threaded host) no longer share them -- a shared ticket mixed
// their partial maxima into a wrong 2^e; a single-block launch needs no slot at all.
constexpr int SCALE_SLOTS = 64;
__device__ float g_scale_part[SCALE_SLOTS][64];
__device__ unsigned g_scale_ticket[SCALE_SLOTS];
__global__ __launch_bounds__(256) void tensor_scale_kernel(const float* x, long long ldx, long long rows, int K, float* updown, int slot) {
    __shared__ float red[4];
    __shared__ int is_last;
    float mx = 0.f;
    const long long n4 = rows * (K / 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long r = i / (K / 4);
        const int c = (int)(i - r * (K / 4));
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c * 4);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (gridDim.x == 1) {                      // one block: its maximum is the tensor's
        if (threadIdx.x == 0) scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), updown, updown + 1);
        return;
    }
    if (threadIdx.x == 0) {
        g_scale_part[slot][blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __threadfence();
        const unsigned t = atomicAdd(&g_scale_ticket[slot], 1u);
        is_last = t == gridDim.x - 1;
        if (is_last) g_scale_ticket[slot] = 0u;
    }
    __syncthreads();
    if (!is_last || threadIdx.x >= 64) return;
    __threadfence();
    float m = threadIdx.x < gridDim.x ? __builtin_nontemporal_load(&g_scale_part[slot][threadIdx.x]) : 0.f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) scale_of(m, updown, updown + 1);
}
inline int next_scale_slot() {
    static std::atomic<unsigned> n{0};
    return (int)(n.fetch_add(1u, std::memory_order_relaxed) % SCALE_SLOTS);
}

// ---- LDS-DMA: 1 KiB per wave-instruction, LDS destination = M0 + 16 lane (lane-linear), source = sbase + voff per lane ----------
// (M0 is written inside the asm block; hipcc uses M0 nowhere else in these kernels -- no LDS instruction needs it on gfx9+,
// there is no dynamic register indexing; checked in the ISA, as for row_chain.hip.)
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_byte) {
    // wave-uniform by construction; readfirstlane makes them SGPRs whatever the divergence analysis concluded
    lds_byte = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte);
    const unsigned long long sb = (unsigned long long)sbase;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32));
    sbase = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 4\n\t"                       // M0 write -> LDS-DMA (1 state) and a readfirstlane'd base -> VMEM (5 states)
        "global_load_lds_dwordx4 %1, %2\n\t"
        :
        : "s"(lds_byte), "v"(voff), "s"(sbase)
        : "memory");
}

// the NIMG images of one (32-row block, k-step): 1 KiB apart in memory AND in the LDS stage -> one M0 / base set-up for all of them
template <int NIMG>
__device__ __forceinline__ void glds16_images(const void* sbase, unsigned voff, unsigned lds_byte) {
    lds_byte = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte);
    const unsigned long long sb = (unsigned long long)sbase;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32));
    sbase = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    if constexpr (NIMG == 3)
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
            :
            : "s"(lds_byte), "v"(voff), "s"(sbase)
            : "memory");
    else
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
            :
            : "s"(lds_byte), "v"(voff), "s"(sbase)
            : "memory");
}

// ==== generator-shaped GEMM ======================================================================================================
struct GenArgs {
    const unsigned short* Xs;    // [ceil(M/32)][K/16][NIMG][64][8] bf16 fragments (sbev_pack_bf16s_frags)
    const unsigned short* Ws;    // [N/32][K/16][NIMG][64][8]
    const float* bias;           // [N] or null
    float* Y;                    // [M, ldy]
    int M, N, K;
    long long ldy;
    int relu;
    int ntm, base, rem;          // row tiles: the first `rem` have base + 1 fragments of 32 rows, the others `base`
    const float* colscale;       // fp16 modes: [N] 2^-ew of W's rows (the output columns); null otherwise
    const float* xscale;         // fp16 modes: {2^ex, 2^-ex} of X (device memory: written by the pack launch before this one)
};

constexpr int G_COLS = 256;                     // columns of a workgroup tile (8 fragments); rows: 128 or 256 (RF)

// ---- generator -----------------------------------------------------------------------------------------------------------------
// History (c2, bf16x6; DESIGN_HISTORY.md section 4), every step measured on the MI355X:
//   v1  one 128 x 256 tile per workgroup, 32-k slabs, two LDS stages filled from ROW-MAJOR bf16 planes, a barrier per slab: 116 us;
//       without its stores 93, without its MFMAs 73, with neither 50 -- the parts add up, nothing overlapped.
//   v2  persistent workgroups, 16-k stages in a 4-deep LDS-DMA ring, fragments read one stage ahead between the MFMAs
//       (sched_group_barrier), epilogue under the next tile: 132 us (!) -- the 32-byte-per-row pieces of a row-major operand made
//       every LDS-DMA instruction touch 32 cache lines.  With both operands pre-packed in MFMA fragment order (whole 1-KiB pieces,
//       lane-linear, conflict-free ds_read_b128 at lane x 16, no swizzle): 106 us.  PMC: matrix pipe busy 49 % of the waves' lifetime,
//       48 % of it stalled on issue, 32 % in waits -- the two waves of a SIMD in lock-step behind the shared barrier.
//   v3  ping-pong (below), 128 x 256 tiles: 108 us.  An in-kernel cycle trace shows why: FETCH = 1150 cycles (LDS-DMA issue 510 +
//       counted wait 360 + fragment reads 200 + barrier 84) against COMPUTE = 850 (24 MFMAs); the L2 -> CU path delivers only
//       ~21 B/clk/CU to this kernel (the out-projection's W stream: ~30), and a 128 x 256 x 16-k stage needs 36 KB per 1536 MFMA
//       cycles = 23 B/clk.  Both kernels were bound by operand delivery, not by latency or issue order.
//   v4  256 x 256 tiles (wave = 128 x 64, 128 accumulator registers): 48 KB per 3072 MFMA cycles = 16 B/clk.
// ---- generator, ping-pong version: the two row-halves of a workgroup run one barrier phase apart ------------------------------------
// PMC of the kernel above (c2, bf16x6, 106 us): the matrix pipe is busy for 49 % of the waves' lifetime, the waves spend 48 % of it
// stalled on instruction issue and 32 % in s_waitcnt / s_barrier -- the two waves of a SIMD (rows 0-63 and 64-127 of the tile: waves
// w and w + 4) run in lock-step behind the shared barrier, want the matrix pipe in the same cycles and leave it idle together.
// Here they alternate: per 16-k stage a wave has a FETCH phase (its share of the LDS-DMA loads three stages ahead, a counted wait,
// the stage's fragments LDS -> registers, a finished tile's stores) and a COMPUTE phase (the stage's MFMAs, nothing else), with a
// barrier after each; the second row-half simply takes one extra barrier before its first phase (and the first one after its
// last), so on every SIMD one wave computes while the other fetches.  A single fragment set per wave is enough (nothing is
// loaded during COMPUTE), which pays for a second accumulator set: the five small products of a k-step (<= 2^-8 of the sum) are
// summed apart from the hi x hi product and added once per tile, so the full-magnitude accumulator is rounded once per k-step
// instead of six times -- measured on the out-projection the error of these kernels is accumulation rounding (it falls as
// 1 / sqrt(K chunks)), not the dropped 2^-24-class products.
// phase boundary of the ping-pong kernels: a barrier no instruction may be scheduled across (hipcc otherwise moves register-only
// MFMAs of the COMPUTE phase over a plain __syncthreads() into the FETCH phase, where they collide with the partner wave's)
__device__ __forceinline__ void phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_n(int n) {          // n wave-uniform, 0 .. 10
    switch (n) {
        case 0: wait_vmcnt_imm<0>(); break;
        case 2: wait_vmcnt_imm<2>(); break;
        case 3: wait_vmcnt_imm<3>(); break;
        case 4: wait_vmcnt_imm<4>(); break;
        case 5: wait_vmcnt_imm<5>(); break;
        case 6: wait_vmcnt_imm<6>(); break;
        case 8: wait_vmcnt_imm<8>(); break;
        case 10: wait_vmcnt_imm<10>(); break;
        case 12: wait_vmcnt_imm<12>(); break;
        default: wait_vmcnt_imm<0>(); break;
    }
}

// RF = row fragments (32 rows) per wave: 2 -> 128 x 256 tiles, 4-deep ring, split accumulators; 4 -> 256 x 256 tiles, 3-deep ring
template <int MODE, int RF>
__global__ __launch_bounds__(512) void gemm_bf16s_gen3_kernel(const GenArgs a) {
    typedef Fmt<MODE> PR;
    constexpr int NIMG = PR::NIMG;
    constexpr bool F16 = PR::F16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];    // the only LDS object: its byte address is 0
    constexpr int AF = 2 * RF;                                       // row fragments of a tile (A side); 8 column fragments (B side)
    constexpr int ST_A = AF * 1024, ST_B = 8 * 1024;                 // bytes of one image of one 16-k stage
    constexpr int STAGE = NIMG * (ST_A + ST_B);
    // LDS ring depth; loads run NST - 1 stages ahead.  256-row tiles: 3 stages of 48 KB with three images; with two images (32 KB) a
    // fourth fits, and the f16x3 trace wants it: its 24 MFMAs per stage no longer cover the landing of a stage issued 2 stages ago
    // (380 cycles of vmcnt wait in every FETCH phase, FETCH 1050 > COMPUTE 830)
    constexpr int NST = RF == 2 ? 4 : 3;        // (a 4th stage for two images x 256 rows fits but measured nothing: 57.2 -> 58.3 us)
    // Two images x 256-row tiles: the LDS the third image would take carries a per-wave 32 x 32 transpose patch for the epilogue.
    // The direct epilogue (a lane owns ONE output column: 128 dword stores of two 128-byte lines per wave and tile) held the tile
    // boundary for 5.3k cycles per phase group -- with the partner group idle at the barrier, 2 x 2 x 5.3k of a 90k-cycle
    // workgroup (f16x3 trace).  Through the patch a wave stores 32 x dwordx4 (8 full 128-byte row segments per instruction).
    constexpr bool EPI_LDS = NIMG == 2 && RF == 4;
    constexpr int EPI_LD = 36;                                       // patch row stride in floats (16-byte aligned rows, 4-bank shift per row)
    constexpr int EPI_BYTES = EPI_LDS ? 8 * 32 * EPI_LD * 4 : 0;
    // Experiment kept as a switch: the LDS-DMA requests of the stage NST - 1 ahead issued in the COMPUTE phase, between the MFMAs (they
    // touch no vector register), instead of in the FETCH phase, because with 24 MFMAs per stage FETCH (issue 380 + reads 190 + wait +
    // barrier) is longer than COMPUTE (816) and sets the pace.
    constexpr bool ISSUE_IN_COMPUTE = false;       // (measured: 61 us instead of 56 -- an LDS-DMA instruction holds the wave's issue ~100 cycles, MFMAs wait behind it)
    constexpr bool SPLIT_ACC = RF == 2;                              // second accumulator set for the small products
    constexpr int NBLK = AF + 8;                                     // 32-row blocks per stage (A rows, then B columns): NIMG KiB each
    constexpr int NLMAX = (NBLK + 7) / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                         // wr: row half = phase group (waves w, w + 4 share a SIMD)
    SBEV_WGTIME(0, 0)
    const int M = a.M, nk = a.K / 16;
    const int lw = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int rt = lw % a.ntm, ct0 = lw / a.ntm, cstep = (int)gridDim.x / a.ntm, nct = a.N / G_COLS;
    const int my_tiles = ct0 < nct ? (nct - ct0 + cstep - 1) / cstep : 0;
    const int G = my_tiles * nk;                                     // stages of this workgroup
    if (G == 0) return;
    const int f0 = rt * a.base + (rt < a.rem ? rt : a.rem);
    const int nf = a.base + (rt < a.rem ? 1 : 0);
    const int m0 = f0 * 32;
    int nfa = nf - RF * wr;                                          // this wave's row fragments: 0 .. RF (fixed for its life)
    nfa = nfa < 0 ? 0 : (nfa > RF ? RF : nfa);
    const int nlb = (NBLK - wave + 7) / 8;                           // this wave's blocks per stage (NIMG loads each)
    const int nl = nlb * NIMG;

    const unsigned voff = (unsigned)lane * 16u;
    const unsigned char* gbase[NLMAX];
    unsigned ldst[NLMAX];
    long long gstep[NLMAX];
    const int nfrag = (M + 31) / 32;
    const long long blkbytes = (long long)nk * NIMG * 1024;          // one 32-row block, all k-steps and images
#pragma unroll
    for (int j = 0; j < NLMAX; ++j) {
        const int q = wave + 8 * j;                                  // block q of the stage: the stage image is [block][image][1 KiB]
        ldst[j] = (unsigned)(q * NIMG * 1024);
        if (q < AF) {
            int fb = f0 + q;
            fb = fb < nfrag ? fb : nfrag - 1;
            gbase[j] = reinterpret_cast<const unsigned char*>(a.Xs) + fb * blkbytes;
            gstep[j] = -(long long)nk * NIMG * 1024;                  // next tile: the same rows again
        } else {
            gbase[j] = reinterpret_cast<const unsigned char*>(a.Ws) + (long long)(ct0 * 8 + (q - AF)) * blkbytes;
            gstep[j] = (long long)(cstep * 8) * blkbytes - (long long)nk * NIMG * 1024;      // to the next column tile of this workgroup
        }
    }
    int lk = 0, lg = 0;                                              // load cursor: k-step in its tile, next stage to issue
    auto issue_next = [&]() {
        if (lg >= G) return;
        const unsigned sb = (unsigned)((lg % NST) * STAGE);
#pragma unroll
        for (int j = 0; j < NLMAX; ++j) {
#ifndef SBEV_EXP_NOGLDS
            if (j < nlb) glds16_images<NIMG>(gbase[j], voff, sb + ldst[j]);
#endif
            gbase[j] += NIMG * 1024;
        }
        ++lg;
        if (++lk == nk) {
            lk = 0;
#pragma unroll
            for (int j = 0; j < NLMAX; ++j) gbase[j] += gstep[j];
        }
    };
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned aoff = (unsigned)(wr * RF) * (NIMG * 1024u) + voff;
    const unsigned boff = (unsigned)(AF + wc * 2) * (NIMG * 1024u) + voff;

    auto run = [&](auto nfa_c) {
        constexpr int NFA = decltype(nfa_c)::value;
        constexpr int NFR = NFA > 0 ? NFA : 1;
        bf16x8 xf[NFR][NIMG], wf[2][NIMG];
        constexpr int NFS = SPLIT_ACC ? NFR : 1;
        f32x16 acc[NFR][2], accs[NFS][2];                            // hi x hi (+ bias) | the small products (SPLIT_ACC)
        // D rows = X rows, D columns = W rows (output columns): a lane holds ONE output column (n = lane & 31 of its fragment) and
        // 16 rows, so a register of the 64 lanes is two full 128-byte lines of the output -- the epilogue stores whole lines
        // (the transposed orientation's 16-byte stores wrote 32-byte pieces of 32 different rows per instruction: 330 cycles
        // each, 19k cycles of a 55k-cycle tile in the cycle trace)
        auto init_acc = [&](int ti) {                                // tile ti of this workgroup: its bias slice waits in LDS
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                // (fp16 modes: the accumulators are in scaled units -- bias and the 2^-(ex + ew) factor are applied by store_tile)
                const float bv = F16 ? 0.f : reinterpret_cast<const float*>(lds + NST * STAGE + EPI_BYTES)[ti * G_COLS + (wc * 2 + fb) * 32 + l31];
#pragma unroll
                for (int fa = 0; fa < NFR; ++fa)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc[fa][fb][e] = bv;
                        if constexpr (SPLIT_ACC) accs[fa][fb][e] = 0.f;
                    }
            }
        };
        auto store_tile = [&](int n0, int tix) {                     // acc already holds hi x hi + small products
            if constexpr (NFA > 0) {
                if constexpr (F16) {                                 // y = acc 2^-(ex + ew[n]) + b[n]: a lane holds two output columns
                    const float* bs = reinterpret_cast<const float*>(lds + NST * STAGE + EPI_BYTES) + tix * G_COLS + wc * 64 + l31;
                    const float* cs = bs + my_tiles * G_COLS;
                    const float b0 = bs[0], b1 = bs[32], c0 = cs[0], c1 = cs[32];
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            acc[fa][0][e] = fmaf(acc[fa][0][e], c0, b0);
                            acc[fa][1][e] = fmaf(acc[fa][1][e], c1, b1);
                        }
                }
                if (a.relu) {                                        // one uniform branch, not one per store
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[fa][fb][e] = fmaxf(acc[fa][fb][e], 0.f);
                }
                if constexpr (EPI_LDS) {
                    float* patch = reinterpret_cast<float*>(lds + NST * STAGE) + wave * (32 * EPI_LD);
                    const int pr = lane >> 3, pc = (lane & 7) * 4;   // read-back: lane -> row pr + 8 i, 4 columns at pc
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa) {
                        const int r0 = m0 + (wr * RF + fa) * 32;
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + 4 * lh) * EPI_LD + l31] = acc[fa][fb][e];
                            __builtin_amdgcn_wave_barrier();         // (wave-private patch: the LDS pipe keeps a wave's accesses in order)
                            f32x4 v[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(patch + (pr + 8 * i) * EPI_LD + pc);
                            __builtin_amdgcn_wave_barrier();
                            float* y = a.Y + (long long)(r0 + pr) * a.ldy + n0 + wc * 64 + fb * 32 + pc;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (r0 + pr + 8 * i < M SBEV_EXP_STORE_COND) *reinterpret_cast<f32x4*>(y + (long long)(8 * i) * a.ldy) = v[i];
                        }
                    }
                    return;
                }
#pragma unroll
                for (int fa = 0; fa < NFA; ++fa) {
                    const int r0 = m0 + (wr * RF + fa) * 32;         // register e: row r0 + (e & 3) + 8 (e >> 2) + 4 lh
                    float* y = a.Y + (long long)(r0 + 4 * lh) * a.ldy + n0 + wc * 64 + l31;
                    // a running row pointer, opaque to the optimiser (it otherwise keeps 64 row addresses per tile in spilled SGPRs)
                    if (r0 + 32 <= M SBEV_EXP_STORE_COND) {          // whole fragment inside the matrix: no per-lane masks
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            y[0] = acc[fa][0][e];
                            y[32] = acc[fa][1][e];
                            y += (e & 3) == 3 ? 5 * a.ldy : a.ldy;
                            asm volatile("" : "+v"(y));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            if (r0 + 4 * lh + (e & 3) + 8 * (e >> 2) < M SBEV_EXP_STORE_COND) {
                                y[0] = acc[fa][0][e];
                                y[32] = acc[fa][1][e];
                            }
                            y += (e & 3) == 3 ? 5 * a.ldy : a.ldy;
                            asm volatile("" : "+v"(y));
                        }
                    }
                }
            }
        };
        int cn0 = ct0 * G_COLS, ck = 0, ti = 0;
        bool pending = false;                                        // a finished tile (columns cn0) waits for its stores
        // the bias slices of this workgroup's column tiles -> LDS (behind the stage ring), once: a global load per tile would
        // make hipcc wait vmcnt(0) in the middle of the LDS-DMA pipeline
        for (int i = tid; i < my_tiles * G_COLS; i += 512) {
            const int t = i / G_COLS, c = i - t * G_COLS;
            reinterpret_cast<float*>(lds + NST * STAGE + EPI_BYTES)[i] = a.bias ? a.bias[(ct0 + t * cstep) * G_COLS + c] : 0.f;
            if constexpr (F16)        // behind the bias slices: the columns' output scales 2^-ew[n] 2^-ex
                reinterpret_cast<float*>(lds + NST * STAGE + EPI_BYTES)[my_tiles * G_COLS + i] = a.colscale[(ct0 + t * cstep) * G_COLS + c] * a.xscale[1];
        }
        __syncthreads();
        init_acc(0);
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) issue_next();
        wait_vmcnt_imm<0>();
        __syncthreads();
        if (wr == 1) phase_barrier();                                // the second row-half runs one phase behind
        for (int g = 0; g < G; ++g) {
            // ---- FETCH(g): loads of stage g + NST - 1; stage g + 1 of this wave landed (only newer loads may be outstanding: vector
            // loads return in order; a store in flight can only make the wait longer); fragments of stage g -> registers
            SBEV_TRACE(g, 0)
            if constexpr (NFA > 0) {                                 // (stage g landed and was published a phase ago)
                const unsigned char* st = lds + (g % NST) * STAGE;
#pragma unroll
                for (int img = 0; img < NIMG; ++img) {
                    wf[0][img] = *reinterpret_cast<const bf16x8*>(st + boff + img * 1024);
                    wf[1][img] = *reinterpret_cast<const bf16x8*>(st + boff + (NIMG + img) * 1024);
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa) xf[fa][img] = *reinterpret_cast<const bf16x8*>(st + aoff + (fa * NIMG + img) * 1024);
                }
            }
            SBEV_TRACE(g, 1)
            if constexpr (!ISSUE_IN_COMPUTE) issue_next();
            SBEV_TRACE(g, 2)
            {
                const int hi = lg - 1, need = g + 1 < G ? g + 1 : g;
                wait_vmcnt_n(hi > need ? (hi - need) * nl : 0);
            }
            if (pending) {                                           // the previous tile's stores ride in this phase
                store_tile(cn0, ti);
                cn0 += cstep * G_COLS;
                ++ti;
                init_acc(ti < my_tiles ? ti : 0);
                pending = false;
            }
            SBEV_TRACE(g, 3)
            phase_barrier();
            SBEV_TRACE(g, 4)
            // ---- COMPUTE(g): nothing but MFMAs (the partner wave of this SIMD is in its FETCH phase)
            if constexpr (ISSUE_IN_COMPUTE) issue_next();
            if constexpr (NFA > 0) {
#pragma unroll
                for (int p = 0; p < PR::N - 1; ++p)
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb) {
                            if constexpr (SPLIT_ACC) accs[fa][fb] = SBEV_MFMA(xf[fa][PR::ia(p)], wf[fb][PR::ib(p)], accs[fa][fb]);
                            else acc[fa][fb] = SBEV_MFMA(xf[fa][PR::ia(p)], wf[fb][PR::ib(p)], acc[fa][fb]);
                        }
#pragma unroll
                for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb)
                        acc[fa][fb] = SBEV_MFMA(xf[fa][0], wf[fb][0], acc[fa][fb]);
            }
            if (++ck == nk) {
                ck = 0;
                pending = true;
                if constexpr (NFA > 0 && SPLIT_ACC) {
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb) acc[fa][fb] += accs[fa][fb];
                }
            }
            SBEV_TRACE(g, 5)
            phase_barrier();
            SBEV_TRACE(g, 6)
        }
        if (wr == 0) phase_barrier();
        if (pending) store_tile(cn0, ti);
        SBEV_WGTIME(0, 1)
    };
    if (nfa == RF) run(std::integral_constant<int, RF>{});
    else if (nfa == RF - 1) run(std::integral_constant<int, RF - 1>{});
    else if (nfa == 0) run(std::integral_constant<int, 0>{});
    else if constexpr (RF == 4) {
        if (nfa == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 1>{});
    }
}

// ---- generator, WEIGHT-STATIONARY version (round 4; K = 256, two images) ------------------------------------------------------------
// What bounded the kernel above (DESIGN_HISTORY.md section 9.7): a 256 x 256 tile streams 32 KB per 16-k stage through the L2 -> CU path
// (~21 B/clk/CU by LDS-DMA) for 1632 cycles of MFMAs, alternately -- the FETCH phase of one wave group is as long as the COMPUTE phase
// of the other, and both operands of every tile come through that path again (X: once per column tile).  With M ~ 10^3 rows and
// K = 256 the weights of 32 output columns are 16 k-steps x 2 images x 4 registers = 128 VGPRs: a wave can HOLD its B operands for
// the whole kernel.  So here a workgroup owns 256 output columns (wave w: columns 32 w .. 32 w + 31, its 32 KB of W fragments loaded
// once into registers) and walks row fragments: the only stream is X -- one 32-row fragment (all K: 32 KB, contiguous in the packed
// operand) per 48 MFMAs of each of the 8 waves, i.e. 32 KB per 3072 matrix-pipe cycles of a SIMD = 10.7 B/clk/CU, half of what the
// tiled kernel needs, read by every wave from a 4-slot LDS ring the workgroup fills by LDS-DMA up to three fragments ahead.  The
// finished fragment's 16 stores ride between the next fragment's MFMAs (a lane owns one output column: a wave-store is two full
// 128-byte lines), through a buffer resource of M ldy 4 bytes, so rows >= M of the last fragment are dropped by the range check and
// need no branch.  Each wave sums a fragment's products in TWO interleaved accumulator chains (a dependent v_mfma_f32_32x32x16 issues
// every ~73 cycles against 32 for independent ones): same image products as gemm_bf16s_gen3_kernel<MODE, 4>, other summation order --
// equal to it to fp32 round-off, closer to fp64 (tests/test_gpu_bf16s.py).
// LDS-DMA ordering (guide: read a staged buffer one barrier after the wait that retires it): see the comment in the kernel.
struct GenWsArgs {
    const unsigned short* Xs;    // [ceil(M/32)][16][2][64][8]
    const unsigned short* Ws;    // [N/32][16][2][64][8]
    const float* bias;           // [N] or null
    float* Y;                    // [M, ldy]
    int M, N;
    long long ldy;
    int relu;
    int nrs, base, rem;          // row splits: the first `rem` own base + 1 fragments, the others `base`
    int ntask;                   // (N / 256) * nrs
    const float* colscale;       // fp16 modes: [N] 2^-ew; null otherwise
    const float* xscale;         // fp16 modes: {2^ex, 2^-ex}
    // on-demand relayout (round 6): layers 1..5 find and move the feature units their sample points marked and no earlier launch of the
    // step moved -- a few hundred 16-KB units at config 2.  As a launch of its own that is 7-20 us of launch + dependent round trips between
    // the attention chain and this GEMM; here every workgroup does its share BEFORE its first request of the GEMM (the only kernel between
    // the marks and the gather that does not touch the features): one round trip for the flags, a wave per found unit.
    int lazy_on, lazy_esize;
    LazyArgs lazy;
};

#ifndef SBEV_WS_NCH
#define SBEV_WS_NCH 2
#endif
constexpr int WS_KS = 16;                        // k-steps of 16: K = 256
constexpr int WS_FRAG = WS_KS * 2 * 1024;        // bytes of one row fragment of X (all K, both images)
constexpr int WS_SLOTS = 4;

// workgroup barrier with LDS-DMA in flight: a bare s_barrier behind this wave's LDS traffic -- __syncthreads()' release fence would
// also wait vmcnt(0) for the compiler-visible Y stores (and with them for the prefetched fragments)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, bool RELU>
__global__ __launch_bounds__(512) void gemm_f16s_gen_ws_kernel(const GenWsArgs a) {
    typedef Fmt<MODE> PR;
    static_assert(PR::NIMG == 2, "two-image modes only: the stationary weights are 128 registers");
    constexpr bool F16 = PR::F16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];    // the only LDS object: [WS_SLOTS][WS_FRAG]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned voff = (unsigned)lane * 16u;
    // Y through a buffer resource: per-lane byte offset of (row 4 lh, column n) in a 32-bit register, rows >= M fall outside num_records
    const unsigned long long yb = (unsigned long long)a.Y;
    __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(yb >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)yb)),
        0, (int)(unsigned)((long long)a.M * a.ldy * 4), 0x00020000);
    const unsigned ldyb = (unsigned)(a.ldy * 4);
    [[maybe_unused]] const int tid = (int)threadIdx.x;
    SBEV_WGTIME(1, 0)
    if (a.lazy_on) {
        // (LDS: 8 wave tiles of 8448 B, then the list of up to 4 x 512 units and its counter -- 76 KB of the 128-KB ring, before any DMA)
        float* wt = reinterpret_cast<float*>(lds);
        unsigned* list = reinterpret_cast<unsigned*>(lds + 8 * (LZ_TS * WLD * 4));
        unsigned* n_list = list + 4 * 512;
        if (a.lazy_esize == 4) lazy_scan_share<float>(a.lazy, blockIdx.x, gridDim.x, 512, wt, list, n_list);
        else lazy_scan_share<unsigned short>(a.lazy, blockIdx.x, gridDim.x, 512, wt, list, n_list);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the counted waits below start from "nothing outstanding")
        __syncthreads();
    }

    // One barrier per fragment, all eight waves in the same phase.  (Tried, measured, not kept: the two waves of a SIMD half a fragment
    // apart -- the second group meeting the barrier in the MIDDLE of its fragment, so that one wave's boundary work lies beside its partner's
    // MFMA stream -- and the LDS-DMA issued by one group only: 49.0 vs 49.3 us at c2, 189 vs 190 at c3.  The kernel is power-capped, not
    // schedule-bound: DESIGN_HISTORY.md section 10.2.)  What the barrier b_i at the end of fragment i orders: before it every wave waits for its own
    // pieces of fragment i + 1 (loads return in order: at most the 4 pieces of fragment i + 2 may be outstanding; stores still in flight only
    // make the wait longer); after it every wave refills slot (i + 3) % 4 = (i - 1) % 4, which everybody has finished reading.  The stores of
    // fragment i - 1 ride in the first half of fragment i: behind the barrier, never in front of the counted wait (hipcc otherwise sinks them
    // there and the wait sits out their write latency).
    {
        for (int task = (int)xcd_contiguous(blockIdx.x, gridDim.x); task < a.ntask; task += (int)gridDim.x) {
            const int ct = task / a.nrs, rs = task - ct * a.nrs;
            const int f0 = rs * a.base + (rs < a.rem ? rs : a.rem);
            const int nf = a.base + (rs < a.rem ? 1 : 0);
            if (nf <= 0) continue;
            const int n = ct * G_COLS + wave * 32 + l31;                   // this lane's output column
            const float bv = a.bias ? a.bias[n] : 0.f;
            const float cv = F16 ? a.colscale[n] * a.xscale[1] : 1.f;
            const unsigned ycol = (unsigned)(4 * lh) * ldyb + (unsigned)n * 4u;
            // ---- X stream: fragment f0 + i -> slot i % 4; wave w copies pieces 4 w .. 4 w + 3 (4 KB, contiguous on both sides)
            const unsigned char* xg = reinterpret_cast<const unsigned char*>(a.Xs) + (long long)f0 * WS_FRAG + wave * 4096;
            auto issue = [&](int i) {                                     // (callers guarantee i < nf)
                {
                    const unsigned long long sb = (unsigned long long)(xg + (long long)i * WS_FRAG);
                    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32));
                    const void* sbase = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
                    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((i % WS_SLOTS) * WS_FRAG + wave * 4096);
#ifndef SBEV_EXP_NOGLDS
                    asm volatile(
                        "s_mov_b32 m0, %0\n\t"
                        "s_nop 4\n\t"
                        "global_load_lds_dwordx4 %1, %2\n\t"
                        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                        :
                        : "s"(dst), "v"(voff), "s"(sbase)
                        : "memory");
#endif
                }
            };
            issue(0);
            if (nf > 1) issue(1);
            if (nf > 2) issue(2);
            __builtin_amdgcn_sched_barrier(0);
            // ---- the wave's weights: 32 columns x K, both images -> 128 registers.  Requested BEHIND the first fragments' DMA: loads
            // return in order, so "at most these 32 outstanding" = the fragments have landed, and the first MFMAs start as soon as
            // k-step 0's weights arrive (the compiler's own counted waits)
            bf16x8 wf[WS_KS][2];
            {
                const unsigned short* wb = a.Ws + ((long long)(ct * 8 + wave) * (WS_KS * 2) * 64 + lane) * 8;
#pragma unroll
                for (int ks = 0; ks < WS_KS; ++ks)
#pragma unroll
                    for (int img = 0; img < 2; ++img) wf[ks][img] = *reinterpret_cast<const bf16x8*>(wb + (ks * 2 + img) * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt_imm<32>();
            lds_barrier();
            f32x16 prev;                                                  // the finished fragment, scaled + biased, waiting for its stores
#pragma unroll
            for (int e = 0; e < 16; ++e) prev[e] = 0.f;
            for (int i = 0; i < nf; ++i) {
                SBEV_TRACE(i, 0)
                const unsigned char* st = lds + (i % WS_SLOTS) * WS_FRAG + voff;
                // fragment i - 1; at i = 0 there is none: bit 31 puts the 16 stores outside the buffer (M ldy 4 < 2^31, host-checked) and
                // the hardware drops them -- no branch in the k loop
                const unsigned yprev = i > 0 ? ycol + (unsigned)((f0 + i - 1) * 32) * ldyb : (ycol | 0x80000000u);
                // NCH accumulator chains (k-step ks -> chain ks % NCH, summed at the end).  A dependent v_mfma_f32_32x32x16 (same accumulator)
                // issues only every ~100 cycles against 32 for independent ones: with ONE chain per wave the two waves of a SIMD kept the
                // matrix pipe ~60 % busy whatever else was taken out of the kernel (r4_run2 / r4_run3: LDS reads, stores, DMA, stagger).
                // The partial sums also round less than one long chain (max error vs fp64 at c2: 1.6e-6 against 2.8e-6).
                constexpr int NCH = SBEV_WS_NCH;
                f32x16 accs[NCH];                                         // (bf16 modes: chain 0 starts from the bias, like the tiled kernel's accumulators)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) accs[c][e] = (F16 || c > 0) ? 0.f : bv;
                // blocks of NCH k-steps: the block's MFMAs go round the NCH chains product by product, so consecutive MFMAs of a wave never
                // share an accumulator; the NEXT block's fragments are requested before them (double-buffered registers)
                constexpr int NBLK = WS_KS / NCH;
                static_assert(WS_KS % NCH == 0 && NBLK % 2 == 0 && 16 % (NBLK / 2) == 0, "NCH divides the k-steps into an even number of blocks");
                constexpr int SPB = 16 / (NBLK / 2);                      // stores per block of the half that carries them
                bf16x8 xr[2][NCH][2];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    xr[0][c][0] = *reinterpret_cast<const bf16x8*>(st + (c * 2) * 1024);
                    xr[0][c][1] = *reinterpret_cast<const bf16x8*>(st + (c * 2 + 1) * 1024);
                }
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
#ifndef SBEV_EXP_NOREAD
                    if (blk + 1 < NBLK) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            xr[(blk + 1) & 1][c][0] = *reinterpret_cast<const bf16x8*>(st + (((blk + 1) * NCH + c) * 2) * 1024);
                            xr[(blk + 1) & 1][c][1] = *reinterpret_cast<const bf16x8*>(st + (((blk + 1) * NCH + c) * 2 + 1) * 1024);
                        }
                    }
#endif
                    if (blk < NBLK / 2) {                                 // SPB stores of the previous fragment per block of the first half
#pragma unroll
                        for (int e = SPB * (blk % (NBLK / 2)); e < SPB * (blk % (NBLK / 2)) + SPB; ++e) {
                            const unsigned ro = (unsigned)((e & 3) + 8 * (e >> 2)) * ldyb;
#ifndef SBEV_EXP_NOSTORE
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[e]), yrs, (int)(yprev + ro), 0, 0);
#endif
                        }
                    }
#ifdef SBEV_EXP_NOREAD
                    const auto& xc = xr[0];
#else
                    const auto& xc = xr[blk & 1];
#endif
#pragma unroll
                    for (int p = 0; p < PR::N; ++p)
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const int ks = blk * NCH + c;
                            if (p < PR::N - 1) accs[c] = SBEV_MFMA(xc[c][PR::ia(p)], wf[ks][PR::ib(p)], accs[c]);
                            else accs[c] = SBEV_MFMA(xc[c][0], wf[ks][0], accs[c]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (blk == NBLK - 1) {
                        // b_i: this wave's pieces of fragment i + 1 have landed (only fragment i + 2's may be outstanding), then published;
                        // slot (i + 3) % 4 is free behind it
                        SBEV_TRACE(i, 1)
                        if (i + 2 < nf) wait_vmcnt_imm<4>();
                        else wait_vmcnt_imm<0>();
                        SBEV_TRACE(i, 2)
                        lds_barrier();
                        SBEV_TRACE(i, 3)
                        if (i + 3 < nf) issue(i + 3);
                        __builtin_amdgcn_sched_barrier(0);
                        SBEV_TRACE(i, 4)
                    }
                }
                f32x16 acc = accs[0];
#pragma unroll
                for (int c = 1; c < NCH; ++c) acc += accs[c];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = F16 ? fmaf(acc[e], cv, bv) : acc[e];
                    prev[e] = RELU ? fmaxf(v, 0.f) : v;
                }
            }
            {                                                             // the last fragment's stores
                const unsigned ylast = ycol + (unsigned)((f0 + nf - 1) * 32) * ldyb;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[e]), yrs, (int)(ylast + (unsigned)((e & 3) + 8 * (e >> 2)) * ldyb), 0, 0);
            }
        }
    }
    SBEV_WGTIME(1, 1)
}

// ==== out-projection-shaped split-K GEMM (N = 256) ================================================================================
struct OutArgs {
    const float* X;              // [M, ldx] fp32
    const unsigned short* Wp;    // [8][K/16][NIMG][64][8] bf16 fragments
    float* P;                    // [S, M, 256] partial slabs
    int M, K;
    long long ldx;
    int nrt, S;                  // row tiles of 64 rows, K chunks
    float xup;                   // fp16 modes: X is multiplied by this power of two before the split (the caller's bound on |X|)
    const float* nscale;         // fp16 modes: [256] 2^-ew[n] / xup, applied to the slab values (exact); null otherwise
    const float* xdev;           // fp16 modes, optional: X's {2^e, 2^-e} in device memory (replaces xup; the slabs are also multiplied by 2^-e)
};

constexpr int O_IMG = 64 * 64;                  // bytes of one image of one half's stage (64 rows x 32 k)

// ---- out-projection, ping-pong version: the two K halves of a workgroup run one barrier phase apart ----------------------------------
// PMC of the kernel above (c2, bf16x6, 100 us): matrix pipe busy 62 % of the waves' lifetime, 55 % of it spent stalled on issue, 27 %
// in waits -- again the two waves of a SIMD (the same 64 columns of the two K halves: waves w and w + 4) in lock-step.  As in the
// generator a slab now has a FETCH phase (fragments of the slab LDS -> registers; split + LDS write of the slab two ahead; the X
// loads four ahead) and a COMPUTE phase (its 48 MFMAs, then the W fragment loads of the next slab, which land during the following
// FETCH), a barrier after each, and the second K half takes one extra barrier up front: on every SIMD one wave computes while the
// other fetches.  The halves never touch each other's LDS ring; they meet only in the final fold.
// XPRE (fp16 modes): X already holds (fp16 hi, fp16 lo) pairs of x 2^e in its 32-bit slots (the mixing kernel's epilogue made them:
// sbev_adaptive_mixing_pairs_f16 / sbev_sample_mix_pairs_f16) -- staging a slab is then 8 byte-permutes per thread instead of the
// ~60 conversion instructions of the split, which the 24 MFMAs of an f16x3 slab no longer hide (trace: COMPUTE 1100 .. 1800 cycles
// for 792 cycles of matrix work)
template <int MODE, bool XPRE = false>
__global__ __launch_bounds__(512) void gemm_bf16s_out3_kernel(const OutArgs a) {
    typedef Fmt<MODE> PR;
    static_assert(!XPRE || PR::F16, "pre-split X is the fp16 pair format");
    constexpr int NIMG = PR::NIMG;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int HSTAGE = NIMG * O_IMG;        // one half's stage
    constexpr int NST = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wc = wave & 3;  // K half of the chunk = phase group, 64-column quarter
    SBEV_WGTIME(2, 0)
    const unsigned logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int chunk = (int)(logical / (unsigned)a.nrt), rt = (int)(logical % (unsigned)a.nrt);
    const int M = a.M;
    const int m0 = rt * 64;
    const int nfa = (M - m0) > 32 ? 2 : 1;      // row fragments of this tile
    const int nslab = a.K / 32;
    const int c0 = (int)((long long)nslab * chunk / a.S), c1 = (int)((long long)nslab * (chunk + 1) / a.S);
    const int n_all = c1 - c0, n0h = (n_all + 1) / 2;
    const int sb = half == 0 ? c0 : c0 + n0h;            // first slab of this half
    const int nh = half == 0 ? n0h : n_all - n0h;        // its slabs (half 0 may have one more)

    const int th = tid & 255;
    const int srow = th >> 2, skq = th & 3;
    int grow = m0 + srow;
    grow = grow < M ? grow : M - 1;
    const float* xp = a.X + (long long)grow * a.ldx + skq * 8;
    const unsigned wofs = (unsigned)(srow * 64 + ((skq ^ ((srow >> 2) & 3)) * 16));
    unsigned char* hst = lds + half * (NST * HSTAGE);    // this half's stage ring
    const int last_slab = nh > 0 ? sb + nh - 1 : c1 - 1;
    auto loadx = [&](int i, f32x4& v0, f32x4& v1) {      // slab i of this half (clamped: a dummy past the end)
        int sl = sb + i;
        sl = sl < last_slab ? sl : last_slab;
#ifdef SBEV_EXP_HOTX
        sl = sl & 7;
#endif
        const float* p = xp + (long long)sl * 32;
        v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));         // streamed once: keep L2 for W
        v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
    };
    const float xup = a.xdev ? a.xdev[0] : a.xup;
    auto stagex = [&](int i, const f32x4 v0, const f32x4 v1) {       // slab i -> ring slot i % 3
        u32x4 im[NIMG];
        if constexpr (XPRE) {
            const u32x4 p = __builtin_bit_cast(u32x4, v0), q = __builtin_bit_cast(u32x4, v1);
            im[0] = (u32x4){__builtin_amdgcn_perm(p.y, p.x, 0x05040100u), __builtin_amdgcn_perm(p.w, p.z, 0x05040100u),
                            __builtin_amdgcn_perm(q.y, q.x, 0x05040100u), __builtin_amdgcn_perm(q.w, q.z, 0x05040100u)};
            im[1] = (u32x4){__builtin_amdgcn_perm(p.y, p.x, 0x07060302u), __builtin_amdgcn_perm(p.w, p.z, 0x07060302u),
                            __builtin_amdgcn_perm(q.y, q.x, 0x07060302u), __builtin_amdgcn_perm(q.w, q.z, 0x07060302u)};
        } else {
            split8m<MODE>(v0, v1, xup, im);
        }
        unsigned char* st = hst + (i % NST) * HSTAGE + wofs;
#pragma unroll
        for (int img = 0; img < NIMG; ++img) *reinterpret_cast<u32x4*>(st + img * O_IMG) = im[img];
    };
    const int KS = a.K / 16;
    const unsigned short* wb0 = a.Wp + ((long long)(2 * wc) * KS * NIMG * 64 + lane) * 8;
    const unsigned short* wb1 = a.Wp + ((long long)(2 * wc + 1) * KS * NIMG * 64 + lane) * 8;
    auto loadw = [&](int i, int j, bf16x8 (&w)[2][NIMG]) {   // k-step j of slab i of this half (clamped)
        int sl = sb + i;
        sl = sl < last_slab ? sl : last_slab;
#ifdef SBEV_EXP_HOTW
        sl = sl & 7;
#endif
        const long long o = (long long)(2 * sl + j) * NIMG * 64 * 8;
#pragma unroll
        for (int img = 0; img < NIMG; ++img) {
            w[0][img] = *reinterpret_cast<const bf16x8*>(wb0 + o + img * 512);
            w[1][img] = *reinterpret_cast<const bf16x8*>(wb1 + o + img * 512);
        }
    };
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned swz = (unsigned)((lane >> 2) & 3);
    const unsigned fo0 = (unsigned)l31 * 64u + (((unsigned)lh) ^ swz) * 16u;
    const unsigned fo1 = (unsigned)l31 * 64u + ((2u + (unsigned)lh) ^ swz) * 16u;

    auto run = [&](auto nfa_c) {
        constexpr int NFA = decltype(nfa_c)::value;
        f32x16 acc[NFA][2];
#pragma unroll
        for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[fa][fb][e] = 0.f;
        f32x4 xa0, xa1, xb0, xb1;                             // X register ring: slab s + 2 waits in a (s even) / b (s odd)
        bf16x8 w[2][2][NIMG], xf[2][NFA][NIMG];               // [k-step][fragment][image]
        // prologue: slabs 0 and 1 staged, 2 and 3 requested, W of slab 0 requested
        loadx(0, xa0, xa1);
        loadx(1, xb0, xb1);
        loadw(0, 0, w[0]);
        loadw(0, 1, w[1]);
        stagex(0, xa0, xa1);
        loadx(2, xa0, xa1);
        stagex(1, xb0, xb1);
        loadx(3, xb0, xb1);
        __syncthreads();
        if (half == 1) phase_barrier();                       // the second K half runs one phase behind
#define SBEV_O3_SLAB(S_, X0, X1)                                                                            \
        {                                                                                                   \
            /* FETCH */                                                                                     \
            {                                                                                               \
                SBEV_TRACE(S_, 0)                                                                           \
                const unsigned char* A = hst + ((S_) % NST) * HSTAGE;                                       \
                _Pragma("unroll") for (int img = 0; img < NIMG; ++img)                                      \
                    _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa) {                                    \
                        xf[0][fa][img] = *reinterpret_cast<const bf16x8*>(A + fo0 + img * O_IMG + fa * 32 * 64); \
                        xf[1][fa][img] = *reinterpret_cast<const bf16x8*>(A + fo1 + img * O_IMG + fa * 32 * 64); \
                    }                                                                                       \
                if ((S_) >= nh) {      /* the unequal last slab: this half has none left and multiplies zeros */ \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
                        _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                  \
                            _Pragma("unroll") for (int img = 0; img < NIMG; ++img)                          \
                                _Pragma("unroll") for (int e = 0; e < 8; ++e) xf[j][fa][img][e] = (__bf16)0.f; \
                }                                                                                           \
                SBEV_TRACE(S_, 1)                                                                           \
                if ((S_) > 0) loadw((S_), 1, w[1]);   /* needed a FETCH + half a COMPUTE from now */        \
                SBEV_TRACE(S_, 2)                                                                           \
                SBEV_TRACE(S_, 3)                                                                           \
            }                                                                                               \
            phase_barrier();                                                                                \
            SBEV_TRACE(S_, 4)                                                                               \
            /* COMPUTE: k-step 0 with the split + LDS write of the slab two ahead and the X loads four ahead riding between its */ \
            /* MFMAs (a 32-cycle MFMA hides ~5 single-issue instructions; in the FETCH phase the same ~60 VALU ops cost 1000     */ \
            /* cycles of the partner's matrix time), then k-step 1 with the next slab's k-step-0 W loads between its MFMAs (the  */ \
            /* registers they overwrite are dead once k-step 0 has issued); the k-step-1 W loads go out in the next FETCH        */ \
            stagex((S_) + 2, X0, X1);                                                                       \
            loadx((S_) + 4, X0, X1);                                                                        \
            _Pragma("unroll") for (int p = 0; p < PR::N; ++p)                                               \
                _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                          \
                    _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                        \
                        acc[fa][fb] = SBEV_MFMA(w[0][fb][PR::ib(p)], xf[0][fa][PR::ia(p)], acc[fa][fb]);    \
            _Pragma("unroll") for (int i = 0; i < PR::N * NFA * 2 - 2; ++i) {                               \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                          \
            }                                                                                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
            __builtin_amdgcn_sched_group_barrier(0x200, NIMG, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                              \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            loadw((S_) + 1, 0, w[0]);                                                                       \
            _Pragma("unroll") for (int p = 0; p < PR::N; ++p)                                               \
                _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                          \
                    _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                        \
                        acc[fa][fb] = SBEV_MFMA(w[1][fb][PR::ib(p)], xf[1][fa][PR::ia(p)], acc[fa][fb]);    \
            _Pragma("unroll") for (int i = 0; i < 2 * NIMG; ++i) {                                          \
                __builtin_amdgcn_sched_group_barrier(0x008, (PR::N * NFA * 2) / (2 * NIMG), 0);             \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                          \
            }                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            SBEV_TRACE(S_, 5)                                                                               \
            phase_barrier();                                                                                \
            SBEV_TRACE(S_, 6)                                                                               \
        }
        int sl = 0;
        for (; sl + 1 < n0h; sl += 2) {
            SBEV_O3_SLAB(sl, xa0, xa1)
            SBEV_O3_SLAB(sl + 1, xb0, xb1)
        }
        if (sl < n0h) SBEV_O3_SLAB(sl, xa0, xa1)
#undef SBEV_O3_SLAB
        if (half == 0) phase_barrier();
        // fold the two K halves (fixed order: bit-reproducible) and write the chunk's slab
        __syncthreads();
        f32x4* fold = reinterpret_cast<f32x4*>(lds) + (wc * 16) * 64 + lane;       // [wc][fa][fb][g][lane] float4
        if (half == 1) {
#pragma unroll
            for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        fold[((fa * 2 + fb) * 4 + g) * 64] = (f32x4){acc[fa][fb][4 * g], acc[fa][fb][4 * g + 1], acc[fa][fb][4 * g + 2], acc[fa][fb][4 * g + 3]};
        }
        __syncthreads();
        if (half == 1) return;
        float* out = a.P + (long long)chunk * M * 256 + wc * 64 + 4 * lh;
#pragma unroll
        for (int fa = 0; fa < NFA; ++fa) {
            const int row = m0 + fa * 32 + l31;
            if (row < M SBEV_EXP_STORE_COND) {
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = fold[((fa * 2 + fb) * 4 + g) * 64];
                        f32x4 v = {acc[fa][fb][4 * g] + o[0], acc[fa][fb][4 * g + 1] + o[1], acc[fa][fb][4 * g + 2] + o[2], acc[fa][fb][4 * g + 3] + o[3]};
                        if constexpr (PR::F16) {      // exact: powers of two
                            v *= *reinterpret_cast<const f32x4*>(a.nscale + wc * 64 + 4 * lh + fb * 32 + 8 * g);
                            if (a.xdev) v *= a.xdev[1];
                        }
                        *reinterpret_cast<f32x4*>(out + (long long)row * 256 + fb * 32 + 8 * g) = v;
                    }
            }
        }
        SBEV_WGTIME(2, 1)
    };
    if (nfa == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
}

// ---- out-projection, 128-row tiles (fp16 modes, pre-split X) ---------------------------------------------------------------------
// Ablation of the kernel above in f16x3 (c2, 66 us): without its MFMAs 63, without MFMAs and stores 58 -- with three products the
// kernel is bound by operand DELIVERY: every 64-row tile streams its K chunk of all 256 W rows from L2, 15 row tiles x 33.5 MB =
// 503 MB (+ 118 MB of X) per launch through ~22 B/clk/CU.  Here a workgroup owns up to 128 rows (4 fragments, balanced like the
// generator's tiles: 900 rows = 5 x 4 + 3 x 3 fragments) x all 256 columns x one K chunk: half the W bytes per MFMA.  A slab is ONE
// 16-k step (24 MFMAs per wave with 4 row fragments, as before with 2 fragments x 2 k-steps), so a wave holds 128 accumulator
// registers + one k-step of fragments (32) + two W sets (32); X arrives as (hi, lo) pairs (the mixing kernel's epilogue), staging is
// 8 byte-permutes per thread.  Same ping-pong of the two K halves, same fixed-order fold: bit-reproducible.
struct Out4Args {
    const unsigned* Xp;          // [M, ldx] (fp16 hi, fp16 lo) pairs
    const unsigned short* Wp;    // [8][K/16][2][64][8] fp16 fragments
    float* P;                    // [S, M, 256] partial slabs
    int M, K;
    long long ldx;
    int ntm, base, rem, S;       // row tiles: the first `rem` have base + 1 fragments of 32 rows, the others `base`; K chunks
    const float* nscale;         // [256] 2^-ew[n] 2^-ex
    // in-launch fold of the S slabs (round 6): the S chunk-workgroups of a row tile are co-resident (one round of workgroups: host-checked),
    // meet at fold_sync[row tile] and each sums a share of the tile's rows over all slabs, in slab order, into `folded` [M, 256]
    unsigned* fold_sync;         // null: off (the consumer sums the slabs)
    float* folded;
    int debug_drop;              // test hook (sbev_debug_out_fold_drop): chunk 1 of row tile 0 never arrives -- its tile must time out, not hang
};

// a chunk-workgroup whose row tile never became complete within the poll bound (the device was shared: not every workgroup of the launch
// was resident): counted here and in the host-mapped word the decoder's fault gate reads (csrc/row_chain.hip installs both pointers)
__device__ unsigned g_fold_timeouts;
__device__ unsigned* g_fold_fault_host;
constexpr unsigned FOLD_POLL_LIMIT = 1u << 20;
constexpr int FOLD_CPOL = 17;    // sc0 | sc1: write-through stores, loads served past this XCD's L2 (MI355X guide, inter-workgroup visibility)

template <int MODE>
__global__ __launch_bounds__(512) void gemm_bf16s_out4_kernel(const Out4Args a) {
    typedef Fmt<MODE> PR;
    static_assert(PR::F16 && PR::NIMG == 2, "pre-split fp16 operands");
    constexpr int NIMG = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int IMG = 128 * 32;               // bytes of one image of one half's stage: 128 rows x 16 k
    constexpr int HSTAGE = NIMG * IMG;
    constexpr int NST = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wc = wave & 3;  // K half of the chunk = phase group, 64-column quarter
    SBEV_WGTIME(3, 0)
    const unsigned logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int chunk = (int)(logical / (unsigned)a.ntm), rt = (int)(logical % (unsigned)a.ntm);
    const int M = a.M;
    const int f0 = rt * a.base + (rt < a.rem ? rt : a.rem);
    const int nfa = a.base + (rt < a.rem ? 1 : 0);               // row fragments of this tile: 1 .. 4
    const int m0 = f0 * 32;
    const int KS = a.K / 16;
    const int c0 = (int)((long long)KS * chunk / a.S), c1 = (int)((long long)KS * (chunk + 1) / a.S);
    const int n_all = c1 - c0, n0h = (n_all + 1) / 2;
    const int sb = half == 0 ? c0 : c0 + n0h;            // first k-step of this half
    const int nh = half == 0 ? n0h : n_all - n0h;        // its k-steps (half 0 may have one more)

    const int th = tid & 255;
    const int srow = th >> 1, skq = th & 1;              // staging: thread -> (row of the tile, 8-k half of the step)
    int grow = m0 + srow;
    grow = grow < M ? grow : M - 1;
    const unsigned* xp = a.Xp + (long long)grow * a.ldx + skq * 8;
    const unsigned wofs = (unsigned)(srow * 32 + skq * 16);
    unsigned char* hst = lds + half * (NST * HSTAGE);    // this half's stage ring
    const int last = nh > 0 ? sb + nh - 1 : c1 - 1;
    auto loadx = [&](int i, u32x4& v0, u32x4& v1) {      // k-step i of this half (clamped: a dummy past the end)
        int sl = sb + i;
        sl = sl < last ? sl : last;
#ifdef SBEV_EXP_HOTX
        sl = sl & 7;
#endif
        const unsigned* p = xp + (long long)sl * 16;
#ifdef SBEV_O4_XNT
        v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 4));
#else
        // (plain loads: a k-step uses 64 of a line's 128 bytes, the next step the rest -- the line has to survive in L2 until then)
        v0 = *reinterpret_cast<const u32x4*>(p);
        v1 = *reinterpret_cast<const u32x4*>(p + 4);
#endif
    };
    auto stagex = [&](int i, const u32x4 p, const u32x4 q) {                     // k-step i -> ring slot i % 3: de-interleave hi | lo
        unsigned char* st = hst + (i % NST) * HSTAGE + wofs;
        *reinterpret_cast<u32x4*>(st) = (u32x4){__builtin_amdgcn_perm(p.y, p.x, 0x05040100u), __builtin_amdgcn_perm(p.w, p.z, 0x05040100u),
                                                __builtin_amdgcn_perm(q.y, q.x, 0x05040100u), __builtin_amdgcn_perm(q.w, q.z, 0x05040100u)};
        *reinterpret_cast<u32x4*>(st + IMG) = (u32x4){__builtin_amdgcn_perm(p.y, p.x, 0x07060302u), __builtin_amdgcn_perm(p.w, p.z, 0x07060302u),
                                                      __builtin_amdgcn_perm(q.y, q.x, 0x07060302u), __builtin_amdgcn_perm(q.w, q.z, 0x07060302u)};
    };
    // W: this wave's two column fragments of a k-step (2 x 2 KiB, hi | lo images adjacent) travel global -> LDS by LDS-DMA into a
    // wave-private 3-slot ring, requested in the FETCH phase two steps ahead, and are read into registers in the step's own FETCH
    // phase: no vector register is in flight for W and the COMPUTE phase is nothing but MFMAs.  (W requests between the MFMAs cost
    // ~200 cycles per phase -- every VMEM issue delays the next MFMA; requested at the start of the FETCH phase straight into
    // registers their L2 latency did not fit the phase: 69 us instead of 62.)
    constexpr int WSLOT = 2 * NIMG * 1024;               // one step of one wave: 2 column fragments x 2 images
    constexpr int WRING0 = 2 * NST * HSTAGE;             // behind the two halves' X stages
    const unsigned wring = (unsigned)(WRING0 + wave * (NST * WSLOT));
    const unsigned char* wg0 = reinterpret_cast<const unsigned char*>(a.Wp) + (long long)(2 * wc) * KS * (NIMG * 1024);
    const unsigned voff = (unsigned)lane * 16u;
    auto issue_w = [&](int i) {                          // k-step i of this half (clamped) -> ring slot i % 3
        int sl = sb + i;
        sl = sl < last ? sl : last;
#ifdef SBEV_EXP_HOTW
        sl = sl & 7;
#endif
        const unsigned dst = wring + (unsigned)((i % NST) * WSLOT);
        glds16_images<NIMG>(wg0 + (long long)sl * (NIMG * 1024), voff, dst);
        glds16_images<NIMG>(wg0 + ((long long)KS + sl) * (NIMG * 1024), voff, dst + NIMG * 1024);
    };
    auto readw = [&](int i, bf16x8 (&w)[2][NIMG]) {
        const unsigned char* src = lds + wring + (i % NST) * WSLOT + voff;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int img = 0; img < NIMG; ++img) w[fb][img] = *reinterpret_cast<const bf16x8*>(src + (fb * NIMG + img) * 1024);
    };
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned fo = (unsigned)l31 * 32u + (unsigned)lh * 16u;       // a fragment = 1 KiB of the image, read as one b128 per lane

    auto run = [&](auto nfa_c) {
        constexpr int NFA = decltype(nfa_c)::value;
        f32x16 acc[NFA][2];
#pragma unroll
        for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[fa][fb][e] = 0.f;
        u32x4 xa0, xa1, xb0, xb1;                             // X register ring: k-step s + 2 waits in a (s even) / b (s odd)
        bf16x8 wa[2][NIMG], xf[NFA][NIMG];                    // one W set: the next step's request goes out behind this step's last MFMA
                                                              // and lands during the partner's COMPUTE phase (a second set spilled)
        // prologue: k-steps 0 and 1 staged, X of 2 and 3 and W of 0 and 1 requested -- and landed: the counted waits below then
        // start from "nothing outstanding" (per step a wave issues 4 W requests, then 2 X requests)
        loadx(0, xa0, xa1);
        loadx(1, xb0, xb1);
        issue_w(0);
        issue_w(1);
        stagex(0, xa0, xa1);
        loadx(2, xa0, xa1);
        stagex(1, xb0, xb1);
        loadx(3, xb0, xb1);
        wait_vmcnt_imm<0>();
        __syncthreads();
        if (half == 1) phase_barrier();                       // the second K half runs one phase behind
#define SBEV_O4_STEP(S_, X0, X1)                                                                    \
        {                                                                                                   \
            /* FETCH: the step's fragments LDS -> registers */                                              \
            {                                                                                               \
                SBEV_TRACE(S_, 0)                                                                           \
                /* W of this step (requested two steps ago) has landed once at most the 8 younger requests are outstanding: X of */ \
                /* step + 2 (2), W of step + 1 (4), X of step + 3 (2) -- vector memory operations complete in order             */ \
                wait_vmcnt_imm<8>();                                                                        \
                readw((S_), wa);                                                                            \
                const unsigned char* A = hst + ((S_) % NST) * HSTAGE + fo;                                  \
                _Pragma("unroll") for (int img = 0; img < NIMG; ++img)                                      \
                    _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                      \
                        xf[fa][img] = *reinterpret_cast<const bf16x8*>(A + img * IMG + fa * 1024);          \
                if ((S_) >= nh) {      /* the unequal last step: this half has none left and multiplies zeros */ \
                    _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                      \
                        _Pragma("unroll") for (int img = 0; img < NIMG; ++img)                              \
                            _Pragma("unroll") for (int e = 0; e < 8; ++e) xf[fa][img][e] = (__bf16)0.f;     \
                }                                                                                           \
                SBEV_TRACE(S_, 1)                                                                           \
                /* the staging of step + 2 and the X request of step + 4 also ride here: this phase otherwise waits ~800 cycles at its */ \
                /* barrier for the partner's MFMAs, and in the COMPUTE phase every VMEM issue delays the next MFMA                    */ \
                stagex((S_) + 2, X0, X1);                                                                   \
                __builtin_amdgcn_sched_barrier(0);     /* (the W requests stay behind the staging: the compiler's wait for X counts only its own loads) */ \
                issue_w((S_) + 2);                                                                          \
                loadx((S_) + 4, X0, X1);                                                                    \
            }                                                                                               \
            phase_barrier();                                                                                \
            SBEV_TRACE(S_, 4)                                                                               \
            /* COMPUTE: the step's MFMAs, nothing else */                                                   \
            _Pragma("unroll") for (int p = 0; p < PR::N; ++p)                                               \
                _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                          \
                    _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                        \
                        acc[fa][fb] = SBEV_MFMA(wa[fb][PR::ib(p)], xf[fa][PR::ia(p)], acc[fa][fb]);         \
            SBEV_TRACE(S_, 5)                                                                               \
            phase_barrier();                                                                                \
            SBEV_TRACE(S_, 6)                                                                               \
        }
        int sl = 0;
        for (; sl + 1 < n0h; sl += 2) {
            SBEV_O4_STEP(sl, xa0, xa1)
            SBEV_O4_STEP(sl + 1, xb0, xb1)
        }
        if (sl < n0h) SBEV_O4_STEP(sl, xa0, xa1)
#undef SBEV_O4_STEP
        if (half == 0) phase_barrier();
        // fold the two K halves (fixed order: bit-reproducible) and write the chunk's slab -- through an [m][n] image of the whole tile in
        // LDS: K half 1 writes its accumulators there (a lane holds one output ROW m and 4-column pieces), half 0 adds its own in
        // place, then all 8 waves read whole rows back and store 1 KiB per instruction.  (Storing from the accumulator layout wrote
        // 32-byte pieces of 32 rows per instruction: 19k of a workgroup's 83k cycles went into the fold + stores -- f16x3 trace.)
        wait_vmcnt_imm<0>();                                   // the last (dummy, clamped) W requests still write LDS this image reuses
        __syncthreads();
        constexpr int FLD = 256 + 4;                           // image row stride in floats
        float* img = reinterpret_cast<float*>(lds);
        if (half == 1) {
#pragma unroll
            for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(img + (fa * 32 + l31) * FLD + wc * 64 + fb * 32 + 8 * g + 4 * lh) =
                            (f32x4){acc[fa][fb][4 * g], acc[fa][fb][4 * g + 1], acc[fa][fb][4 * g + 2], acc[fa][fb][4 * g + 3]};
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4* q = reinterpret_cast<f32x4*>(img + (fa * 32 + l31) * FLD + wc * 64 + fb * 32 + 8 * g + 4 * lh);
                        const f32x4 o = *q;
                        *q = (f32x4){acc[fa][fb][4 * g] + o[0], acc[fa][fb][4 * g + 1] + o[1], acc[fa][fb][4 * g + 2] + o[2], acc[fa][fb][4 * g + 3] + o[3]};
                    }
        }
        __syncthreads();
        if (a.fold_sync == nullptr) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.nscale + lane * 4);          // a lane owns 4 fixed columns: exact powers of two
            float* out = a.P + (long long)chunk * M * 256 + lane * 4;
            for (int r = wave; r < NFA * 32; r += 8) {
                const int row = m0 + r;
                if (row < M SBEV_EXP_STORE_COND)
                    *reinterpret_cast<f32x4*>(out + (long long)row * 256) = *reinterpret_cast<const f32x4*>(img + r * FLD + lane * 4) * sc;
            }
        } else {
            // ---- the same slab, written through; then the row tile's S workgroups meet and fold ------------------------------------
            // The consumer (the tail row chain) summed the S slabs of its rows itself: 32 slabs x 8 rows = 256 KB per workgroup, read by
            // BOTH members of a pair -- 59 MB through the fabric in the first 10 us of a 48-us launch.  Here every chunk-workgroup sums
            // ceil(rows / S) rows of its tile over all S slabs (slab order 0 .. S - 1 from +0: the consumer's own order, bit for bit)
            // and the consumer reads ONE row block.
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.nscale + lane * 4);
            const __amdgpu_buffer_rsrc_t slab_rs = __builtin_amdgcn_make_buffer_rsrc(a.P, 0, 0x7fffffff, 0x00020000);
            const unsigned row_b = 1024u, slab_b = (unsigned)M * 1024u;                   // bytes (S * M * 1 KiB < 2^31: host-checked)
            for (int r = wave; r < NFA * 32; r += 8) {
                const int row = m0 + r;
                if (row < M) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(img + r * FLD + lane * 4) * sc;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slab_rs, (int)((unsigned)chunk * slab_b + (unsigned)row * row_b + (unsigned)lane * 16u), 0, FOLD_CPOL);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its slab rows have left
            __syncthreads();
            if (a.debug_drop && rt == 0 && chunk == 1) return;
            if (tid == 0) {
                unsigned* const word = a.fold_sync + rt;
                __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned n = 0;
                while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.S) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++n > FOLD_POLL_LIMIT) {
                        __hip_atomic_fetch_add(&g_fold_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        unsigned* const host_word = g_fold_fault_host;
                        if (host_word) __hip_atomic_fetch_add(host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
            const int rp = (NFA * 32 + a.S - 1) / a.S;                // rows of the tile per chunk-workgroup (4 at 128 rows x 32 chunks)
            for (int j = wave; j < rp; j += 8) {
                const int r = chunk * rp + j, row = m0 + r;
                if (r >= NFA * 32 || row >= M) continue;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                const unsigned base_b = (unsigned)row * row_b + (unsigned)lane * 16u;
                for (int z0 = 0; z0 < a.S; z0 += 32) {
                    f32x4 q[32];
#pragma unroll
                    for (int z = 0; z < 32; ++z) {
                        const int zz = z0 + z < a.S ? z0 + z : a.S - 1;
                        q[z] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(slab_rs, (int)((unsigned)zz * slab_b + base_b), 0, FOLD_CPOL));
                    }
#pragma unroll
                    for (int z = 0; z < 32; ++z) {
                        const float m = z0 + z < a.S ? 1.f : 0.f;
                        t[0] += q[z][0] * m; t[1] += q[z][1] * m; t[2] += q[z][2] * m; t[3] += q[z][3] * m;
                    }
                }
                *reinterpret_cast<f32x4*>(a.folded + (long long)row * 256 + lane * 4) = t;
            }
        }
        SBEV_WGTIME(3, 1)
    };
    if (nfa == 4) run(std::integral_constant<int, 4>{});
    else if (nfa == 3) run(std::integral_constant<int, 3>{});
    else if (nfa == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
}

// ---- out-projection, 256-row tiles (fp16 modes, pre-split X; round 6: from 1024 rows -- the batch shapes and the 1600-query config) -------------------------------------
// At 3200 / 3600 rows the 128-row kernel above reaches 0.28 / 0.31 of the matrix peak while the generator reaches 0.46 on the same rows:
// every 128-row tile streams its K chunk of ALL 256 W rows through the L2 -> LDS path -- 25 tiles x 33.5 MB of W + 420 MB of X per launch at
// 3200 rows -- and that delivery adds to the matrix time instead of hiding under it (DESIGN_HISTORY.md section 11.7).  Here a workgroup owns up to
// 256 rows x 256 columns x one K chunk: the two phase groups are the two ROW halves (not the two K halves), each with its own X stage ring,
// and they SHARE one W ring per 64-column quarter -- the group-0 wave of a quarter requests a k-step's W fragments (LDS-DMA, two steps
// ahead), waits for them in its own FETCH phase, and the group-1 wave reads the same slot one phase later, behind the barrier in between.
// Per 16-k step a workgroup takes 16 KB of X + 16 KB of W for twice the MFMAs the 128-row kernel gets out of 8 + 16 KB: a third less
// operand delivery per product, no fold of K halves.  Same products, same accumulation order over k as the 128-row kernel WITHOUT its
// K-half split -- results differ from it by fp32 summation order (a chunk's two halves are added k-ascending here), not in class.
template <int MODE>
__global__ __launch_bounds__(512) void gemm_bf16s_out8_kernel(const Out4Args a) {
    typedef Fmt<MODE> PR;
    static_assert(PR::F16 && PR::NIMG == 2, "pre-split fp16 operands");
    constexpr int NIMG = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int IMG = 128 * 32;               // bytes of one image of one group's stage: 128 rows x 16 k
    constexpr int HSTAGE = NIMG * IMG;
    constexpr int NST = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wave >> 2, wc = wave & 3;    // row half of the tile = phase group, 64-column quarter
    const unsigned logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int chunk = (int)(logical / (unsigned)a.ntm), rt = (int)(logical % (unsigned)a.ntm);
    const int M = a.M;
    const int f0 = rt * a.base + (rt < a.rem ? rt : a.rem);
    const int nf = a.base + (rt < a.rem ? 1 : 0);                // row fragments of this tile: 2 .. 8
    const int nfa0 = (nf + 1) / 2, nfa1 = nf / 2;                // ... of row half 0 / 1
    const int nfa = rh == 0 ? nfa0 : nfa1;
    const int m0 = f0 * 32, mh = m0 + (rh == 0 ? 0 : nfa0 * 32);
    const int KS = a.K / 16;
    const int c0 = (int)((long long)KS * chunk / a.S), c1 = (int)((long long)KS * (chunk + 1) / a.S);
    const int n_all = c1 - c0;                  // both groups walk every k-step of the chunk

    const int th = tid & 255;
    const int srow = th >> 1, skq = th & 1;     // staging: thread -> (row of its half, 8-k half of the step)
    int grow = mh + srow;
    grow = grow < M ? grow : M - 1;
    const unsigned* xp = a.Xp + (long long)grow * a.ldx + skq * 8;
    const unsigned wofs = (unsigned)(srow * 32 + skq * 16);
    unsigned char* hst = lds + rh * (NST * HSTAGE);              // this group's X stage ring
    const int last = c1 - 1;
    auto loadx = [&](int i, u32x4& v0, u32x4& v1) {              // k-step i (clamped: a dummy past the end)
        int sl = c0 + i;
        sl = sl < last ? sl : last;
        const unsigned* p = xp + (long long)sl * 16;
        v0 = *reinterpret_cast<const u32x4*>(p);
        v1 = *reinterpret_cast<const u32x4*>(p + 4);
    };
    auto stagex = [&](int i, const u32x4 p, const u32x4 q) {     // k-step i -> ring slot i % 3: de-interleave hi | lo
        unsigned char* st = hst + (i % NST) * HSTAGE + wofs;
        *reinterpret_cast<u32x4*>(st) = (u32x4){__builtin_amdgcn_perm(p.y, p.x, 0x05040100u), __builtin_amdgcn_perm(p.w, p.z, 0x05040100u),
                                                __builtin_amdgcn_perm(q.y, q.x, 0x05040100u), __builtin_amdgcn_perm(q.w, q.z, 0x05040100u)};
        *reinterpret_cast<u32x4*>(st + IMG) = (u32x4){__builtin_amdgcn_perm(p.y, p.x, 0x07060302u), __builtin_amdgcn_perm(p.w, p.z, 0x07060302u),
                                                      __builtin_amdgcn_perm(q.y, q.x, 0x07060302u), __builtin_amdgcn_perm(q.w, q.z, 0x07060302u)};
    };
    // W: ONE 3-slot ring per column quarter, filled by the quarter's group-0 wave, read by both of its waves
    constexpr int WSLOT = 2 * NIMG * 1024;               // one step of one quarter: 2 column fragments x 2 images
    constexpr int WRING0 = 2 * NST * HSTAGE;             // behind the two groups' X stages
    const unsigned wring = (unsigned)(WRING0 + wc * (NST * WSLOT));
    const unsigned char* wg0 = reinterpret_cast<const unsigned char*>(a.Wp) + (long long)(2 * wc) * KS * (NIMG * 1024);
    const unsigned voff = (unsigned)lane * 16u;
    auto issue_w = [&](int i) {                          // (group 0 only) k-step i (clamped) -> ring slot i % 3
        int sl = c0 + i;
        sl = sl < last ? sl : last;
        const unsigned dst = wring + (unsigned)((i % NST) * WSLOT);
        glds16_images<NIMG>(wg0 + (long long)sl * (NIMG * 1024), voff, dst);
        glds16_images<NIMG>(wg0 + ((long long)KS + sl) * (NIMG * 1024), voff, dst + NIMG * 1024);
    };
    auto readw = [&](int i, bf16x8 (&w)[2][NIMG]) {
        const unsigned char* src = lds + wring + (i % NST) * WSLOT + voff;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int img = 0; img < NIMG; ++img) w[fb][img] = *reinterpret_cast<const bf16x8*>(src + (fb * NIMG + img) * 1024);
    };
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned fo = (unsigned)l31 * 32u + (unsigned)lh * 16u;

    auto run = [&](auto nfa_c) {
        constexpr int NFA = decltype(nfa_c)::value;
        f32x16 acc[NFA][2];
#pragma unroll
        for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[fa][fb][e] = 0.f;
        u32x4 xa0, xa1, xb0, xb1;
        bf16x8 wa[2][NIMG], xf[NFA][NIMG];
        loadx(0, xa0, xa1);
        loadx(1, xb0, xb1);
        if (rh == 0) { issue_w(0); issue_w(1); }
        stagex(0, xa0, xa1);
        loadx(2, xa0, xa1);
        stagex(1, xb0, xb1);
        loadx(3, xb0, xb1);
        wait_vmcnt_imm<0>();
        __syncthreads();
        if (rh == 1) phase_barrier();                         // the second row half runs one phase behind
#define SBEV_O8_STEP(S_, X0, X1)                                                                    \
        {                                                                                                   \
            {   /* FETCH */                                                                                 \
                /* group 0: W of this step (its own request of two steps ago) has landed once at most the 8 younger requests are */ \
                /* outstanding (X of step + 2, W of step + 1, X of step + 3); group 1 reads the slot one phase -- one barrier -- later */ \
                wait_vmcnt_imm<8>();                                                                        \
                readw((S_), wa);                                                                            \
                const unsigned char* A = hst + ((S_) % NST) * HSTAGE + fo;                                  \
                _Pragma("unroll") for (int img = 0; img < NIMG; ++img)                                      \
                    _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                      \
                        xf[fa][img] = *reinterpret_cast<const bf16x8*>(A + img * IMG + fa * 1024);          \
                stagex((S_) + 2, X0, X1);                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                if (rh == 0) issue_w((S_) + 2);                                                             \
                loadx((S_) + 4, X0, X1);                                                                    \
                /* the slot read above is overwritten by group 0's request of the NEXT phase: the reads must have returned by the barrier */ \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
            }                                                                                               \
            phase_barrier();                                                                                \
            _Pragma("unroll") for (int p = 0; p < PR::N; ++p)                                               \
                _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                          \
                    _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                        \
                        acc[fa][fb] = SBEV_MFMA(wa[fb][PR::ib(p)], xf[fa][PR::ia(p)], acc[fa][fb]);         \
            phase_barrier();                                                                                \
        }
        int sl = 0;
        for (; sl + 1 < n_all; sl += 2) {
            SBEV_O8_STEP(sl, xa0, xa1)
            SBEV_O8_STEP(sl + 1, xb0, xb1)
        }
        if (sl < n_all) SBEV_O8_STEP(sl, xa0, xa1)
#undef SBEV_O8_STEP
        if (rh == 0) phase_barrier();
        // the chunk's slab: one row half at a time through an [m][n] image in LDS (a lane holds one output ROW m and 4-column pieces), all 8
        // waves then read whole rows back and store 1 KiB per instruction
        wait_vmcnt_imm<0>();                                   // the last (dummy, clamped) W requests still write LDS this image reuses
        __syncthreads();
        constexpr int FLD = 256 + 4;
        float* img = reinterpret_cast<float*>(lds);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.nscale + lane * 4);
        float* out = a.P + (long long)chunk * M * 256 + lane * 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (rh == pass) {
#pragma unroll
                for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<f32x4*>(img + (fa * 32 + l31) * FLD + wc * 64 + fb * 32 + 8 * g + 4 * lh) =
                                (f32x4){acc[fa][fb][4 * g], acc[fa][fb][4 * g + 1], acc[fa][fb][4 * g + 2], acc[fa][fb][4 * g + 3]};
            }
            __syncthreads();
            const int prow0 = m0 + (pass == 0 ? 0 : nfa0 * 32), prows = (pass == 0 ? nfa0 : nfa1) * 32;
            for (int r = wave; r < prows; r += 8) {
                const int row = prow0 + r;
                if (row < M SBEV_EXP_STORE_COND)
                    *reinterpret_cast<f32x4*>(out + (long long)row * 256) = *reinterpret_cast<const f32x4*>(img + r * FLD + lane * 4) * sc;
            }
            __syncthreads();
        }
    };
    if (nfa == 4) run(std::integral_constant<int, 4>{});
    else if (nfa == 3) run(std::integral_constant<int, 3>{});
    else if (nfa == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
}

template <typename Kern>
int reserve_lds(Kern k, int bytes, const char* what) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        sbev::set_error("%s: cannot reserve %d B of LDS: %s", what, bytes, hipGetErrorString(e));
        return SBEV_ELAUNCH;
    }
    return SBEV_OK;
}

// K chunks of the out-projection: fill the 256 CUs with (64-row tile x chunk) workgroups in whole rounds, at least 8 slabs each
int out_chunks(long long M, int K) {
    static const int forced = getenv("SBEV_BF16S_OUT_CHUNKS") ? atoi(getenv("SBEV_BF16S_OUT_CHUNKS")) : 0;      // experiments
    if (forced > 0 && forced <= K / 32 / 8) return forced;
    const long long nrt = (M + 63) / 64;
    const int max_s = K / 32 / 8 < 1 ? 1 : K / 32 / 8;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 64 && s <= max_s; ++s) {
        const long long wgs = nrt * s;
        const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
    }
    return best;
}

// the 128-row kernel's plan: row tiles of <= 4 fragments (balanced), K chunks of >= 8 k-steps that fill the CUs in whole rounds
struct Out4Plan { int ntm, base, rem, S; };
Out4Plan out4_plan(long long M, int K) {
    const int nfrag = (int)((M + 31) / 32);
    Out4Plan p{};
    p.ntm = (nfrag + 3) / 4;
    p.base = nfrag / p.ntm;
    p.rem = nfrag % p.ntm;
    const int max_s = K / 16 / 8 < 1 ? 1 : K / 16 / 8;
    p.S = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 64 && s <= max_s; ++s) {
        const long long wgs = (long long)p.ntm * s;
        const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
        if (eff > best_eff + 1e-9) { best_eff = eff; p.S = s; }
    }
    return p;
}

// the 256-row kernel's plan: row tiles of <= 8 fragments (balanced, >= 2), K chunks of >= 8 k-steps, ONE round of workgroups (more chunks
// would only add slabs for the consumer to sum)
// (measured, samples/s with / without: 3200 rows 1289 / 1189, 3600 rows 506 / 481, 1600 rows 194.6 / 187.2; 900 rows 536.6 / 538.8 -- the
// kernel itself is 4 us faster there too, but 4 row tiles x 64 chunks leave the tail 64 slabs to sum instead of 32: from 1024 rows)
constexpr int OUT8_HARD_MIN = 1024;       // (workspaces are sized for the 256-row plan from here on, whatever the switch says later)
int out8_clamp(int rows) { return rows <= 0 ? 0 : (rows < OUT8_HARD_MIN ? OUT8_HARD_MIN : rows); }
std::atomic<int> g_out8_min_rows{out8_clamp(getenv("SBEV_OUT8_MIN_ROWS") ? atoi(getenv("SBEV_OUT8_MIN_ROWS")) : OUT8_HARD_MIN)};
Out4Plan out8_plan(long long M, int K) {
    const int nfrag = (int)((M + 31) / 32);
    Out4Plan p{};
    p.ntm = (nfrag + 7) / 8;
    p.base = nfrag / p.ntm;
    p.rem = nfrag % p.ntm;
    const int max_s = K / 16 / 8 < 1 ? 1 : K / 16 / 8;
    p.S = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 64 && s <= max_s && (long long)p.ntm * s <= 256; ++s) {
        const double eff = (double)p.ntm * s / 256.0;
        if (eff > best_eff + 1e-9) { best_eff = eff; p.S = s; }
    }
    return p;
}
bool out8_takes(long long M, int K) {
    const int mr = g_out8_min_rows.load(std::memory_order_relaxed);
    if (mr <= 0 || M < mr) return false;
    const Out4Plan p = out8_plan(M, K);
    return p.base >= 2 && p.ntm <= 256;
}

}  // namespace

extern "C" int64_t sbev_bf16s_image_elems(int64_t rows, int K, int nimg) { return (rows + 31) / 32 * 32 * K * (nimg == 3 ? 3 : 2); }

extern "C" int sbev_split_bf16s_rows(const float* X, int64_t ldx, uint16_t* out, int64_t rows, int K, int nimg, sbev_stream_t stream) {
    SBEV_REQUIRE(rows >= 0 && K >= 8 && K % 8 == 0 && (nimg == 2 || nimg == 3), "sbev_split_bf16s_rows: K=%d (multiple of 8), nimg=%d (2 or 3)", K, nimg);
    if (rows == 0) return SBEV_OK;
    SBEV_REQUIRE(X && out && ldx >= K && ldx % 4 == 0 && (((uintptr_t)X | (uintptr_t)out) & 15) == 0, "sbev_split_bf16s_rows: null / unaligned pointer or bad ldx");
    const long long n = rows * (K / 8);
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (nimg == 3) hipLaunchKernelGGL(split_rows_kernel<3>, grid, dim3(256), 0, s, X, (long long)ldx, out, (long long)rows, K);
    else hipLaunchKernelGGL(split_rows_kernel<2>, grid, dim3(256), 0, s, X, (long long)ldx, out, (long long)rows, K);
    return sbev::check_launch("sbev_split_bf16s_rows");
}

extern "C" int sbev_pack_bf16s_frags(const float* W, int64_t ldw, uint16_t* out, int N, int K, int nimg, sbev_stream_t stream) {
    SBEV_REQUIRE(N >= 1 && K >= 16 && K % 16 == 0 && (nimg == 2 || nimg == 3), "sbev_pack_bf16s_frags: N=%d, K=%d (multiple of 16), nimg=%d (2 or 3)", N, K, nimg);
    SBEV_REQUIRE(W && out && ldw >= K && ldw % 4 == 0 && (((uintptr_t)W | (uintptr_t)out) & 15) == 0, "sbev_pack_bf16s_frags: null / unaligned pointer or bad ldw");
    const long long n = (long long)((N + 31) / 32) * (K / 16) * 64;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (nimg == 3) hipLaunchKernelGGL(pack_frags_kernel<1>, grid, dim3(256), 0, s, W, (long long)ldw, out, N, K, (const float*)nullptr, 0);
    else hipLaunchKernelGGL(pack_frags_kernel<0>, grid, dim3(256), 0, s, W, (long long)ldw, out, N, K, (const float*)nullptr, 0);
    return sbev::check_launch("sbev_pack_bf16s_frags");
}

// fp16 hi + lo fragments of W [N, ldw] in the same fragment order, scaled by a power of two per row (per_tensor = 0: scales =
// [2][N], up then down) or by one for the whole matrix (per_tensor = 1: scales = [2]); the scales are computed here (device side).
// per_tensor = 2: scales [2] is an INPUT (the caller's power of two for a bounded operand: no pass over W for its maximum)
extern "C" int sbev_pack_f16s_frags(const float* W, int64_t ldw, uint16_t* out, float* scales, int N, int K, int per_tensor, sbev_stream_t stream) {
    SBEV_REQUIRE(N >= 1 && K >= 16 && K % 16 == 0, "sbev_pack_f16s_frags: N=%d, K=%d (multiple of 16)", N, K);
    SBEV_REQUIRE(W && out && scales && ldw >= K && ldw % 4 == 0 && (((uintptr_t)W | (uintptr_t)out) & 15) == 0, "sbev_pack_f16s_frags: null / unaligned pointer or bad ldw");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (per_tensor == 1) {
        const long long n4 = (long long)N * (K / 4);
        const unsigned nb = (unsigned)(n4 / 1024 < 1 ? 1 : n4 / 1024 > 64 ? 64 : n4 / 1024);
        hipLaunchKernelGGL(tensor_scale_kernel, dim3(nb), dim3(256), 0, s, W, (long long)ldw, (long long)N, K, scales, next_scale_slot());
    }
    else if (per_tensor == 0) hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, (long long)ldw, N, K, scales, scales + N);
    const long long n = (long long)((N + 31) / 32) * (K / 16) * 64;
    hipLaunchKernelGGL(pack_frags_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, (long long)ldw, out, N, K, (const float*)scales, per_tensor ? 0 : 1);
    return sbev::check_launch("sbev_pack_f16s_frags");
}

// {2^e, 2^-e} of a whole fp32 matrix [rows, ldx] (K columns used): the scale sbev_gemm_tn_f16s takes for an operand
extern "C" int sbev_f16s_tensor_scale(const float* X, int64_t ldx, int64_t rows, int K, float* updown, sbev_stream_t stream) {
    SBEV_REQUIRE(X && updown && rows >= 1 && K >= 4 && K % 4 == 0 && ldx >= K && ldx % 4 == 0 && (((uintptr_t)X) & 15) == 0,
                 "sbev_f16s_tensor_scale: rows=%lld, K=%d (multiple of 4), ldx=%lld (multiple of 4), X 16-byte aligned", (long long)rows, K, (long long)ldx);
    const long long n4 = (long long)rows * (K / 4);
    const unsigned nb = (unsigned)(n4 / 1024 < 1 ? 1 : n4 / 1024 > 64 ? 64 : n4 / 1024);
    hipLaunchKernelGGL(tensor_scale_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, (long long)ldx, (long long)rows, K, updown, next_scale_slot());
    return sbev::check_launch("sbev_f16s_tensor_scale");
}

// the weight-stationary generator kernel where it applies (K = 256, two images); sbev_linear_gen_weight_stationary(0) / SBEV_NO_GEN_WS=1
// restore the tiled ping-pong kernel everywhere (A/B; results are bit-identical)
static std::atomic<int> g_gen_ws{getenv("SBEV_NO_GEN_WS") ? 0 : 1};
extern "C" int sbev_linear_gen_weight_stationary(int enable) { return g_gen_ws.exchange(enable ? 1 : 0, std::memory_order_relaxed); }

static int ntm_of(int64_t M) { return (int)(((M + 31) / 32 + 7) / 8); }      // row tiles of <= 8 fragments

extern "C" int sbev_linear_bf16s_gen_ok(int64_t M, int N, int K) {
    return M >= 1 && M <= 0x7fffffffLL / 1024 && N >= 256 && N % 256 == 0 && K >= 32 && K % 32 == 0 && K <= 4096 &&
           ntm_of(M) <= 256;
}

static bool gen_ws_takes(int64_t M, int N, int K, int64_t ldy, int nimg) {
    return g_gen_ws.load(std::memory_order_relaxed) != 0 && K == 16 * WS_KS && nimg != 3 && M * ldy * 4 < 0x7fffffffLL;
}

static int gen_launch(const uint16_t* Xs, const uint16_t* Ws, const float* bias, float* Y, int64_t M, int N, int K, int64_t ldy, int relu,
                      int nimg, const float* xscale, const float* colscale, sbev_stream_t stream, const sbev::LazyScan* lz = nullptr) {
    SBEV_REQUIRE(M >= 0 && sbev_linear_bf16s_gen_ok(M > 0 ? M : 1, N, K), "sbev_linear_bf16s_gen: needs N %% 256 == 0, K %% 32 == 0, K <= 4096 (M=%lld N=%d K=%d)", (long long)M, N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(Xs && Ws && Y && ldy >= N && ldy % 4 == 0, "sbev_linear_bf16s_gen: bad pointers / leading dimension");
    SBEV_REQUIRE((((uintptr_t)Xs | (uintptr_t)Ws | (uintptr_t)Y) & 15) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0), "sbev_linear_bf16s_gen: 16-byte alignment");
    const int nfrag = (int)((M + 31) / 32);
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    // weight-stationary kernel (round 4): K = 256, two images, Y addressable by a 31-bit byte offset
    SBEV_REQUIRE(!lz || gen_ws_takes(M, N, K, ldy, nimg), "generator: the on-demand relayout's scan rides in the weight-stationary kernel only");
    if (gen_ws_takes(M, N, K, ldy, nimg)) {
        // row splits: tasks = column tiles x splits walked by <= one workgroup per CU; a task costs its fragments + ~4 fragments' worth
        // of weight load (256 KB that nothing overlaps).  c2 (29 fragments, 128 column tiles): 2 splits = 256 tasks of 15 / 14 fragments
        const int nct = N / G_COLS;
        int nrs = 1;
        double best = 1e30;
        for (int r = 1; r <= 16 && r <= nfrag; ++r) {
            const long long tasks = (long long)nct * r;
            const double cost = (double)((tasks + cus - 1) / cus) * ((nfrag + r - 1) / r + 4.0);
            if (cost < best - 1e-9) { best = cost; nrs = r; }
        }
        GenWsArgs w{Xs, Ws, bias, Y, (int)M, N, (long long)ldy, relu, nrs, nfrag / nrs, nfrag % nrs, nct * nrs, colscale, xscale, 0, 4, {}};
        if (lz) {
            w.lazy_on = 1; w.lazy_esize = lz->esize;
            LazyArgs& la = w.lazy;
            la.table = lz->table; la.n_levels = lz->plan->n_levels; la.R = lz->plan->R; la.need = lz->need; la.done = lz->done;
            la.first = 0; la.last = lz->last ? 1 : 0;
            for (int l = 0; l < lz->plan->n_levels; ++l) {
                la.index[l] = lz->table ? lz->index[l] : 0;
                la.src[l] = lz->table ? nullptr : lz->src[l];
                la.out[l] = lz->out[l];
                la.S[l] = lz->plan->S[l]; la.tiles[l] = lz->plan->tiles[l]; la.base[l] = lz->plan->base[l];
            }
            la.base[lz->plan->n_levels] = lz->plan->base[lz->plan->n_levels];
        }
        const unsigned grid = (unsigned)(w.ntask < cus ? w.ntask : cus);
        const int lds = WS_SLOTS * WS_FRAG;
        int st;
        const bool prof = sbev::profile_begin(s, &e0, &e1, 1);
#define SBEV_LAUNCH_WS(MD)                                                                                  \
        {                                                                                                   \
            auto kern = relu ? gemm_f16s_gen_ws_kernel<MD, true> : gemm_f16s_gen_ws_kernel<MD, false>;       \
            st = reserve_lds(kern, lds, "sbev_linear_bf16s_gen");                                           \
            if (st != SBEV_OK) return st;                                                                   \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, w);                                     \
        }
        if (nimg == 2) SBEV_LAUNCH_WS(0)
        else if (nimg == 4) SBEV_LAUNCH_WS(2)
        else SBEV_LAUNCH_WS(3)
#undef SBEV_LAUNCH_WS
        if (prof) sbev::profile_end(s, e0, e1, 1);
        return sbev::check_launch("sbev_linear_bf16s_gen");
    }
    // 256-row tiles (wave = 128 x 64) carry 1.5x the MFMA work per operand byte; 128-row tiles only where they fill the chip
    // better (few rows) -- SBEV_BF16S_GEN_RF=2/4 forces one (A/B runs)
    static const int forced_rf = getenv("SBEV_BF16S_GEN_RF") ? atoi(getenv("SBEV_BF16S_GEN_RF")) : 0;
    const int rf = forced_rf == 2 || forced_rf == 4 ? forced_rf : (nfrag > 4 ? 4 : 2);
    const int tf = 2 * rf;
    const int ntm = (nfrag + tf - 1) / tf;
    GenArgs a{Xs, Ws, bias, Y, (int)M, N, K, (long long)ldy, relu, ntm, nfrag / ntm, nfrag % ntm, colscale, xscale};
    const int nim = nimg == 3 ? 3 : 2;          // images per operand
    // one row tile per workgroup for life: grid = a multiple of ntm, at most the CU count, at most the tile count
    long long per = cus / ntm < 1 ? 1 : cus / ntm;
    if (per > N / G_COLS) per = N / G_COLS;
    // the bias slices of a workgroup's column tiles wait in LDS behind the stage ring: at most 16 tiles (16 KiB) per launch,
    // wider matrices take several launches over column ranges
    const int nct = N / G_COLS;
    const int nst = rf == 2 ? 4 : 3;
    const int ring_bytes = nst * nim * (tf + 8) * 1024 + (nim == 2 && rf == 4 ? 8 * 32 * 36 * 4 : 0);      // + the epilogue's transpose patches
    int tiles_per_wg = (160 * 1024 - ring_bytes) / (G_COLS * 4 * (nimg >= 4 ? 2 : 1));      // bias (+ scale) slices behind the ring
    tiles_per_wg = tiles_per_wg > 16 ? 16 : tiles_per_wg;
    const long long max_ct = per * tiles_per_wg;
    for (long long c0 = 0; c0 < nct; c0 += max_ct) {
        const int nc = (int)(nct - c0 < max_ct ? nct - c0 : max_ct);
        GenArgs ac = a;
        ac.Ws = Ws + c0 * 8 * (long long)(K / 16) * nim * 512;       // 8 fragment blocks of 32 columns per tile
        ac.colscale = colscale ? colscale + c0 * G_COLS : nullptr;
        ac.bias = bias ? bias + c0 * G_COLS : nullptr;
        ac.Y = Y + c0 * G_COLS;
        ac.N = nc * G_COLS;
        const long long pc = per < nc ? per : nc;
        const unsigned gridc = (unsigned)(pc * ntm);
        const int bias_bytes = (int)((nc + pc - 1) / pc) * G_COLS * 4 * (nimg >= 4 ? 2 : 1);     // (fp16 modes: + the column scales)
        const int lds = ring_bytes + bias_bytes;
        int st;
        const bool prof = sbev::profile_begin(s, &e0, &e1, 1);
#define SBEV_LAUNCH_GEN(NI, RFV)                                                                            \
        {                                                                                                   \
            st = reserve_lds(gemm_bf16s_gen3_kernel<NI, RFV>, lds, "sbev_linear_bf16s_gen");                \
            if (st != SBEV_OK) return st;                                                                   \
            hipLaunchKernelGGL((gemm_bf16s_gen3_kernel<NI, RFV>), dim3(gridc), dim3(512), lds, s, ac);      \
        }
        if (nimg == 3 && rf == 4) SBEV_LAUNCH_GEN(1, 4)
        else if (nimg == 3) SBEV_LAUNCH_GEN(1, 2)
        else if (nimg == 2 && rf == 4) SBEV_LAUNCH_GEN(0, 4)
        else if (nimg == 2) SBEV_LAUNCH_GEN(0, 2)
        else if (nimg == 4 && rf == 4) SBEV_LAUNCH_GEN(2, 4)
        else if (nimg == 4) SBEV_LAUNCH_GEN(2, 2)
        else if (rf == 4) SBEV_LAUNCH_GEN(3, 4)
        else SBEV_LAUNCH_GEN(3, 2)
#undef SBEV_LAUNCH_GEN
        if (prof) sbev::profile_end(s, e0, e1, 1);
    }
    return sbev::check_launch("sbev_linear_bf16s_gen");
}

extern "C" int sbev_linear_bf16s_gen(const uint16_t* Xs, const uint16_t* Ws, const float* bias, float* Y, int64_t M, int N, int K,
                                     int64_t ldy, int relu, int nimg, sbev_stream_t stream) {
    SBEV_REQUIRE(nimg == 2 || nimg == 3, "sbev_linear_bf16s_gen: nimg=%d (2 = bf16x3, 3 = bf16x6)", nimg);
    return gen_launch(Xs, Ws, bias, Y, M, N, K, ldy, relu, nimg, nullptr, nullptr, stream);
}

// fp16 hi + lo images (sbev_pack_f16s_frags): xscale = X's {up, down} (per tensor), wdown = the [N] down-scales of W's rows;
// nprod = 3 (hl, lh, hh: drops the 2^-24-class lo x lo) or 4
extern "C" int sbev_linear_f16s_gen(const uint16_t* Xs, const float* xscale, const uint16_t* Ws, const float* wdown, const float* bias, float* Y,
                                    int64_t M, int N, int K, int64_t ldy, int relu, int nprod, sbev_stream_t stream) {
    SBEV_REQUIRE(nprod == 3 || nprod == 4, "sbev_linear_f16s_gen: nprod=%d (3 or 4 image products)", nprod);
    SBEV_REQUIRE(M == 0 || (xscale && wdown), "sbev_linear_f16s_gen: null scale pointer");
    return gen_launch(Xs, Ws, bias, Y, M, N, K, ldy, relu, nprod + 1, xscale, wdown, stream);
}

namespace sbev {
bool linear_f16s_gen_takes_scan(int64_t M, int N, int K, int64_t ldy, int nprod) { return gen_ws_takes(M, N, K, ldy, nprod + 1); }
int linear_f16s_gen_scan(const uint16_t* Xs, const float* xscale, const uint16_t* Ws, const float* wdown, const float* bias, float* Y, int64_t M,
                         int N, int K, int64_t ldy, int relu, int nprod, const LazyScan& lz, hipStream_t stream) {
    SBEV_REQUIRE(nprod == 3 || nprod == 4, "sbev_linear_f16s_gen: nprod=%d (3 or 4 image products)", nprod);
    SBEV_REQUIRE(M == 0 || (xscale && wdown), "sbev_linear_f16s_gen: null scale pointer");
    return gen_launch(Xs, Ws, bias, Y, M, N, K, ldy, relu, nprod + 1, xscale, wdown, reinterpret_cast<sbev_stream_t>(stream), &lz);
}
}  // namespace sbev

extern "C" int sbev_linear_bf16s_out_ok(int64_t M, int N, int K) {
    return M >= 1 && M <= 0x7fffffffLL / 512 && N == 256 && K >= 256 && K % 32 == 0;
}

extern "C" int sbev_linear_bf16s_out_plan(int64_t M, int N, int K) {      // slabs to provide: the larger of the two kernels' plans
    if (!sbev_linear_bf16s_out_ok(M, N, K)) return 0;
    const int a = out_chunks(M, K), b = out4_plan(M, K).S, c = M >= OUT8_HARD_MIN ? out8_plan(M, K).S : 0;      // (whichever kernel a call picks)
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

namespace sbev {
// the GEMM half: *used partial slabs [used, M, 256] (to be summed by sbev_splitk_reduce_f32 or the row-chain tail)
// the out-projection's in-launch fold is possible for this shape on the current device: pre-split operand plan whose workgroups all fit
// the device at once (one workgroup per CU: 144 KiB of LDS), slabs addressable through one buffer resource
// OFF by default: measured at config 2 (profiles/r6_out_fold_ab.txt) the fold costs the out-projection +12.6 us (56.2 -> 68.8: write-through
// drain, the arrival round trip, 29.5 MB of slab reads past the L2) and saves the tail 4 us (51 -> 47) -- 512 vs 535 samples/s.  Kept as an
// A/B switch (SBEV_OUT_FOLD=1 / sbev_decoder_out_fold(1)), bit-identical either way.
std::atomic<int> g_out_fold{getenv("SBEV_OUT_FOLD") ? 1 : 0};
std::atomic<int> g_out_fold_drop{0};
int out_fold_drop(int enable) { return g_out_fold_drop.exchange(enable ? 1 : 0, std::memory_order_relaxed); }
bool out_fold_ok(long long M, int K) {
    if (g_out_fold.load(std::memory_order_relaxed) == 0 || M < 1 || M > 4096) return false;
    const Out4Plan pl = out4_plan(M, K);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return pl.S > 1 && (long long)pl.ntm * pl.S <= cus && pl.ntm <= 64 && (long long)pl.S * M * 1024 < 0x7fffffffLL;
}
// the fold's fault words (see g_fold_fault_host): `host_word_dev` = device address of the decoder's host-mapped fault word
bool out_fold_install(void* host_word_dev) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fold_fault_host), &host_word_dev, sizeof(host_word_dev)) == hipSuccess;
}
int out8_min_rows(int rows) { return g_out8_min_rows.exchange(out8_clamp(rows), std::memory_order_relaxed); }
int out_fold_switch(int enable) { return g_out_fold.exchange(enable ? 1 : 0, std::memory_order_relaxed); }
long long out_fold_timeouts() {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_fold_timeouts), sizeof(v)) != hipSuccess) return -1;
    return (long long)v;
}

int launch_splitk_slabs_bf16s(const float* X, const uint16_t* Wp, int64_t M, int K, int64_t ldx, int nimg, float* slabs, int* used,
                              hipStream_t s, int x_up_log2, const float* nscale, bool x_pairs, const float* xdev, unsigned* fold_sync,
                              float* folded) {
    SBEV_REQUIRE(nimg < 4 || nscale, "sbev_linear_splitk_f16s: null scale pointer");
    SBEV_REQUIRE(!xdev || (nimg >= 4 && !x_pairs), "sbev_linear_splitk_f16s: a device-side X scale needs an fp16 mode and fp32 X");
    if (x_pairs) {                              // fp16 modes with the pre-split operand: 128-row tiles
        SBEV_REQUIRE(nimg >= 4, "sbev_linear_splitk_f16s: pre-split X needs an fp16 mode");
        const bool big = out8_takes(M, K);                               // batch shapes: 256-row tiles (a third less operand delivery)
        const Out4Plan pl = big ? out8_plan(M, K) : out4_plan(M, K);
        const bool fold = !big && fold_sync && folded && out_fold_ok(M, K);      // (the caller asked AND the shape / device allow it: else S slabs as before)
        *used = fold ? 1 : pl.S;                                        // folded: the consumer reads `folded` as ONE slab
        Out4Args a4{reinterpret_cast<const unsigned*>(X), Wp, slabs, (int)M, K, (long long)ldx, pl.ntm, pl.base, pl.rem, pl.S, nscale,
                    fold ? fold_sync : nullptr, fold ? folded : nullptr, g_out_fold_drop.load(std::memory_order_relaxed)};
        const long long wgs4 = (long long)pl.ntm * pl.S;
        SBEV_REQUIRE(wgs4 <= 0x7fffffffLL, "sbev_linear_splitk_f16s: too many workgroups");
        constexpr int LDS4 = 144 * 1024;        // 48 KiB of X stages + 8 waves x 12 KiB of W ring (the 128 KiB fold buffer reuses them)
        hipEvent_t f0, f1;
        int st4;
#define SBEV_LAUNCH_OUT4(KERN)                                                                   \
        {                                                                                        \
            st4 = reserve_lds(KERN, LDS4, "sbev_linear_splitk_f16s");                            \
            if (st4 != SBEV_OK) return st4;                                                      \
            const bool prof = profile_begin(s, &f0, &f1, 2);                                     \
            hipLaunchKernelGGL(KERN, dim3((unsigned)wgs4), dim3(512), LDS4, s, a4);              \
            if (prof) profile_end(s, f0, f1, 2);                                                 \
        }
        if (big) {
            if (nimg == 4) SBEV_LAUNCH_OUT4(gemm_bf16s_out8_kernel<2>) else SBEV_LAUNCH_OUT4(gemm_bf16s_out8_kernel<3>)
        } else {
            if (nimg == 4) SBEV_LAUNCH_OUT4(gemm_bf16s_out4_kernel<2>) else SBEV_LAUNCH_OUT4(gemm_bf16s_out4_kernel<3>)
        }
#undef SBEV_LAUNCH_OUT4
        return check_launch("sbev_linear_splitk_f16s (gemm, 128-row tiles)");
    }
    const int S = out_chunks(M, K);
    *used = S;
    OutArgs a{X, Wp, slabs, (int)M, K, (long long)ldx, (int)((M + 63) / 64), S, ldexpf(1.f, x_up_log2), nscale, xdev};
    const long long wgs = (long long)a.nrt * S;
    SBEV_REQUIRE(wgs <= 0x7fffffffLL, "sbev_linear_splitk_bf16s: too many workgroups");
    hipEvent_t e0, e1;
    int st;
    constexpr int LDS3 = 2 * 3 * 3 * O_IMG;     // 2 halves x 3 stages x 3 images x 4 KiB = 72 KiB (>= the 64 KiB fold buffer)
    constexpr int LDS2 = 65536;                 // two images: 48 KiB of stages, 64 KiB fold buffer
#define SBEV_LAUNCH_OUT(KERN, LDSB)                                                              \
    {                                                                                            \
        st = reserve_lds(KERN, LDSB, "sbev_linear_splitk_bf16s");                                \
        if (st != SBEV_OK) return st;                                                            \
        const bool prof = profile_begin(s, &e0, &e1, 2);                                         \
        hipLaunchKernelGGL(KERN, dim3((unsigned)wgs), dim3(512), LDSB, s, a);                    \
        if (prof) profile_end(s, e0, e1, 2);                                                     \
    }
    if (nimg == 3) SBEV_LAUNCH_OUT(gemm_bf16s_out3_kernel<1>, LDS3)
    else if (nimg == 2) SBEV_LAUNCH_OUT(gemm_bf16s_out3_kernel<0>, LDS2)
    else if (nimg == 4) SBEV_LAUNCH_OUT(gemm_bf16s_out3_kernel<2>, LDS2)
    else SBEV_LAUNCH_OUT(gemm_bf16s_out3_kernel<3>, LDS2)
#undef SBEV_LAUNCH_OUT
    return check_launch("sbev_linear_splitk_bf16s (gemm)");
}
}  // namespace sbev

extern "C" int sbev_linear_splitk_bf16s(const float* X, const uint16_t* Wp, const float* bias, const float* residual,
                                        const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                                        int64_t M, int N, int K, int64_t ldx, int relu, int nimg, float* workspace,
                                        sbev_stream_t stream) {
    SBEV_REQUIRE(nimg == 2 || nimg == 3, "sbev_linear_splitk_bf16s: nimg=%d (2 = bf16x3, 3 = bf16x6)", nimg);
    SBEV_REQUIRE(M >= 0 && sbev_linear_bf16s_out_ok(M > 0 ? M : 1, N, K), "sbev_linear_splitk_bf16s: needs N == 256, K %% 32 == 0, K >= 256 (N=%d K=%d)", N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && Wp && Y && workspace && ldx % 4 == 0 && ldx >= K, "sbev_linear_splitk_bf16s: bad pointers");
    SBEV_REQUIRE((((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)workspace) & 15) == 0, "sbev_linear_splitk_bf16s: 16-byte alignment");
    int used = 0;
    const int st = sbev::launch_splitk_slabs_bf16s(X, Wp, M, K, ldx, nimg, workspace, &used, reinterpret_cast<hipStream_t>(stream), 0, nullptr, false);
    if (st != SBEV_OK) return st;
    return sbev_splitk_reduce_f32(workspace, used, bias, residual, ln_w, ln_b, ln_eps, Y, M, N, relu, stream);
}

// fp16 hi + lo: X (fp32) is multiplied by 2^x_up_log2 and split in the kernel -- the CALLER guarantees |X| 2^x_up_log2 < 65504 (an
// overflow shows as Inf / NaN in Y, never silently); nscale = [256] 2^-ew[n] 2^-x_up_log2 (sbev_f16s_out_scale).
// x_is_pairs: X already holds the (hi, lo) pairs of x 2^x_up_log2 in its 32-bit slots (sbev_f16s_pairs, the *_pairs_f16 mixing launches)
extern "C" int sbev_linear_splitk_f16s(const float* X, int x_is_pairs, int x_up_log2, const uint16_t* Wp, const float* nscale, const float* bias, const float* residual,
                                       const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                                       int64_t M, int N, int K, int64_t ldx, int relu, int nprod, float* workspace, sbev_stream_t stream) {
    SBEV_REQUIRE(nprod == 3 || nprod == 4, "sbev_linear_splitk_f16s: nprod=%d (3 or 4 image products)", nprod);
    SBEV_REQUIRE(x_up_log2 >= -100 && x_up_log2 <= 100, "sbev_linear_splitk_f16s: x_up_log2=%d", x_up_log2);
    SBEV_REQUIRE(M >= 0 && sbev_linear_bf16s_out_ok(M > 0 ? M : 1, N, K), "sbev_linear_splitk_f16s: needs N == 256, K %% 32 == 0, K >= 256 (N=%d K=%d)", N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && Wp && Y && workspace && nscale && ldx % 4 == 0 && ldx >= K, "sbev_linear_splitk_f16s: bad pointers");
    SBEV_REQUIRE((((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)workspace | (uintptr_t)nscale) & 15) == 0, "sbev_linear_splitk_f16s: 16-byte alignment");
    int used = 0;
    const int st = sbev::launch_splitk_slabs_bf16s(X, Wp, M, K, ldx, nprod + 1, workspace, &used, reinterpret_cast<hipStream_t>(stream), x_up_log2, nscale,
                                                   x_is_pairs != 0);
    if (st != SBEV_OK) return st;
    return sbev_splitk_reduce_f32(workspace, used, bias, residual, ln_w, ln_b, ln_eps, Y, M, N, relu, stream);
}

// The same with X's scale in DEVICE memory (x_scale = {2^e, 2^-e}: sbev_f16s_tensor_scale or the maxima a producer kernel left):
// no host-side bound needed -- grad_x = grad_y . W of the parameter generator in training, whose grad_y has no a-priori magnitude.
// wdown = W's [N] down-scales (sbev_pack_f16s_frags scales + N); the kernel applies 2^-e itself.  X fp32 only.
extern "C" int sbev_linear_splitk_f16s_xdev(const float* X, const float* x_scale, const uint16_t* Wp, const float* wdown, const float* bias,
                                            const float* residual, const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                                            int64_t M, int N, int K, int64_t ldx, int relu, int nprod, float* workspace, sbev_stream_t stream) {
    SBEV_REQUIRE(nprod == 3 || nprod == 4, "sbev_linear_splitk_f16s_xdev: nprod=%d (3 or 4 image products)", nprod);
    SBEV_REQUIRE(M >= 0 && sbev_linear_bf16s_out_ok(M > 0 ? M : 1, N, K), "sbev_linear_splitk_f16s_xdev: needs N == 256, K %% 32 == 0, K >= 256 (N=%d K=%d)", N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && x_scale && Wp && Y && workspace && wdown && ldx % 4 == 0 && ldx >= K, "sbev_linear_splitk_f16s_xdev: bad pointers");
    SBEV_REQUIRE((((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)workspace | (uintptr_t)wdown) & 15) == 0, "sbev_linear_splitk_f16s_xdev: 16-byte alignment");
    int used = 0;
    const int st = sbev::launch_splitk_slabs_bf16s(X, Wp, M, K, ldx, nprod + 1, workspace, &used, reinterpret_cast<hipStream_t>(stream), 0, wdown, false, x_scale);
    if (st != SBEV_OK) return st;
    return sbev_splitk_reduce_f32(workspace, used, bias, residual, ln_w, ln_b, ln_eps, Y, M, N, relu, stream);
}

// nscale[n] = wdown[n] 2^-x_up_log2 (exact), the per-column factor of the fp16 out-projection's slabs
namespace {
__global__ void out_scale_kernel(const float* wdown, float f, float* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = wdown[i] * f;
}
__global__ void pairs_kernel(const float* x, unsigned* out, long long n, float up) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i] * up;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    out[i] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
}
// out[i] = (fp16 hi, fp16 lo) of x[i] 2^up_log2 in one 32-bit slot (hi in the low half): the pre-split operand format
extern "C" int sbev_f16s_pairs(const float* X, void* out, int64_t n, int up_log2, sbev_stream_t stream) {
    SBEV_REQUIRE(n >= 0 && up_log2 >= -100 && up_log2 <= 100, "sbev_f16s_pairs: bad arguments");
    if (n == 0) return SBEV_OK;
    SBEV_REQUIRE(X && out, "sbev_f16s_pairs: null pointer");
    hipLaunchKernelGGL(pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, static_cast<unsigned*>(out),
                       (long long)n, ldexpf(1.f, up_log2));
    return sbev::check_launch("sbev_f16s_pairs");
}
extern "C" int sbev_f16s_out_scale(const float* wdown, int x_up_log2, float* nscale, int N, sbev_stream_t stream) {
    SBEV_REQUIRE(wdown && nscale && N >= 1 && x_up_log2 >= -100 && x_up_log2 <= 100, "sbev_f16s_out_scale: bad arguments");
    hipLaunchKernelGGL(out_scale_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), wdown, ldexpf(1.f, -x_up_log2), nscale, N);
    return sbev::check_launch("sbev_f16s_out_scale");
}

// the out-projection's in-launch slab fold inside sbev_decoder_forward (default 0: measured slower, see g_out_fold; SBEV_OUT_FOLD=1 starts with 1); returns the previous setting
extern "C" int sbev_decoder_out_fold(int enable) { return sbev::out_fold_switch(enable); }
// test hook for the fold's poll bound (never set in production): one chunk-workgroup of row tile 0 leaves without arriving
extern "C" int sbev_debug_out_fold_drop(int enable) { return sbev::out_fold_drop(enable); }

// rows from which the pre-split out-projection runs on 256-row tiles (gemm_bf16s_out8_kernel; default 1024, SBEV_OUT8_MIN_ROWS in the
// environment; 0 = never, anything else is raised to 1024): returns the previous threshold.  A/B switch; results differ by fp32 summation order only.
extern "C" int sbev_linear_out8_min_rows(int rows) { return sbev::out8_min_rows(rows); }
