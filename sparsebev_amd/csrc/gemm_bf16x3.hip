// Opt-in fast "Linear" for the two 15-GFLOP GEMMs of adaptive mixing: fp32 operands split into bf16 (hi, lo)
// pairs, Y = Xhi.Whi + Xhi.Wlo + Xlo.Whi accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate,
// so 3 products = ~5x faster).  hi = RNE_bf16(x), lo = RNE_bf16(x - hi):
// x = hi + lo to 2^-18 relative, the dropped lo.lo term is <= 2^-18 relative per product -- an fp32-class result
// (measured 1.2e-5 max abs error on the reference's AdaptiveMixing fixture vs 6e-3 for plain bf16), NOT bit-equal
// to fp32 math: the exact path (gemm.hip) stays the default, this one is selected by sbev_decoder_config.gemm_mode.
//
// W is pre-split once (sbev_split_bf16x3_weights) into [N][K/8][hi 8 | lo 8] so that a 32-k slice of a row is one
// 128-B line; X (activations) is split on the fly while staging to LDS.  Tile 128x128x32, 2x2 waves x 2x2 MFMA
// tiles, double-buffered LDS (80-B rows: conflict-free ds_read_b128 fragments), one barrier per K-step.
#include "sbev_common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                       // LDS row stride in bytes (32 bf16 = 64 B + 16 B pad)
constexpr int ARR = 128 * ROWB;                // one [128][32] bf16 array
constexpr int STAGE = 4 * ARR;                 // A_hi, A_lo, B_hi, B_lo

struct G3Args {
    const float* X;              // [M, ldx] fp32
    const unsigned short* W2;    // [N, K/8, 2, 8] bf16 (hi block, lo block)
    const float* bias;
    const float* res;
    float* Y;                    // [M, ldy]  (split-K: [splits, M, N])
    long long M;
    int N, K;
    long long ldx, ldy;
    int k_per_split, relu;
};

__device__ __forceinline__ unsigned xcd_swizzle(unsigned id, unsigned n) {
    const unsigned q = n / 8, r = n % 8, x = id % 8, s = id / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// x = hi + lo with hi = RNE_bf16(x), lo = RNE_bf16(x - hi): |x - hi - lo| <= 2^-18 |x| and |lo| <= 2^-9 |x|, so the
// dropped lo.lo product is <= 2^-18 relative.  (A truncated hi would leave |lo| <= 2^-7 |x| and a 2^-14 lo.lo term.)
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
    hi.x = pack_bf16(v.x, v.y);
    hi.y = pack_bf16(v.z, v.w);
    const float r0 = v.x - __uint_as_float(hi.x << 16), r1 = v.y - __uint_as_float(hi.x & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(hi.y << 16), r3 = v.w - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = pack_bf16(r0, r1);
    lo.y = pack_bf16(r2, r3);
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const G3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 stages x 4 arrays, reused by the epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const unsigned tiles_n = (a.N + BN - 1) / BN, tiles_m = (unsigned)((a.M + BM - 1) / BM);
    const unsigned t = xcd_swizzle(blockIdx.x, tiles_m * tiles_n);
    const bool m_fast = tiles_n >= tiles_m;
    const unsigned tm = m_fast ? t % tiles_m : t / tiles_n;
    const unsigned tn = m_fast ? t / tiles_m : t % tiles_n;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;
    const int kbeg = SPLIT ? blockIdx.z * a.k_per_split : 0;
    const int kend = SPLIT ? min(a.K, kbeg + a.k_per_split) : a.K;
    const int nk = (kend - kbeg) / BK;

    // staging: A rows = tid/8 + 32 i (4 floats at (tid%8)*4); B rows = tid/8 + 32 i (16-B unit tid%8 of the 128-B slice)
    const int srow = tid >> 3, su = tid & 7;
    const float* xp[4];
    const unsigned short* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long long r = m0 + srow + 32 * i;
        r = r < a.M ? r : a.M - 1;
        xp[i] = a.X + r * a.ldx + su * 4;
        int c = n0 + srow + 32 * i;
        c = c < a.N ? c : a.N - 1;
        wp[i] = a.W2 + (long long)c * a.K * 2 + su * 8;      // row stride = K/8 * 16 elements = 2K
    }
    // staging registers are clang vector types on purpose: with HIP's struct-based uint4 this array went to
    // scratch memory (and every K-step waited for its own global loads before the MFMAs)
    f32x4 ra[4];
    u32x4 rb[4];
    auto gload = [&](int kt) {
        const int k = kbeg + kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(xp[i] + k);
            rb[i] = *reinterpret_cast<const u32x4*>(wp[i] + (long long)k * 2);
        }
    };
    const int a_off = srow * ROWB + su * 8;
    const int b_off = (2 + (su & 1)) * ARR + srow * ROWB + (su >> 1) * 16;   // unit su: k-block su/2, hi (even) or lo (odd)
    auto lstore = [&](int buf) {
        unsigned char* st = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 hi, lo;
            split4(ra[i], hi, lo);
            *reinterpret_cast<u32x2*>(st + a_off + 32 * i * ROWB) = hi;              // A_hi
            *reinterpret_cast<u32x2*>(st + ARR + a_off + 32 * i * ROWB) = lo;        // A_lo
            *reinterpret_cast<u32x4*>(st + b_off + 32 * i * ROWB) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    if (nk > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* st = lds + buf * STAGE;
        const unsigned char* Ab = st + (wr * 64 + fr) * ROWB + fh * 16;
        const unsigned char* Bb = st + 2 * ARR + (wc * 64 + fr) * ROWB + fh * 16;
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * ROWB + c * 32);
                al[i] = *reinterpret_cast<const bf16x8*>(Ab + ARR + i * 32 * ROWB + c * 32);
                bh[i] = *reinterpret_cast<const bf16x8*>(Bb + i * 32 * ROWB + c * 32);
                bl[i] = *reinterpret_cast<const bf16x8*>(Bb + ARR + i * 32 * ROWB + c * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue (same as gemm.hip): accumulators -> LDS -> row-major float4 pass
    constexpr int LDC = BN + 4;
    static_assert(BM * LDC * 4 <= 2 * STAGE, "C tile must fit in the staging LDS");
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                Cs[(wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * LDC + wc * 64 + j * 32 + fr] = acc[i][j][e];
    __syncthreads();
    float* Y = SPLIT ? a.Y + (long long)blockIdx.z * a.M * a.ldy : a.Y;
    const int er = tid / 32, ec = (tid % 32) * 4;
    const int n = n0 + ec;
    const bool vec = (a.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) && (n + 3 < a.N) &&
                     (SPLIT || !a.res || (reinterpret_cast<uintptr_t>(a.res) & 15) == 0);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (!SPLIT && a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (n + e) < a.N ? a.bias[n + e] : 0.f;
    }
#pragma unroll 4
    for (int it = 0; it < BM / 8; ++it) {
        const int r = er + it * 8;
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


// ---- W-stationary strip kernel for the parameter generator ([M, 256] x [N >= 16384, 256]^T), bf16x3 -----------------------
// The tile kernel above re-splits the same X rows in every one of the 256 column tiles and, with K = 256, spends its 8 K-steps
// mostly in prologue / epilogue (83 us for 45 GFLOP of bf16 MFMA: 20 % of the 2.5 PF peak; round-1 review item 8).  Same
// decomposition as gemm.hip's exact strip kernel instead: a WAVE owns a 64-column strip of W -- its (hi, lo) bf16 images, 256
// registers, at one wave per SIMD -- for half of all 16-row fragments, and the X rows stream past it straight from L2, already
// split ONCE per layer (sbev_split_bf16x3_weights on the [M, 256] activation: the same [row][K/8][hi 8 | lo 8] image as the
// weights, so a lane's operand for one MFMA is one 16-byte load).  v_mfma_f32_16x16x32_bf16 with W as the row operand: a lane
// ends up with 4 consecutive output columns of one row (float4 stores).  Per fragment and wave 8 k-steps x 4 column blocks x 3
// products = 96 MFMAs of 16 cycles against 256 x 32 cycles in the exact kernel; the floor is the 118 MB output write.
#define SBEV_ONE_WAVE_PER_EU __attribute__((amdgpu_waves_per_eu(1, 1)))
#ifdef SBEV_EXP_NOSTORE          // experiments (never in the product build): drop the stores / keep the X rows L1-resident
#define SBEV_EXP_STORE_COND && a.M < 0
#else
#define SBEV_EXP_STORE_COND
#endif
#ifdef SBEV_EXP_NOLOAD
#define SBEV_EXP_ROW(r) ((r) & 15)
#else
#define SBEV_EXP_ROW(r) (r)
#endif
struct StripArgs {
    const unsigned short* X2;    // [M, 32, 2, 8] bf16 (K = 256)
    const unsigned short* W2;    // [N, 32, 2, 8]
    const float* bias;           // [N] or null
    float* Y;                    // [M, ldy]
    int M;
    long long ldy;
};

template <bool RELU>
__global__ __launch_bounds__(256) SBEV_ONE_WAVE_PER_EU void gemm_bf16x3_strip_kernel(const StripArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fk = lane >> 4;
    const long long n0 = (long long)blockIdx.x * 128 + (wave >> 1) * 64;
    const int half = wave & 1;

    // k order of an MFMA: lane group fk owns k = 32 s + 8 fk + (0..7) = block 4 s + fk of the row's image, for both operands
    bf16x8 wh[8][4], wl[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
            const unsigned short* p = a.W2 + ((n0 + cf * 16 + fi) * 32 + 4 * s + fk) * 16;
            wh[s][cf] = *reinterpret_cast<const bf16x8*>(p);
            wl[s][cf] = *reinterpret_cast<const bf16x8*>(p + 8);
        }
    // the bias (the accumulators' start value) waits in LDS: 16 registers less at a budget of 512
    __shared__ __attribute__((aligned(16))) float bias_s[4][64];
    bias_s[wave][lane] = a.bias ? a.bias[n0 + lane] : 0.f;
    __builtin_amdgcn_wave_barrier();

    const int M = a.M;
    const int last = ((M + 15) >> 4) - 1;
    bf16x8 xah[8], xal[8], xbh[8], xbl[8];
#define SBEV_LOAD_X3(dh, dl, f)                                                                     \
    {                                                                                               \
        int row_ = SBEV_EXP_ROW(16 * (f) + fi);                                                     \
        row_ = row_ < M ? row_ : M - 1;                                                             \
        const unsigned short* p_ = a.X2 + ((long long)row_ * 32 + fk) * 16;                         \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                             \
            dh[s] = *reinterpret_cast<const bf16x8*>(p_ + s * 64);                                  \
            dl[s] = *reinterpret_cast<const bf16x8*>(p_ + s * 64 + 8);                              \
        }                                                                                           \
    }
    // small terms first (lo.hi, hi.lo), then hi.hi; the four column blocks rotate so that no MFMA waits for its predecessor
#define SBEV_STRIP3(sh, sl)                                                                         \
    {                                                                                               \
        _Pragma("unroll") for (int cf = 0; cf < 4; ++cf) acc[cf] = *reinterpret_cast<const f32x4*>(&bias_s[wave][cf * 16 + 4 * fk]); \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                             \
            _Pragma("unroll") for (int cf = 0; cf < 4; ++cf) acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s][cf], sh[s], acc[cf], 0, 0, 0); \
            _Pragma("unroll") for (int cf = 0; cf < 4; ++cf) acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s][cf], sl[s], acc[cf], 0, 0, 0); \
            _Pragma("unroll") for (int cf = 0; cf < 4; ++cf) acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s][cf], sh[s], acc[cf], 0, 0, 0); \
        }                                                                                           \
    }
#define SBEV_STORE3(f)                                                                              \
    {                                                                                               \
        const int row_ = 16 * (f) + fi;                                                             \
        if (row_ < M SBEV_EXP_STORE_COND) {                                                         \
            float* y_ = a.Y + (long long)row_ * a.ldy + n0 + 4 * fk;                                \
            _Pragma("unroll") for (int cf = 0; cf < 4; ++cf) {                                      \
                f32x4 v_ = acc[cf];                                                                 \
                if (RELU) { v_[0] = fmaxf(v_[0], 0.f); v_[1] = fmaxf(v_[1], 0.f); v_[2] = fmaxf(v_[2], 0.f); v_[3] = fmaxf(v_[3], 0.f); } \
                *reinterpret_cast<f32x4*>(y_ + cf * 16) = v_;                                       \
            }                                                                                       \
        }                                                                                           \
    }
    // fragments alternate between the two waves of a strip; two register sets used alternately, the next fragment's rows
    // requested before each MFMA block, the stores of fragment f issued at the top of block f + 1 (gemm.hip's strip kernel)
    f32x4 acc[4];
    int f = half ^ ((wave >> 1) & 1);
    if (f > last) return;
    int fprev = f;
    SBEV_LOAD_X3(xah, xal, f);
    { const int fn = f + 2 <= last ? f + 2 : last; SBEV_LOAD_X3(xbh, xbl, fn); }
    __builtin_amdgcn_sched_barrier(0);
    SBEV_STRIP3(xah, xal);
    f += 2;
    while (f <= last) {
        __builtin_amdgcn_sched_barrier(0);
        { const int fn = f + 2 <= last ? f + 2 : last; SBEV_LOAD_X3(xah, xal, fn); }
        SBEV_STORE3(fprev);
        __builtin_amdgcn_sched_barrier(0);
        SBEV_STRIP3(xbh, xbl);
        fprev = f;
        f += 2;
        if (f > last) break;
        __builtin_amdgcn_sched_barrier(0);
        { const int fn = f + 2 <= last ? f + 2 : last; SBEV_LOAD_X3(xbh, xbl, fn); }
        SBEV_STORE3(fprev);
        __builtin_amdgcn_sched_barrier(0);
        SBEV_STRIP3(xah, xal);
        fprev = f;
        f += 2;
    }
    SBEV_STORE3(fprev);
#undef SBEV_STORE3
#undef SBEV_LOAD_X3
#undef SBEV_STRIP3
}

// W [N,K] fp32 -> [N, K/8, 2, 8] bf16: one thread per 8-k block
__global__ void split_weights_kernel(const float* w, unsigned short* out, long long n_blocks) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + i * 8 + 4);
    u32x2 h0, l0, h1, l1;
    split4(a, h0, l0);
    split4(b, h1, l1);
    u32x4* o = reinterpret_cast<u32x4*>(out + i * 16);
    o[0] = (u32x4){h0.x, h0.y, h1.x, h1.y};
    o[1] = (u32x4){l0.x, l0.y, l1.x, l1.y};
}

int set_lds(const void* fn) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    if (e != hipSuccess) {
        sbev::set_error("gemm_bf16x3: cannot reserve %d B of LDS: %s", 2 * STAGE, hipGetErrorString(e));
        return SBEV_ELAUNCH;
    }
    return SBEV_OK;
}

}  // namespace

extern "C" int sbev_split_bf16x3_weights(const float* W, uint16_t* W2, int64_t N, int K, sbev_stream_t stream) {
    SBEV_REQUIRE(N >= 0 && K >= 8 && K % 8 == 0, "sbev_split_bf16x3_weights: K=%d must be a multiple of 8", K);
    if (N == 0) return SBEV_OK;
    SBEV_REQUIRE(W && W2 && (((uintptr_t)W | (uintptr_t)W2) & 15) == 0, "sbev_split_bf16x3_weights: null / unaligned pointer");
    const long long nb = N * (K / 8);
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, W2, nb);
    return sbev::check_launch("sbev_split_bf16x3_weights");
}

extern "C" int sbev_linear_bf16x3(const float* X, const uint16_t* W2, const float* bias, const float* residual, float* Y,
                                  int64_t M, int N, int K, int64_t ldx, int64_t ldy, int relu, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 1 && K >= BK && K % BK == 0, "sbev_linear_bf16x3: K=%d must be a multiple of %d", K, BK);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && W2 && Y && ldx % 4 == 0 && ldx >= K && ldy >= N, "sbev_linear_bf16x3: bad pointers / leading dimensions");
    SBEV_REQUIRE((((uintptr_t)X | (uintptr_t)W2) & 15) == 0, "sbev_linear_bf16x3: 16-byte alignment");
    G3Args a{X, W2, bias, residual, Y, M, N, K, ldx, ldy, K, relu};
    const long long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    SBEV_REQUIRE(tiles <= 0x7fffffffLL, "sbev_linear_bf16x3: too many tiles");
    auto k = gemm_bf16x3_kernel<false>;
    int st = set_lds(reinterpret_cast<const void*>(k));
    if (st != SBEV_OK) return st;
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256), 2 * STAGE, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_linear_bf16x3");
}


// The strip kernel's shape: the parameter generator of adaptive mixing (K = 256, wide N), any M.
extern "C" int sbev_linear_bf16x3_strip_ok(int64_t M, int N, int K) { return M >= 1 && M < 0x7fffffffLL / 16 && K == 256 && N >= 1024 && N % 128 == 0; }

// Y = X W^T + bias with X ALREADY split (X2 = sbev_split_bf16x3_weights image of the [M, 256] activation)
extern "C" int sbev_linear_bf16x3_strip(const uint16_t* X2, const uint16_t* W2, const float* bias, float* Y, int64_t M, int N, int K,
                                        int64_t ldy, int relu, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && sbev_linear_bf16x3_strip_ok(M > 0 ? M : 1, N, K), "sbev_linear_bf16x3_strip: needs K = 256 and N %% 128 == 0, N >= 1024 (N=%d K=%d)", N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X2 && W2 && Y && ldy >= N && ldy % 4 == 0, "sbev_linear_bf16x3_strip: bad pointers / leading dimension");
    SBEV_REQUIRE((((uintptr_t)X2 | (uintptr_t)W2 | (uintptr_t)Y) & 15) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0), "sbev_linear_bf16x3_strip: 16-byte alignment");
    StripArgs a{X2, W2, bias, Y, (int)M, (long long)ldy};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    const bool prof = sbev::profile_begin(s, &e0, &e1, 1);
    if (relu) hipLaunchKernelGGL(gemm_bf16x3_strip_kernel<true>, dim3((unsigned)(N / 128)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gemm_bf16x3_strip_kernel<false>, dim3((unsigned)(N / 128)), dim3(256), 0, s, a);
    if (prof) sbev::profile_end(s, e0, e1, 1);
    return sbev::check_launch("sbev_linear_bf16x3_strip");
}

namespace sbev {
// the GEMM half of sbev_linear_splitk_bf16x3: *used partial slabs [used, M, N], not reduced
int launch_splitk_slabs_bf16x3(const float* X, const uint16_t* W2, int64_t M, int N, int K, int64_t ldx, int splits,
                               float* workspace, int* used, hipStream_t s) {
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    *used = (K + kps - 1) / kps;
    G3Args a{X, W2, nullptr, nullptr, workspace, M, N, K, ldx, (long long)N, kps, 0};
    const long long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    auto k = gemm_bf16x3_kernel<true>;
    int st = set_lds(reinterpret_cast<const void*>(k));
    if (st != SBEV_OK) return st;
    hipLaunchKernelGGL(k, dim3((unsigned)tiles, 1, (unsigned)*used), dim3(256), 2 * STAGE, s, a);
    return check_launch("sbev_linear_splitk_bf16x3 (gemm)");
}
}  // namespace sbev

extern "C" int sbev_linear_splitk_bf16x3(const float* X, const uint16_t* W2, const float* bias, const float* residual,
                                         const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                                         int64_t M, int N, int K, int64_t ldx, int relu, int splits, float* workspace,
                                         sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024 && K >= BK && K % BK == 0, "sbev_linear_splitk_bf16x3: bad sizes");
    SBEV_REQUIRE(splits >= 1 && splits <= 1024, "sbev_linear_splitk_bf16x3: splits=%d", splits);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(X && W2 && Y && workspace && ldx % 4 == 0 && ldx >= K, "sbev_linear_splitk_bf16x3: bad pointers");
    int used;
    int st = sbev::launch_splitk_slabs_bf16x3(X, W2, M, N, K, ldx, splits, workspace, &used, reinterpret_cast<hipStream_t>(stream));
    if (st != SBEV_OK) return st;
    // the slab reducer (bias + residual + LayerNorm) is shared with the exact path
    return sbev_splitk_reduce_f32(workspace, used, bias, residual, ln_w, ln_b, ln_eps, Y, M, N, relu, stream);
}
