// Split-bf16 Linears for the two 15-GFLOP GEMMs of adaptive mixing on the bf16 matrix core (round 3):
//   NIMG = 3  "bf16x6": x = hi + mid + lo (three RNE bf16 images, 8 + 8 + 8 significand bits = an exact split of an fp32
//             value), Y = sum of the SIX image products whose weight is >= 2^-16 (hh, hm, mh, mm, hl, lh), fp32 accumulation
//             on v_mfma_f32_32x32x16_bf16.  Dropped: ml, lm, ll <= 2^-23 |a b| per product -- below the rounding an fp32 fma
//             chain commits per step, so the result is fp32-class ("not narrower than fp32": tests/test_gpu_bf16s.py compares
//             both against fp64).  6 x 15.1 GFLOP at the 2.5 PF bf16 peak = 36 us, against a 96 us floor on the f32 MFMA.
//   NIMG = 2  "bf16x3": hi + lo, three products (2^-16 class), the opt-in fast mode of rounds 1-2 on the same kernels.
// Replaces nothing in the reference (torch.nn.Linear, models/sparsebev_transformer.py:358,378); same semantics as
// sbev_linear_f32 / sbev_linear_splitk_f32.
//
// Why these kernels and not gemm_bf16x3.hip's: a 16-byte-per-lane VGPR write-back (global load or ds_read_b128) costs the issuing
// wave ~85 matrix-pipe cycles when it is alone on its SIMD and 23-38 with a partner wave, so the
// one-wave-per-SIMD register-stationary strips that win for 32-cycle-per-16x16x4 f32 MFMAs are load-issue-bound at bf16 rates
// (72 us for 45 GFLOP = 25 % of peak).  Here every workgroup is 8 waves = two per SIMD, operands are read as whole 1-KiB
// fragments, and a wave owns a 64 x 64 output tile so that one fragment read feeds 2 (x3) ... 6 (x6) MFMAs of 32 cycles.
//
//   generator  Y[M, N] = X[M, K] W[N, K]^T + b   (K = 256, N = 32768 ...):  workgroup tile <= 128 rows x 256 columns, K slabs of
//              32 through a double-buffered LDS stage filled by global_load_lds_dwordx4 (both operands pre-split into row-major
//              bf16 planes: X once per layer by sbev_split_bf16s_rows, W once per weight update), 16-B chunks XOR-swizzled on
//              the SOURCE address so that the lane-linear LDS image is conflict-free for ds_read_b128 fragments.  Row tiles
//              are 3 or 4 fragments of 32 rows, balanced (900 rows = 5 x 4 + 3 x 3 fragments: 3 % padding instead of 12 %).
//   out-proj   slabs[S, M, 256] = X[M, K] W[256, K]^T over S K-chunks (K = 32768): workgroup = 64 rows x all 256 columns x one
//              chunk, its two wave quartets take the two halves of the chunk and fold through LDS (S slabs instead of 2 S).
//              X (fp32, the mixing kernel's output) is split in the kernel -- each element exactly once -- into an LDS stage;
//              W fragments come pre-packed in MFMA order (sbev_pack_bf16s_frags: 1 KiB per (32 columns, 16 k, image)), straight
//              from L2 into registers (no sharing between waves to exploit: every wave owns its own 64 columns).
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
#define SBEV_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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

// products in the order they are accumulated (small terms first inside a k-step): image of X, image of W
//   x3: (h,l) (l,h) (h,h)                 x6: (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
template <int NIMG>
struct Prods {
    static constexpr int N = NIMG == 2 ? 3 : 6;
    __host__ __device__ static constexpr int ia(int p) { return NIMG == 2 ? (p == 1 ? 1 : 0) : (p == 1 ? 2 : (p == 2 || p == 4) ? 1 : 0); }
    __host__ __device__ static constexpr int ib(int p) { return NIMG == 2 ? (p == 0 ? 1 : 0) : (p == 0 ? 2 : (p == 2 || p == 3) ? 1 : 0); }
};

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
template <int NIMG>
__global__ void pack_frags_kernel(const float* w, long long ldw, unsigned short* out, int N, int K) {
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
    split8<NIMG>(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), im);
#pragma unroll
    for (int img = 0; img < NIMG; ++img)
        *reinterpret_cast<u32x4*>(out + ((f * NIMG + img) * 64 + lane) * 8) = im[img];
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
};

constexpr int G_ROWS = 128, G_COLS = 256;

// ---- generator: persistent workgroups, 16-k stages in a 4-deep LDS ring, fragments one stage ahead ----------------------------------
// History (c2, bf16x6): the first version -- one 128 x 256 tile per workgroup, 32-k slabs, two LDS stages
// filled from ROW-MAJOR bf16 planes, one barrier per slab -- ran 116 us; without its stores 93, without its MFMAs 73, with neither
// 50: the three parts add up, NOTHING overlapped (one workgroup per CU, every wave in the same phase behind the barrier, the
// fragment reads of a k-step in front of its MFMAs, the 128 KB epilogue of a tile before the next tile's first load), and the
// 64-byte-per-row pieces of a row-major operand made every LDS-DMA instruction touch 16 cache lines (32 at 16-k stages).  Here
//   * both operands are pre-packed in MFMA fragment order (sbev_pack_bf16s_frags): a stage is made of whole 1-KiB fragments, an
//     LDS-DMA instruction copies one of them verbatim (8 full cache lines, lane-linear) and a ds_read_b128 at lane x 16 reads
//     it back conflict-free -- no swizzle, no address arithmetic per lane;
//   * a workgroup is persistent (grid = CUs) and walks its tiles as ONE stream of 16-k stages: the LDS-DMA loads run three stages
//     ahead across tile boundaries, the epilogue stores of a tile drain under the next tile's MFMAs;
//   * the fragments of stage g + 1 are read (interleaved 1 : 2 by sched_group_barrier) among the MFMAs of stage g -- legal because
//     the ring is 4 deep: stage g + 1 was published by the barrier at the top of iteration g, stage g + 3 is being filled;
//   * counted waits: at the top of iteration g only the newest stage's loads of this wave may be outstanding (vector loads
//     return in order, so "at most n outstanding" = everything older than the newest n has landed; stores in flight can only
//     make the wait more conservative, and the two iterations behind an epilogue skip it: the epilogue drained the queue).
constexpr int G2_ST_A = G_ROWS * 32, G2_ST_B = G_COLS * 32;      // bytes of one image of one 16-k stage (32 B per row)
constexpr int G2_NST = 4;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
}

template <int NIMG>
__global__ __launch_bounds__(512) void gemm_bf16s_gen2_kernel(const GenArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];    // the only LDS object: its byte address is 0
    constexpr int STAGE = NIMG * (G2_ST_A + G2_ST_B);
    constexpr int NQ = NIMG * 12;                                    // wave-loads (32 rows x 32 B) per stage
    constexpr int NLMAX = (NQ + 7) / 8;
    typedef Prods<NIMG> PR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int M = a.M, K = a.K, nk = K / 16;
    // A workgroup keeps ONE row tile for its whole life (grid is a multiple of ntm) and walks column tiles: the fragment count
    // of each wave is then a kernel-lifetime constant and the stage loop is instantiated per count (a per-tile branch around
    // the MFMA block made hipcc copy all 64 accumulators per stage).
    const int lw = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int rt = lw % a.ntm, ct0 = lw / a.ntm, cstep = (int)gridDim.x / a.ntm, nct = a.N / G_COLS;
    const int my_tiles = ct0 < nct ? (nct - ct0 + cstep - 1) / cstep : 0;
    const int G = my_tiles * nk;                                     // stages of this workgroup (even: K % 32 == 0)
    if (G == 0) return;
    const int f0 = rt * a.base + (rt < a.rem ? rt : a.rem);
    const int nf = a.base + (rt < a.rem ? 1 : 0);
    const int m0 = f0 * 32;
    int nfa = nf - 2 * wr;                                           // this wave's row fragments: 0, 1 or 2
    nfa = nfa < 0 ? 0 : (nfa > 2 ? 2 : nfa);
    const int nl = (NQ - wave + 7) / 8;                              // this wave's loads per stage

    // ---- load cursor: load q = wave + 8 j is one fragment = (image, 32-row block) of A (this tile's rows: fixed) or of B (the
    // column tile); in memory the fragments of a row block are [k-step][image][1 KiB] --------------------------------------------
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned char* gbase[NLMAX];
    unsigned ldst[NLMAX];
    long long gstep[NLMAX];
    const int nfrag = (M + 31) / 32;
    const long long blkbytes = (long long)nk * NIMG * 1024;          // one 32-row block, all k-steps and images
#pragma unroll
    for (int j = 0; j < NLMAX; ++j) {
        const int q = wave + 8 * j;
        if (q < NIMG * 4) {
            const int img = q >> 2, blk = q & 3;
            int fb = f0 + blk;
            fb = fb < nfrag ? fb : nfrag - 1;
            gbase[j] = reinterpret_cast<const unsigned char*>(a.Xs) + fb * blkbytes + img * 1024;
            ldst[j] = (unsigned)(img * G2_ST_A + blk * 1024);
            gstep[j] = -(long long)nk * NIMG * 1024;                  // next tile: the same rows again
        } else {
            const int q2 = q - NIMG * 4;
            const int img = q2 >> 3, blk = q2 & 7;
            gbase[j] = reinterpret_cast<const unsigned char*>(a.Ws) + (long long)(ct0 * 8 + blk) * blkbytes + img * 1024;
            ldst[j] = (unsigned)(NIMG * G2_ST_A + img * G2_ST_B + blk * 1024);
            gstep[j] = (long long)(cstep * 8) * blkbytes - (long long)nk * NIMG * 1024;      // to the next column tile of this workgroup
        }
    }
    int lk = 0, lg = 0;                                              // load cursor: k-step in its tile, stage index
    auto issue_next = [&]() {
        if (lg >= G) return;
        const unsigned sb = (unsigned)((lg & (G2_NST - 1)) * STAGE);
#pragma unroll
        for (int j = 0; j < NLMAX; ++j) {
            if (j < nl) glds16(gbase[j], voff, sb + ldst[j]);
            gbase[j] += NIMG * 1024;                                 // next k-step of the same row block
        }
        ++lg;
        if (++lk == nk) {
            lk = 0;
#pragma unroll
            for (int j = 0; j < NLMAX; ++j) gbase[j] += gstep[j];
        }
    };

    // ---- compute side -----------------------------------------------------------------------------------------------------------
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned aoff = (unsigned)(wr * 2) * 1024u + voff;
    const unsigned boff = (unsigned)(NIMG * G2_ST_A) + (unsigned)(wc * 2) * 1024u + voff;

    auto run = [&](auto nfa_c) {
        constexpr int NFA = decltype(nfa_c)::value;
        constexpr int NFR = NFA > 0 ? NFA : 1;
        bf16x8 xf[2][NFR][NIMG], wf[2][2][NIMG];                     // [set][fragment][image]
        f32x16 acc[NFR][2];
        auto read_frags = [&](int g, int set) {
            if constexpr (NFA > 0) {
                const unsigned char* st = lds + (g & (G2_NST - 1)) * STAGE;
#pragma unroll
                for (int img = 0; img < NIMG; ++img) {
                    wf[set][0][img] = *reinterpret_cast<const bf16x8*>(st + boff + img * G2_ST_B);
                    wf[set][1][img] = *reinterpret_cast<const bf16x8*>(st + boff + img * G2_ST_B + 1024);
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa) xf[set][fa][img] = *reinterpret_cast<const bf16x8*>(st + aoff + img * G2_ST_A + fa * 1024);
                }
            }
        };
        auto init_acc = [&](int n0) {
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n0 + (wc * 2 + fb) * 32 + 8 * gq + 4 * lh);
#pragma unroll
                    for (int fa = 0; fa < NFR; ++fa)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[fa][fb][4 * gq + e] = bv[e];
                }
        };
        int skip = 0;
        auto top = [&](int g) {
            // stage g + 1 (or g, at the very end) of this wave has landed: only the newest stage's loads may be outstanding
            if (skip > 0) --skip;
            else if (lg - 1 > g + 1 || (g + 1 >= G && lg - 1 > g)) {
                if (nl == 5) wait_vmcnt<5>(); else if (nl == 4) wait_vmcnt<4>(); else wait_vmcnt<3>();
            } else wait_vmcnt<0>();
            __syncthreads();
            issue_next();                                            // stage g + 3 into the buffer stage g - 1 left
        };
        // MFMAs of the current set with the reads of the next set interleaved (1 read per 2 MFMAs at two fragments)
#define SBEV_G2_BODY(CUR)                                                                                   \
        if constexpr (NFA > 0) {                                                                            \
            _Pragma("unroll") for (int p = 0; p < PR::N; ++p)                                               \
                _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                          \
                    _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                        \
                        acc[fa][fb] = SBEV_MFMA(wf[CUR][fb][PR::ib(p)], xf[CUR][fa][PR::ia(p)], acc[fa][fb]); \
            _Pragma("unroll") for (int i = 0; i < (2 + NFA) * NIMG; ++i) {                                  \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
                __builtin_amdgcn_sched_group_barrier(0x008, (PR::N * NFA * 2) / ((2 + NFA) * NIMG), 0);     \
            }                                                                                               \
        }
        int cn0 = ct0 * G_COLS, ck = 0;
        init_acc(cn0);
        issue_next();
        issue_next();
        issue_next();
        top(0);
        read_frags(0, 0);
        for (int g = 0; g < G; g += 2) {
            read_frags(g + 1, 1);
            SBEV_G2_BODY(0)
            top(g + 1);
            if (g + 2 < G) read_frags(g + 2, 0);
            SBEV_G2_BODY(1)
            ck += 2;
            if (ck == nk) {
                // epilogue of this tile: drain this wave's loads first (the next two top-of-iteration waits are then skipped and
                // never wait for these stores), then 16-byte stores: a lane holds 4 consecutive columns of one row per 4 registers
                wait_vmcnt<0>();
                if constexpr (NFA > 0) {
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa) {
                        const int row = m0 + (wr * 2 + fa) * 32 + l31;
                        if (row < M SBEV_EXP_STORE_COND) {
                            float* y = a.Y + (long long)row * a.ldy + cn0 + wc * 64 + 4 * lh;
#pragma unroll
                            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                                for (int gq = 0; gq < 4; ++gq) {
                                    f32x4 v = {acc[fa][fb][4 * gq], acc[fa][fb][4 * gq + 1], acc[fa][fb][4 * gq + 2], acc[fa][fb][4 * gq + 3]};
                                    if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                                    *reinterpret_cast<f32x4*>(y + fb * 32 + 8 * gq) = v;
                                }
                        }
                    }
                }
                ck = 0;
                skip = 2;
                cn0 += cstep * G_COLS;
                if (cn0 < a.N) init_acc(cn0);
            }
            if (g + 2 < G) top(g + 2);
        }
#undef SBEV_G2_BODY
    };
    if (nfa == 2) run(std::integral_constant<int, 2>{});
    else if (nfa == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
}

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
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_n(int n) {          // n wave-uniform, 0 .. 10
    switch (n) {
        case 0: wait_vmcnt_imm<0>(); break;
        case 3: wait_vmcnt_imm<3>(); break;
        case 4: wait_vmcnt_imm<4>(); break;
        case 5: wait_vmcnt_imm<5>(); break;
        case 6: wait_vmcnt_imm<6>(); break;
        case 8: wait_vmcnt_imm<8>(); break;
        case 10: wait_vmcnt_imm<10>(); break;
        default: wait_vmcnt_imm<0>(); break;
    }
}

template <int NIMG>
__global__ __launch_bounds__(512) void gemm_bf16s_gen3_kernel(const GenArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];    // the only LDS object: its byte address is 0
    constexpr int STAGE = NIMG * (G2_ST_A + G2_ST_B);
    constexpr int NQ = NIMG * 12;                                    // fragments (1 KiB) per stage
    constexpr int NLMAX = (NQ + 7) / 8;
    typedef Prods<NIMG> PR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                         // wr: row half = phase group (waves w, w + 4 share a SIMD)
    const int M = a.M, nk = a.K / 16;
    const int lw = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int rt = lw % a.ntm, ct0 = lw / a.ntm, cstep = (int)gridDim.x / a.ntm, nct = a.N / G_COLS;
    const int my_tiles = ct0 < nct ? (nct - ct0 + cstep - 1) / cstep : 0;
    const int G = my_tiles * nk;                                     // stages of this workgroup
    if (G == 0) return;
    const int f0 = rt * a.base + (rt < a.rem ? rt : a.rem);
    const int nf = a.base + (rt < a.rem ? 1 : 0);
    const int m0 = f0 * 32;
    int nfa = nf - 2 * wr;                                           // this wave's row fragments: 0, 1 or 2 (fixed for its life)
    nfa = nfa < 0 ? 0 : (nfa > 2 ? 2 : nfa);
    const int nl = (NQ - wave + 7) / 8;                              // this wave's loads per stage

    const unsigned voff = (unsigned)lane * 16u;
    const unsigned char* gbase[NLMAX];
    unsigned ldst[NLMAX];
    long long gstep[NLMAX];
    const int nfrag = (M + 31) / 32;
    const long long blkbytes = (long long)nk * NIMG * 1024;          // one 32-row block, all k-steps and images
#pragma unroll
    for (int j = 0; j < NLMAX; ++j) {
        const int q = wave + 8 * j;
        if (q < NIMG * 4) {
            const int img = q >> 2, blk = q & 3;
            int fb = f0 + blk;
            fb = fb < nfrag ? fb : nfrag - 1;
            gbase[j] = reinterpret_cast<const unsigned char*>(a.Xs) + fb * blkbytes + img * 1024;
            ldst[j] = (unsigned)(img * G2_ST_A + blk * 1024);
            gstep[j] = -(long long)nk * NIMG * 1024;
        } else {
            const int q2 = q - NIMG * 4;
            const int img = q2 >> 3, blk = q2 & 7;
            gbase[j] = reinterpret_cast<const unsigned char*>(a.Ws) + (long long)(ct0 * 8 + blk) * blkbytes + img * 1024;
            ldst[j] = (unsigned)(NIMG * G2_ST_A + img * G2_ST_B + blk * 1024);
            gstep[j] = (long long)(cstep * 8) * blkbytes - (long long)nk * NIMG * 1024;
        }
    }
    int lk = 0, lg = 0;                                              // load cursor: k-step in its tile, next stage to issue
    auto issue_next = [&]() {
        if (lg >= G) return;
        const unsigned sb = (unsigned)((lg & (G2_NST - 1)) * STAGE);
#pragma unroll
        for (int j = 0; j < NLMAX; ++j) {
            if (j < nl) glds16(gbase[j], voff, sb + ldst[j]);
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
    const unsigned aoff = (unsigned)(wr * 2) * 1024u + voff;
    const unsigned boff = (unsigned)(NIMG * G2_ST_A) + (unsigned)(wc * 2) * 1024u + voff;

    auto run = [&](auto nfa_c) {
        constexpr int NFA = decltype(nfa_c)::value;
        constexpr int NFR = NFA > 0 ? NFA : 1;
        bf16x8 xf[NFR][NIMG], wf[2][NIMG];
        f32x16 acc[NFR][2], accs[NFR][2];                            // hi x hi (+ bias) | the small products
        auto init_acc = [&](int ti) {                                // tile ti of this workgroup: its bias slice waits in LDS
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(lds + G2_NST * STAGE + (ti * G_COLS + (wc * 2 + fb) * 32 + 8 * gq + 4 * lh) * 4);
#pragma unroll
                    for (int fa = 0; fa < NFR; ++fa)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc[fa][fb][4 * gq + e] = bv[e]; accs[fa][fb][4 * gq + e] = 0.f; }
                }
        };
        auto store_tile = [&](int n0) {                              // acc already holds hi x hi + small products
            if constexpr (NFA > 0) {
                if (a.relu) {                                        // one uniform branch, not one per store
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[fa][fb][e] = fmaxf(acc[fa][fb][e], 0.f);
                }
#pragma unroll
                for (int fa = 0; fa < NFA; ++fa) {
                    const int row = m0 + (wr * 2 + fa) * 32 + l31;
                    if (row < M SBEV_EXP_STORE_COND) {
                        float* y = a.Y + (long long)row * a.ldy + n0 + wc * 64 + 4 * lh;
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                const f32x4 v = {acc[fa][fb][4 * gq], acc[fa][fb][4 * gq + 1], acc[fa][fb][4 * gq + 2], acc[fa][fb][4 * gq + 3]};
                                *reinterpret_cast<f32x4*>(y + fb * 32 + 8 * gq) = v;
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
            reinterpret_cast<float*>(lds + G2_NST * STAGE)[i] = a.bias ? a.bias[(ct0 + t * cstep) * G_COLS + c] : 0.f;
        }
        __syncthreads();
        init_acc(0);
        issue_next();
        issue_next();
        issue_next();
        wait_vmcnt_imm<0>();
        __syncthreads();
        if (wr == 1) __syncthreads();                                // the second row-half runs one phase behind
        for (int g = 0; g < G; ++g) {
            // ---- FETCH(g): loads of stage g + 3; stage g + 1 of this wave landed (only newer loads may be outstanding: vector
            // loads return in order; a store in flight can only make the wait longer); fragments of stage g -> registers
            issue_next();
            {
                const int hi = lg - 1, need = g + 1 < G ? g + 1 : g;
                wait_vmcnt_n(hi > need ? (hi - need) * nl : 0);
            }
            if constexpr (NFA > 0) {
                const unsigned char* st = lds + (g & (G2_NST - 1)) * STAGE;
#pragma unroll
                for (int img = 0; img < NIMG; ++img) {
                    wf[0][img] = *reinterpret_cast<const bf16x8*>(st + boff + img * G2_ST_B);
                    wf[1][img] = *reinterpret_cast<const bf16x8*>(st + boff + img * G2_ST_B + 1024);
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa) xf[fa][img] = *reinterpret_cast<const bf16x8*>(st + aoff + img * G2_ST_A + fa * 1024);
                }
            }
            if (pending) {                                           // the previous tile's stores ride in this phase
                store_tile(cn0);
                cn0 += cstep * G_COLS;
                ++ti;
                init_acc(ti < my_tiles ? ti : 0);
                pending = false;
            }
            __syncthreads();
            // ---- COMPUTE(g): nothing but MFMAs (the partner wave of this SIMD is in its FETCH phase)
            if constexpr (NFA > 0) {
#pragma unroll
                for (int p = 0; p < PR::N - 1; ++p)
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb)
                            accs[fa][fb] = SBEV_MFMA(wf[fb][PR::ib(p)], xf[fa][PR::ia(p)], accs[fa][fb]);
#pragma unroll
                for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb)
                        acc[fa][fb] = SBEV_MFMA(wf[fb][0], xf[fa][0], acc[fa][fb]);
            }
            if (++ck == nk) {
                ck = 0;
                pending = true;
                if constexpr (NFA > 0) {
#pragma unroll
                    for (int fa = 0; fa < NFA; ++fa)
#pragma unroll
                        for (int fb = 0; fb < 2; ++fb) acc[fa][fb] += accs[fa][fb];
                }
            }
            __syncthreads();
        }
        if (wr == 0) __syncthreads();
        if (pending) store_tile(cn0);
    };
    if (nfa == 2) run(std::integral_constant<int, 2>{});
    else if (nfa == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
}

// ==== out-projection-shaped split-K GEMM (N = 256) ================================================================================
struct OutArgs {
    const float* X;              // [M, ldx] fp32
    const unsigned short* Wp;    // [8][K/16][NIMG][64][8] bf16 fragments
    float* P;                    // [S, M, 256] partial slabs
    int M, K;
    long long ldx;
    int nrt, S;                  // row tiles of 64 rows, K chunks
};

constexpr int O_IMG = 64 * 64;                  // bytes of one image of one half's stage (64 rows x 32 k)

// ---- out-projection, version 2: 3-deep X stage ring, fragments one k-step ahead, X two slabs ahead in registers ------------------
// Same ablation as for the generator (c2, bf16x6: 104 us; without MFMAs 61, with neither MFMAs nor stores 57): the X stream (118 MB
// of fp32 from HBM, one 16 KB slab in flight per workgroup), the split, the fragment reads and the MFMAs ran one after the other.
// Here a k-step's MFMAs are issued with the NEXT k-step's fragment reads and W loads in front of them, the split + LDS write of slab
// it + 2 rides in the second k-step, and two slabs of X are in flight per thread.  The fragment count of a workgroup's row tile
// is a template parameter of the loop (a branch around the MFMAs costs accumulator copies), as is the unequal last iteration.
template <int NIMG>
__global__ __launch_bounds__(512) void gemm_bf16s_out2_kernel(const OutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int HSTAGE = NIMG * O_IMG;        // one half's stage
    constexpr int NST = 3;
    typedef Prods<NIMG> PR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wc = wave & 3;  // K half of the chunk, 64-column quarter
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
    const int nmin = n_all - n0h;                        // iterations both halves compute in

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
    auto stagex = [&](int i, const f32x4 v0, const f32x4 v1) {       // slab i -> ring slot i % 3
        u32x4 im[NIMG];
        split8<NIMG>(v0, v1, im);
        unsigned char* st = hst + (i % NST) * HSTAGE + wofs;
#pragma unroll
        for (int img = 0; img < NIMG; ++img) *reinterpret_cast<u32x4*>(st + img * O_IMG) = im[img];
    };
    const int KS = a.K / 16;
    const unsigned short* wb0 = a.Wp + ((long long)(2 * wc) * KS * NIMG * 64 + lane) * 8;
    const unsigned short* wb1 = a.Wp + ((long long)(2 * wc + 1) * KS * NIMG * 64 + lane) * 8;
    const int last_ks = 2 * last_slab + 1;
    auto loadw = [&](int kk, bf16x8 (&w)[2][NIMG]) {     // k-step kk of this half (clamped)
        int ks = 2 * sb + kk;
        ks = ks < last_ks ? ks : last_ks;
#ifdef SBEV_EXP_HOTW
        ks = ks & 15;
#endif
        const long long o = (long long)ks * NIMG * 64 * 8;
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
        auto readx = [&](int i, int j, bf16x8 (&xf)[NFA][NIMG]) {      // fragments of slab i, k-step j
            const unsigned char* A = hst + (i % NST) * HSTAGE + (j ? fo1 : fo0);
#pragma unroll
            for (int img = 0; img < NIMG; ++img)
#pragma unroll
                for (int fa = 0; fa < NFA; ++fa) xf[fa][img] = *reinterpret_cast<const bf16x8*>(A + img * O_IMG + fa * 32 * 64);
        };
#define SBEV_O2_MMA(XF, W)                                                                                  \
        _Pragma("unroll") for (int p = 0; p < PR::N; ++p)                                                   \
            _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                              \
                _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                            \
                    acc[fa][fb] = SBEV_MFMA(W[fb][PR::ib(p)], XF[fa][PR::ia(p)], acc[fa][fb]);
        f32x4 xa0, xa1, xb0, xb1;                             // X register ring: slabs it + 2 (even it: a, odd: b) ...
        bf16x8 wA[2][NIMG], wB[2][NIMG], xfA[NFA][NIMG], xfB[NFA][NIMG];
        // prologue: slabs 0 and 1 staged, 2 and 3 requested, W of k-step 0 requested, fragments of (slab 0, k-step 0) read
        loadx(0, xa0, xa1);
        loadx(1, xb0, xb1);
        loadw(0, wA);
        stagex(0, xa0, xa1);
        loadx(2, xa0, xa1);
        stagex(1, xb0, xb1);
        loadx(3, xb0, xb1);
        __syncthreads();
        readx(0, 0, xfA);
        // one iteration = one slab; `it` even uses the a registers for slab it + 2, odd the b registers
        // (ZERO: the unequal last iteration -- half 1 has no slab left and multiplies zeros instead of branching around the MFMAs)
#define SBEV_O2_ZERO(XF)                                                                                    \
        if (half == 1) {                                                                                    \
            _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                              \
                _Pragma("unroll") for (int img = 0; img < NIMG; ++img)                                      \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) XF[fa][img][e] = (__bf16)0.f;             \
        }
#define SBEV_O2_ITER(IT, X0, X1, ZERO)                                                                      \
        {                                                                                                   \
            loadw(2 * (IT) + 1, wB);                                                                        \
            readx((IT), 1, xfB);                                                                            \
            if (ZERO) { SBEV_O2_ZERO(xfA) SBEV_O2_ZERO(xfB) }                                               \
            SBEV_O2_MMA(xfA, wA)                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            loadw(2 * (IT) + 2, wA);                                                                        \
            stagex((IT) + 2, X0, X1);                                                                       \
            loadx((IT) + 4, X0, X1);                                                                        \
            __syncthreads();              /* slab IT + 1 (staged one iteration ago) is published */         \
            readx((IT) + 1, 0, xfA);                                                                        \
            SBEV_O2_MMA(xfB, wB)                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                              \
        }
        int it = 0;
        for (; it + 1 < nmin; it += 2) {
            SBEV_O2_ITER(it, xa0, xa1, false)
            SBEV_O2_ITER(it + 1, xb0, xb1, false)
        }
        if (it < nmin) {
            SBEV_O2_ITER(it, xa0, xa1, false)
            ++it;
            if (it < n0h) SBEV_O2_ITER(it, xb0, xb1, true)
        } else if (it < n0h) {
            SBEV_O2_ITER(it, xa0, xa1, true)
        }
#undef SBEV_O2_ZERO
#undef SBEV_O2_ITER
#undef SBEV_O2_MMA
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
                        const f32x4 v = {acc[fa][fb][4 * g] + o[0], acc[fa][fb][4 * g + 1] + o[1], acc[fa][fb][4 * g + 2] + o[2], acc[fa][fb][4 * g + 3] + o[3]};
                        *reinterpret_cast<f32x4*>(out + (long long)row * 256 + fb * 32 + 8 * g) = v;
                    }
            }
        }
    };
    if (nfa == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
}

// ---- out-projection, ping-pong version: the two K halves of a workgroup run one barrier phase apart ----------------------------------
// PMC of the kernel above (c2, bf16x6, 100 us): matrix pipe busy 62 % of the waves' lifetime, 55 % of it spent stalled on issue, 27 %
// in waits -- again the two waves of a SIMD (the same 64 columns of the two K halves: waves w and w + 4) in lock-step.  As in the
// generator a slab now has a FETCH phase (fragments of the slab LDS -> registers; split + LDS write of the slab two ahead; the X
// loads four ahead) and a COMPUTE phase (its 48 MFMAs, then the W fragment loads of the next slab, which land during the following
// FETCH), a barrier after each, and the second K half takes one extra barrier up front: on every SIMD one wave computes while the
// other fetches.  The halves never touch each other's LDS ring; they meet only in the final fold.
template <int NIMG>
__global__ __launch_bounds__(512) void gemm_bf16s_out3_kernel(const OutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int HSTAGE = NIMG * O_IMG;        // one half's stage
    constexpr int NST = 3;
    typedef Prods<NIMG> PR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wc = wave & 3;  // K half of the chunk = phase group, 64-column quarter
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
    auto stagex = [&](int i, const f32x4 v0, const f32x4 v1) {       // slab i -> ring slot i % 3
        u32x4 im[NIMG];
        split8<NIMG>(v0, v1, im);
        unsigned char* st = hst + (i % NST) * HSTAGE + wofs;
#pragma unroll
        for (int img = 0; img < NIMG; ++img) *reinterpret_cast<u32x4*>(st + img * O_IMG) = im[img];
    };
    const int KS = a.K / 16;
    const unsigned short* wb0 = a.Wp + ((long long)(2 * wc) * KS * NIMG * 64 + lane) * 8;
    const unsigned short* wb1 = a.Wp + ((long long)(2 * wc + 1) * KS * NIMG * 64 + lane) * 8;
    auto loadw = [&](int i, bf16x8 (&w)[2][2][NIMG]) {   // both k-steps of slab i of this half (clamped)
        int sl = sb + i;
        sl = sl < last_slab ? sl : last_slab;
#ifdef SBEV_EXP_HOTW
        sl = sl & 7;
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long o = (long long)(2 * sl + j) * NIMG * 64 * 8;
#pragma unroll
            for (int img = 0; img < NIMG; ++img) {
                w[j][0][img] = *reinterpret_cast<const bf16x8*>(wb0 + o + img * 512);
                w[j][1][img] = *reinterpret_cast<const bf16x8*>(wb1 + o + img * 512);
            }
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
        loadw(0, w);
        stagex(0, xa0, xa1);
        loadx(2, xa0, xa1);
        stagex(1, xb0, xb1);
        loadx(3, xb0, xb1);
        __syncthreads();
        if (half == 1) __syncthreads();                       // the second K half runs one phase behind
#define SBEV_O3_SLAB(S_, X0, X1)                                                                            \
        {                                                                                                   \
            /* FETCH */                                                                                     \
            {                                                                                               \
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
                stagex((S_) + 2, X0, X1);                                                                   \
                loadx((S_) + 4, X0, X1);                                                                    \
            }                                                                                               \
            __syncthreads();                                                                                \
            /* COMPUTE */                                                                                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
                _Pragma("unroll") for (int p = 0; p < PR::N; ++p)                                           \
                    _Pragma("unroll") for (int fa = 0; fa < NFA; ++fa)                                      \
                        _Pragma("unroll") for (int fb = 0; fb < 2; ++fb)                                    \
                            acc[fa][fb] = SBEV_MFMA(w[j][fb][PR::ib(p)], xf[j][fa][PR::ia(p)], acc[fa][fb]); \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            loadw((S_) + 1, w);                                                                             \
            __syncthreads();                                                                                \
        }
        int sl = 0;
        for (; sl + 1 < n0h; sl += 2) {
            SBEV_O3_SLAB(sl, xa0, xa1)
            SBEV_O3_SLAB(sl + 1, xb0, xb1)
        }
        if (sl < n0h) SBEV_O3_SLAB(sl, xa0, xa1)
#undef SBEV_O3_SLAB
        if (half == 0) __syncthreads();
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
                        const f32x4 v = {acc[fa][fb][4 * g] + o[0], acc[fa][fb][4 * g + 1] + o[1], acc[fa][fb][4 * g + 2] + o[2], acc[fa][fb][4 * g + 3] + o[3]};
                        *reinterpret_cast<f32x4*>(out + (long long)row * 256 + fb * 32 + 8 * g) = v;
                    }
            }
        }
    };
    if (nfa == 2) run(std::integral_constant<int, 2>{});
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

}  // namespace

extern "C" int64_t sbev_bf16s_image_elems(int64_t rows, int K, int nimg) { return (rows + 31) / 32 * 32 * K * nimg; }

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
    if (nimg == 3) hipLaunchKernelGGL(pack_frags_kernel<3>, grid, dim3(256), 0, s, W, (long long)ldw, out, N, K);
    else hipLaunchKernelGGL(pack_frags_kernel<2>, grid, dim3(256), 0, s, W, (long long)ldw, out, N, K);
    return sbev::check_launch("sbev_pack_bf16s_frags");
}

static int ntm_of(int64_t M) { return (int)(((M + 31) / 32 + 3) / 4); }      // row tiles of <= 4 fragments

extern "C" int sbev_linear_bf16s_gen_ok(int64_t M, int N, int K) {
    return M >= 1 && M <= 0x7fffffffLL / 1024 && N >= 256 && N % 256 == 0 && K >= 32 && K % 32 == 0 && K <= 4096 &&
           ntm_of(M) <= 256;
}

extern "C" int sbev_linear_bf16s_gen(const uint16_t* Xs, const uint16_t* Ws, const float* bias, float* Y, int64_t M, int N, int K,
                                     int64_t ldy, int relu, int nimg, sbev_stream_t stream) {
    SBEV_REQUIRE(nimg == 2 || nimg == 3, "sbev_linear_bf16s_gen: nimg=%d (2 = bf16x3, 3 = bf16x6)", nimg);
    SBEV_REQUIRE(M >= 0 && sbev_linear_bf16s_gen_ok(M > 0 ? M : 1, N, K), "sbev_linear_bf16s_gen: needs N %% 256 == 0, K %% 32 == 0, K <= 4096 (M=%lld N=%d K=%d)", (long long)M, N, K);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(Xs && Ws && Y && ldy >= N && ldy % 4 == 0, "sbev_linear_bf16s_gen: bad pointers / leading dimension");
    SBEV_REQUIRE((((uintptr_t)Xs | (uintptr_t)Ws | (uintptr_t)Y) & 15) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0), "sbev_linear_bf16s_gen: 16-byte alignment");
    const int nfrag = (int)((M + 31) / 32);
    const int ntm = (nfrag + 3) / 4;
    GenArgs a{Xs, Ws, bias, Y, (int)M, N, K, (long long)ldy, relu, ntm, nfrag / ntm, nfrag % ntm};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    int st;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    // one row tile per workgroup for life: grid = a multiple of ntm, at most the CU count, at most the tile count
    long long per = cus / ntm < 1 ? 1 : cus / ntm;
    if (per > N / G_COLS) per = N / G_COLS;
    const unsigned grid = (unsigned)(per * ntm);
    static const bool v2 = getenv("SBEV_BF16S_GEN_V2") != nullptr;       // A/B switch: the lock-step kernel
    if (!v2) {
        // the bias slices of a workgroup's column tiles wait in LDS behind the stage ring: at most 16 tiles (16 KiB) per launch,
        // wider matrices take several launches over column ranges
        const int nct = N / G_COLS;
        const long long max_ct = per * 16;
        for (long long c0 = 0; c0 < nct; c0 += max_ct) {
            const int nc = (int)(nct - c0 < max_ct ? nct - c0 : max_ct);
            GenArgs ac = a;
            ac.Ws = Ws + c0 * 8 * (long long)(K / 16) * nimg * 512;      // 8 fragment blocks of 32 columns per tile
            ac.bias = bias ? bias + c0 * G_COLS : nullptr;
            ac.Y = Y + c0 * G_COLS;
            ac.N = nc * G_COLS;
            const long long pc = per < nc ? per : nc;
            const unsigned gridc = (unsigned)(pc * ntm);
            const int bias_bytes = (int)((nc + pc - 1) / pc) * G_COLS * 4;
            const bool prof = sbev::profile_begin(s, &e0, &e1, 1);
            if (nimg == 3) {
                const int LDS = G2_NST * 3 * (G2_ST_A + G2_ST_B) + bias_bytes;
                st = reserve_lds(gemm_bf16s_gen3_kernel<3>, LDS, "sbev_linear_bf16s_gen");
                if (st != SBEV_OK) return st;
                hipLaunchKernelGGL(gemm_bf16s_gen3_kernel<3>, dim3(gridc), dim3(512), LDS, s, ac);
            } else {
                const int LDS = G2_NST * 2 * (G2_ST_A + G2_ST_B) + bias_bytes;
                st = reserve_lds(gemm_bf16s_gen3_kernel<2>, LDS, "sbev_linear_bf16s_gen");
                if (st != SBEV_OK) return st;
                hipLaunchKernelGGL(gemm_bf16s_gen3_kernel<2>, dim3(gridc), dim3(512), LDS, s, ac);
            }
            if (prof) sbev::profile_end(s, e0, e1, 1);
        }
        return sbev::check_launch("sbev_linear_bf16s_gen");
    }
    if (nimg == 3) {
        constexpr int LDS = G2_NST * 3 * (G2_ST_A + G2_ST_B);
        st = reserve_lds(gemm_bf16s_gen2_kernel<3>, LDS, "sbev_linear_bf16s_gen");
        if (st != SBEV_OK) return st;
        const bool prof = sbev::profile_begin(s, &e0, &e1, 1);
        hipLaunchKernelGGL(gemm_bf16s_gen2_kernel<3>, dim3(grid), dim3(512), LDS, s, a);
        if (prof) sbev::profile_end(s, e0, e1, 1);
    } else {
        constexpr int LDS = G2_NST * 2 * (G2_ST_A + G2_ST_B);
        st = reserve_lds(gemm_bf16s_gen2_kernel<2>, LDS, "sbev_linear_bf16s_gen");
        if (st != SBEV_OK) return st;
        const bool prof = sbev::profile_begin(s, &e0, &e1, 1);
        hipLaunchKernelGGL(gemm_bf16s_gen2_kernel<2>, dim3(grid), dim3(512), LDS, s, a);
        if (prof) sbev::profile_end(s, e0, e1, 1);
    }
    return sbev::check_launch("sbev_linear_bf16s_gen");
}

extern "C" int sbev_linear_bf16s_out_ok(int64_t M, int N, int K) {
    return M >= 1 && M <= 0x7fffffffLL / 512 && N == 256 && K >= 256 && K % 32 == 0;
}

extern "C" int sbev_linear_bf16s_out_plan(int64_t M, int N, int K) {
    if (!sbev_linear_bf16s_out_ok(M, N, K)) return 0;
    return out_chunks(M, K);
}

namespace sbev {
// the GEMM half: *used partial slabs [used, M, 256] (to be summed by sbev_splitk_reduce_f32 or the row-chain tail)
int launch_splitk_slabs_bf16s(const float* X, const uint16_t* Wp, int64_t M, int K, int64_t ldx, int nimg, float* slabs, int* used,
                              hipStream_t s) {
    const int S = out_chunks(M, K);
    *used = S;
    OutArgs a{X, Wp, slabs, (int)M, K, (long long)ldx, (int)((M + 63) / 64), S};
    const long long wgs = (long long)a.nrt * S;
    SBEV_REQUIRE(wgs <= 0x7fffffffLL, "sbev_linear_splitk_bf16s: too many workgroups");
    hipEvent_t e0, e1;
    int st;
    static const bool v2 = getenv("SBEV_BF16S_OUT_V2") != nullptr;       // A/B switch: the lock-step kernel
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
    if (!v2) {
        if (nimg == 3) SBEV_LAUNCH_OUT(gemm_bf16s_out3_kernel<3>, LDS3) else SBEV_LAUNCH_OUT(gemm_bf16s_out3_kernel<2>, LDS2)
    } else {
        if (nimg == 3) SBEV_LAUNCH_OUT(gemm_bf16s_out2_kernel<3>, LDS3) else SBEV_LAUNCH_OUT(gemm_bf16s_out2_kernel<2>, LDS2)
    }
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
    const int st = sbev::launch_splitk_slabs_bf16s(X, Wp, M, K, ldx, nimg, workspace, &used, reinterpret_cast<hipStream_t>(stream));
    if (st != SBEV_OK) return st;
    return sbev_splitk_reduce_f32(workspace, used, bias, residual, ln_w, ln_b, ln_eps, Y, M, N, relu, stream);
}
