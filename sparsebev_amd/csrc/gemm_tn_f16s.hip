// Weight gradients of AdaptiveMixing's two big Linears on the fp16 matrix core (training; gfx950).
//
//   C[M,N] (+)= sum_k A[k*lda + m] * B[k*ldb + n]          both operands k-major: grad_W = grad_y^T . x, reduced over the B*Q rows
//
// models/sparsebev_transformer.py:358-379 leaves these to autograd / cuBLAS.  At config 2 they are [256 x 32768] and [32768 x 256]
// over K = 900 rows: 15.1 GFLOP each, 174 us on gemm_any.hip's exact v_mfma_f32_32x32x2_f32 tiles (99 us of matrix-core time at
// 2 workgroups per CU).  Here every fp32 operand value is multiplied by the caller's power of two (max |x| 2^e in [2^14, 2^15), see
// gemm_bf16s.hip) and split into fp16 hi + lo ON THE WAY INTO LDS; the products hl + lh + hh run on v_mfma_f32_32x32x16_f16 with
// fp32 accumulation (18.6 us of matrix-core time for the same tile plan), the result is multiplied by 2^-(ea + eb).  Error against
// fp64: below the exact f32-MFMA kernel's (tests/test_gpu_backward.py), as for the forward GEMMs.
//
// 128 x 128 x 32 tiles, 2 x 2 waves x 2 x 2 MFMA tiles, register-staged, double-buffered LDS, one barrier per K step (the plan
// of gemm_any.hip).  A thread stages a 4 (outer) x 4 (k) block: four float4 loads along the contiguous outer index (512 B per
// wave and k row), then per outer index one 8-byte LDS write of 4 consecutive k per image.  LDS image: [128 outer][32 k] fp16,
// rows of 72 bytes (18 banks), row m stored at physical row  m ^ ((m >> 4) & 3).  Both the staging writes (ds_write2st64_b64) and the
// fragment reads (ds_read2_b64) are serviced in groups of 16 consecutive lanes on 32 banks, 8 bytes per lane: conflict-free iff the 16
// rows of a group are distinct mod 16 -- a write group is rows 4 o4 + e over 16 consecutive o4 (bits 2..5 of m vary), a read group
// rows base + 0..15 (bits 0..3 vary); the XOR moves bits 4..5 into bits 0..1, so both are.  (The first layout was conflict-free for
// 32-lane groups on 64 banks -- ds_read_b64's rule -- but the compiler fuses the two halves of a fragment into ds_read2_b64.)
// K is zero-filled to a multiple of 32 on the way in (K = 900).
#include "sbev_common.hpp"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int ROWB = 72;                 // bytes per LDS row: 32 fp16 + 8 pad
constexpr int IMG = TM * ROWB;           // one image (hi or lo) of one operand tile
constexpr int OPER = 2 * IMG;            // hi, lo
constexpr int BUF = 2 * OPER;            // A, B
constexpr int LDS_BYTES = 2 * BUF;       // double buffer: 73,728

struct TnArgs {
    const float* A;
    const float* B;
    const float* a_scale;    // [2]: 2^ea, 2^-ea
    const float* b_scale;
    float* C;
    long long M, N, K;
    long long lda, ldb, ldc;
    int accumulate;
};

// workgroup barrier for the LDS tiles only: the global loads of the next two K steps stay in flight across it (__syncthreads waits
// for vmcnt(0))
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int phys_row(int m) { return m ^ ((m >> 4) & 3); }

// Request a 4 (k) x 4 (outer) block of K step `kt`: buffer loads -- the matrix's descriptor and the tile's row offset in SGPRs, a
// per-lane 32-bit offset that does not change over the loop (64-bit per-lane address arithmetic was a third of the loop's instructions, and by the PMC model of these GEMMs --
// wave cycles = a + 12.5 cycles x VALU instructions -- instructions are what the kernel's time is made of).  The LAST tile is
// shifted back to rows K - 32 .. K - 1 (tiles past K too): every load is in bounds and UNCONDITIONAL (a load under a branch makes
// every later wait a vmcnt(0)); store_tile zeroes the rows below kt * 32 that the shift brought in again.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ void stage_load(rsrc_t P, unsigned ld4, int K, int kt, const unsigned (&voff)[4], u32x4 (&r)[4]) {
    const int k0 = kt * TK < K - TK ? kt * TK : K - TK;                                          // (uniform)
    const unsigned soff = (unsigned)k0 * ld4;                                                    // SGPR offset of the tile's first row
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(P, (int)voff[i], (int)soff, 0));
}
#ifdef SBEV_TN_NO_LOAD
#define SBEV_TN_LOOP_LOAD(...)
#else
#define SBEV_TN_LOOP_LOAD(...) stage_load(__VA_ARGS__)
#endif
#ifdef SBEV_TN_NO_MFMA
#define SBEV_TN_MFMA(A_, B_, C_) ([&]() { asm volatile("" ::"v"(A_), "v"(B_)); return C_; }())
#else
#define SBEV_TN_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, C_, 0, 0, 0)
#endif

// (f16(a.x * s.x), f16(a.y * s.y)) packed, and the same of the remainders a * s - hi: v_fma_mix{lo,hi}_f16 round fma(a, s, c) to
// fp16 once and write one half of the destination -- 2 instructions per value, no separate multiply, convert-back, subtract or pack
// (the compiler's own selection for the C expression: 3.4 per value).  a * s is exact (s a power of two), a * s - hi is exact
// in the fma: the images are bit for bit those of (_Float16)(a s), (_Float16)(a s - hi).
__device__ __forceinline__ void split_pair(unsigned a0, unsigned a1, float s0, float s1, unsigned& hi, unsigned& lo) {
#ifdef SBEV_TN_NO_CONV            // ablations (tools/exp/ablate_tn.py): timing only, wrong numbers
    hi = a0 ^ __builtin_bit_cast(unsigned, s0); lo = a1 ^ __builtin_bit_cast(unsigned, s1);
    return;
#endif
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(a0), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(a1), "v"(s1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a0), "v"(s0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(a1), "v"(s1), "v"(hi));
}

// r[i][e] = value at (k row 4 kq + i of the tile, outer = 4 o4 + e)  ->  hi / lo images, 4 consecutive k (8 bytes) per outer index.
// upk[i]: the operand's scale, or 0 for a row that is not this tile's.  Only A's rows are zeroed (B's then multiply zeros), and
// clamped columns are not zeroed at all (their outputs are never stored): such an element is a copy of an element of the same
// column of the same operand, so a non-finite copy can only reach outputs that the original already makes non-finite.  No branch
// here: a conditional block that touches the staging registers makes the statically inserted waits drain the prefetch.
__device__ __forceinline__ void stage_store_e(unsigned char* S, int o4, int kq, const u32x4 (&r)[4], const float (&upk)[4], int e) {
    const unsigned b0 = r[0][e], b1 = r[1][e], b2 = r[2][e], b3 = r[3][e];
    unsigned h01, l01, h23, l23;
    split_pair(b0, b1, upk[0], upk[1], h01, l01);
    split_pair(b2, b3, upk[2], upk[3], h23, l23);
    const int prow = phys_row(4 * o4 + e);
    unsigned char* d = S + prow * ROWB + kq * 8;
#ifdef SBEV_TN_NO_LDSW
    asm volatile("" ::"v"(h01), "v"(h23), "v"(l01), "v"(l23), "v"(d));
#else
    *reinterpret_cast<u32x2*>(d) = (u32x2){h01, h23};
    *reinterpret_cast<u32x2*>(d + IMG) = (u32x2){l01, l23};
#endif
}
__device__ __forceinline__ void stage_store(unsigned char* S, int o4, int kq, const u32x4 (&r)[4], const float (&upk)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) stage_store_e(S, o4, kq, r, upk, e);
}

// operand of v_mfma_f32_32x32x16_f16: lane (fr, fh) supplies (row fr, k = 8 fh .. 8 fh + 7) of the 16-k step
__device__ __forceinline__ f16x8 frag(const unsigned char* S, int row_off, int ks, int fh) {
#ifdef SBEV_TN_NO_LDSR
    unsigned q = (unsigned)(unsigned long long)S + row_off + ks + fh;
    asm volatile("" : "+v"(q));
    return __builtin_bit_cast(f16x8, (u32x4){q, q, q, q});
#endif
    const u32x2* p = reinterpret_cast<const u32x2*>(S + row_off + ks * 32 + fh * 16);
    const u32x2 a = p[0], b = p[1];
    return __builtin_bit_cast(f16x8, (u32x4){a.x, a.y, b.x, b.y});
}

__global__ __launch_bounds__(256, 2) void gemm_tn_f16s_kernel(const TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int o4 = tid & 31, kq = tid >> 5;
    // Workgroup b runs on XCD b % 8.  Give each XCD a CONTIGUOUS range of logical tile ids, the short side of the tile grid
    // fastest: the (two) workgroups that share a tile of the long operand are dispatched back to back into the same L2, and the
    // 32 consecutive long-side tiles of an XCD read one contiguous 16 KiB piece of every row of the long operand at about the same
    // time (first version: id % tiles along the long side -- neighbouring 512-byte pieces of a row went to eight different L2s;
    // in the training step, operands coming from HBM: 88 us against 64 us with the operands hot in the 256 MB cache)
    const unsigned tiles_m = (unsigned)((a.M + TM - 1) / TM), tiles_n = (unsigned)((a.N + TN - 1) / TN);
    const unsigned nb = gridDim.x, full = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;
    const unsigned lid = xcd * full + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    unsigned tm, tn;
    if (tiles_n < tiles_m) { tm = lid / tiles_n; tn = lid % tiles_n; }
    else { tn = lid / tiles_m; tm = lid % tiles_m; }
    const long long m0 = (long long)tm * TM, n0 = (long long)tn * TN;
    const int nk = (int)((a.K + TK - 1) / TK);
    const float upa = a.a_scale[0], upb = a.b_scale[0];

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int arow[2], brow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = phys_row(wr * 64 + i * 32 + fr) * ROWB;
        brow[i] = phys_row(wc * 64 + i * 32 + fr) * ROWB;
    }

    // operand tiles are requested TWO K steps ahead (register sets 0 / 1): a K step is ~0.6 us of matrix-core + conversion work per
    // wave, the loads' round trip under load is several times that (one step ahead: 102 us per launch at config 2)
    u32x4 ra[2][4], rb[2][4];
    const bool a_ok = m0 + 4 * o4 < a.M, b_ok = n0 + 4 * o4 < a.N;
    const unsigned lda4 = (unsigned)a.lda * 4u, ldb4 = (unsigned)a.ldb * 4u;
    const unsigned acol4 = (unsigned)(a_ok ? m0 + 4 * o4 : a.M - 4) * 4u, bcol4 = (unsigned)(b_ok ? n0 + 4 * o4 : a.N - 4) * 4u;
    unsigned voa[4], vob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        voa[i] = (unsigned)(4 * kq + i) * lda4 + acol4;
        vob[i] = (unsigned)(4 * kq + i) * ldb4 + bcol4;
    }
    const int K = (int)a.K;
    const rsrc_t Ad = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)((unsigned)K * lda4), 0x00020000);
    const rsrc_t Bd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, (int)((unsigned)K * ldb4), 0x00020000);
    stage_load(Ad, lda4, K, 0, voa, ra[0]);
    stage_load(Bd, ldb4, K, 0, vob, rb[0]);
    stage_load(Ad, lda4, K, 1, voa, ra[1]);
    stage_load(Bd, ldb4, K, 1, vob, rb[1]);
    const float upb4[4] = {upb, upb, upb, upb};
    const int kend = (int)a.K - TK;                  // first row of the shifted last tile
    auto store_tile = [&](unsigned char* Sn, const u32x4 (&xa)[4], const u32x4 (&xb)[4], int kt) {
        const int skip = kt * TK - (kt * TK < kend ? kt * TK : kend);      // (uniform) rows of the tile that belong to earlier tiles; >= 32 past K
        float upa4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) upa4[i] = 4 * kq + i >= skip ? upa : 0.f;
        stage_store(Sn, o4, kq, xa, upa4);
        stage_store(Sn + OPER, o4, kq, xb, upb4);
    };
    store_tile(lds, ra[0], rb[0], 0);
    lds_barrier();
    // step kt: LDS buffer kt & 1 holds tile kt, register set (kt + 1) & 1 tile kt + 1 (in flight), register set kt & 1 is free
    auto step = [&](auto par, int kt) {
        constexpr int P = decltype(par)::value;
        SBEV_TN_LOOP_LOAD(Ad, lda4, K, kt + 2, voa, ra[P]);
        SBEV_TN_LOOP_LOAD(Bd, ldb4, K, kt + 2, vob, rb[P]);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* As = lds + P * BUF;
        const unsigned char* Bs = As + OPER;
        unsigned char* An = lds + (P ^ 1) * BUF;
        // all fragments of the K step first, then six groups of four MFMAs (2 k-steps x 3 products; small terms first, hi x hi last:
        // the order of gemm_bf16s.hip's 3-product mode), each followed by its share of the conversion of tile kt + 1 into the other
        // buffer (8 chunks: 4 outer indices x 2 operands).  The conversion's VALU work issues in the shadow of the MFMAs before
        // it; with everything of one kind in one block (first version) the waves of a workgroup -- barrier-synchronised -- were
        // all reading LDS, then all on the matrix core, then all converting: the three pipes' busy times ADDED up to the K step's
        // duration (PMC: LDS 34 %, MFMA 28 %, VALU 13-25 %).  The conversion is unconditional (a skipped one would leave its
        // register set "pending" on one path into the loop head and the static waits would drain the prefetch); past K the
        // tiles are zero.
        f16x8 fa[2][2][2], fb[2][2][2];       // [k-step][tile][image]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int img = 0; img < 2; ++img) {
                    fa[ks][i][img] = frag(As + img * IMG, arow[i], ks, fh);
                    fb[ks][i][img] = frag(Bs + img * IMG, brow[i], ks, fh);
                }
        const int skip = (kt + 1) * TK - ((kt + 1) * TK < kend ? (kt + 1) * TK : kend);   // (uniform) rows of tile kt + 1 that belong to earlier tiles; >= 32 past K
        float upa4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) upa4[i] = 4 * kq + i >= skip ? upa : 0.f;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int ks = g / 3, pr = g % 3, ia = pr == 1 ? 1 : 0, ib = pr == 0 ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = SBEV_TN_MFMA(fa[ks][i][ia], fb[ks][j][ib], acc[i][j]);
            if (g < 4) stage_store_e(An, o4, kq, ra[P ^ 1], upa4, g);
            else {
                stage_store_e(An + OPER, o4, kq, rb[P ^ 1], upb4, 2 * (g - 4));
                stage_store_e(An + OPER, o4, kq, rb[P ^ 1], upb4, 2 * (g - 4) + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
    };
    // tile 1 has landed before the loop is entered: otherwise the registers the prologue's loads target count as pending at the
    // loop head on every iteration (static s_waitcnt insertion) and the loop waits for its loads a K step early
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    for (int kt = 0; kt < nk; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
    }

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
    // accumulate: ALL 64 old values are requested before the first is needed (clamped addresses, unconditional loads) -- a
    // load + add + store per element was 64 serialised round trips, 30 us of a 90 us launch.
    const float down = a.a_scale[1] * a.b_scale[1];
    if (a.accumulate) {
        float old[2][2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long long n = n0 + wc * 64 + j * 32 + fr, nc = n < a.N ? n : a.N - 1;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long long m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh, mc = m < a.M ? m : a.M - 1;
                    old[i][j][e] = a.C[mc * a.ldc + nc];
                }
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = old[i][j][e] + acc[i][j][e] * down;
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] *= down;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n = n0 + wc * 64 + j * 32 + fr;
            if (n >= a.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (m < a.M) a.C[m * a.ldc + n] = acc[i][j][e];
            }
        }
}

}  // namespace

extern "C" int sbev_gemm_tn_f16s_ok(int64_t M, int64_t N, int64_t K) {
    // enough tiles to fill the chip without a split of K (else: sbev_gemm_f32 and its split-K plan)
    return M >= 4 && N >= 4 && M % 4 == 0 && N % 4 == 0 && K >= TK && K < (1LL << 26) && ((M + TM - 1) / TM) * ((N + TN - 1) / TN) >= 256 &&
           ((M + TM - 1) / TM) * ((N + TN - 1) / TN) <= 0x7fffffffLL;
}

extern "C" int sbev_gemm_tn_f16s(const float* A, int64_t lda, const float* a_scale, const float* B, int64_t ldb, const float* b_scale,
                                 float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, sbev_stream_t stream) {
    SBEV_REQUIRE(sbev_gemm_tn_f16s_ok(M, N, K), "sbev_gemm_tn_f16s: M=%lld, N=%lld (multiples of 4, >= 256 tiles of 128 x 128), K=%lld (>= 32)",
                 (long long)M, (long long)N, (long long)K);
    SBEV_REQUIRE(A && B && C && a_scale && b_scale, "sbev_gemm_tn_f16s: null pointer");
    SBEV_REQUIRE(lda >= M && ldb >= N && ldc >= N && lda % 4 == 0 && ldb % 4 == 0 && K * lda < (1LL << 29) && K * ldb < (1LL << 29) &&
                     (((uintptr_t)A | (uintptr_t)B) & 15) == 0,
                 "sbev_gemm_tn_f16s: leading dimensions must cover the operands and be multiples of 4, an operand < 2 GiB (32-bit buffer offsets); A and B 16-byte aligned");
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_f16s_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
    if (!attr_ok) {
        (void)hipGetLastError();
        sbev::set_error("sbev_gemm_tn_f16s: cannot reserve %d bytes of LDS", LDS_BYTES);
        return SBEV_ELAUNCH;
    }
    const TnArgs a{A, B, a_scale, b_scale, C, M, N, K, lda, ldb, ldc, accumulate};
    const long long tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    hipLaunchKernelGGL(gemm_tn_f16s_kernel, dim3((unsigned)tiles), dim3(256), LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_gemm_tn_f16s");
}
