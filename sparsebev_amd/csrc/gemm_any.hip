// Layout-generic fp32 GEMM on the gfx950 matrix cores -- the two products every Linear needs in the BACKWARD pass.
//
//   C[M,N] (+)= sum_k A(m,k) * B(k,n)      A(m,k) = a_kmajor ? A[k*lda + m] : A[m*lda + k]
//                                          B(k,n) = b_kmajor ? B[k*ldb + n] : B[n*ldb + k]
//
// For y = x W^T + b (nn.Linear, W stored [out, in]):
//   grad_x [M, in]  = grad_y [M, out] . W [out, in]        -> A = grad_y (row-major, k contiguous), B = W   (k-major)
//   grad_W [out,in] = grad_y^T [out, M] . x [M, in]        -> A = grad_y (k-major),                 B = x   (k-major)
// i.e. the reduction runs over `out` or over the B*Q rows, and NEITHER product needs a transposed copy of a 33.5 MB
// weight or a 118 MB activation (the reference lets cuBLAS do these through autograd, models/sparsebev_transformer.py:
// 358-379).  Same arithmetic as gemm.hip's forward kernel: v_mfma_f32_32x32x2_f32, exact fp32 (an fmaf chain).
//
// 128x128x32 tiles, 2x2 waves x 2x2 MFMA tiles, register-staged double-buffered LDS, one barrier per K step.  A k-major
// operand is staged as [32 k][128 m (+4)] and read back as two conflict-free ds_read_b32 per fragment (consecutive lanes
// = consecutive m); a row-major operand as [128 m][32 k (+4)] with one ds_read_b64, as in the forward kernel.  Ragged M, N
// and K are zero-filled on the way into LDS (K = B*Q = 900 rows is not a multiple of 32).  Operands that are not 16-byte
// addressable (ld % 4 != 0: the 10-wide box / class heads, the 3-wide position input) take the element-wise staging
// instantiation.  Few-tile / long-K shapes (grad of the parameter generator's input: 900 x 256 x 32768) split K over
// grid.z into fp32 slabs summed by a second tiny kernel in a fixed order (bit-reproducible, unlike atomics).
#include "sbev_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int LDR = TK + 4;      // row-major staging: [128][36]
constexpr int LDK = TM + 4;      // k-major staging:   [32][132]
constexpr int OPER_FLOATS = TM * LDR > TK * LDK ? TM * LDR : TK * LDK;   // 4608

struct AnyArgs {
    const float* A;
    const float* B;
    float* C;            // [M, ldc]  (split: slabs [splits][M][N], ldc = N)
    long long M, K;
    int N;
    long long lda, ldb, ldc;
    long long k_per_split;   // multiple of TK
    int accumulate;
    // multi-segment reduction (sbev_gemm_f32_multi): C = sum over segments of A_s B_s, every segment with the same layout / M / N / K;
    // split z = segment * splits_per_seg + sub-split of that segment's K range.  nseg = 0: the single (A, B) above.
    int nseg, splits_per_seg;
    const float* Aseg[8];
    const float* Bseg[8];
};

// Stage one operand tile (128 "outer" x 32 k) from global memory into registers.  OUTER = rows of C this operand indexes
// (m for A, n for B); `o0` the tile's first outer index, `k0` the K step's first k.
template <bool KMAJOR, bool VEC>
__device__ __forceinline__ void stage_load(const float* __restrict__ P, long long ld, long long outer, long long kend,
                                           long long o0, long long k0, int tid, float4 (&r)[4]) {
    if (KMAJOR) {
        const int kk = tid >> 5, o4 = (tid & 31) * 4;                      // 8 k rows x 32 float4 per pass, 4 passes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long k = k0 + kk + 8 * i, o = o0 + o4;
            if (VEC) {
                r[i] = (k < kend && o < outer) ? *reinterpret_cast<const float4*>(P + k * ld + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (k < kend && o + e < outer) ? P[k * ld + o + e] : 0.f;
                r[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else {
        const int oo = tid >> 3, k4 = (tid & 7) * 4;                       // 32 outer rows x 8 float4 per pass, 4 passes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long o = o0 + oo + 32 * i, k = k0 + k4;
            if (VEC) {
                r[i] = (o < outer && k < kend) ? *reinterpret_cast<const float4*>(P + o * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (o < outer && k + e < kend) ? P[o * ld + k + e] : 0.f;
                r[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void stage_store(float* __restrict__ S, int tid, const float4 (&r)[4]) {
    if (KMAJOR) {
        const int kk = tid >> 5, o4 = (tid & 31) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&S[(kk + 8 * i) * LDK + o4]) = r[i];
    } else {
        const int oo = tid >> 3, k4 = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&S[(oo + 32 * i) * LDR + k4]) = r[i];
    }
}

// fragment of the 32x32x2 MFMA: lane (fr = lane & 31, fh = lane >> 5) supplies operand(row fr, k = 4*kk + 2*fh + {0, 1})
template <bool KMAJOR>
__device__ __forceinline__ float2 frag(const float* __restrict__ S, int o, int kk, int fh) {
    const int k = 4 * kk + 2 * fh;
    if (KMAJOR) return make_float2(S[k * LDK + o], S[(k + 1) * LDK + o]);
    return *reinterpret_cast<const float2*>(&S[o * LDR + k]);
}

template <bool AK, bool BKM, bool VEC, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_any_kernel(const AnyArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[4 * OPER_FLOATS];   // [2 buffers][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned tiles_n = (unsigned)((a.N + TN - 1) / TN);
    const unsigned tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const long long m0 = (long long)tm * TM;
    const long long n0 = (long long)tn * TN;
    const float* Aop = a.A;
    const float* Bop = a.B;
    unsigned zk = blockIdx.z;
    if (SPLIT && a.nseg > 0) {
        const unsigned seg = blockIdx.z / (unsigned)a.splits_per_seg;
        zk = blockIdx.z - seg * (unsigned)a.splits_per_seg;
        Aop = a.Aseg[seg];
        Bop = a.Bseg[seg];
    }
    const long long kbeg = SPLIT ? (long long)zk * a.k_per_split : 0;
    const long long kend = SPLIT ? (kbeg + a.k_per_split < a.K ? kbeg + a.k_per_split : a.K) : a.K;
    const int nk = (int)((kend - kbeg + TK - 1) / TK);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[4], rb[4];
    if (nk > 0) {
        stage_load<AK, VEC>(Aop, a.lda, a.M, kend, m0, kbeg, tid, ra);
        stage_load<BKM, VEC>(Bop, a.ldb, a.N, kend, n0, kbeg, tid, rb);
        stage_store<AK>(lds, tid, ra);
        stage_store<BKM>(lds + OPER_FLOATS, tid, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            stage_load<AK, VEC>(Aop, a.lda, a.M, kend, m0, kbeg + (long long)(kt + 1) * TK, tid, ra);
            stage_load<BKM, VEC>(Bop, a.ldb, a.N, kend, n0, kbeg + (long long)(kt + 1) * TK, tid, rb);
        }
        const float* As = lds + buf * 2 * OPER_FLOATS;
        const float* Bs = As + OPER_FLOATS;
#pragma unroll
        for (int kk = 0; kk < TK / 4; ++kk) {
            float2 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = frag<AK>(As, wr * 64 + i * 32 + fr, kk, fh);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = frag<BKM>(Bs, wc * 64 + j * 32 + fr, kk, fh);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            float* An = lds + (buf ^ 1) * 2 * OPER_FLOATS;
            stage_store<AK>(An, tid, ra);
            stage_store<BKM>(An + OPER_FLOATS, tid, rb);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5): the 32 lanes of a
    // half-wave store 32 consecutive floats of one row.
    float* C = SPLIT ? a.C + (long long)blockIdx.z * a.M * a.ldc : a.C;
    if (!SPLIT && a.accumulate) {      // all 64 old values requested together (clamped addresses): element-wise load + add + store
        float old[2][2][16];           // was 64 serialised round trips per lane (30 us of the big grad_W launches)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long long n = n0 + wc * 64 + j * 32 + fr, nc = n < a.N ? n : a.N - 1;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long long m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh, mc = m < a.M ? m : a.M - 1;
                    old[i][j][e] = C[mc * a.ldc + nc];
                }
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += old[i][j][e];
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
                if (m < a.M) C[m * a.ldc + n] = acc[i][j][e];
            }
        }
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ C, long long M, int N,
                                                       long long ldc, int splits, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(long long)z * M * N + i];
    float* p = C + (i / N) * ldc + (i % N);
    *p = accumulate ? *p + s : s;
}

int plan_splits(long long M, int N, long long K) {
    // fewer tiles than CUs and a reduction long enough to cut: split K so that ~256 workgroups exist, each with at least
    // two 32-wide K steps (a 256 x 256 weight gradient over K = B*Q = 900 rows is 4 tiles: 72 us on 4 CUs unsplit)
    const long long tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    if (tiles >= 128 || K < 256) return 1;
    long long s = 256 / tiles;
    const long long max_s = K / 64;
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    return s < 2 ? 1 : (int)s;
}

template <bool AK, bool BKM, bool VEC>
int launch_any(const AnyArgs& a, int splits, hipStream_t s) {
    const long long tiles = ((a.M + TM - 1) / TM) * ((a.N + TN - 1) / TN);
    if (tiles > 0x7fffffffLL) {
        sbev::set_error("sbev_gemm_f32: too many tiles");
        return SBEV_EINVAL;
    }
    if (splits > 1)
        hipLaunchKernelGGL((gemm_any_kernel<AK, BKM, VEC, true>), dim3((unsigned)tiles, 1, splits), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_any_kernel<AK, BKM, VEC, false>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    return sbev::check_launch("sbev_gemm_f32");
}

template <bool VEC>
int launch_layout(const AnyArgs& a, int ak, int bk, int splits, hipStream_t s) {
    if (ak && bk) return launch_any<true, true, VEC>(a, splits, s);
    if (ak) return launch_any<true, false, VEC>(a, splits, s);
    if (bk) return launch_any<false, true, VEC>(a, splits, s);
    return launch_any<false, false, VEC>(a, splits, s);
}

}  // namespace

extern "C" int64_t sbev_gemm_f32_workspace(int64_t M, int N, int64_t K) {
    if (M < 0 || N < 0 || K < 0) return -1;
    const int s = plan_splits(M, N, K);
    return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int sbev_gemm_f32(const float* A, int a_kmajor, int64_t lda, const float* B, int b_kmajor, int64_t ldb,
                             float* C, int64_t ldc, int64_t M, int N, int64_t K, int accumulate,
                             float* workspace, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sbev_gemm_f32: negative size");
    if (M == 0 || N == 0) return SBEV_OK;
    SBEV_REQUIRE(A && B && C, "sbev_gemm_f32: null pointer");
    SBEV_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N, "sbev_gemm_f32: leading dimension too small");
    AnyArgs a{};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.accumulate = accumulate;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool vec = lda % 4 == 0 && ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0 &&
                     (a_kmajor ? M % 4 == 0 : K % 4 == 0) && (b_kmajor ? N % 4 == 0 : K % 4 == 0);
    int splits = plan_splits(M, N, K);
    if (splits > 1 && !workspace) splits = 1;
    if (splits > 1) {
        a.k_per_split = ((K + splits - 1) / splits + TK - 1) / TK * TK;
        splits = (int)((K + a.k_per_split - 1) / a.k_per_split);
        AnyArgs p = a;
        p.C = workspace; p.ldc = N;
        int st = vec ? launch_layout<true>(p, a_kmajor, b_kmajor, splits, s) : launch_layout<false>(p, a_kmajor, b_kmajor, splits, s);
        if (st != SBEV_OK) return st;
        const long long n = M * N;
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, workspace, C, M, N, ldc, splits, accumulate);
        return sbev::check_launch("sbev_gemm_f32 (slab sum)");
    }
    a.k_per_split = K;
    return vec ? launch_layout<true>(a, a_kmajor, b_kmajor, 1, s) : launch_layout<false>(a, a_kmajor, b_kmajor, 1, s);
}

// C[M,N] (+)= sum_s sum_k A_s(m,k) B_s(k,n): nseg <= 8 operand pairs of identical layout and shape reduced in ONE launch (+ the slab
// sum) -- the weight gradient of a Linear that several layers of a decoder call share (training: 6 x (GEMM + slab sum + add) -> 2
// launches).  workspace: sbev_gemm_f32_multi_workspace(M, N, K, nseg) bytes.  Slabs are added segment by segment, split by split:
// bit-reproducible.
static int multi_splits(int64_t M, int N, int64_t K, int nseg) {
    const long long tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    long long s = 256 / (tiles * nseg);
    const long long max_s = K / 64 < 1 ? 1 : K / 64;
    if (s > max_s) s = max_s;
    if (s > 16) s = 16;
    return s < 1 ? 1 : (int)s;
}
extern "C" int64_t sbev_gemm_f32_multi_workspace(int64_t M, int N, int64_t K, int nseg) {
    if (M < 0 || N < 0 || K < 0 || nseg < 1 || nseg > 8) return -1;
    return (int64_t)nseg * multi_splits(M, N, K, nseg) * M * N * (int64_t)sizeof(float);
}
extern "C" int sbev_gemm_f32_multi(const float* const* A, int a_kmajor, int64_t lda, const float* const* B, int b_kmajor, int64_t ldb,
                                   int nseg, float* C, int64_t ldc, int64_t M, int N, int64_t K, int accumulate,
                                   float* workspace, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nseg >= 1 && nseg <= 8, "sbev_gemm_f32_multi: bad sizes (1 .. 8 segments)");
    if (M == 0 || N == 0) return SBEV_OK;
    SBEV_REQUIRE(A && B && C && workspace, "sbev_gemm_f32_multi: null pointer");
    SBEV_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N, "sbev_gemm_f32_multi: leading dimension too small");
    AnyArgs a{};
    a.A = A[0]; a.B = B[0]; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.accumulate = 0;
    bool vec = lda % 4 == 0 && ldb % 4 == 0 && (a_kmajor ? M % 4 == 0 : K % 4 == 0) && (b_kmajor ? N % 4 == 0 : K % 4 == 0);
    for (int i = 0; i < nseg; ++i) {
        SBEV_REQUIRE(A[i] && B[i], "sbev_gemm_f32_multi: null segment pointer");
        a.Aseg[i] = A[i]; a.Bseg[i] = B[i];
        vec = vec && (reinterpret_cast<uintptr_t>(A[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(B[i]) & 15) == 0;
    }
    int sps = multi_splits(M, N, K, nseg);
    a.k_per_split = ((K + sps - 1) / sps + TK - 1) / TK * TK;
    sps = (int)((K + a.k_per_split - 1) / a.k_per_split);
    a.nseg = nseg; a.splits_per_seg = sps;
    a.C = workspace; a.ldc = N;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int st = vec ? launch_layout<true>(a, a_kmajor, b_kmajor, nseg * sps, s) : launch_layout<false>(a, a_kmajor, b_kmajor, nseg * sps, s);
    if (st != SBEV_OK) return st;
    const long long n = M * N;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, workspace, C, M, N, ldc, nseg * sps, accumulate);
    return sbev::check_launch("sbev_gemm_f32_multi (slab sum)");
}
