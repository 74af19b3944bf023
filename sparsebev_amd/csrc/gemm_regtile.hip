// Register-tiled split-K GEMM for the adaptive-mixing out-projection (own translation unit: it is compiled with
// `-mllvm -amdgpu-mfma-vgpr-form=1`).  All 224 live registers (96 accumulators + two operand sets + pointers) fit the
// 256 architectural VGPRs; with the default AGPR-form MFMAs hipcc keeps half of the accumulators in VGPRs across the
// loop back-edge and copies them to AGPRs and back EVERY iteration (144 serial v_accvgpr_* per 192 MFMAs: 140 -> 130 us
// at Q = 900 once they are gone).
#include <cstdlib>
#include "sbev_common.hpp"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef SBEV_RT_FR
#define SBEV_RT_FR 3
#endif
constexpr int FR = SBEV_RT_FR;            // 16-row fragments per wave task
constexpr int TROWS = 16 * FR;            // rows per wave task
constexpr int WPE = FR >= 3 ? 1 : 2;      // waves per SIMD the register budget allows (FR = 2 at two waves per SIMD measured
                                          // 2x SLOWER: 10 fragment-shaped loads per 64 MFMAs saturate the load path)

// ---- register-tiled split-K kernel for the out-projection [M, K = 32768] x [256, K]^T --------------------------------
// Same reasoning as the strip kernel, other shape: 8 x 2 tiles of 128 x 128 times 32 K-splits are exactly one round of
// 512 workgroups, so the 12 % row padding of Q = 900 is paid in full.  Here the unit of work is a WAVE task =
// (48 rows = 3 fragments) x (128 columns = 8 fragments) x (a K range): 900 rows are 19 x 48 = 912 (1.3 % padding),
// 19 x 2 x 26 splits = 988 tasks fill the 1024 one-wave-per-SIMD slots in a single round.  The 96 accumulator
// registers stay put, both operands stream from L2 straight into registers in 16-k steps (11 x 16-byte loads per 96
// MFMAs of 32 cycles, double buffered one step ahead): no LDS, no barrier.  Task order is (split, column group, row
// group) with the row group fastest and the workgroup index remapped so that consecutive tasks share an XCD: the 19
// row groups that re-read one W slice and the 2 column groups that re-read one X slice hit that XCD's L2.
struct RegTileArgs {
    const float* X;   // [M, ldx]
    const float* W;   // [N, ldw]
    float* P;         // [splits, M, N] partial slabs
    long long M;
    int N, K;
    long long ldx, ldw;
    int rgs, cgs, splits, pairs;   // row groups (48), column groups (128), K splits, K / 32
    unsigned tasks;
    int fold;                      // 2: the two waves of a pair own K splits 2s / 2s + 1 of ONE tile and write their sum (slab s)
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void gemm_nt_f32_regtile_kernel(const RegTileArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = lane & 15, fk = lane >> 4;
    // workgroup b runs on XCD b % 8: give each XCD a contiguous range of logical workgroups
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned full = nb >> 3, rem = nb & 7, x = b & 7;
    const unsigned logical = x * full + (x < rem ? x : rem) + (b >> 3);
    const unsigned task = logical * 4 + wave;
    // fold == 2: waves (0, 1) and (2, 3) of a workgroup are PAIRS on one output tile with adjacent K splits; the odd wave
    // hands its partial tile over through LDS and the even one writes the sum: half the slabs to write here and to read in
    // the reducer (24 -> 12 MB each at Q = 900), in a fixed order (bit-reproducible, unlike atomics).
    __shared__ f32x4v fold_buf[2][FR * 8][64];
    const bool active = task < a.tasks;
    if (a.fold == 1 && !active) return;
    const unsigned tc = active ? task : a.tasks - 1;       // a spare wave of the last workgroup repeats the last task (it has
    const unsigned unit = a.fold == 2 ? tc >> 1 : tc;      // to reach the barrier below) and writes nothing
    const int rg = unit % a.rgs;
    const unsigned t2 = unit / a.rgs;
    const int cg = t2 % a.cgs;
    const int sp_out = t2 / a.cgs;
    const int sp = a.fold == 2 ? 2 * sp_out + (int)(tc & 1) : sp_out;
    const int p0 = (a.pairs * sp) / a.splits, p1 = (a.pairs * (sp + 1)) / a.splits;     // pairs * splits < 2^31 (host-checked)
    const int M = (int)a.M;

    const float* xp[FR];
#pragma unroll
    for (int fr = 0; fr < FR; ++fr) {
        int row = rg * TROWS + fr * 16 + fi;
        row = row < M ? row : M - 1;
        xp[fr] = a.X + (long long)row * a.ldx + 4 * fk;
    }
    const float* wp = a.W + (long long)(cg * 128 + fi) * a.ldw + 4 * fk;
    const long long w16 = 16 * a.ldw;

    f32x4v acc[FR][8];
#pragma unroll
    for (int fr = 0; fr < FR; ++fr)
#pragma unroll
        for (int cf = 0; cf < 8; ++cf) acc[fr][cf] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    f32x4v xa[FR], wa[8], xb[FR], wb[8];
#define SBEV_RT_LOAD(xd, wd, kk)                                                                    \
    {                                                                                               \
        const int k_ = (kk) < kend ? (kk) : kend;       /* clamped: the load past the split is a dummy */ \
        _Pragma("unroll") for (int fr = 0; fr < FR; ++fr) xd[fr] = *reinterpret_cast<const f32x4v*>(xp[fr] + k_); \
        _Pragma("unroll") for (int cf = 0; cf < 8; ++cf) wd[cf] = *reinterpret_cast<const f32x4v*>(wp + cf * w16 + k_); \
        __builtin_amdgcn_sched_barrier(0);              /* loads stay ahead of the MFMA block */    \
    }
#define SBEV_RT_MMA(xs, ws)                                                                         \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int fr = 0; fr < FR; ++fr)                                       \
                _Pragma("unroll") for (int cf = 0; cf < 8; ++cf)                                    \
                    acc[fr][cf] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[cf][i], xs[fr][i], acc[fr][cf], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    }
    // 16-k steps, two register sets: a step's operands are requested one MFMA block (96 x 32 cycles) before use (a
    // 3-deep ring measured 4 % slower: 496 registers, no gain -- the loads are L2 hits)
    const int kend = p1 * 32 - 16;                  // last valid 16-k step of this split
    int k = p0 * 32;
    SBEV_RT_LOAD(xa, wa, k);
    for (int p = p0; p < p1; ++p) {
        SBEV_RT_LOAD(xb, wb, k + 16);
        SBEV_RT_MMA(xa, wa);
        k += 32;
        SBEV_RT_LOAD(xa, wa, k);
        SBEV_RT_MMA(xb, wb);
    }
#undef SBEV_RT_LOAD
#undef SBEV_RT_MMA
    if (a.fold == 2) {
        const int pr = wave >> 1;
        if (active && (wave & 1)) {
#pragma unroll
            for (int fr = 0; fr < FR; ++fr)
#pragma unroll
                for (int cf = 0; cf < 8; ++cf) fold_buf[pr][fr * 8 + cf][lane] = acc[fr][cf];
        }
        __syncthreads();
        if (!active || (wave & 1)) return;
#pragma unroll
        for (int fr = 0; fr < FR; ++fr)
#pragma unroll
            for (int cf = 0; cf < 8; ++cf) acc[fr][cf] += fold_buf[pr][fr * 8 + cf][lane];
    }
    float* out = a.P + ((long long)sp_out * a.M) * a.N + cg * 128 + 4 * fk;
#pragma unroll
    for (int fr = 0; fr < FR; ++fr) {
        const int row = rg * TROWS + fr * 16 + fi;
        if (row < M) {
#pragma unroll
            for (int cf = 0; cf < 8; ++cf)
                *reinterpret_cast<f32x4v*>(out + (long long)row * a.N + cf * 16) = acc[fr][cf];
        }
    }
}

}  // namespace

namespace sbev {
int launch_splitk_regtile(const float* X, const float* W, float* slabs, int64_t M, int N, int K, int64_t ldx, int64_t ldw,
                          int splits, int* slabs_written, hipStream_t stream) {
    SBEV_REQUIRE((long long)(K / 32) * (splits + 1) < 0x7fffffffLL, "sbev_linear_splitk_f32: K * splits too large");
    static const bool no_fold = getenv("SBEV_NO_SPLIT_FOLD") != nullptr;       // A/B switch
    const int fold = (splits % 2 == 0 && !no_fold) ? 2 : 1;
    *slabs_written = splits / fold;
    RegTileArgs t{X, W, slabs, M, N, K, ldx, ldw, (int)((M + TROWS - 1) / TROWS), N / 128, splits, K / 32, 0u, fold};
    const long long tasks = (long long)t.rgs * t.cgs * splits;
    SBEV_REQUIRE(tasks <= 0x3fffffffLL, "sbev_linear_splitk_f32: too many tasks");
    t.tasks = (unsigned)tasks;
    hipEvent_t e0, e1;
    const bool prof = profile_begin(stream, &e0, &e1, 2);
    hipLaunchKernelGGL(gemm_nt_f32_regtile_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, stream, t);
    if (prof) profile_end(stream, e0, e1, 2);
    return check_launch("sbev_linear_splitk_f32 (gemm)");
}

// K splits that fill the wave slots (1024 SIMDs x WPE) exactly once with (TROWS rows x 128 columns x K/splits) tasks
int regtile_plan(int64_t M, int N, int K) {
    const long long units = ((M + TROWS - 1) / TROWS) * (N / 128);
    long long s = (1024LL * WPE) / units;
    if (s > K / 512) s = K / 512;
    return (int)(s < 1 ? 1 : s);
}
}  // namespace sbev
