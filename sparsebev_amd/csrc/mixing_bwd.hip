// Adaptive mixing core, BACKWARD (gfx950).  SURVEY.md section 8f rank 4.
//
// Forward (mixing.hip; models/sparsebev_transformer.py:362-374), per (query, group):
//     y1 = x[Pin,C] @ M[C,C]         h1 = LN_all(y1)      n1 = relu(h1)
//     y2 = S[Pout,Pin] @ n1[Pin,C]   h2 = LN_all(y2)      out = relu(h2)        (LN over the whole block, no affine)
// Backward, given g_out[Pout,C] (the forward is recomputed here -- nothing but x and the dynamic parameters is kept from
// the forward pass, the same policy as the reference's activation checkpoint around inner_forward, :383-387):
//     g2  = g_out * (h2 > 0)         dy2 = rstd2 * (g2 - mean(g2) - h2 * mean(g2 * h2))
//     dS  = dy2 @ n1^T [Pout,Pin]    dn1 = S^T @ dy2 [Pin,C]
//     g1  = dn1 * (h1 > 0)           dy1 = rstd1 * (g1 - mean(g1) - h1 * mean(g1 * h1))
//     dM  = x^T @ dy1 [C,C]          dx  = dy1 @ M^T [Pin,C]
// One 256-thread workgroup per (b*Q + q, g); x, h1, dy1 and dy2 live in LDS (rows padded to 65 floats so row- and column-
// walking fragment reads are conflict-free), M and S are read as MFMA fragments through L1 / L2.  All six products run on
// v_mfma_f32_16x16x4_f32 (exact fp32), wave w owning the 16-column slab w of every 64-wide result (two 16-row tiles of dS):
// 288 MFMAs per wave.  The first version did them with plain FMAs fed by LDS broadcasts: 1.05 ms per layer at config 2 (150
// us per workgroup); this one is bounded by its 112 KiB of HBM traffic per item (x, params, g_out in; g_x, g_params out).
//
// FAST (in_points == 32, the decoder's T * P = 8 * 4): every global operand of the item is requested in the first instructions of
// the kernel -- x and S as float4 into LDS (S [128][36]: 16 KiB read once per workgroup instead of once per wave per product), the
// wave's slabs of M, M^T and g_out into registers -- and the workgroup synchronises on LDS traffic only (s_waitcnt lgkmcnt(0) +
// s_barrier: __syncthreads' vmcnt(0) made every LayerNorm reduction wait for the gradient stores of the product before it).
// The generic path exposes one L2 / HBM round trip per product and per 8-step operand batch (~16 per workgroup, 39 us per
// workgroup at 2 workgroups per CU: 274 us per layer at config 2); FAST exposes one.  LDS 76.7 KiB: still 2 workgroups per CU.
#include "sbev_common.hpp"
#include <cstdlib>

namespace {

constexpr int C = 64, POUT = 128, LD = 65;

struct MixBwdArgs {
    const float* x;        // [BQ, G, Pin, C]
    const float* params;   // [BQ, G, C*C + Pout*Pin]
    const float* gout;     // [BQ, G, Pout, C]
    float* gx;             // [BQ, G, Pin, C]
    float* gparams;        // [BQ, G, C*C + Pout*Pin]
    float* item_max;       // [BQ * G][4] partial maxima of |gparams| per item (one per wave; the fp16 GEMMs' scale of grad_params comes from these), or NULL
    long long n_items;
    int Pin;
    float eps;
};

// workgroup barrier for data exchanged through LDS only (global loads / stores of this wave stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool FAST>
__device__ __forceinline__ void wg_sync() {
    if constexpr (FAST) lds_barrier();
    else __syncthreads();
}

template <bool FAST>
__device__ __forceinline__ float block_sum(float v, float* red, int wave, int lane) {
    v = sbev::wave_sum_dpp(v);
    wg_sync<FAST>();
    if (lane == 0) red[wave] = v;
    wg_sync<FAST>();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

// v_mfma_f32_16x16x4_f32 operand convention used below: lane = (fi = lane & 15, fk = lane >> 4);
//   A operand: A[i = fi][k = 4s + fk];  B operand: B[k = 4s + fk][j = fi];  C/D: rows fk*4 + e (e = 0..3), column fi.
constexpr int LDS_S = 36;   // FAST: row stride of S in LDS (floats): 16-byte rows, row- and column-walking fragment reads conflict-free per half wave

template <int RT, bool FAST>      // RT = ceil(Pin / 16) row tiles of the Pin-row matrices (rows Pin..16*RT-1 are zero padding in LDS)
__global__ __launch_bounds__(256) void mixing_bwd_kernel(const MixBwdArgs a) {
    static_assert(!FAST || RT == 2, "FAST is the in_points == 32 instance");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Pin = FAST ? 32 : a.Pin;       // FAST: a compile-time constant (store addresses by shifts, no per-element bounds branches)
    constexpr int PR = RT * 16;
    float* xs = smem;                    // [PR][LD]
    float* h1 = xs + PR * LD;            // [PR][LD]   h1 = LN(y1) (pre-ReLU; pad rows 0)
    float* d1 = h1 + PR * LD;            // [PR][LD]   dy1
    float* y2 = d1 + PR * LD;            // [POUT][LD] dy2
    float* red = y2 + POUT * LD;         // [8]
    float* sl = red + 8;                 // FAST: S [POUT][LDS_S]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fk = lane >> 4;
    const int cw = wave * 16;            // this wave's 16-column slab of every 64-wide matrix
    const long long item = blockIdx.x;
    const int NP = C * C + POUT * Pin;
    const float* __restrict__ x = a.x + item * Pin * C;
    const float* __restrict__ Mg = a.params + item * NP;
    const float* __restrict__ Sg = Mg + C * C;
    const float* __restrict__ go = a.gout + item * POUT * C;
    float* __restrict__ gM = a.gparams + item * NP;
    float* __restrict__ gS = gM + C * C;
    float* __restrict__ gx = a.gx + item * Pin * C;
    const float n1cnt = (float)(Pin * C), n2cnt = (float)(POUT * C);

    float mB[C / 4], mT[C / 4];
    f32x4 gor[POUT / 16];
    if constexpr (FAST) {        // all global operands of the item, in the order they are consumed
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        const f32x4* s4 = reinterpret_cast<const f32x4*>(Sg);
        f32x4 xr[2], sr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) xr[i] = x4[tid + 256 * i];
#pragma unroll
        for (int i = 0; i < 4; ++i) sr[i] = s4[tid + 256 * i];
#pragma unroll
        for (int q = 0; q < C / 4; ++q) mB[q] = Mg[(4 * q + fk) * C + cw + fi];
#pragma unroll
        for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) gor[r][e] = go[(r * 16 + fk * 4 + e) * C + cw + fi];
#pragma unroll
        for (int q = 0; q < C / 4; ++q) mT[q] = Mg[(cw + fi) * C + 4 * q + fk];        // B[k = co][j = ci] = M[ci][co]
        __builtin_amdgcn_sched_barrier(0);       // every request is out before the first wait (the LDS writes of x)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e0 = (tid + 256 * i) * 4;
            float* d = xs + (e0 >> 6) * LD + (e0 & 63);
            d[0] = xr[i][0]; d[1] = xr[i][1]; d[2] = xr[i][2]; d[3] = xr[i][3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e0 = (tid + 256 * i) * 4;
            *reinterpret_cast<f32x4*>(sl + (e0 >> 5) * LDS_S + (e0 & 31)) = sr[i];
        }
    } else {
        for (int i = tid; i < PR * C; i += 256) xs[(i >> 6) * LD + (i & 63)] = (i >> 6) < Pin ? x[i] : 0.f;
    }
    wg_sync<FAST>();
    // ---- (1) y1 = x M, LayerNorm statistics, h1 -> LDS ----------------------------------------------------------------
    f32x4 acc1[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc1[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // operands that come from global memory are requested as ONE batch of independent loads per product (a load inside the
    // MFMA chain made every MFMA wait for its own L2 round trip: 47 us per workgroup for 4 us of matrix-core work)
    if constexpr (!FAST) {
#pragma unroll
        for (int s4 = 0; s4 < C / 4; ++s4) mB[s4] = Mg[(4 * s4 + fk) * C + cw + fi];
    }
#pragma unroll
    for (int s4 = 0; s4 < C / 4; ++s4) {
#pragma unroll
        for (int r = 0; r < RT; ++r) acc1[r] = MFMA16(xs[(r * 16 + fi) * LD + 4 * s4 + fk], mB[s4], acc1[r]);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += (r * 16 + fk * 4 + e) < Pin ? acc1[r][e] : 0.f;
    const float mean1 = block_sum<FAST>(s, red, wave, lane) / n1cnt;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = acc1[r][e] - mean1;
            s += (r * 16 + fk * 4 + e) < Pin ? d * d : 0.f;
        }
    const float rstd1 = rsqrtf(block_sum<FAST>(s, red + 4, wave, lane) / n1cnt + a.eps);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = r * 16 + fk * 4 + e;
            h1[row * LD + cw + fi] = row < Pin ? (acc1[r][e] - mean1) * rstd1 : 0.f;
        }
    wg_sync<FAST>();
    // ---- (2) y2 = S relu(h1), LayerNorm statistics, dy2 -> LDS -------------------------------------------------------
    f32x4 acc2[POUT / 16];
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r) acc2[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (FAST) {
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const float bq = fmaxf(h1[(4 * s4 + fk) * LD + cw + fi], 0.f);
#pragma unroll
            for (int r = 0; r < POUT / 16; ++r) acc2[r] = MFMA16(sl[(r * 16 + fi) * LDS_S + 4 * s4 + fk], bq, acc2[r]);
        }
    } else {
        float sA[POUT / 16], sN[POUT / 16];
#pragma unroll
        for (int r = 0; r < POUT / 16; ++r) sA[r] = Sg[(r * 16 + fi) * Pin + fk];
        for (int s4 = 0; s4 < Pin / 4; ++s4) {
            if (s4 + 1 < Pin / 4) {
#pragma unroll
                for (int r = 0; r < POUT / 16; ++r) sN[r] = Sg[(r * 16 + fi) * Pin + 4 * (s4 + 1) + fk];      // next step's operands in flight
            }
            const float bq = fmaxf(h1[(4 * s4 + fk) * LD + cw + fi], 0.f);
#pragma unroll
            for (int r = 0; r < POUT / 16; ++r) acc2[r] = MFMA16(sA[r], bq, acc2[r]);
#pragma unroll
            for (int r = 0; r < POUT / 16; ++r) sA[r] = sN[r];
        }
    }
    s = 0.f;
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc2[r][e];
    const float mean2 = block_sum<FAST>(s, red, wave, lane) / n2cnt;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = acc2[r][e] - mean2;
            s += d * d;
        }
    const float rstd2 = rsqrtf(block_sum<FAST>(s, red + 4, wave, lane) / n2cnt + a.eps);
    f32x4 g2[POUT / 16];
    float sg = 0.f, sgh = 0.f;
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = (acc2[r][e] - mean2) * rstd2;
            acc2[r][e] = h;
            float gq;
            if constexpr (FAST) gq = gor[r][e];
            else gq = go[(r * 16 + fk * 4 + e) * C + cw + fi];
            const float g = h > 0.f ? gq : 0.f;
            g2[r][e] = g;
            sg += g;
            sgh += g * h;
        }
    const float mg2 = block_sum<FAST>(sg, red, wave, lane) / n2cnt;
    const float mgh2 = block_sum<FAST>(sgh, red + 4, wave, lane) / n2cnt;
#pragma unroll
    for (int r = 0; r < POUT / 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) y2[(r * 16 + fk * 4 + e) * LD + cw + fi] = rstd2 * (g2[r][e] - mg2 - acc2[r][e] * mgh2);
    wg_sync<FAST>();
    float gmx = 0.f;
    // ---- (3) dS = dy2 n1^T : wave w owns the 16-row tiles w and w + 4 of the 128 output rows -----------------------------
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int rt = wave + 4 * rr;
        f32x4 acc[RT];
#pragma unroll
        for (int ct = 0; ct < RT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int s4 = 0; s4 < C / 4; ++s4) {
            const float aq = y2[(rt * 16 + fi) * LD + 4 * s4 + fk];
#pragma unroll
            for (int ct = 0; ct < RT; ++ct) acc[ct] = MFMA16(aq, fmaxf(h1[(ct * 16 + fi) * LD + 4 * s4 + fk], 0.f), acc[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < RT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (FAST || ct * 16 + fi < Pin) {
                    gS[(rt * 16 + fk * 4 + e) * Pin + ct * 16 + fi] = acc[ct][e];
                    gmx = fmaxf(gmx, fabsf(acc[ct][e]));
                }
    }
    // ---- (4) dn1 = S^T dy2, ReLU mask, LayerNorm-1 backward -> dy1 in LDS ------------------------------------------------
    f32x4 acc4[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc4[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (FAST) {
#pragma unroll 8
        for (int s4 = 0; s4 < POUT / 4; ++s4) {
            const float bq = y2[(4 * s4 + fk) * LD + cw + fi];
#pragma unroll
            for (int r = 0; r < RT; ++r) acc4[r] = MFMA16(sl[(4 * s4 + fk) * LDS_S + r * 16 + fi], bq, acc4[r]);
        }
    } else {
        constexpr int AHEAD = 8;                                  // steps of S^T operands requested together
        float tA[AHEAD][RT];
        for (int s0 = 0; s0 < POUT / 4; s0 += AHEAD) {
#pragma unroll
            for (int u = 0; u < AHEAD; ++u)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const int p = r * 16 + fi;
                    tA[u][r] = p < Pin ? Sg[(4 * (s0 + u) + fk) * Pin + p] : 0.f;
                }
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) {
                const float bq = y2[(4 * (s0 + u) + fk) * LD + cw + fi];
#pragma unroll
                for (int r = 0; r < RT; ++r) acc4[r] = MFMA16(tA[u][r], bq, acc4[r]);
            }
        }
    }
    sg = 0.f; sgh = 0.f;
    f32x4 hh[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = h1[(r * 16 + fk * 4 + e) * LD + cw + fi];       // pad rows hold 0 -> masked out
            hh[r][e] = h;
            const float g = h > 0.f ? acc4[r][e] : 0.f;
            acc4[r][e] = g;
            sg += g;
            sgh += g * h;
        }
    const float mg1 = block_sum<FAST>(sg, red, wave, lane) / n1cnt;
    const float mgh1 = block_sum<FAST>(sgh, red + 4, wave, lane) / n1cnt;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = r * 16 + fk * 4 + e;
            d1[row * LD + cw + fi] = row < Pin ? rstd1 * (acc4[r][e] - mg1 - hh[r][e] * mgh1) : 0.f;
        }
    wg_sync<FAST>();
    // ---- (5) dM = x^T dy1 : wave w owns output columns [cw, cw + 16), the four 16-row tiles of ci ------------------------
    {
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s4 = 0; s4 < Pin / 4; ++s4) {
            const float bq = d1[(4 * s4 + fk) * LD + cw + fi];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMA16(xs[(4 * s4 + fk) * LD + ct * 16 + fi], bq, acc[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gM[(ct * 16 + fk * 4 + e) * C + cw + fi] = acc[ct][e];
                gmx = fmaxf(gmx, fabsf(acc[ct][e]));
            }
    }
    // ---- (6) dx = dy1 M^T : wave w owns the input-channel slab ci in [cw, cw + 16) ---------------------------------------
    {
        f32x4 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (!FAST) {
#pragma unroll
            for (int s4 = 0; s4 < C / 4; ++s4) mT[s4] = Mg[(cw + fi) * C + 4 * s4 + fk];    // B[k = co][j = ci] = M[ci][co]
        }
#pragma unroll
        for (int s4 = 0; s4 < C / 4; ++s4) {
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = MFMA16(d1[(r * 16 + fi) * LD + 4 * s4 + fk], mT[s4], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = r * 16 + fk * 4 + e;
                if (row < Pin) gx[row * C + cw + fi] = acc[r][e];
            }
    }
    if (a.item_max) {       // one partial maximum per wave: no workgroup barrier at the end of the kernel
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) gmx = fmaxf(gmx, __shfl_xor(gmx, o));
        if (lane == 0) a.item_max[item * 4 + wave] = gmx;
    }
}

template <int RT, bool FAST = false>
int launch_mix_bwd(const MixBwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)((3 * RT * 16 + POUT) * LD + 8 + (FAST ? POUT * LDS_S : 0)) * sizeof(float);
    auto k = mixing_bwd_kernel<RT, FAST>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            sbev::set_error("sbev_adaptive_mixing_bwd_f32: cannot reserve %zu bytes of LDS", lds);
            return SBEV_ELAUNCH;
        }
    }
    hipLaunchKernelGGL(k, dim3((unsigned)a.n_items), dim3(256), lds, s, a);
    return sbev::check_launch("sbev_adaptive_mixing_bwd_f32");
}

}  // namespace

static int mixing_bwd(const float* x, const float* params, const float* grad_y, float* grad_x, float* grad_params, float* item_max,
                      int64_t BQ, int G, int Pin, int Cg, int Pout, float eps, sbev_stream_t stream) {
    SBEV_REQUIRE(BQ >= 0 && G >= 1, "sbev_adaptive_mixing_bwd_f32: bad sizes");
    SBEV_REQUIRE(Cg == C && Pout == POUT, "sbev_adaptive_mixing_bwd_f32: built for C=64 channels per group and 128 out points (got %d, %d)", Cg, Pout);
    SBEV_REQUIRE(Pin >= 4 && Pin % 4 == 0 && Pin <= 120, "sbev_adaptive_mixing_bwd_f32: in_points=%d must be a multiple of 4 in 4..120 (as the forward)", Pin);
    if (BQ == 0) return SBEV_OK;
    SBEV_REQUIRE(x && params && grad_y && grad_x && grad_params, "sbev_adaptive_mixing_bwd_f32: null pointer");
    SBEV_REQUIRE(BQ * G <= 0x7fffffffLL, "sbev_adaptive_mixing_bwd_f32: too many items");
    SBEV_REQUIRE((((uintptr_t)params) & 15) == 0 && (Pin * Pout) % 4 == 0, "sbev_adaptive_mixing_bwd_f32: params must be 16-byte aligned");
    MixBwdArgs a{x, params, grad_y, grad_x, grad_params, item_max, BQ * G, Pin, eps};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch ((Pin + 15) / 16) {
        case 1: return launch_mix_bwd<1>(a, s);
        case 2:
        {
            static const bool generic = getenv("SBEV_MIX_BWD_GENERIC") != nullptr;      // A/B switch: the per-product operand loads
            if (Pin == 32 && (((uintptr_t)x) & 15) == 0 && !generic) return launch_mix_bwd<2, true>(a, s);
        }
            return launch_mix_bwd<2>(a, s);
        case 3: return launch_mix_bwd<3>(a, s);
        case 4: return launch_mix_bwd<4>(a, s);
        case 5: return launch_mix_bwd<5>(a, s);
        case 6: return launch_mix_bwd<6>(a, s);
        case 7: return launch_mix_bwd<7>(a, s);
        default: return launch_mix_bwd<8>(a, s);
    }
}

extern "C" int sbev_adaptive_mixing_bwd_f32(const float* x, const float* params, const float* grad_y, float* grad_x, float* grad_params,
                                            int64_t BQ, int G, int Pin, int Cg, int Pout, float eps, sbev_stream_t stream) {
    return mixing_bwd(x, params, grad_y, grad_x, grad_params, nullptr, BQ, G, Pin, Cg, Pout, eps, stream);
}

extern "C" int sbev_adaptive_mixing_bwd_max_f32(const float* x, const float* params, const float* grad_y, float* grad_x, float* grad_params,
                                                float* item_max, int64_t BQ, int G, int Pin, int Cg, int Pout, float eps, sbev_stream_t stream) {
    SBEV_REQUIRE(item_max || BQ == 0, "sbev_adaptive_mixing_bwd_max_f32: null item_max");
    return mixing_bwd(x, params, grad_y, grad_x, grad_params, item_max, BQ, G, Pin, Cg, Pout, eps, stream);
}
