// Row-wise operator chains of a decoder layer in ONE launch (gfx950).
//
// Between the four big kernels of a layer (attention, sampler, mixing GEMMs) the reference runs ~20 tiny per-query
// ops: Linear(256->256/512/776/112/10), LayerNorm, ReLU, residual adds, refine_bbox (models/sparsebev_transformer.py:
// 116-153,162-187 and the mmcv FFN / MultiheadAttention projections).  As separate kernels each costs 5-7 us of launch
// + dependency latency for < 1 us of work (20 launches = 100 us per layer at Q = 900).  All of them are independent per
// query ROW, so here a workgroup owns 16 rows and runs a whole chain of them back to back:
//   * activations stay in LDS (three 16 x 512 fp32 buffers), only tensors other kernels need are written to HBM;
//   * every Linear is v_mfma_f32_16x16x4_f32 with the weight matrix as the row operand, its fragments streamed from
//     L2 straight into an 8-step register ring (no LDS staging, requested 8 k-steps = ~2000 MFMA cycles ahead and
//     ACROSS op boundaries, so LayerNorm / barrier phases never restart the stream cold);
//   * a wave owns 64 output columns per pass (4 accumulators -> dependent MFMAs are 4 apart), LayerNorm is a 16-lane
//     DPP row reduction (16 threads per row).
// One wave per SIMD by design (the ring + pointers want ~200 registers and 57 workgroups cannot fill the chip anyway);
// the chain is MFMA-bound at 3.6 us per 256x256 Linear, i.e. ~50 us per layer instead of ~100 us of launches.
#include "sbev_common.hpp"

namespace {

typedef float f32x4c __attribute__((ext_vector_type(4)));

constexpr int ROWS = 16, LDA = 516, MAX_OPS = SBEV_CHAIN_MAX_OPS;   // LDA: 512 + 4 floats (16-byte aligned rows)

struct ChainArgs {
    sbev_chain_op op[MAX_OPS];
    int n_ops;
    int M;
    float eps;
};

__device__ __forceinline__ float row16_sum(float v) {       // sum over the 16 lanes of a DPP row; every lane gets it
#define SBEV_DPP_F(x, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), ctrl, 0xf, 0xf, true))
    v += SBEV_DPP_F(v, 0x140);
    v += SBEV_DPP_F(v, 0x141);
    v += SBEV_DPP_F(v, 0x4e);
    v += SBEV_DPP_F(v, 0xb1);
#undef SBEV_DPP_F
    return v;
}

// next LINEAR pass of this wave after (oi, quad): passes of one op are quads wave, wave+4, ...; returns false at the end
__device__ __forceinline__ bool next_pass(const ChainArgs& a, int wave, int& oi, int& quad) {
    int o = oi, q = quad + 4;
    for (;;) {
        if (o >= a.n_ops) return false;
        if (a.op[o].kind == SBEV_CHAIN_LINEAR) {
            const int nq = (a.op[o].N + 63) / 64;
            if (q < nq) { oi = o; quad = q; return true; }
        }
        ++o;
        q = wave;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void row_chain_kernel(const ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float buf[3][ROWS][LDA];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fk = lane >> 4;
    const int r0 = blockIdx.x * ROWS;
    const int M = a.M;
    // row-phase mapping: 16 threads per row
    const int prow = tid >> 4, pl = tid & 15;
    const int grow = r0 + prow;                       // global row of the row phases
    const bool prow_ok = grow < M;
    const long long grow_c = prow_ok ? grow : M - 1;  // clamped for loads

    // ---- weight ring: wr[slot][c] = W[(quad*4 + c)*16 + fi][16*step + 4*fk .. +3], slot = step & 7 ------------------
    f32x4c wr[8][4];
    const float* wp[4];                               // row pointers of the pass being PREFETCHED
    int pf_oi = 0, pf_quad = wave - 4;                // lookahead cursor
    bool pf_ok = next_pass(a, wave, pf_oi, pf_quad);
#define SBEV_SET_WP(oi_, quad_)                                                             \
    {                                                                                      \
        const sbev_chain_op& o_ = a.op[oi_];                                               \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                    \
            int n_ = ((quad_) * 4 + c) * 16 + fi;                                          \
            n_ = n_ < o_.N ? n_ : o_.N - 1;       /* rows past N: clamped, never stored */ \
            wp[c] = o_.W + (long long)n_ * o_.K + 4 * fk;                                  \
        }                                                                                  \
    }
    int pf_K = 0;
    if (pf_ok) {
        SBEV_SET_WP(pf_oi, pf_quad);
        pf_K = a.op[pf_oi].K;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) wr[s][c] = *reinterpret_cast<const f32x4c*>(wp[c] + 16 * s);
    }
    int pf_step = 8;                                  // next step of the prefetched pass to request (K/16 steps per pass)

    for (int oi = 0; oi < a.n_ops; ++oi) {
        const sbev_chain_op& o = a.op[oi];
        if (o.kind == SBEV_CHAIN_LINEAR) {
            const int nq = (o.N + 63) / 64, nsteps = o.K / 16;
            for (int quad = wave; quad < nq; quad += 4) {
                // (pf_oi, pf_quad) == (oi, quad) here: the ring holds steps [pf_step - 8, pf_step) of THIS pass
                f32x4c acc[4];
                bool fvalid[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = (f32x4c){0.f, 0.f, 0.f, 0.f};
                    fvalid[c] = (quad * 4 + c) * 16 < o.N;
                }
                const float* arow = &buf[o.src][fi][4 * fk];
                for (int s0 = 0; s0 < nsteps; s0 += 8) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const f32x4c av = *reinterpret_cast<const f32x4c*>(arow + 16 * (s0 + s));
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (fvalid[c]) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][c][i], av[i], acc[c], 0, 0, 0);
                        // refill this slot: the same pass 8 steps on, or the first steps of the wave's next pass
                        if (pf_step >= pf_K / 16 && pf_ok) {       // the prefetched pass is fully requested: move on
                            pf_ok = next_pass(a, wave, pf_oi, pf_quad);
                            if (pf_ok) {
                                SBEV_SET_WP(pf_oi, pf_quad);
                                pf_K = a.op[pf_oi].K;
                                pf_step = 0;
                            }
                        }
                        if (pf_ok) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) wr[s][c] = *reinterpret_cast<const f32x4c*>(wp[c] + 16 * pf_step);
                            ++pf_step;
                        }
                    }
                }
                // ---- pass epilogue: bias, ReLU, residual, to LDS / HBM; lane: row m = fi, columns nb + 0..3 ------------
                const int grow_m = r0 + fi;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int nb = (quad * 4 + c) * 16 + 4 * fk;
                    if (!fvalid[c]) continue;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n = nb + i;
                        float x = acc[c][i];
                        if (n < o.N) {
                            if (o.bias) x += o.bias[n];
                            if (o.relu) x = fmaxf(x, 0.f);
                            if (o.res_buf >= 0) x += buf[o.res_buf][fi][n];
                            if (o.res_g) x += o.res_g[(long long)(grow_m < M ? grow_m : M - 1) * o.ld_res + n];
                        }
                        v[i] = x;
                    }
                    if (o.to_lds && nb < o.N) *reinterpret_cast<f32x4c*>(&buf[o.dst][fi][nb]) = (f32x4c){v[0], v[1], v[2], v[3]};
                    if (o.out_g && !o.ln && grow_m < M) {
                        float* og = o.out_g + (long long)grow_m * o.ld_out + nb;
                        if (nb + 3 < o.N && (o.ld_out & 3) == 0) {
                            *reinterpret_cast<f32x4c*>(og) = (f32x4c){v[0], v[1], v[2], v[3]};
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (nb + i < o.N) og[i] = v[i];
                        }
                    }
                }
            }
            __syncthreads();
        } else if (o.kind == SBEV_CHAIN_LOAD) {
            // rows [M, ld_in] (N columns, N % 4 == 0) -> LDS buffer dst
            const float* src = o.W + grow_c * o.ld_in;
            for (int n = 4 * pl; n < o.N; n += 64)
                *reinterpret_cast<f32x4c*>(&buf[o.dst][prow][n]) = *reinterpret_cast<const f32x4c*>(src + n);
            __syncthreads();
        } else if (o.kind == SBEV_CHAIN_LINEAR3) {
            // Linear(3 -> N) on the first 3 columns of buffer src: 3 FMAs per output are not a GEMM
            const float x0 = buf[o.src][prow][0], x1 = buf[o.src][prow][1], x2 = buf[o.src][prow][2];
            for (int n = 4 * pl; n < o.N; n += 64) {
                f32x4c r;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* w = o.W + (long long)(n + i) * 3;
                    float y = o.bias ? o.bias[n + i] : 0.f;
                    y = fmaf(w[0], x0, y);
                    y = fmaf(w[1], x1, y);
                    y = fmaf(w[2], x2, y);
                    r[i] = o.relu ? fmaxf(y, 0.f) : y;
                }
                *reinterpret_cast<f32x4c*>(&buf[o.dst][prow][n]) = r;
            }
            __syncthreads();
        } else if (o.kind == SBEV_CHAIN_REFINE) {
            // refine_bbox + velocity / time_diff (models/sparsebev_transformer.py:155-160,179-183): reg = buffer src
            // columns 0..N-1, previous boxes = rows [M, ld_in] of W; result -> out_g rows and (to_lds) buffer dst
            if (pl < 1) {
                const float* bb = o.W + grow_c * o.ld_in;
                for (int d = 0; d < o.N; ++d) {
                    float v = buf[o.src][prow][d];
                    if (d < 3) {
                        const float p = fminf(fmaxf(bb[d], 0.f), 1.f);
                        const float logit = logf(fmaxf(p, 1e-5f) / fmaxf(1.f - p, 1e-5f));
                        v = 1.f / (1.f + expf(-(v + logit)));
                    } else if (d >= 8 && o.aux) {
                        v = v / o.aux[grow_c / o.aux_i];
                    }
                    if (o.to_lds) buf[o.dst][prow][d] = v;
                    if (o.out_g && prow_ok) o.out_g[(long long)grow * o.ld_out + d] = v;
                }
            }
            __syncthreads();
        }
        // ---- post phase of LINEAR / LINEAR3 / LOAD: LayerNorm(N) [+ ReLU] [+ add] [-> HBM], 16 threads per row -------
        if (o.ln && o.kind != SBEV_CHAIN_REFINE) {
            float* x = &buf[o.dst][prow][0];
            const float inv_n = 1.f / (float)o.N;
            float s = 0.f;
            for (int n = 4 * pl; n < o.N; n += 64) {
                const f32x4c v = *reinterpret_cast<const f32x4c*>(x + n);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
            const float mean = row16_sum(s) * inv_n;
            float q = 0.f;
            for (int n = 4 * pl; n < o.N; n += 64) {
                const f32x4c v = *reinterpret_cast<const f32x4c*>(x + n);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            const float rstd = rsqrtf(row16_sum(q) * inv_n + a.eps);
            for (int n = 4 * pl; n < o.N; n += 64) {
                f32x4c v = *reinterpret_cast<const f32x4c*>(x + n);
                const f32x4c g = *reinterpret_cast<const f32x4c*>(o.ln_w + n);
                const f32x4c b = *reinterpret_cast<const f32x4c*>(o.ln_b + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float y = (v[i] - mean) * rstd * g[i] + b[i];
                    if (o.ln == 2) y = fmaxf(y, 0.f);
                    v[i] = y;
                }
                if (o.add_buf >= 0) v += *reinterpret_cast<const f32x4c*>(&buf[o.add_buf][prow][n]);
                if (o.add_g) v += *reinterpret_cast<const f32x4c*>(o.add_g + grow_c * o.ld_add + n);
                *reinterpret_cast<f32x4c*>(x + n) = v;
                if (o.out_g && prow_ok) *reinterpret_cast<f32x4c*>(o.out_g + (long long)grow * o.ld_out + n) = v;
            }
            __syncthreads();
        }
    }
}

}  // namespace

extern "C" int sbev_row_chain(const sbev_chain_op* ops, int n_ops, int64_t M, float ln_eps, sbev_stream_t stream) {
    SBEV_REQUIRE(ops && n_ops >= 1 && n_ops <= MAX_OPS, "sbev_row_chain: 1..%d ops (got %d)", MAX_OPS, n_ops);
    SBEV_REQUIRE(M >= 0 && M < (1LL << 31) - 16, "sbev_row_chain: bad row count");
    if (M == 0) return SBEV_OK;
    ChainArgs a{};
    a.n_ops = n_ops; a.M = (int)M; a.eps = ln_eps;
    for (int i = 0; i < n_ops; ++i) {
        const sbev_chain_op& o = ops[i];
        SBEV_REQUIRE(o.kind >= SBEV_CHAIN_LOAD && o.kind <= SBEV_CHAIN_REFINE, "sbev_row_chain: op %d: unknown kind %d", i, o.kind);
        SBEV_REQUIRE(o.dst >= 0 && o.dst <= 2 && o.src >= 0 && o.src <= 2, "sbev_row_chain: op %d: LDS buffers are 0..2", i);
        SBEV_REQUIRE(o.res_buf <= 2 && o.add_buf <= 2, "sbev_row_chain: op %d: LDS buffers are 0..2", i);
        SBEV_REQUIRE(o.W != nullptr, "sbev_row_chain: op %d: null source / weight pointer", i);
        if (o.kind == SBEV_CHAIN_LINEAR) {
            SBEV_REQUIRE(o.K >= 128 && o.K <= 512 && o.K % 128 == 0, "sbev_row_chain: op %d: K = %d (need 128, 256, 384 or 512)", i, o.K);
            SBEV_REQUIRE(o.N >= 1 && o.N <= 1024 && (!o.to_lds || (o.N <= 512 && o.N % 4 == 0)), "sbev_row_chain: op %d: N = %d", i, o.N);
            SBEV_REQUIRE(o.src != o.dst || !o.to_lds, "sbev_row_chain: op %d: a Linear cannot write the buffer it reads", i);
            SBEV_REQUIRE(((uintptr_t)o.W & 15) == 0, "sbev_row_chain: op %d: W not 16-byte aligned", i);
            SBEV_REQUIRE(!o.ln || o.to_lds, "sbev_row_chain: op %d: LayerNorm needs the result in LDS", i);
        } else if (o.kind == SBEV_CHAIN_LOAD) {
            SBEV_REQUIRE(o.N >= 4 && o.N <= 512 && o.N % 4 == 0 && o.ld_in % 4 == 0 && ((uintptr_t)o.W & 15) == 0, "sbev_row_chain: op %d: LOAD of %d columns (ld %lld)", i, o.N, (long long)o.ld_in);
        } else if (o.kind == SBEV_CHAIN_LINEAR3) {
            SBEV_REQUIRE(o.N >= 4 && o.N <= 512 && o.N % 4 == 0 && o.src != o.dst, "sbev_row_chain: op %d: LINEAR3 N = %d", i, o.N);
        } else {
            SBEV_REQUIRE(o.N >= 10 && o.N <= 64 && (!o.aux || o.aux_i >= 1) && (o.src != o.dst || !o.to_lds), "sbev_row_chain: op %d: REFINE code size %d", i, o.N);
        }
        if (o.ln) {
            SBEV_REQUIRE(o.kind != SBEV_CHAIN_REFINE && o.ln_w && o.ln_b && o.N % 4 == 0 && o.N <= 512, "sbev_row_chain: op %d: LayerNorm arguments", i);
            SBEV_REQUIRE((((uintptr_t)o.ln_w | (uintptr_t)o.ln_b | (uintptr_t)o.add_g | (uintptr_t)o.out_g) & 15) == 0 && o.ld_add % 4 == 0 && o.ld_out % 4 == 0,
                         "sbev_row_chain: op %d: LayerNorm operands must be 16-byte aligned", i);
        }
        a.op[i] = o;
    }
    const long long blocks = (M + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(row_chain_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_row_chain");
}
