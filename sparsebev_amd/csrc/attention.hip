// Scale-adaptive self attention core (gfx950): flash-style, the [B,8,Q,Q] bias tensor never exists.
//
// Replaces SparseBEVSelfAttention.inner_forward minus its Linear layers (models/sparsebev_transformer.py:
// 210-228,236-248) and the softmax(QK^T/sqrt(d) + mask)V core of torch.nn.MultiheadAttention that mmcv's
// MultiheadAttention wraps:
//     logits[h,i,j] = (q_i . k_j)/sqrt(d) - ||c_i - c_j||_2 * tau[i,h]      (+ -inf where the DN mask is set)
//     out[i, h*d:(h+1)*d] = softmax_j(logits) @ v
// c = decoded (x, y) box centres in metres.  The in-projection (with gen_tau's 8 rows appended: one GEMM,
// N = 3D + H) runs before this kernel and the out-projection + residual + norm1 after it (sbev_linear_f32).
//
// One workgroup = 2 row groups x 16 query rows of one (batch, head), x 2 key halves.  K/V tiles of 64 keys are staged in LDS
// (16-B coalesced loads: a key's 32-float head slice is one 128-B line), S = QK^T and O += PV run on
// v_mfma_f32_16x16x4_f32 (exact fp32), the distance bias is recomputed from the centres on the fly, the
// softmax is the online (running max / running sum) form, and P goes from the MFMA C layout to the A layout
// through a per-wave 4-KiB LDS patch.  1.3 GFLOP per layer-sample: latency-, not throughput-critical.
#include "sbev_common.hpp"
#include "small_ops.hpp"

namespace {

using sbev_ops::MiscArgs;
using sbev_ops::refine_rows;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HD = 32;        // head dim (embed 256 / 8 heads)
constexpr int KT = 64;        // keys per tile
constexpr int NQ = 2;         // query row groups (16 rows each) per workgroup
constexpr int KS = 4;         // key splits per workgroup: wave (ks, qg) walks key tiles ks, ks+KS, ... of row group qg
constexpr int NWAVES = NQ * KS;
constexpr int LDK = HD + 4;   // K tile row stride (B operand of QK^T is read along d: rows = keys)
constexpr int LDV = KT + 4;    // V tile is staged TRANSPOSED, [dim][key]: the B operand of PV is then one 16-byte read along the keys
constexpr int LDP = KT + 4;   // P patch row stride

#ifdef SBEV_SASA_TRACE           // phase stamps of wave 0 of workgroup 0 (tools/exp/r4_sasa_trace.py; never in the product build)
__device__ long long g_sasa_trace[64];
#define SASA_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_sasa_trace[i] = (long long)__builtin_readcyclecounter();
#else
#define SASA_STAMP(i)
#endif

struct AttnArgs {
    const float* qkvt;          // [B, Q, ld]: q | k | v | tau
    const float* bbox;          // [B, Q, 10]: columns 0, 1 = normalised centre -> metres via lo/span (decode_bbox)
    float lo[2], span[2];
    const unsigned char* mask;  // [Q, Q] (1 = masked) or null
    float* out;                 // [B, Q, H*HD]
    int B, Q, H, ld;
    float scale;                // 1/sqrt(HD)
};

// All-reduce over the 16 lanes that share (lane >> 4), on the VALU's DPP path (no LDS round trips -- with one or
// two waves per SIMD every ds_bpermute latency would be exposed): row_mirror pairs i <-> 15-i, row_half_mirror pairs
// i <-> 7-i inside each half, then the two quad permutes; after the four steps every lane holds the full result.
#define SBEV_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl, 0xf, 0xf, true))
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, SBEV_DPP(v, 0x140));   // row_mirror
    v = fmaxf(v, SBEV_DPP(v, 0x141));   // row_half_mirror
    v = fmaxf(v, SBEV_DPP(v, 0x4e));    // quad_perm [2,3,0,1]
    v = fmaxf(v, SBEV_DPP(v, 0xb1));    // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += SBEV_DPP(v, 0x140);
    v += SBEV_DPP(v, 0x141);
    v += SBEV_DPP(v, 0x4e);
    v += SBEV_DPP(v, 0xb1);
    return v;
}

// With Q = 900 and 8 heads there are only 232 (head, 32-row) work items -- fewer than CUs -- so the workgroup
// also splits the KEYS: 8 waves = 2 row groups x 4 key quarters (two waves per SIMD), each with its own running
// (max, sum, O) that are merged through LDS at the end (the flash-decoding combine).
template <bool MASK>
__global__ __launch_bounds__(64 * NWAVES) void sasa_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[KS * KT * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[KS * HD * LDV];
    __shared__ __attribute__((aligned(16))) float Cs[KS * KT * 2];
    __shared__ __attribute__((aligned(16))) float Ps[NWAVES * 16 * LDP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SASA_STAMP(0)
    const int qg = wave % NQ, ks = wave / NQ;
    const int fi = lane & 15, fk = lane >> 4;
    const int qtiles = (a.Q + 16 * NQ - 1) / (16 * NQ);
    const int qt = blockIdx.x % qtiles;
    const int h = (blockIdx.x / qtiles) % a.H;
    const int b = blockIdx.x / (qtiles * a.H);
    const int D = a.H * HD;
    const float* base = a.qkvt + (long long)b * a.Q * a.ld;
    const int q0 = qt * 16 * NQ + qg * 16;                   // this wave's first query row

    // Q fragments (A operand: row = fi, k = 4s + fk), pre-scaled like torch's MHA (q * head_dim^-0.5)
    // k order of every MFMA chain below: lane group fk owns k = 16 blk + 4 fk + j (j = 0..3) -- any k <-> (step, lane group)
    // bijection is valid as long as A and B agree -- so that an operand row is ONE 16-byte LDS read per 4 MFMAs (the kernel
    // is LDS-latency-bound: 80 ds_read_b32 per key tile and wave before, 20 ds_read_b128 now)
    f32x4 qf[HD / 16];
    {
        const int qi = min(q0 + fi, a.Q - 1);
#pragma unroll
        for (int blk = 0; blk < HD / 16; ++blk) {
            qf[blk] = *reinterpret_cast<const f32x4*>(base + (long long)qi * a.ld + h * HD + 16 * blk + 4 * fk);
#pragma unroll
            for (int j = 0; j < 4; ++j) qf[blk][j] *= a.scale;
        }
    }
    // per-lane rows of the C layout: row r = fk*4 + e  ->  query q0 + r
    float cx[4], cy[4], tau[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qi = min(q0 + fk * 4 + e, a.Q - 1);
        cx[e] = a.bbox[((long long)b * a.Q + qi) * 10 + 0] * a.span[0] + a.lo[0];
        cy[e] = a.bbox[((long long)b * a.Q + qi) * 10 + 1] * a.span[1] + a.lo[1];
        tau[e] = base[(long long)qi * a.ld + 3 * D + h];
    }
    float m_run[4], l_run[4];
    f32x4 o_acc[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m_run[e] = -INFINITY; l_run[e] = 0.f; }
    o_acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o_acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* Pw = Ps + wave * 16 * LDP;
    const float* Kw = Ks + ks * KT * LDK;
    const float* Vw = Vs + ks * HD * LDV;
    const float* Cw = Cs + ks * KT * 2;

    // K/V staging: the workgroup fetches KS tiles (KS*64 keys) per iteration; thread -> SLOTS x (key row, float4
    // column) of K and of V, plus one key centre for tid < KS*KT.  Tiles are fetched into registers one iteration
    // AHEAD (issue-early / write-late), so their L2 latency hides under this iteration's MFMAs and softmax.
    constexpr int SLOTS = KS * KT * (HD / 4) / (64 * NWAVES);
    // clang vector types, NOT HIP's struct float4: arrays of the struct type captured by the lambdas below are not
    // promoted to registers (hipcc parked rk in LDS and rv in scratch, and waited for every load right after issuing it)
    f32x4 rk[SLOTS], rv[SLOTS];
    f32x2 rc = {0.f, 0.f};
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int i = tid + j * 64 * NWAVES;
            const int r = i / (HD / 4), c4 = (i % (HD / 4)) * 4;
            const int kj = min(k0 + r, a.Q - 1);
            const float* row = base + (long long)kj * a.ld + h * HD + c4;
            rk[j] = *reinterpret_cast<const f32x4*>(row + D);
            rv[j] = *reinterpret_cast<const f32x4*>(row + 2 * D);
        }
        // unconditional on purpose (threads >= KS*KT re-read a valid centre and drop it): a guarded load makes hipcc
        // wait vmcnt(0) right here, which would drain the K/V prefetch it was issued with
        const int kj = min(k0 + (tid % (KS * KT)), a.Q - 1);
        rc = *reinterpret_cast<const f32x2*>(a.bbox + ((long long)b * a.Q + kj) * 10);
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int i = tid + j * 64 * NWAVES;
            const int r = i / (HD / 4), c4 = (i % (HD / 4)) * 4;      // r in [0, KS*KT): tile r / KT, row r % KT
            *reinterpret_cast<f32x4*>(&Ks[r * LDK + c4]) = rk[j];
            float* vt = Vs + (r / KT) * HD * LDV + (r % KT);        // [tile][dim][key]
#pragma unroll
            for (int e = 0; e < 4; ++e) vt[(c4 + e) * LDV] = rv[j][e];
        }
        if (tid < KS * KT) { Cs[2 * tid] = rc.x * a.span[0] + a.lo[0]; Cs[2 * tid + 1] = rc.y * a.span[1] + a.lo[1]; }
    };
    fetch(0);
    stash();
    __syncthreads();
    SASA_STAMP(1)
    [[maybe_unused]] int it_ = 0;

    for (int k0 = 0; k0 < a.Q; k0 += KS * KT, ++it_) {
        const bool more = k0 + KS * KT < a.Q;
        // in flight during this iteration's compute.  Unconditional (the last iteration re-fetches tile 0 and drops
        // it): under `if (more)` hipcc copies the loaded registers at the join and waits for them right here.
        fetch(more ? k0 + KS * KT : 0);
        SASA_STAMP(2 + 8 * it_)
        const int kbase = k0 + ks * KT;                        // this wave's key tile
        if (kbase < a.Q) {
            // S = (Q/sqrt(d)) K^T : 4 key sub-tiles of 16
            f32x4 s_acc[KT / 16];
#pragma unroll
            for (int c = 0; c < KT / 16; ++c) {
                s_acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int blk = 0; blk < HD / 16; ++blk) {
                    const f32x4 kb = *reinterpret_cast<const f32x4*>(&Kw[(c * 16 + fi) * LDK + 16 * blk + 4 * fk]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s_acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[blk][j], kb[j], s_acc[c], 0, 0, 0);
                }
            }
            SASA_STAMP(3 + 8 * it_)
            // + distance bias, masks; tile row max.  C layout: column (key) = fi, row (query) = fk*4 + e
            float tmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int c = 0; c < KT / 16; ++c) {
                const int kj = kbase + c * 16 + fi;
                const float kx = Cw[2 * (c * 16 + fi)], ky = Cw[2 * (c * 16 + fi) + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dx = cx[e] - kx, dy = cy[e] - ky;
                    // v_sqrt_f32 (1 ulp): |d(dist*tau)| <= ~2e-5 on logits that are O(10), far inside the budget
                    float v = s_acc[c][e] - __builtin_amdgcn_sqrtf(dx * dx + dy * dy) * tau[e];
                    bool dead = kj >= a.Q;
                    if (MASK) {
                        const int qi = min(q0 + fk * 4 + e, a.Q - 1);
                        dead = dead || a.mask[(long long)qi * a.Q + min(kj, a.Q - 1)] != 0;
                    }
                    v = dead ? -INFINITY : v;
                    s_acc[c][e] = v;
                    tmax[e] = fmaxf(tmax[e], v);
                }
            }
            float alpha[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m_new = fmaxf(m_run[e], row16_max(tmax[e]));
                const float m_use = m_new == -INFINITY ? 0.f : m_new;   // fully masked so far: keep everything 0
                alpha[e] = __expf(m_run[e] - m_use);                    // exp(-inf) = 0 on the first tile
                m_run[e] = m_new;
                float psum = 0.f;
#pragma unroll
                for (int c = 0; c < KT / 16; ++c) {
                    const float p = __expf(s_acc[c][e] - m_use);
                    s_acc[c][e] = p;
                    psum += p;
                }
                l_run[e] = l_run[e] * alpha[e] + row16_sum(psum);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) o_acc[t][e] *= alpha[e];
            SASA_STAMP(4 + 8 * it_)
            // P: C layout -> LDS -> A layout (row = fi, k = key)
#pragma unroll
            for (int c = 0; c < KT / 16; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) Pw[(fk * 4 + e) * LDP + c * 16 + fi] = s_acc[c][e];
            // the patch is private to this wave and a wave's DS operations execute in issue order, so the reads
            // below see the writes above without a workgroup barrier; only keep the compiler from reordering them
            __builtin_amdgcn_wave_barrier();
            // O += P V : 2 column tiles of 16 dims, K = 64 keys
#pragma unroll
            for (int blk = 0; blk < KT / 16; ++blk) {
                const f32x4 pa = *reinterpret_cast<const f32x4*>(&Pw[fi * LDP + 16 * blk + 4 * fk]);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Vw[fi * LDV + 16 * blk + 4 * fk]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Vw[(16 + fi) * LDV + 16 * blk + 4 * fk]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o_acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[j], v0[j], o_acc[0], 0, 0, 0);
                    o_acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[j], v1[j], o_acc[1], 0, 0, 0);
                }
            }
        }
        SASA_STAMP(5 + 8 * it_)
        __syncthreads();                                       // every wave is done with these K/V/centre tiles
        SASA_STAMP(6 + 8 * it_)
        if (more) {
            stash();
            SASA_STAMP(7 + 8 * it_)
            __syncthreads();
        }
        SASA_STAMP(8 + 8 * it_)
    }
    // merge the KS key-split partials of each row group (flash-decoding combine) through LDS, then normalise.
    // Ks is free now: slot layout [ks][qg][12 values][64 lanes]
    float* mg = Ks;
    if (ks > 0) {
        float* d = mg + ((ks - 1) * NQ + qg) * 12 * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[(0 + e) * 64 + lane] = m_run[e];
            d[(4 + e) * 64 + lane] = o_acc[0][e];
            d[(8 + e) * 64 + lane] = o_acc[1][e];
        }
    }
    float* lg = Vs;                                            // l_run partials, same indexing with 4 values
    if (ks > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) lg[(((ks - 1) * NQ + qg) * 4 + e) * 64 + lane] = l_run[e];
    }
    __syncthreads();
    if (ks == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m = m_run[e], l = l_run[e], o0 = o_acc[0][e], o1 = o_acc[1][e];
#pragma unroll
            for (int z = 1; z < KS; ++z) {
                const float* d = mg + ((z - 1) * NQ + qg) * 12 * 64;
                const float m2 = d[(0 + e) * 64 + lane];
                const float l2 = lg[(((z - 1) * NQ + qg) * 4 + e) * 64 + lane];
                const float mn = fmaxf(m, m2);
                const float mu = mn == -INFINITY ? 0.f : mn;
                const float f1 = __expf(m - mu), f2 = __expf(m2 - mu);
                o0 = o0 * f1 + d[(4 + e) * 64 + lane] * f2;
                o1 = o1 * f1 + d[(8 + e) * 64 + lane] * f2;
                l = l * f1 + l2 * f2;
                m = mn;
            }
            const int qi = q0 + fk * 4 + e;
            if (qi < a.Q) {
                const float inv = 1.f / l;
                float* o = a.out + ((long long)b * a.Q + qi) * D + h * HD;
                o[fi] = o0 * inv;                               // C layout: column = dim (fi), row = query
                o[16 + fi] = o1 * inv;
            }
        }
    }
    SASA_STAMP(63)
}

#ifdef SBEV_SASA_TRACE
}  // namespace
extern "C" int sbev_debug_sasa_trace_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sasa_trace), sizeof(g_sasa_trace)); }
namespace {
#endif

__global__ __launch_bounds__(256) void refine_kernel(const MiscArgs a) { refine_rows(a, blockIdx.x); }

}  // namespace

extern "C" int sbev_sasa_f32(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                             const uint8_t* mask, float* out, int B, int Q, int H, int head_dim, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && H >= 1, "sbev_sasa_f32: bad sizes");
    SBEV_REQUIRE(head_dim == HD, "sbev_sasa_f32: built for head_dim 32 (got %d)", head_dim);
    SBEV_REQUIRE(ld >= 3 * H * HD + H && ld % 4 == 0, "sbev_sasa_f32: row stride %lld must be >= 3*H*32 + H and a multiple of 4", (long long)ld);
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(qkvt && query_bbox && pc_range && out, "sbev_sasa_f32: null pointer");
    SBEV_REQUIRE((((uintptr_t)qkvt) & 15) == 0 && (((uintptr_t)query_bbox) & 7) == 0, "sbev_sasa_f32: qkvt must be 16-byte, query_bbox 8-byte aligned");
    AttnArgs a{};
    a.qkvt = qkvt; a.bbox = query_bbox; a.mask = mask; a.out = out;
    a.B = B; a.Q = Q; a.H = H; a.ld = (int)ld; a.scale = 1.0f / sqrtf((float)HD);
    for (int i = 0; i < 2; ++i) {           // decode_bbox: python-float scalars cast to fp32 (models/bbox/utils.py:69-70)
        a.lo[i] = (float)pc_range[i];
        a.span[i] = (float)(pc_range[3 + i] - pc_range[i]);
    }
    const long long blocks = (long long)B * H * ((Q + 16 * NQ - 1) / (16 * NQ));
    SBEV_REQUIRE(blocks <= 0x7fffffffLL, "sbev_sasa_f32: too many blocks");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (mask)
        hipLaunchKernelGGL(sasa_kernel<true>, dim3((unsigned)blocks), dim3(64 * NWAVES), 0, s, a);
    else
        hipLaunchKernelGGL(sasa_kernel<false>, dim3((unsigned)blocks), dim3(64 * NWAVES), 0, s, a);
    return sbev::check_launch("sbev_sasa_f32");
}

extern "C" int sbev_refine_bbox(const float* query_bbox, const float* reg, const float* vel_div, float* out,
                                int B, int Q, int code_size, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && code_size >= 3, "sbev_refine_bbox: bad sizes");
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(query_bbox && reg && out, "sbev_refine_bbox: null pointer");
    MiscArgs a{};
    a.bbox = query_bbox; a.reg = reg; a.vel_div = vel_div; a.out = out;
    a.BQ = (long long)B * Q; a.Q = Q; a.code = code_size;
    SBEV_REQUIRE(a.BQ <= 0x7fffffffLL, "sbev_refine_bbox: too many rows");
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)((a.BQ + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_refine_bbox");
}
