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
// (16-B coalesced loads: a key's 32-float head slice is one 128-B line), S^T = K Q^T and O^T += V^T P^T run on
// v_mfma_f32_16x16x4_f32 (exact fp32), the distance bias is recomputed from the centres on the fly, the
// softmax is the online (running max / running sum) form, and P^T stays in the registers it was computed in: in the transposed
// formulation it IS the B operand of the second product.  1.3 GFLOP per layer-sample: latency-, not throughput-critical.
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

// All-reduce over the 4 lanes {fi, fi + 16, fi + 32, fi + 48} that hold one query's keys, without LDS: v_permlane32_swap(x, x) leaves
// [lo, lo] / [hi, hi] in the two results (the halves of the wave exchanged), v_permlane16_swap(x, x) does the same with the odd / even
// 16-lane rows -- two swaps and two combines for the four rows (the 16-lane DPP reductions of the round-1..3 kernel took 4 + 4 per
// row, and there were four rows per lane).
__device__ __forceinline__ float rows4_max(float v) {
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// With Q = 900 and 8 heads there are only 232 (head, 32-row) work items -- fewer than CUs -- so the workgroup
// also splits the KEYS: 8 waves = 2 row groups x 4 key quarters (two waves per SIMD), each with its own running
// (max, sum, O) that are merged through LDS at the end (the flash-decoding combine).
//
// Round 4: the TRANSPOSED formulation.  A cycle trace (tools/exp/r4_sasa_trace.py) put the MFMAs at 30 % of a key-tile iteration;
// the rest was VALU -- the distance bias and the softmax bookkeeping of FOUR query rows per lane (4 x (16-lane DPP max + sum), 4
// alphas), the P patch through LDS (C layout -> A layout) -- and barriers.  S^T = K Q^T puts the keys in the C layout's rows and ONE
// query in a lane (column fi): the row statistics are in-lane over 16 values + a 4-row all-reduce (2 permlane swaps), there is one
// (m, l, alpha, tau, centre) per lane instead of four, and P^T is ALREADY the B operand of O^T = V^T P^T (lane (fk, fi) holds keys
// 16 c + 4 fk + e of query fi = exactly B[k][n] of MFMA step (c, e)) -- no LDS round trip, no P patch.  O^T leaves a lane with 4
// consecutive dims of its query: the output is two 16-byte stores per lane.
template <bool MASK>
__global__ __launch_bounds__(64 * NWAVES) void sasa_kernel(const AttnArgs a) {
    // K / V / centre tiles, DOUBLE-buffered (the P patch's 35 KB are gone): iteration i computes from buffer i & 1 while the tiles of
    // iteration i + 1 are written to the other one -- one workgroup barrier per iteration instead of two, and the two waves of a SIMD
    // (w and w + 4) need not be in the same phase: waves 4 .. 7 do their tile I/O (stash the prefetched registers, request the
    // tiles after next) at the START of an iteration, waves 0 .. 3 at its end, so one wave's MFMA phases run beside the other's
    // softmax / stash / fetch (in lock-step behind two barriers both waves wanted the matrix pipe, then both left it idle)
    __shared__ __attribute__((aligned(16))) float Ks[2 * KS * KT * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[2 * KS * HD * LDV];
    __shared__ __attribute__((aligned(16))) float Cs[2 * KS * KT * 2];

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
    const int qi = min(q0 + fi, a.Q - 1);                    // this lane's query (clamped: a ragged last tile repeats the last row)

    // Q^T fragments (B operand of S^T = K Q^T: n = query fi, k = d), pre-scaled like torch's MHA (q * head_dim^-0.5)
    // k order of every MFMA chain below: lane group fk owns k = 16 blk + 4 fk + j (j = 0..3) -- any k <-> (step, lane group)
    // bijection is valid as long as A and B agree -- so that an operand row is ONE 16-byte LDS read per 4 MFMAs
    f32x4 qf[HD / 16];
#pragma unroll
    for (int blk = 0; blk < HD / 16; ++blk) {
        qf[blk] = *reinterpret_cast<const f32x4*>(base + (long long)qi * a.ld + h * HD + 16 * blk + 4 * fk);
#pragma unroll
        for (int j = 0; j < 4; ++j) qf[blk][j] *= a.scale;
    }
    const float cx = a.bbox[((long long)b * a.Q + qi) * 10 + 0] * a.span[0] + a.lo[0];
    const float cy = a.bbox[((long long)b * a.Q + qi) * 10 + 1] * a.span[1] + a.lo[1];
    const float tau = base[(long long)qi * a.ld + 3 * D + h];
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o_acc[2];                                          // O^T: dims 16 t + 4 fk + e of query fi
    o_acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o_acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int KBUF = KS * KT * LDK, VBUF = KS * HD * LDV, CBUF = KS * KT * 2;

    // K/V staging: the workgroup fetches KS tiles (KS*64 keys) per iteration; thread -> SLOTS x (key row, float4
    // column) of K and of V, plus one key centre for tid < KS*KT.  Tiles are fetched into registers one iteration
    // AHEAD (issue-early / write-late), so their L2 latency hides under this iteration's MFMAs and softmax.
    constexpr int SLOTS = KS * KT * (HD / 4) / (64 * NWAVES);
    // clang vector types, NOT HIP's struct float4: arrays of the struct type captured by the lambdas below are not
    // promoted to registers (hipcc parked rk in LDS and rv in scratch, and waited for every load right after issuing it)
    f32x4 rk[SLOTS], rv[SLOTS];
    f32x2 rc = {0.f, 0.f};
    // 32-bit element offsets on the wave-uniform base (round 4: the 64-bit address arithmetic of 9 loads was 1 000 cycles per iteration)
    const unsigned hoff = (unsigned)(h * HD);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int i = tid + j * 64 * NWAVES;
            const int r = i / (HD / 4), c4 = (i % (HD / 4)) * 4;
            const unsigned kj = (unsigned)min(k0 + r, a.Q - 1);
            const unsigned off = kj * (unsigned)a.ld + hoff + (unsigned)c4;
            rk[j] = *reinterpret_cast<const f32x4*>(base + off + (unsigned)D);
            rv[j] = *reinterpret_cast<const f32x4*>(base + off + 2u * (unsigned)D);
        }
        // unconditional on purpose (threads >= KS*KT re-read a valid centre and drop it): a guarded load makes hipcc
        // wait vmcnt(0) right here, which would drain the K/V prefetch it was issued with
        const unsigned kj = (unsigned)min(k0 + (tid % (KS * KT)), a.Q - 1);
        rc = *reinterpret_cast<const f32x2*>(a.bbox + ((long long)b * a.Q) * 10 + kj * 10u);
    };
    auto stash = [&](int buf) {
        float* Kb = Ks + buf * KBUF;
        float* Vb = Vs + buf * VBUF;
        float* Cb = Cs + buf * CBUF;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int i = tid + j * 64 * NWAVES;
            const int r = i / (HD / 4), c4 = (i % (HD / 4)) * 4;      // r in [0, KS*KT): tile r / KT, row r % KT
            *reinterpret_cast<f32x4*>(&Kb[r * LDK + c4]) = rk[j];
            float* vt = Vb + (r / KT) * HD * LDV + (r % KT);        // [tile][dim][key]
#pragma unroll
            for (int e = 0; e < 4; ++e) vt[(c4 + e) * LDV] = rv[j][e];
        }
        if (tid < KS * KT) { Cb[2 * tid] = rc.x * a.span[0] + a.lo[0]; Cb[2 * tid + 1] = rc.y * a.span[1] + a.lo[1]; }
    };
    constexpr int STEP = KS * KT;
    const bool late_io = wave < NWAVES / 2;                    // waves 0 .. 3: tile I/O at the END of an iteration (their SIMD partners: at the start)
    fetch(0);
    stash(0);
    fetch(STEP < a.Q ? STEP : 0);                              // the registers now hold iteration 1's tiles (a dummy if there is none)
    __syncthreads();
    SASA_STAMP(1)
    [[maybe_unused]] int it_ = 0;

    for (int k0 = 0; k0 < a.Q; k0 += STEP, ++it_) {
        // tile I/O of this iteration: the prefetched registers (iteration it_ + 1) -> the other buffer, then the request for iteration
        // it_ + 2.  Unconditional loads (past the end: tile 0, dropped): under a branch hipcc copies the loaded registers at the join and
        // waits for them right there.  The other buffer was last read in iteration it_ - 1: everybody is past the barrier that ended it.
        auto tile_io = [&]() {
            if (k0 + STEP < a.Q) stash((it_ + 1) & 1);
            fetch(k0 + 2 * STEP < a.Q ? k0 + 2 * STEP : 0);
        };
        if (!late_io) tile_io();
        SASA_STAMP(2 + 8 * it_)
        const float* Kw = Ks + (it_ & 1) * KBUF + ks * KT * LDK;
        const float* Vw = Vs + (it_ & 1) * VBUF + ks * HD * LDV;
        const float* Cw = Cs + (it_ & 1) * CBUF + ks * KT * 2;
        const int kbase = k0 + ks * KT;                        // this wave's key tile
        if (kbase < a.Q) {
            // S^T = K (Q/sqrt(d))^T : 4 key sub-tiles of 16; C layout: row (key) = 16 c + 4 fk + e, column (query) = fi
            f32x4 s_acc[KT / 16];
#pragma unroll
            for (int c = 0; c < KT / 16; ++c) s_acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < HD / 16; ++blk) {
                f32x4 kb[KT / 16];
#pragma unroll
                for (int c = 0; c < KT / 16; ++c) kb[c] = *reinterpret_cast<const f32x4*>(&Kw[(c * 16 + fi) * LDK + 16 * blk + 4 * fk]);
                // consecutive MFMAs on DIFFERENT accumulators (c inner): a dependent 16x16x4 chain issues every 40 cycles instead of 32
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < KT / 16; ++c) s_acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[c][j], qf[blk][j], s_acc[c], 0, 0, 0);
            }
            SASA_STAMP(3 + 8 * it_)
            // + distance bias, masks; this query's tile maximum: 16 keys in the lane, the other 48 in lanes fi + 16 / 32 / 48
            const bool ragged = kbase + KT > a.Q;                 // (wave-uniform: only the last tile of the key range pays for the guard)
            float tmax = -INFINITY;
#pragma unroll
            for (int c = 0; c < KT / 16; ++c) {
                const f32x4 k01 = *reinterpret_cast<const f32x4*>(&Cw[2 * (c * 16 + 4 * fk)]);          // centres of keys 4 fk, 4 fk + 1
                const f32x4 k23 = *reinterpret_cast<const f32x4*>(&Cw[2 * (c * 16 + 4 * fk) + 4]);      // ... + 2, + 3
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float kx = e < 2 ? k01[2 * e] : k23[2 * e - 4], ky = e < 2 ? k01[2 * e + 1] : k23[2 * e - 3];
                    const float dx = cx - kx, dy = cy - ky;
                    // v_sqrt_f32 (1 ulp): |d(dist*tau)| <= ~2e-5 on logits that are O(10), far inside the budget
                    float v = s_acc[c][e] - __builtin_amdgcn_sqrtf(dx * dx + dy * dy) * tau;
                    if (MASK || ragged) {
                        const int kj = kbase + c * 16 + 4 * fk + e;
                        bool dead = kj >= a.Q;
                        if (MASK) dead = dead || a.mask[(long long)qi * a.Q + min(kj, a.Q - 1)] != 0;
                        v = dead ? -INFINITY : v;
                    }
                    s_acc[c][e] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
            const float m_new = fmaxf(m_run, rows4_max(tmax));
            const float m_use = m_new == -INFINITY ? 0.f : m_new;   // fully masked so far: keep everything 0
            const float alpha = __expf(m_run - m_use);              // exp(-inf) = 0 on the first tile
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int c = 0; c < KT / 16; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __expf(s_acc[c][e] - m_use);
                    s_acc[c][e] = p;
                    psum += p;
                }
            l_run = l_run * alpha + rows4_sum(psum);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) o_acc[t][e] *= alpha;
            SASA_STAMP(4 + 8 * it_)
            // O^T += V^T P^T : 2 row tiles of 16 dims, K = 64 keys; P^T is in place (B operand of step (c, e): the lane's s_acc[c][e])
#pragma unroll
            for (int blk = 0; blk < KT / 16; ++blk) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Vw[fi * LDV + 16 * blk + 4 * fk]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Vw[(16 + fi) * LDV + 16 * blk + 4 * fk]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o_acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[j], s_acc[blk][j], o_acc[0], 0, 0, 0);
                    o_acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[j], s_acc[blk][j], o_acc[1], 0, 0, 0);
                }
            }
        }
        SASA_STAMP(5 + 8 * it_)
        if (late_io) tile_io();
        SASA_STAMP(6 + 8 * it_)
        __syncthreads();                                       // the next iteration's tiles are complete; this iteration's buffer is free
        SASA_STAMP(7 + 8 * it_)
        SASA_STAMP(8 + 8 * it_)
    }
    // merge the KS key-split partials of each row group (flash-decoding combine) through LDS, then normalise.
    // Ks is free now: slot layout [ks - 1][qg][10 values][64 lanes]
    float* mg = Ks;
    if (ks > 0) {
        float* d = mg + ((ks - 1) * NQ + qg) * 10 * 64;
        d[0 * 64 + lane] = m_run;
        d[1 * 64 + lane] = l_run;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[(2 + e) * 64 + lane] = o_acc[0][e];
            d[(6 + e) * 64 + lane] = o_acc[1][e];
        }
    }
    __syncthreads();
    if (ks == 0) {
        float m = m_run, l = l_run;
        f32x4 o0 = o_acc[0], o1 = o_acc[1];
#pragma unroll
        for (int z = 1; z < KS; ++z) {
            const float* d = mg + ((z - 1) * NQ + qg) * 10 * 64;
            const float m2 = d[0 * 64 + lane], l2 = d[1 * 64 + lane];
            const float mn = fmaxf(m, m2);
            const float mu = mn == -INFINITY ? 0.f : mn;
            const float f1 = __expf(m - mu), f2 = __expf(m2 - mu);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = o0[e] * f1 + d[(2 + e) * 64 + lane] * f2;
                o1[e] = o1[e] * f1 + d[(6 + e) * 64 + lane] * f2;
            }
            l = l * f1 + l2 * f2;
            m = mn;
        }
        if (q0 + fi < a.Q) {
            const float inv = 1.f / l;
            float* o = a.out + ((long long)b * a.Q + q0 + fi) * D + h * HD + 4 * fk;     // this lane: dims 4 fk .. + 3 and 16 + 4 fk .. + 3
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] *= inv; o1[e] *= inv; }
            *reinterpret_cast<f32x4*>(o) = o0;
            *reinterpret_cast<f32x4*>(o + 16) = o1;
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
