// Row chains of the decoder layer: every op between the big mixing GEMMs and the self attention is ROW-LOCAL (a Linear, a
// LayerNorm, a ReLU, refine_bbox, the sample points and their projection: query i's output depends on query i only), so a
// workgroup that owns R = 4 queries runs the whole chain with the activations in LDS instead of one full-chip launch per op
// (sparsebev_transformer.py:166-183 are 14 such ops per layer; the per-op launches cost 5-10 us each for < 1 us of arithmetic).
//   TAIL (+ FRONT of the next layer): split-K slabs of mixing.out_proj -> + bias + residual -> norm2 -> ffn -> norm3 ->
//       cls_branch / reg_branch -> refine_bbox  [-> position_encoder -> x = feat + pos -> attention in-projection]
//   ATT: attention out-projection + residual -> norm1 -> sampling_offset / scale_weights Linear -> sample points ->
//       projection into the T*6 cameras, first-hit selection, level softmax (sample_point.hpp)
//   FRONT: position_encoder + in-projection of layer 0
//
// GEMM engine (M = 4 rows, so neither a 16- nor a 32-row MFMA tile fits): v_mfma_f32_4x4x1_16b_f32 with the A-matrix
// BROADCAST modifier (cbsz = 4, abid = k): the 16 blocks of the instruction are 16 groups of 4 output COLUMNS (lane = one
// of 64 columns), all multiplied by the same 4 rows of block `abid` of the A register -- A holds 16 k-values x 4 rows (one
// ds_read_b32 per 16 k), B is the lane's own weight column, D is 4 rows x 64 columns in 4 VGPRs.  Exact fp32 (fmaf chain)
// at the full f32 MFMA rate (8.6-9.8 cycles per instruction measured), no operand duplication.  The weights are streamed
// from L2 (every workgroup reads every weight: that stream, ~50 B/clk/CU through global_load_lds_dwordx4, is the bound), so
// they are PRE-PACKED (sbev_decoder_chain_pack) in the order the lanes consume them: a wave-load is 1 KB of contiguous memory.
// Work split: 8 waves; a unit (one or two independent Linears reading LDS rows) is cut into items of 64 columns x 128 k;
// item i goes to wave i % 8 in round i / 8 and leaves its partial sums in LDS slot i; the unit's epilogue (wave = row)
// adds the k-halves in a fixed order, + bias, ReLU / residual / LayerNorm, and writes the next unit's input rows.
// Results equal the op-by-op launches to fp32 round-off (other summation order), not bit for bit; DESIGN_HISTORY.md section 4.
#include "sbev_common.hpp"
#include "sample_point.hpp"
#include <mutex>
#include <atomic>
#include <cstdlib>

namespace {

constexpr int NWAVE = 8;
constexpr int DM = 256, FF = 512;    // embed_dims, ffn width (host-checked)
constexpr int LDX = DM + 8;          // LDS row strides: == 8 (mod 32) so the A read (4 rows x 16 k per 32 lanes) is conflict-free
constexpr int LDH = FF + 8;
static_assert(2 * LDX >= LDH, "branch rows alias the ffn hidden rows");

// The small vectors (biases, LayerNorm weights, the 3 -> 256 position-encoder weight) live behind the packed matrices in the
// chain_pack image, grouped per chain, and are copied into LDS at kernel start (one contiguous copy; every epilogue would
// otherwise pay a cold global load).  Float offsets inside the LDS parameter block:
constexpr int PV_OP_B = 0, PV_N2G = 256, PV_N2B = 512, PV_FFN0_B = 768, PV_FFN1_B = 1280, PV_N3G = 1536, PV_N3B = 1792,
              PV_CLS0_B = 2048, PV_CLS1G = 2304, PV_CLS1B = 2560, PV_CLS3_B = 2816, PV_CLS4G = 3072, PV_CLS4B = 3328,
              PV_CLS6_B = 3584, PV_REG0_B = 3648, PV_REG2_B = 3904, PV_REG4_B = 4160, PV_TAIL_END = 4224;
constexpr int PV_PE0_W = PV_TAIL_END, PV_PE0_B = PV_PE0_W + 768, PV_PE1G = PV_PE0_B + 256, PV_PE1B = PV_PE1G + 256,
              PV_PE3_B = PV_PE1B + 256, PV_PE4G = PV_PE3_B + 256, PV_PE4B = PV_PE4G + 256, PV_QKV_B = PV_PE4B + 256, PV_QKV_PAD = 832,
              PV_FRONT_END = PV_QKV_B + PV_QKV_PAD;
// attention chain (its own launch: offsets from the block start again)
constexpr int PV_AOUT_B = 0, PV_N1G = 256, PV_N1B = 512, PV_SAMP_B = 768, PV_ATTN_END = 1024;
constexpr int PV_IMAGE_FLOATS = PV_FRONT_END + PV_ATTN_END;      // in the image: [tail | front | attention]

enum Epi { EPI_FFN0, EPI_FFN1, EPI_BR1, EPI_BR2, EPI_OUT, EPI_PE3, EPI_QKV, EPI_AOUT, EPI_SAMP };
enum Pre { PRE_SLABS, PRE_FRONT, PRE_ATT };

struct Lin {
    const float* wp;     // packed weights of this Linear
    int in_off, ld;      // LDS rows it reads
    int KH;              // k-pieces per column group (RG = 1: K / 128)
    int items;           // 64-column groups x KH
    int chunks;          // 16-k chunks per item = K / (16 KHT); a multiple of 4
    int kh0, KHT;        // this Linear covers the k-pieces [kh0, kh0 + KH) of the KHT pieces of a full row (KHT == KH, kh0 == 0 but for
                         // the k-half a pair member takes of ffn.layers.1)
};
struct Unit {
    Lin a, b;            // b.items == 0: one Linear
    int epi;
    int col0;            // EPI_QKV: first output column of this unit
};

#ifdef SBEV_CHAIN_TRACE
#define SBEV_TRACE(i) if ((blockIdx.x == 7 || blockIdx.x == 15) && lane == 0) a.trace[(blockIdx.x == 15 ? 64 * 512 : 0) + wave * 64 + (i)] = (long long)__builtin_amdgcn_s_memrealtime();   // 100 MHz
#else
#define SBEV_TRACE(i)
#endif

struct ChainArgs {
#ifdef SBEV_CHAIN_TRACE
    long long* trace;
#endif
    Unit units[8];
    Unit units_b[8];             // pair mode: the second member's units (same count, same epilogues)
    int n_units, pre, front;     // front: EPI_OUT continues with the next layer's position encoder
    int n_pairs;                 // pair mode: pairs of workgroups (= row blocks)
    float* pair_x;               // pair mode: exchange rows [n_pairs][3][R][256] (ffn.1 partial sums of member 0 / 1, x rows)
    unsigned* pair_sync;         // pair mode: one arrival counter per pair; the ATTENTION chain of the same layer zeroes them:
    unsigned* zero_words; int n_zero;
    int debug_drop;              // test hook (sbev_debug_chain_pair_drop): member 1 of pair 0 leaves at once -- its partner must time out, not hang
    const float* warm;           // the launch's packed weights: one contiguous span of the chain_pack image ...
    int warm_lines;              // ... of this many 128-byte lines (L2 warm-up, see warm_l2)
    const float* vec;            // the launch's small vectors in the image -> LDS parameter block [vec_off, vec_off + vec_n)
    int vec_off, vec_n;
    long long M;                 // rows (B * Q)
    int Q;
    const float* slabs; int splits;      // PRE_SLABS: [splits, M, 256]
    const float* x1;             // [M, 256] residual of mixing.out_proj
    float* x3;                   // [M, 256] the layer's output rows
    const float* bbox;           // [M, 10] this layer's query boxes
    const float* vel_div;        // [B] or null
    float* cls_out;              // [M, num_classes]
    float* box_out;              // [M, 10]
    int num_classes;
    const float* feat;           // PRE_FRONT: [M, 256] query_feat
    float* x;                    // [M, 256]
    float* qkvt; int attn_in_rows;
    const float* att;            // PRE_ATT: [M, 256]
    float* x1_out;               // [M, 256]
    unsigned short* x1_frag;     // fp16 GEMM modes: x1 also as the generator's operand -- scaled fp16 hi | lo images in MFMA fragment order
    const float* x1_scale;       //   (gemm_bf16s.hip: [M/32][16 k-steps][2][64 lanes][8]), scaled by x1_scale[0] = 2^e; null otherwise
    float* so; int soN;          // so: null when the sample points are projected in here (proj.loc_bp set)
    sbev_ops::SamplePointArgs proj;
    float eps;
};

// ---- the weight stream --------------------------------------------------------------------------------------------------
// A wave's items (in program order) form one stream of 4 KB chunks (64 columns x 16 k, 4 wave-loads of 1 KB).  The chunks
// go global -> LDS directly (global_load_lds_dwordx4) into a private 3-slot ring per wave; a chunk is read into registers
// and its slot refilled at once, so THREE chunks (12 KB per wave, 96 KB per CU) are in flight while one multiplies, also
// across item / unit boundaries (weights do not depend on activations).  The loads are issued
// and waited for by hand (inline asm, counted s_waitcnt vmcnt): hipcc waits vmcnt(0) before every LDS read that follows a
// compiler-visible LDS-DMA load, which would serialise the stream.  Vector-memory loads return in order, so "at most 4
// outstanding" means everything older than the newest chunk has landed; compiler-issued loads / stores in between only
// make these waits (and the compiler's own) more conservative.
constexpr int RING_SLOTS = 3;
constexpr int CHUNK_FLOATS = 1024;

// Rows per workgroup: R = 4 RG, RG = 1, 2 or 4 row groups of the 4-row MFMA (template parameter of the kernels).
//   RG = 1 (<= 1024 rows, round 2): the LDS-DMA ring above, items of 64 columns x 128 k, 16 partial-sum slots.
//   RG = 2 / 4 (round 3: up to 2048 / 4096 rows in ONE round of workgroups -- the batch configs, 3200 and 3600 rows, and the
//     1600-query config): every weight chunk multiplies RG row groups (RG x the MFMAs per byte streamed: at RG = 4 the launch
//     is as much MFMA- as stream-bound), so the LDS goes to the rows, the chunks travel global -> VGPR (a ring of 4 chunks =
//     64 registers, waits counted by the compiler), and a wave keeps a column group's k-pieces in its accumulators: items are
//     64 columns x (K / KH) k with KH chosen by pieces() so that a unit has ~8 items (<= 12 partial-sum slots).
// LDS map (float offsets):
template <int RG>
struct Lay {
    static constexpr int R = 4 * RG;
    static constexpr bool WIDE = RG > 1;
    static constexpr int MAX_SLOTS = WIDE ? 12 : 16;
    static constexpr int OFF_X2 = 0;                    // x2 (norm2 output) / x (attention input rows) / att rows
    static constexpr int OFF_X3 = OFF_X2 + R * LDX;     // x3 = the layer's output rows (next layer's query_feat) / x1
    static constexpr int OFF_H = OFF_X3 + R * LDX;      // ffn hidden rows; dead after ffn.layers.1, then:
    static constexpr int OFF_C = OFF_H;                 //   classification-branch rows / position-encoder rows
    static constexpr int OFF_R = OFF_H + R * LDX;       //   regression-branch rows
    static constexpr int OFF_P = OFF_H + 2 * R * LDX;   // partial sums [MAX_SLOTS][R][64]     (2 R LDX >= R LDH)
    static constexpr int LDS_FLOATS = OFF_P + MAX_SLOTS * R * 64;
    static constexpr int OFF_RING = LDS_FLOATS;                                                      // [NWAVE][RING_SLOTS][CHUNK_FLOATS]
    static constexpr int OFF_DUMP = OFF_RING + (WIDE ? 0 : NWAVE * RING_SLOTS * CHUNK_FLOATS);     // [NWAVE][64] sink of the warm-up loads
    static constexpr int OFF_PARAM = OFF_DUMP + NWAVE * 64;                                          // the launch's small vectors (PV_* offsets)
    static constexpr int LDS_TOTAL_FLOATS = OFF_PARAM + PV_FRONT_END;
    static constexpr int RPW = (R + NWAVE - 1) / NWAVE;                                              // input rows per wave in the prologue
    static_assert(LDS_TOTAL_FLOATS * 4 <= 160 * 1024, "LDS budget of one CU");
};
struct Offs { int x2, x3, h, c, r; };        // the same offsets for the host-side unit tables
template <int RG> constexpr Offs offs_of() { return Offs{Lay<RG>::OFF_X2, Lay<RG>::OFF_X3, Lay<RG>::OFF_H, Lay<RG>::OFF_C, Lay<RG>::OFF_R}; }

// L2 warm-up.  The L2 is cold for the weights at every launch (a whole layer of GEMM / gather traffic went through it), and
// the workgroups of an XCD stream the same lines in near lock-step, so each line's HBM / Infinity-Cache miss latency would be
// seen by all of them, every round.  Instead the workgroups of an XCD (workgroup i runs on XCD i % 8) split the launch's
// weight span between them and touch one dword per 128-byte line of their slice up front, while the prologue runs; the
// loads go to an LDS sink (no register to keep alive) and are older than every chunk load, so the counted waits stay valid.
template <int RG>
__device__ __forceinline__ void warm_l2(const ChainArgs& a, int wave, int lane) {
    const int per_xcd = ((int)gridDim.x + 7) / 8;
    const int parts = per_xcd < 32 ? per_xcd : 32;
    const int part = ((int)blockIdx.x / 8) % parts;
    const int n = (a.warm_lines + parts - 1) / parts;
    const int l0 = part * n;
    const int l1 = min(a.warm_lines, l0 + n);
    const unsigned dump = (unsigned)(Lay<RG>::OFF_DUMP + wave * 64) * 4u;
    for (int l = l0 + wave * 64; l < l1; l += NWAVE * 64) {
        const int line = min(l + lane, l1 - 1);
        const unsigned voff = (unsigned)line * 128u;
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dword %1, %2\n\t"
            :
            : "s"(dump), "v"(voff), "s"(a.warm)
            : "memory");
    }
}

struct Walk {            // position of a wave in its item stream
    int u, it, step;
};

__device__ __forceinline__ int unit_items(const Unit* U, int u) { return U[u].a.items + U[u].b.items; }

// first item of this wave at or after (u, it)
__device__ __forceinline__ void walk_settle(const ChainArgs& a, const Unit* U, Walk& w, int wave) {
    while (w.u < a.n_units && w.it >= unit_items(U, w.u)) {
        ++w.u;
        w.it = wave;
    }
}

__device__ __forceinline__ const float* item_weights(const Unit& un, int it) {
    const bool second = it >= un.a.items;
    const Lin& l = second ? un.b : un.a;
    const int j = it - (second ? un.a.items : 0);
    const int piece = l.KHT == l.KH ? j : (j / l.KH) * l.KHT + l.kh0 + j % l.KH;
    return l.wp + (long long)piece * (l.chunks * CHUNK_FLOATS);
}

// (M0 carries the LDS destination of an LDS-DMA load and is written inside the asm block; hipcc does not use M0 anywhere else
// in these kernels -- LDS instructions need no M0 on gfx9+, there is no dynamic register indexing -- checked in the ISA.)
__device__ __forceinline__ void issue_chunk(const float* g, unsigned lds_byte, unsigned voff) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        :
        : "s"(lds_byte), "v"(voff), "s"(g)
        : "memory");
}

struct Stream {
    const Unit* U;       // the workgroup's unit table (kernel arguments)
    Walk iw;             // item of the next chunk to issue
    const float* g;      // its weights (next chunk)
    int nchunk;          // chunks of that item
    int issued, used;    // chunk counters
    int islot, uslot;    // issued % RING_SLOTS, used % RING_SLOTS
    unsigned ring_byte;  // LDS byte address of this wave's ring
    unsigned voff;       // lane * 16
};

__device__ __forceinline__ void stream_seek(const ChainArgs& a, Stream& st, int wave) {
    walk_settle(a, st.U, st.iw, wave);
    st.iw.step = 0;
    if (st.iw.u < a.n_units) {
        const Unit& un = st.U[st.iw.u];
        st.g = item_weights(un, st.iw.it);
        st.nchunk = st.iw.it >= un.a.items ? un.b.chunks : un.a.chunks;
    }
}

__device__ __forceinline__ void stream_issue(const ChainArgs& a, Stream& st, int wave) {
    if (st.iw.u >= a.n_units) return;
    issue_chunk(st.g + st.iw.step * CHUNK_FLOATS, st.ring_byte + (unsigned)st.islot * (CHUNK_FLOATS * 4), st.voff);
    ++st.issued;
    st.islot = st.islot == RING_SLOTS - 1 ? 0 : st.islot + 1;
    if (++st.iw.step == st.nchunk) {
        st.iw.it += NWAVE;
        stream_seek(a, st, wave);
    }
}

// RG > 1: the next chunk of the stream into 16 registers (four 1 KB wave-loads; the compiler counts the waits)
struct RChunk { float4 q[4]; };
// The loads are UNCONDITIONAL (past the end of the stream a wave re-reads the first chunk of its last item, <= 16 KB per wave):
// a load under a branch would make the number of younger loads unknown to the compiler, which then waits vmcnt(0) before every
// chunk instead of vmcnt(12).
__device__ __forceinline__ void stream_load(const ChainArgs& a, Stream& st, int wave, int lane, RChunk& c) {
    const float* g = st.g + st.iw.step * CHUNK_FLOATS + lane * 4;
#pragma unroll
    for (int m = 0; m < 4; ++m) c.q[m] = *reinterpret_cast<const float4*>(g + 256 * m);
    if (st.iw.u < a.n_units && ++st.iw.step == st.nchunk) {
        st.iw.it += NWAVE;
        stream_seek(a, st, wave);
    }
}
constexpr int RDEPTH = 4;        // chunks in registers per wave (one multiplying, three on their way)

#define SBEV_MF(KK, BV) acc[(KK) & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, BV, acc[(KK) & 1], 4, KK, 0);
#define SBEV_MSTEP(m, q) SBEV_MF(4 * (m) + 0, q.x) SBEV_MF(4 * (m) + 1, q.y) SBEV_MF(4 * (m) + 2, q.z) SBEV_MF(4 * (m) + 3, q.w)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one item (64 columns x 128 k) of Linear `l`: 8 chunks from the ring; partial sums -> P[slot]
__device__ __forceinline__ void mma_item(const ChainArgs& a, Stream& st, const Lin& l, int j, float* smem, int pslot, int lane, int wave) {
    constexpr int R = 4, OFF_RING = Lay<1>::OFF_RING, OFF_P = Lay<1>::OFF_P;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int kh = j % l.KH;
    const float* arow = smem + l.in_off + kh * 128 + (lane & 3) * l.ld + (lane >> 2);
    const float* ring = smem + OFF_RING + wave * (RING_SLOTS * CHUNK_FLOATS) + lane * 4;
    for (int s = 0; s < 8; ++s) {
        const int ahead = st.issued - st.used;             // chunks in flight incl. this one
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float av = arow[s * 16];
        const float* slot = ring + st.uslot * CHUNK_FLOATS;
        const float4 q0 = *reinterpret_cast<const float4*>(slot);
        const float4 q1 = *reinterpret_cast<const float4*>(slot + 256);
        const float4 q2 = *reinterpret_cast<const float4*>(slot + 512);
        const float4 q3 = *reinterpret_cast<const float4*>(slot + 768);
        // the chunk is in registers: its slot is refilled right away (three chunks in flight while this one multiplies)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++st.used;
        st.uslot = st.uslot == RING_SLOTS - 1 ? 0 : st.uslot + 1;
        stream_issue(a, st, wave);
        __builtin_amdgcn_sched_barrier(0);
        SBEV_MSTEP(0, q0) SBEV_MSTEP(1, q1) SBEV_MSTEP(2, q2) SBEV_MSTEP(3, q3)
    }
    const f32x4 t = acc[0] + acc[1];
    float* p = smem + OFF_P + pslot * (R * 64) + lane;
    p[0] = t.x; p[64] = t.y; p[128] = t.z; p[192] = t.w;
}

// RG > 1: one item (64 columns x 16 l.chunks k) for the workgroup's RG row groups; the chunk registers rb[d] are consumed and
// refilled in ring order (l.chunks is a multiple of RDEPTH, so slot d of the ring is a compile-time register set)
template <int RG>
__device__ __forceinline__ void mma_item_wide(const ChainArgs& a, Stream& st, RChunk (&rb)[RDEPTH], const Lin& l, int j, float* smem, int pslot,
                                              int lane, int wave) {
    constexpr int NACC = RG >= 4 ? 1 : 2;          // >= 4 independent accumulator chains back to back
    f32x4 acc[RG][NACC];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[rg][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kh = l.kh0 + j % l.KH;
    const float* arow = smem + l.in_off + kh * (l.chunks * 16) + (lane & 3) * l.ld + (lane >> 2);
    float avn[RG];                                 // the A registers (16 k x 4 rows per row group) of the NEXT chunk: read one chunk ahead
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) avn[rg] = arow[rg * 4 * l.ld];
    for (int s0 = 0; s0 < l.chunks; s0 += RDEPTH) {
#pragma unroll
        for (int d = 0; d < RDEPTH; ++d) {
            float av[RG];
            const int nxt = min(s0 + d + 1, l.chunks - 1);
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) { av[rg] = avn[rg]; avn[rg] = arow[nxt * 16 + rg * 4 * l.ld]; }
            __builtin_amdgcn_sched_barrier(0);
#define SBEV_MW(KK, BV)                                                                                                     \
    _Pragma("unroll") for (int rg = 0; rg < RG; ++rg)                                                                        \
        acc[rg][(KK) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[rg], BV, acc[rg][(KK) % NACC], 4, KK, 0);
#define SBEV_MWSTEP(m, q) SBEV_MW(4 * (m) + 0, q.x) SBEV_MW(4 * (m) + 1, q.y) SBEV_MW(4 * (m) + 2, q.z) SBEV_MW(4 * (m) + 3, q.w)
            SBEV_MWSTEP(0, rb[d].q[0]) SBEV_MWSTEP(1, rb[d].q[1]) SBEV_MWSTEP(2, rb[d].q[2]) SBEV_MWSTEP(3, rb[d].q[3])
#undef SBEV_MWSTEP
#undef SBEV_MW
            __builtin_amdgcn_sched_barrier(0);
            stream_load(a, st, wave, lane, rb[d]);       // refill: three chunks stay in flight while the next one multiplies
        }
    }
    float* p = smem + Lay<RG>::OFF_P + pslot * (Lay<RG>::R * 64) + lane;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        f32x4 t = acc[rg][0];
        if (NACC == 2) t += acc[rg][NACC - 1];
        p[(rg * 4 + 0) * 64] = t.x; p[(rg * 4 + 1) * 64] = t.y; p[(rg * 4 + 2) * 64] = t.z; p[(rg * 4 + 3) * 64] = t.w;
    }
}

__device__ __forceinline__ float wsum(float v) { return sbev::wave_sum_dpp(v); }

// LayerNorm over 256 columns held as 4 per lane (column = lane + 64 c).  The weights are fetched (LDS) by LnW's constructor,
// i.e. where the caller declares it -- before the partial sums are gathered, so that their latency is not at the end of the chain.
struct LnW {
    float g[4], b[4];
    __device__ __forceinline__ LnW(const float* gp, const float* bp, int lane) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { g[c] = gp[lane + 64 * c]; b[c] = bp[lane + 64 * c]; }
    }
};
__device__ __forceinline__ void ln4(float (&v)[4], const LnW& w, float eps, bool relu) {
    const float mean = wsum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / DM);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float d = v[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wsum(q) * (1.f / DM) + eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float o = (v[c] - mean) * rstd * w.g[c] + w.b[c];
        v[c] = relu ? fmaxf(o, 0.f) : o;
    }
}

// partial sums of 4 column groups (cg0 .. cg0 + 3) of a Linear whose items start at slot `base`: k-halves added in order, + bias.
// KH is 2 (K = 256) or 4 (K = 512): specialised so that all LDS reads are in flight before the first add.
template <int KH, int R>
__device__ __forceinline__ void gather4_t(float (&v)[4], const float* P, int base, int cg0, int row, int lane, const float* bias, int bias0) {
    float t[4][KH];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) t[c][kh] = P[(base + (cg0 + c) * KH + kh) * (R * 64) + row * 64 + lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = 0.f;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) s += t[c][kh];
        v[c] = s + bias[bias0 + lane + 64 * c];
    }
}
template <int R>
__device__ __forceinline__ void gather4(float (&v)[4], const float* P, int base, int KH, int cg0, int row, int lane, const float* bias,
                                        int bias0) {
    if (KH == 1) gather4_t<1, R>(v, P, base, cg0, row, lane, bias, bias0);
    else if (KH == 2) gather4_t<2, R>(v, P, base, cg0, row, lane, bias, bias0);
    else gather4_t<4, R>(v, P, base, cg0, row, lane, bias, bias0);
}

// the k-pieces of the 4 column groups of a Linear added in order, no bias (pair mode: a member's partial sums)
template <int R>
__device__ __forceinline__ void gather4_raw(float (&v)[4], const float* P, int KH, int row, int lane) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        for (int kh = 0; kh < KH; ++kh) s += P[(c * KH + kh) * (R * 64) + row * 64 + lane];
        v[c] = s;
    }
}

// ---- pair mode: two workgroups share a row block (PAIR instantiations of the tail) --------------------------------------------------
// Every workgroup of a chain streams every weight, and a CU takes ~60 GB/s from the L2 whatever the row count (DESIGN_HISTORY.md
// section 10.3): the only way to stream less per CU is to split a row block's Linears between CUs.  A PAIR of workgroups owns R rows;
// both reduce the out-projection slabs and run norm2 (bit-identical x2 rows), then
//   ffn.layers.0    member s computes the hidden columns [256 s, 256 s + 256)            (its half of the weight's ROWS)
//   ffn.layers.1    member s sums over k in [256 s, 256 s + 256) -- the hidden columns it has -- for all 256 outputs; the partial
//                   sums are EXCHANGED, added member 0 + member 1 in both, + bias + residual -> norm3: x3 in both
//   branches        member 0 runs the classification branch, member 1 the regression branch, refine_bbox and (front) the position
//                   encoder; its x rows (feat + pos) are HANDED to member 0
//   in-projection   member 0 computes column groups 0 .. 7, member 1 the rest
// so a member streams ~1.65 of the 3.25 MB and the pair meets twice.  Hand-off (MI355X guide, inter-workgroup visibility): payload
// with 16-byte sc0 sc1 (write-through) stores, EVERY wave drains them, workgroup barrier, one lane adds to the pair's counter with a
// relaxed agent-scope atomic; the reader polls that word from one lane (s_sleep between polls; BOUNDED: a lost partner raises
// g_chain_pair_timeouts instead of hanging the GPU), workgroup barrier, sc0 sc1 loads.  Valid for any workgroup -> XCD placement; the
// block -> pair map puts both members on one XCD under round-robin dispatch (speed only).  The counters are zeroed by the attention
// chain of the same layer, which always runs between two tails on the stream (no memset node).
__device__ unsigned g_chain_pair_timeouts;
// ... and a STICKY fault word in pinned, device-mapped HOST memory (pair_fault_host() below installs the pointer once per process): a
// timed-out hand-off also bumps it with a system-scope atomic, so the host sees the fault WITHOUT synchronising the device -- at the next
// sbev_decoder_forward / graph replay it is an error (SBEV_EFAULT), not eight silently wrong rows.  Read only on the timeout
// path: no kernel argument, no register in the hot loop.
__device__ unsigned* g_chain_pair_fault_host;
constexpr unsigned PAIR_POLL_LIMIT = 1u << 20;       // x (s_sleep 2 + one memory round trip, ~1 us): about a second

__device__ __forceinline__ void pair_store4(float* p, const float (&v)[4]) {
    const f32x4 q = {v[0], v[1], v[2], v[3]};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void pair_load4(const float* p, float (&v)[4]) {
    f32x4 q;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void pair_arrive(unsigned* word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its payload stores have left
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pair_wait(unsigned* word, unsigned arrivals) {
    if (threadIdx.x == 0) {
        unsigned n = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrivals) {
            __builtin_amdgcn_s_sleep(2);
            if (++n > PAIR_POLL_LIMIT) {
                __hip_atomic_fetch_add(&g_chain_pair_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned* const host_word = g_chain_pair_fault_host;
                if (host_word) __hip_atomic_fetch_add(host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void store_rows(float* smem, int off, int ld, int row, int lane, const float (&v)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) smem[off + row * ld + lane + 64 * c] = v[c];
}

// position_encoder[0..2]: Linear(3 -> 256) + LayerNorm + ReLU of one row (sparsebev_transformer.py:116-119), into LDS
template <int RG>
__device__ __forceinline__ void pe0_row(const ChainArgs& a, float* smem, int row, int lane, float x0, float x1, float x2) {
    constexpr int OFF_PARAM = Lay<RG>::OFF_PARAM, OFF_C = Lay<RG>::OFF_C;
    const float* pv = smem + OFF_PARAM;
    const LnW lw(pv + PV_PE1G, pv + PV_PE1B, lane);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = lane + 64 * c;
        const float* w = pv + PV_PE0_W + n * 3;
        v[c] = ((x0 * w[0] + x1 * w[1]) + x2 * w[2]) + pv[PV_PE0_B + n];
    }
    ln4(v, lw, a.eps, true);
    store_rows(smem, OFF_C, LDX, row, lane, v);
}

// PRE: how the input rows are produced (compile-time: the three chains are three instantiations)
template <int PRE, int RG, bool PAIR = false>
__global__ __launch_bounds__(64 * NWAVE) void row_chain_kernel(const ChainArgs a) {
    using L = Lay<RG>;
    static_assert(!PAIR || (PRE == PRE_SLABS && RG > 1), "pair mode: the tail, wide row blocks");
    constexpr int R = L::R, RPW = L::RPW, OFF_X2 = L::OFF_X2, OFF_X3 = L::OFF_X3, OFF_H = L::OFF_H, OFF_C = L::OFF_C, OFF_R = L::OFF_R,
                  OFF_P = L::OFF_P, OFF_PARAM = L::OFF_PARAM;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // L::LDS_TOTAL_FLOATS (dynamic: > 64 KB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // pair mode: block b -> XCD b % 8 under round-robin dispatch, so the members of pair p are blocks b and b + 8
    int pair = (int)blockIdx.x, member = 0;
    if constexpr (PAIR) {
        const int li = (int)blockIdx.x >> 3;
        pair = (li >> 1) * 8 + ((int)blockIdx.x & 7);
        member = li & 1;
        if (pair >= a.n_pairs) return;
        if (a.debug_drop && pair == 0 && member == 1) return;
    }
    const Unit* const U = (PAIR && member) ? a.units_b : a.units;
    unsigned* const pword = PAIR ? a.pair_sync + pair : nullptr;
    float* const px = PAIR ? a.pair_x + (long long)pair * (3 * R * DM) : nullptr;       // [3][R][256]
    const long long row0 = (long long)pair * R;
    float* P = smem + OFF_P;
    if (PRE == PRE_ATT && a.zero_words && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.n_zero; i += 64 * NWAVE) a.zero_words[i] = 0u;

    // the weight stream starts before anything else: three chunks in flight
    SBEV_TRACE(0)
    warm_l2<RG>(a, wave, lane);
    Stream st;
    st.U = U;
    st.iw = Walk{0, wave, 0};
    st.issued = st.used = st.islot = st.uslot = 0;
    st.ring_byte = (unsigned)(L::OFF_RING + wave * (RING_SLOTS * CHUNK_FLOATS)) * 4u;
    st.voff = (unsigned)lane * 16u;
    st.g = a.warm;           // (valid for stream_load's unconditional loads even if this wave has no item)
    st.nchunk = 8;
    stream_seek(a, st, wave);
    RChunk rb[RDEPTH];
    if constexpr (L::WIDE) {
#pragma unroll
        for (int d = 0; d < RDEPTH; ++d) {
#pragma unroll
            for (int m = 0; m < 4; ++m) rb[d].q[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            stream_load(a, st, wave, lane, rb[d]);
        }
    } else {
        stream_issue(a, st, wave);
        stream_issue(a, st, wave);
        stream_issue(a, st, wave);
    }

    // ---- prologue: every global load of it is issued before the first wait (one memory latency, not one per phase) -----
    const float* pv = smem + OFF_PARAM;
    constexpr int PASSES = (PV_FRONT_END / 4 + 64 * NWAVE - 1) / (64 * NWAVE);
    float4 pq[PASSES];                                   // the launch's small vectors, on their way to LDS
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const int i4 = (int)threadIdx.x + j * 64 * NWAVE;
        pq[j] = *reinterpret_cast<const float4*>(a.vec + 4 * min(i4, a.vec_n / 4 - 1));
    }
    // input rows: wave w owns rows w, w + 8 (RG = 4) of the workgroup
    float4 t[RPW], r4[RPW];
    float v0[RPW][4], v1[RPW][4];
    float bx[RPW], by[RPW], bz[RPW];
    bool plive[RPW];
#pragma unroll
    for (int pi = 0; pi < RPW; ++pi) {
        const int prow = wave + pi * NWAVE;
        const long long pg = row0 + prow;
        plive[pi] = prow < R && pg < a.M;
        t[pi] = make_float4(0.f, 0.f, 0.f, 0.f); r4[pi] = t[pi];
#pragma unroll
        for (int c = 0; c < 4; ++c) v0[pi][c] = v1[pi][c] = 0.f;
        bx[pi] = by[pi] = bz[pi] = 0.f;
        if (PRE == PRE_SLABS) {
            if (plive[pi]) r4[pi] = *reinterpret_cast<const float4*>(a.x1 + pg * DM + lane * 4);
        } else if (plive[pi]) {
            const float* src = PRE == PRE_FRONT ? a.feat : a.att;
#pragma unroll
            for (int c = 0; c < 4; ++c) v0[pi][c] = src[pg * DM + lane + 64 * c];
            if (PRE == PRE_FRONT) { bx[pi] = a.bbox[pg * 10]; by[pi] = a.bbox[pg * 10 + 1]; bz[pi] = a.bbox[pg * 10 + 2]; }
            if (PRE == PRE_ATT) {        // the residual of the out-projection: x rows, kept in LDS until the first epilogue
#pragma unroll
                for (int c = 0; c < 4; ++c) v1[pi][c] = a.x[pg * DM + lane + 64 * c];
            }
        }
    }
    if (PRE == PRE_SLABS) {
        // mixing.out_proj: sum of the split-K slabs -- up to 32 independent loads in flight per wave (all of its rows' together), added in
        // slab order; lane = 4 columns.  A dead row reads the last row's slabs (no branch: hipcc sinks loads into branches) x 0.
        constexpr int NB = 32 / RPW;             // (32 slabs = the fp16 out-projection's plan at 900 rows: one batch, one latency)
        for (int z0 = 0; z0 < a.splits; z0 += NB) {
            float4 q[RPW][NB];
#pragma unroll
            for (int pi = 0; pi < RPW; ++pi) {
                const long long pg = min(row0 + wave + pi * NWAVE, a.M - 1);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    q[pi][j] = *reinterpret_cast<const float4*>(a.slabs + ((long long)min(z0 + j, a.splits - 1) * a.M + pg) * DM + lane * 4);
            }
#pragma unroll
            for (int pi = 0; pi < RPW; ++pi)
#pragma unroll
                for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(q[pi][j].x), "v"(q[pi][j].y), "v"(q[pi][j].z), "v"(q[pi][j].w));   // all issued first
#pragma unroll
            for (int pi = 0; pi < RPW; ++pi)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float m = (z0 + j < a.splits && plive[pi]) ? 1.f : 0.f;
                    t[pi].x += q[pi][j].x * m; t[pi].y += q[pi][j].y * m; t[pi].z += q[pi][j].z * m; t[pi].w += q[pi][j].w * m;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < PASSES; ++j) asm volatile("" ::"v"(pq[j].x), "v"(pq[j].y), "v"(pq[j].z), "v"(pq[j].w));   // keeps the loads up there
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const int i4 = (int)threadIdx.x + j * 64 * NWAVE;
        if (i4 < a.vec_n / 4) *reinterpret_cast<float4*>(smem + OFF_PARAM + a.vec_off + 4 * i4) = pq[j];
    }
    __syncthreads();
#pragma unroll
    for (int pi = 0; pi < RPW; ++pi) {
        const int prow = wave + pi * NWAVE;
        if (prow >= R) continue;
        if (PRE == PRE_SLABS) {
            // + bias + residual x1 -> norm2 (sparsebev_transformer.py:171)
            float4 tt = t[pi];
            const float4 rr = r4[pi];
            const float4 b4 = *reinterpret_cast<const float4*>(pv + PV_OP_B + lane * 4);
            if (plive[pi]) { tt.x = (tt.x + b4.x) + rr.x; tt.y = (tt.y + b4.y) + rr.y; tt.z = (tt.z + b4.z) + rr.z; tt.w = (tt.w + b4.w) + rr.w; }
            const float mean = wsum((tt.x + tt.y) + (tt.z + tt.w)) * (1.f / DM);
            const float dx = tt.x - mean, dy = tt.y - mean, dz = tt.z - mean, dw = tt.w - mean;
            const float rstd = rsqrtf(wsum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / DM) + a.eps);
            const float4 g4 = *reinterpret_cast<const float4*>(pv + PV_N2G + lane * 4);
            const float4 n4 = *reinterpret_cast<const float4*>(pv + PV_N2B + lane * 4);
            *reinterpret_cast<float4*>(smem + OFF_X2 + prow * LDX + lane * 4) =
                make_float4(dx * rstd * g4.x + n4.x, dy * rstd * g4.y + n4.y, dz * rstd * g4.z + n4.z, dw * rstd * g4.w + n4.w);
        } else {
            store_rows(smem, PRE == PRE_FRONT ? OFF_X3 : OFF_X2, LDX, prow, lane, v0[pi]);
            if (PRE == PRE_ATT) store_rows(smem, OFF_R, LDX, prow, lane, v1[pi]);
            if (PRE == PRE_FRONT) pe0_row<RG>(a, smem, prow, lane, bx[pi], by[pi], bz[pi]);
        }
    }
    __syncthreads();

    SBEV_TRACE(1)
    for (int u = 0; u < a.n_units; ++u) {
        const Unit& un = U[u];
        const int items = un.a.items + un.b.items;
        for (int it = wave; it < items; it += NWAVE) {
            const bool second = it >= un.a.items;
            if constexpr (L::WIDE) mma_item_wide<RG>(a, st, rb, second ? un.b : un.a, it - (second ? un.a.items : 0), smem, it, lane, wave);
            else mma_item(a, st, second ? un.b : un.a, it - (second ? un.a.items : 0), smem, it, lane, wave);
        }
        SBEV_TRACE(2 + 4 * u)
        __syncthreads();
        SBEV_TRACE(3 + 4 * u)
        if constexpr (PAIR) {
            if (un.epi == EPI_FFN1) {
                // this member's partial sums of ffn.layers.1 (its k-half) -> exchange rows; lane = columns lane + 64 c, stored as one float4
                for (int row = wave; row < R; row += NWAVE) {
                    float v[4];
                    gather4_raw<R>(v, P, un.a.KH, row, lane);
                    pair_store4(px + (member * R + row) * DM + lane * 4, v);
                }
                pair_arrive(pword);
                pair_wait(pword, 2u);
            } else if (un.epi == EPI_PE3 && member == 0) {
                pair_wait(pword, 3u);      // member 1's x rows
            }
        }
        // ---- epilogue: task = (row, side); wave-local row ops ----------------------------------------------------------
        for (int task = wave; task < 2 * R; task += NWAVE) {
            const int row = task % R, side = task / R;
            const long long g = row0 + row;
            const bool live = g < a.M;
            float v[4];
            if (PAIR && side != member && (un.epi == EPI_FFN0 || un.epi == EPI_BR1 || un.epi == EPI_BR2 || un.epi == EPI_OUT)) continue;
            switch (un.epi) {
            case EPI_FFN0: {     // relu(x2 W0^T + b0): side = column half
                gather4<R>(v, P, 0, un.a.KH, PAIR ? 0 : side * 4, row, lane, pv + PV_FFN0_B, side * 256);
#pragma unroll
                for (int c = 0; c < 4; ++c) smem[OFF_H + row * LDH + side * 256 + lane + 64 * c] = fmaxf(v[c], 0.f);
            } break;
            case EPI_FFN1: {     // + x2 -> norm3 -> x3
                if (side) break;
                const LnW lw(pv + PV_N3G, pv + PV_N3B, lane);
                float res[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) res[c] = smem[OFF_X2 + row * LDX + lane + 64 * c];
                if constexpr (PAIR) {       // member 0's + member 1's partial sums, in that order in both members
                    float own[4], oth[4];
                    gather4_raw<R>(own, P, un.a.KH, row, lane);
                    pair_load4(px + ((1 - member) * R + row) * DM + lane * 4, oth);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = ((member ? oth[c] : own[c]) + (member ? own[c] : oth[c])) + pv[PV_FFN1_B + lane + 64 * c];
                } else {
                    gather4<R>(v, P, 0, un.a.KH, 0, row, lane, pv + PV_FFN1_B, 0);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += res[c];
                ln4(v, lw, a.eps, false);
                store_rows(smem, OFF_X3, LDX, row, lane, v);
                if (live && member == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a.x3[g * DM + lane + 64 * c] = v[c];
                }
            } break;
            case EPI_BR1:        // cls_branch[0..2] | reg_branch[0..1]
            case EPI_BR2: {      // cls_branch[3..5] | reg_branch[2..3]
                const bool first = un.epi == EPI_BR1;
                if (side == 0) {
                    const LnW lw(pv + (first ? PV_CLS1G : PV_CLS4G), pv + (first ? PV_CLS1B : PV_CLS4B), lane);
                    gather4<R>(v, P, 0, un.a.KH, 0, row, lane, pv + (first ? PV_CLS0_B : PV_CLS3_B), 0);
                    ln4(v, lw, a.eps, true);
                    store_rows(smem, OFF_C, LDX, row, lane, v);
                } else {
                    gather4<R>(v, P, un.a.items, un.b.KH, 0, row, lane, pv + (first ? PV_REG0_B : PV_REG2_B), 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                    store_rows(smem, OFF_R, LDX, row, lane, v);
                }
            } break;
            case EPI_OUT: {      // cls_branch[6] -> scores | reg_branch[4] -> refine_bbox (:174-183) [-> next position encoder]
                if (side == 0) {
                    if (lane < a.num_classes) {
                        float s = 0.f;
                        for (int kh = 0; kh < un.a.KH; ++kh) s += P[kh * (R * 64) + row * 64 + lane];
                        if (live) a.cls_out[g * a.num_classes + lane] = s + pv[PV_CLS6_B + lane];
                    }
                } else {
                    float o = 0.f;
                    if (lane < 10) {
                        float s = 0.f;
                        for (int kh = 0; kh < un.b.KH; ++kh) s += P[(un.a.items + kh) * (R * 64) + row * 64 + lane];
                        const float r = s + pv[PV_REG4_B + lane];
                        const long long gi = live ? g : 0;
                        if (lane < 3) {
                            float p = a.bbox[gi * 10 + lane];
                            p = fminf(fmaxf(p, 0.f), 1.f);
                            const float logit = logf(fmaxf(p, 1e-5f) / fmaxf(1.f - p, 1e-5f));
                            o = 1.f / (1.f + expf(-(r + logit)));
                        } else {
                            o = r;
                            if (lane >= 8 && a.vel_div) o = r / a.vel_div[(unsigned)gi / (unsigned)a.Q];
                        }
                        if (live) a.box_out[g * 10 + lane] = o;
                    }
                    if (a.front) {
                        const int ob = (int)__float_as_uint(o);
                        const float x0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(ob, 0));
                        const float x1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(ob, 1));
                        const float x2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(ob, 2));
                        pe0_row<RG>(a, smem, row, lane, x0, x1, x2);
                    }
                }
            } break;
            case EPI_PE3: {      // position_encoder[3..5] + query_feat -> x (:166-167)
                if (side) break;
                if (PAIR && member == 0) {       // the x rows come from member 1
                    pair_load4(px + (2 * R + row) * DM + lane * 4, v);
                    store_rows(smem, OFF_X2, LDX, row, lane, v);
                    break;
                }
                const LnW lw(pv + PV_PE4G, pv + PV_PE4B, lane);
                float res[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) res[c] = smem[OFF_X3 + row * LDX + lane + 64 * c];
                gather4<R>(v, P, 0, un.a.KH, 0, row, lane, pv + PV_PE3_B, 0);
                ln4(v, lw, a.eps, true);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += res[c];
                store_rows(smem, OFF_X2, LDX, row, lane, v);
                if (PAIR) pair_store4(px + (2 * R + row) * DM + lane * 4, v);
                if (live) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a.x[g * DM + lane + 64 * c] = v[c];
                }
            } break;
            case EPI_QKV: {      // attention in-projection + gen_tau rows -> qkvt
                const int ncg = un.a.items / un.a.KH;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cg = side * 4 + c;
                    const int n = un.col0 + cg * 64 + lane;
                    if (cg < ncg && n < a.attn_in_rows) {
                        float s = 0.f;
                        for (int kh = 0; kh < un.a.KH; ++kh) s += P[(cg * un.a.KH + kh) * (R * 64) + row * 64 + lane];
                        if (live) a.qkvt[g * a.attn_in_rows + n] = s + pv[PV_QKV_B + n];
                    }
                }
            } break;
            case EPI_AOUT: {     // attention out-projection + x -> norm1 -> x1 (:169)
                if (side) break;
                const LnW lw(pv + PV_N1G, pv + PV_N1B, lane);
                float res[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) res[c] = smem[OFF_R + row * LDX + lane + 64 * c];     // x rows, staged by the prologue
                gather4<R>(v, P, 0, un.a.KH, 0, row, lane, pv + PV_AOUT_B, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += res[c];
                ln4(v, lw, a.eps, false);
                store_rows(smem, OFF_X3, LDX, row, lane, v);
                if (live) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a.x1_out[g * DM + lane + 64 * c] = v[c];
                }
                if (a.x1_frag && live) {
                    // the generator's operand, saving its pack launch (5 us per layer): lane = (image, 8-k piece) reads the piece's 8
                    // values of the row just written to LDS (same wave: the LDS pipe keeps its accesses in order) and stores 16 bytes
                    // at the piece's place in fragment (g / 32, k-step): row g % 32 is "lane" g % 32 + 32 (k-half) of that fragment
                    __builtin_amdgcn_wave_barrier();
                    const int k8 = lane & 31, im = lane >> 5;
                    const float up = a.x1_scale[0];
                    const float4 p0 = *reinterpret_cast<const float4*>(smem + OFF_X3 + row * LDX + 8 * k8);
                    const float4 p1 = *reinterpret_cast<const float4*>(smem + OFF_X3 + row * LDX + 8 * k8 + 4);
                    const float f[8] = {p0.x * up, p0.y * up, p0.z * up, p0.w * up, p1.x * up, p1.y * up, p1.z * up, p1.w * up};
                    unsigned w4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const _Float16 h0 = (_Float16)f[2 * i], h1 = (_Float16)f[2 * i + 1];
                        const _Float16 l0 = (_Float16)(f[2 * i] - (float)h0), l1 = (_Float16)(f[2 * i + 1] - (float)h1);
                        const _Float16 e0 = im ? l0 : h0, e1 = im ? l1 : h1;
                        w4[i] = (unsigned)__builtin_bit_cast(unsigned short, e0) | ((unsigned)__builtin_bit_cast(unsigned short, e1) << 16);
                    }
                    const long long frag = ((g >> 5) * (DM / 16) + (k8 >> 1)) * 2 + im;            // [row block][k-step][image]
                    uint4* dst = reinterpret_cast<uint4*>(a.x1_frag + frag * 512 + (((int)(g & 31) + 32 * (k8 & 1)) * 8));
                    *dst = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
            } break;
            case EPI_SAMP: {     // sampling_offset | scale_weights Linear -> so (kept in LDS for the projection below)
                if (side) break;
                const int ncg = un.a.items / un.a.KH;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int n = c * 64 + lane;
                    if (c < ncg && n < a.soN) {
                        float s = 0.f;
                        for (int kh = 0; kh < un.a.KH; ++kh) s += P[(c * un.a.KH + kh) * (R * 64) + row * 64 + lane];
                        s += pv[PV_SAMP_B + n];
                        smem[OFF_H + row * 256 + n] = s;
                        if (live && a.so) a.so[g * a.soN + n] = s;
                    }
                }
            } break;
            }
        }
        SBEV_TRACE(4 + 4 * u)
        if (PAIR && un.epi == EPI_PE3 && member == 1) pair_arrive(pword);
        __syncthreads();
        SBEV_TRACE(5 + 4 * u)
    }
    if (PRE == PRE_ATT && a.proj.loc_bp) {
        // adaptive spatio-temporal sampling, projection and camera selection of the rows' sample points (sample_point.hpp: the
        // same function as sample_project_kernel): thread = (frame t, group g, row, point p) -> 48-byte runs of loc_bp
        const int GP = a.proj.G * a.proj.P;
        for (int i = threadIdx.x; i < R * a.proj.T * GP; i += 64 * NWAVE) {
            const int p = i % a.proj.P;
            const int row = (i / a.proj.P) % R;
            const int tg = i / (a.proj.P * R);
            const int gq = tg % a.proj.G, t = tg / a.proj.G;
            const long long g = row0 + row;
            if (g >= a.M) continue;
            const int b = (int)((unsigned)g / (unsigned)a.Q), q = (int)((unsigned)g - (unsigned)b * (unsigned)a.Q);
            const int gp = gq * a.proj.P + p;
            const float* so = smem + OFF_H + row * 256;
            sbev_ops::sample_point(a.proj, b, t, q, gp, a.proj.bbox + g * 10, so + gp * 3, so + GP * 3 + gp * a.proj.L);
        }
    }
}

// weights [N, K] row-major (ld) -> the order the lanes consume them: [ceil(N/64)][K/16][4][64 lanes][4 floats],
// lane = column within the group, (step, m, e) -> k = 16 step + 4 m + e; rows past N are zero
__global__ void pack_kernel(const float* W, long long ld, int N, int K, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // one float4 each
    const long long total = (long long)((N + 63) / 64) * K * 16;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const long long t = i >> 6;
    const int m = (int)(t & 3);
    const long long t2 = t >> 2;
    const int steps = K / 16;
    const int s = (int)(t2 % steps);
    const int cg = (int)(t2 / steps);
    const int n = cg * 64 + lane;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) v = *reinterpret_cast<const float4*>(W + (long long)n * ld + 16 * s + 4 * m);
    reinterpret_cast<float4*>(out)[i] = v;
}

__global__ void copy_pad_kernel(const float* src, int n, float* dst, int slot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < slot) dst[i] = i < n ? src[i] : 0.f;
}

long long packed_floats(int N, int K) { return (long long)((N + 63) / 64) * 64 * K; }

struct PackMap {     // float offsets into the chain_pack blob
    long long ffn0, ffn1, cls0, reg0, cls3, reg2, cls6, reg4, pe3, attn_in, attn_out, samp, vec, total;
};
PackMap pack_map(const sbev_decoder_config& c) {
    PackMap m{};
    long long o = 0;
    auto put = [&](long long& slot, int N, int K) { slot = o; o += packed_floats(N, K); };
    put(m.ffn0, c.ffn, c.D); put(m.ffn1, c.D, c.ffn);
    put(m.cls0, c.D, c.D); put(m.reg0, c.D, c.D); put(m.cls3, c.D, c.D); put(m.reg2, c.D, c.D);
    put(m.cls6, c.num_classes, c.D); put(m.reg4, c.code_size, c.D);
    put(m.pe3, c.D, c.D); put(m.attn_in, c.attn_in_rows, c.D);
    put(m.attn_out, c.D, c.D); put(m.samp, c.G * c.P * (3 + c.L), c.D);
    m.vec = o;                     // the small vectors: [tail | front | attention], PV_* offsets
    m.total = o + PV_IMAGE_FLOATS;
    return m;
}

// column groups [cg0, cg0 + ncg) of a packed [N, K] weight as items of K / KH k each.  kh = 0: K / 128 (the RG = 1 items)
Lin lin(const float* wp, int in_off, int ld, int N, int K, int kh = 0, int cg0 = 0, int ncg = -1) {
    const int KH = kh > 0 ? kh : K / 128;
    const int all = (N + 63) / 64;
    if (ncg < 0) ncg = all - cg0;
    return Lin{wp + (long long)cg0 * K * 64, in_off, ld, KH, ncg * KH, K / (16 * KH), 0, KH};
}
// k-half `half` of a packed [N, K] weight (pair mode): kh pieces of the half, i.e. pieces [half kh, half kh + kh) of 2 kh per row
Lin lin_khalf(const float* wp, int in_off, int ld, int N, int K, int kh, int half) {
    return Lin{wp, in_off, ld, kh, ((N + 63) / 64) * kh, K / (32 * kh), half * kh, 2 * kh};
}
const Lin kNone{nullptr, 0, LDX, 1, 0, 8, 0, 1};

// k-pieces per column group of a unit with `cgs` column groups in all: RG = 1 keeps 128-k items; RG > 1 aims at 8 items per
// unit (one per wave; each piece at least 64 k = RDEPTH chunks)
// -- <= 12 slots; cost of a choice = sum over its rounds of (chunks per item) x (1 for a round of 5 .. 8 items: two waves per SIMD
// share the MFMA pipe; 1/2 for a last round of <= 4 items: waves 0 .. 3 sit on four different SIMDs)
int pieces(int rg, int cgs, int K) {
    if (rg == 1) return 0;
    static const bool tie_large = getenv("SBEV_CHAIN_TIE_SMALL") == nullptr;      // equal cost: more, shorter items (A/B switch)
    int best = 1;
    double best_cost = 1e30;
    for (int kh = 1; kh <= 4 && cgs * kh <= 12 && K / kh >= 16 * RDEPTH; kh *= 2) {
        const int items = cgs * kh, full = items / NWAVE, rest = items % NWAVE;
        const double cost = (K / (16.0 * kh)) * (full + (rest == 0 ? 0.0 : rest <= 4 ? 0.5 : 1.0));
        if (tie_large ? cost < best_cost + 1e-9 : cost < best_cost - 1e-9) { best_cost = cost; best = kh; }
    }
    return best;
}

// rows per workgroup = 4 rg: the smallest of 4, 8, 16 that covers the rows with one round of workgroups (256 CUs)
int row_groups(long long rows) {
    static const int forced = getenv("SBEV_CHAIN_RG") ? atoi(getenv("SBEV_CHAIN_RG")) : 0;      // A/B: 1, 2 or 4 row groups whatever the row count
    if ((forced == 1 && rows <= 256 * 4) || (forced == 2 && rows <= 256 * 8) || forced == 4) return forced;
    return rows <= 256 * 4 ? 1 : rows <= 256 * 8 ? 2 : 4;
}
Offs offs(int rg) { return rg == 1 ? offs_of<1>() : rg == 2 ? offs_of<2>() : offs_of<4>(); }

// Pair mode of the tail (see "pair mode" above): row groups per PAIR of workgroups, 0 = not applicable.  Both members of every
// pair must be resident at once -- one workgroup per CU (88 KB of LDS and more) -- so the launch, 16 blocks per 8 pairs, has to
// fit the device's CUs in one round: 8 rows per pair, <= 1024 rows on 256 CUs.  (16 rows per pair would reach 2048 rows, but
// four row groups per weight chunk are MFMA-bound, not stream-bound: 88 vs 66 us for the single-workgroup tail at 1600 rows;
// SBEV_CHAIN_PAIR16=1 enables it for A/B runs.)
std::atomic<int> g_chain_pair{getenv("SBEV_NO_CHAIN_PAIR") ? 0 : 1};
std::atomic<int> g_chain_pair_drop{0};
// the sticky fault word (see g_chain_pair_fault_host): 64 bytes of pinned host memory mapped into the device, one per process (a process
// drives one GPU: one process per GPU is the deployment model; with several devices the word is shared -- a fault on any of them stops
// pair mode everywhere).  Installed before the first pair-mode launch on a device; nullptr when the allocation fails (the device-side
// counter still works).
std::atomic<unsigned> g_pair_fault_acked{0};
std::atomic<int> g_pair_fault_warned{0};
volatile unsigned* pair_fault_host() {
    // NOT a function-local static initialised by the first call: that call may run where hipHostMalloc is illegal (a caller's stream
    // capture around its first decoder call) and the failure would be cached for the life of the process -- a failed
    // attempt is retried by the next call; sbev_init() (called when the Python binding loads the library) makes the first attempt early
    static std::atomic<volatile unsigned*> word{nullptr};
    static std::mutex mu;
    volatile unsigned* w = word.load(std::memory_order_acquire);
    if (w) return w;
    std::lock_guard<std::mutex> lk(mu);
    w = word.load(std::memory_order_acquire);
    if (w) return w;
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || !h) {
        const hipError_t e = hipGetLastError();
        if (g_pair_fault_warned.exchange(1) == 0) {
            sbev::set_error("pair-mode fault word: hipHostMalloc failed (%s) -- pair mode stays off until a later call succeeds (stream capture active? call sbev_init() first)",
                            hipGetErrorString(e));
            fprintf(stderr, "sparsebev_amd: %s\n", sbev_last_error());
        }
        return nullptr;
    }
    *static_cast<volatile unsigned*>(h) = 0u;
    word.store(static_cast<volatile unsigned*>(h), std::memory_order_release);
    return static_cast<volatile unsigned*>(h);
}
bool pair_fault_install() {            // this device's g_chain_pair_fault_host -> the word; once per device
    static std::atomic<int> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (done[dev & 63].load(std::memory_order_acquire)) return true;
    volatile unsigned* h = pair_fault_host();
    if (!h) return false;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<unsigned*>(h), 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_chain_pair_fault_host), &d, sizeof(d)) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (!sbev::out_fold_install(d)) { (void)hipGetLastError(); return false; }      // (the out-projection's in-launch fold reports into the same word)
    done[dev & 63].store(1, std::memory_order_release);
    return true;
}
int pair_blocks(long long rows, int rg) {
    const long long pairs = (rows + 4 * rg - 1) / (4 * rg);
    return (int)(16 * ((pairs + 7) / 8));
}
int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int v = cus[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}
int pair_row_groups(long long rows) {
    if (g_chain_pair.load(std::memory_order_relaxed) == 0 || rows > 256 * 16) return 0;
    if (!pair_fault_install()) return 0;        // no way to report a lost partner: the single-workgroup tail
    const int cus = device_cus();
    static const int max_rg = getenv("SBEV_CHAIN_PAIR16") ? 4 : 2;
    for (int rg = 2; rg <= max_rg; rg *= 2)
        if (pair_blocks(rows, rg) <= cus) return rg;
    return 0;
}

}  // namespace

namespace sbev {

// pair mode workspace: exchange rows (floats) and arrival counters (words) for `rows` rows, whatever the row block it picks
long long chain_pair_floats(long long rows) { return 3 * (rows + 15) * DM; }
// (+ 64 words behind the pair counters: the out-projection's fold counters, one per row tile -- zeroed by the same attention-chain loop)
long long chain_pair_sync_words(long long rows) { return (rows + 7) / 8 + 64; }
long long chain_fold_sync_offset(long long rows) { return (rows + 7) / 8; }
bool chain_fault_word_ready() { return pair_fault_install(); }
bool chain_pair_enabled() { return g_chain_pair.load(std::memory_order_relaxed) != 0; }

// Install the fault word for the current device NOW (hipHostMalloc + hipMemcpyToSymbol: neither is legal under stream capture, and a
// caller's first pair-mode launch may be a captured one): sbev_decoder_workspace_bytes calls this, which every caller runs before its
// first forward / capture.  Quiet without a device.
void chain_pair_prepare() {
    if (g_chain_pair.load(std::memory_order_relaxed) != 0) (void)pair_fault_install();
}

// hand-offs that timed out and were not acknowledged yet (host memory: no device synchronisation)
unsigned chain_pair_faults_pending() {
    volatile unsigned* h = pair_fault_host();
    return h ? *h - g_pair_fault_acked.load(std::memory_order_relaxed) : 0u;
}

bool row_chain_supported(const sbev_decoder_config& c) {
    return c.D == DM && c.ffn == FF && c.code_size == 10 && c.num_classes >= 1 && c.num_classes <= 64 &&
           c.attn_in_rows <= PV_QKV_PAD && c.G * c.P * (3 + c.L) <= 256 && (long long)c.B * c.Q < 0x7fffffffLL;
}

// Every workgroup streams every weight: that pays while the launch is a single round of workgroups (<= 256 CUs).
// Measured (samples/s with / without the chains), 4 rows per workgroup: config 2 (900 rows) 350 / 328, config 5 387 / 358; config 3
// (3200 rows) 747 / 755, config 4 (3600 rows) 271 / 272 -- so larger batches take 8 or 16 rows per workgroup (round 3), and beyond
// 4096 rows the op-by-op launches, whose tiles amortise the weights, stay.  SBEV_CHAIN_MAX_ROWS overrides the limit (A/B).
bool row_chain_pays(long long rows) {
    static const long long limit = getenv("SBEV_CHAIN_MAX_ROWS") ? atoll(getenv("SBEV_CHAIN_MAX_ROWS")) : 256 * 16;
    return rows <= limit && rows <= 256 * 16;
}

static void fill_common(ChainArgs& a, const sbev_decoder_config& c, float eps) {
    a.M = (long long)c.B * c.Q;
    a.Q = c.Q;
    a.eps = eps;
    a.num_classes = c.num_classes;
    a.attn_in_rows = c.attn_in_rows;
    a.soN = c.G * c.P * (3 + c.L);
}

static int add_front(ChainArgs& a, int n, const sbev_decoder_config& c, const float* pk, const PackMap& m, int rg) {
    const Offs o = offs(rg);
    a.units[n++] = Unit{lin(pk + m.pe3, o.c, LDX, c.D, c.D, pieces(rg, c.D / 64, c.D)), kNone, EPI_PE3, 0};
    const int ncg = (c.attn_in_rows + 63) / 64;
    for (int cg0 = 0; cg0 < ncg; cg0 += 8) {
        const int n_here = ncg - cg0 < 8 ? ncg - cg0 : 8;
        a.units[n++] = Unit{lin(pk + m.attn_in, o.x2, LDX, c.attn_in_rows, c.D, pieces(rg, n_here, c.D), cg0, n_here), kNone, EPI_QKV, cg0 * 64};
    }
    return n;
}

// the kernels use 159 KB of dynamic LDS: raised once per instantiation (also from sbev_decoder_chain_pack, so that the first
// launch may already be inside a stream capture)
template <int PRE, int RG, bool PAIR = false>
static hipError_t lds_attr() {
    static std::atomic<unsigned long long> done{0};      // bit d: raised on device d (a process may drive several devices)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(row_chain_kernel<PRE, RG, PAIR>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, Lay<RG>::LDS_TOTAL_FLOATS * 4);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

template <int PRE, int RG, bool PAIR = false>
static int launch_t(const ChainArgs& a, hipStream_t s, const char* what) {
    constexpr int R = Lay<RG>::R, LDS_TOTAL_FLOATS = Lay<RG>::LDS_TOTAL_FLOATS;
    const hipError_t attr = lds_attr<PRE, RG, PAIR>();
    if (attr != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(%d bytes of LDS): %s", what, LDS_TOTAL_FLOATS * 4, hipGetErrorString(attr));
        return SBEV_ELAUNCH;
    }
    static const bool no_warm = getenv("SBEV_CHAIN_NO_WARM") != nullptr;       // A/B switch
    ChainArgs b = a;
    if (no_warm) b.warm_lines = 0;
#ifdef SBEV_CHAIN_TRACE
    static long long* tr = nullptr;
    static int calls = 0;
    if (!tr) { (void)hipMalloc(&tr, 2 * 64 * 8 * 8 * 64); (void)hipMemset(tr, 0, 2 * 64 * 8 * 8 * 64); }
    b.trace = tr + (calls % 64) * 512;
#endif
    const unsigned grid = PAIR ? (unsigned)pair_blocks(a.M, RG) : (unsigned)((a.M + R - 1) / R);
    hipLaunchKernelGGL((row_chain_kernel<PRE, RG, PAIR>), dim3(grid), dim3(64 * NWAVE), LDS_TOTAL_FLOATS * 4, s, b);
#ifdef SBEV_CHAIN_TRACE
    if (++calls == 13) {        // the first step's 13 launches
        (void)hipDeviceSynchronize();
        static long long h[2 * 64 * 512];
        (void)hipMemcpy(h, tr, sizeof(h), hipMemcpyDeviceToHost);
        for (int c = 0; c < 13; ++c) {
            const long long* t = h + c * 512;
            printf("launch %d (PRE %d RG %d PAIR %d; 10 ns ticks):", c, PRE, RG, (int)PAIR);
            for (int blk = 0; blk < 2; ++blk)
                for (int w = 0; w < 8; w += 7) {
                    const long long* tb = t + blk * 64 * 512;
                    printf(" [b%d w%d]", blk ? 15 : 7, w);
                    for (int i = 0; i < 2 + 4 * 8 && tb[w * 64 + i]; ++i) printf(" %lld", tb[w * 64 + i] - t[0]);
                }
            printf("\n");
        }
    }
#endif
    return check_launch(what);
}

template <int RG>
static bool lds_ready_rg() {
    return lds_attr<PRE_SLABS, RG>() == hipSuccess && lds_attr<PRE_FRONT, RG>() == hipSuccess && lds_attr<PRE_ATT, RG>() == hipSuccess;
}
bool chain_lds_ready() {
    return lds_ready_rg<1>() && lds_ready_rg<2>() && lds_ready_rg<4>() && lds_attr<PRE_SLABS, 2, true>() == hipSuccess &&
           lds_attr<PRE_SLABS, 4, true>() == hipSuccess;
}

template <int RG>
static int launch_rg(const ChainArgs& a, hipStream_t s, const char* what) {
    return a.pre == PRE_SLABS ? launch_t<PRE_SLABS, RG>(a, s, what) : a.pre == PRE_FRONT ? launch_t<PRE_FRONT, RG>(a, s, what)
                                                                                          : launch_t<PRE_ATT, RG>(a, s, what);
}
static int launch(const ChainArgs& a, int rg, hipStream_t s, const char* what) {
    return rg == 1 ? launch_rg<1>(a, s, what) : rg == 2 ? launch_rg<2>(a, s, what) : launch_rg<4>(a, s, what);
}

// position encoder + attention in-projection of the FIRST layer (the later layers' run at the end of the previous tail)
int launch_chain_front(const sbev_decoder_config& c, const sbev_decoder_weights& w, const float* bbox, const float* feat, float* x,
                       float* qkvt, float eps, hipStream_t s) {
    const PackMap m = pack_map(c);
    ChainArgs a{};
    fill_common(a, c, eps);
    a.pre = PRE_FRONT;
    a.bbox = bbox; a.feat = feat; a.x = x; a.qkvt = qkvt;
    const int rg = row_groups(a.M);
    a.n_units = add_front(a, 0, c, w.chain_pack, m, rg);
    a.warm = w.chain_pack + m.pe3;
    a.warm_lines = (int)((m.attn_out - m.pe3) / 32);
    a.vec = w.chain_pack + m.vec + PV_TAIL_END; a.vec_off = PV_TAIL_END; a.vec_n = PV_FRONT_END - PV_TAIL_END;
    return launch(a, rg, s, "row chain (front)");
}

// attention out-projection + residual + norm1 -> x1, sampling Linear -> sample points -> projection (loc, level weights)
int launch_chain_attn(const sbev_decoder_config& c, const sbev_decoder_weights& w, const float* att, const float* x, float* x1,
                      const float* bbox, const float* time_diff, const float* lidar2img, float* loc_bp, float* w_bp, float eps,
                      hipStream_t s, uint16_t* x1_frag, const float* x1_scale, uint32_t* pair_sync, const LazyPlan* touch, uint32_t* touch_need) {
    const PackMap m = pack_map(c);
    ChainArgs a{};
    fill_common(a, c, eps);
    a.pre = PRE_ATT;
    a.zero_words = pair_sync;            // the arrival counters of this layer's tail (pair mode), zeroed here: chain_pair_sync_words()
    a.n_zero = pair_sync ? (int)chain_pair_sync_words(a.M) : 0;
    a.att = att; a.x = const_cast<float*>(x); a.x1_out = x1; a.so = nullptr;
    a.x1_frag = x1_frag; a.x1_scale = x1_scale;
    a.proj = sbev_ops::sample_point_args(bbox, time_diff, lidar2img, c.pc_range, c.B, c.Q, c.T, c.N, c.G, c.P, c.L, c.image_h, c.image_w,
                                         c.eps_homo, loc_bp, w_bp);
    if (touch && touch_need) sbev_ops::sample_point_touch(a.proj, *touch, c.hw, touch_need);      // on-demand relayout: mark what the points read
    const int rg = row_groups(a.M);
    const Offs o = offs(rg);
    a.units[0] = Unit{lin(w.chain_pack + m.attn_out, o.x2, LDX, c.D, c.D, pieces(rg, c.D / 64, c.D)), kNone, EPI_AOUT, 0};
    a.units[1] = Unit{lin(w.chain_pack + m.samp, o.x3, LDX, a.soN, c.D, pieces(rg, (a.soN + 63) / 64, c.D)), kNone, EPI_SAMP, 0};
    a.n_units = 2;
    a.warm = w.chain_pack + m.attn_out;
    a.warm_lines = (int)((m.vec - m.attn_out) / 32);
    a.vec = w.chain_pack + m.vec + PV_FRONT_END; a.vec_off = 0; a.vec_n = PV_ATTN_END;
    return launch(a, rg, s, "row chain (attention)");
}

// out_proj slabs -> norm2 -> ffn -> norm3 -> branches -> refine_bbox [-> the next layer's position encoder + in-projection]
int launch_chain_tail(const sbev_decoder_config& c, const sbev_decoder_weights& w, const float* slabs, int splits, const float* x1,
                      const float* bbox, const float* vel_div, float* x3, float* cls_out, float* box_out, int with_front, float* x,
                      float* qkvt, float eps, hipStream_t s, float* pair_x, uint32_t* pair_sync) {
    const PackMap m = pack_map(c);
    const float* pk = w.chain_pack;
    ChainArgs a{};
    fill_common(a, c, eps);
    a.pre = PRE_SLABS;
    a.slabs = slabs; a.splits = splits; a.x1 = x1; a.bbox = bbox; a.vel_div = vel_div;
    a.x3 = x3; a.cls_out = cls_out; a.box_out = box_out; a.front = with_front; a.x = x; a.qkvt = qkvt;
    a.warm = pk + m.ffn0;
    a.warm_lines = (int)(((with_front ? m.attn_out : m.pe3) - m.ffn0) / 32);
    a.vec = pk + m.vec; a.vec_off = 0; a.vec_n = with_front ? PV_FRONT_END : PV_TAIL_END;
    const int dcg = c.D / 64;
    const int prg = pair_x && pair_sync ? pair_row_groups(a.M) : 0;
    if (prg) {
        // pair mode: two workgroups per row block, each with its own unit table (see "pair mode" above the kernel)
        const int rg = prg;
        const Offs o = offs(rg);
        const int half_cg = c.ffn / 128;                                   // column groups of half the hidden layer
        const int kh1 = pieces(rg, dcg, c.ffn / 2);                        // k-pieces of a member's half of ffn.layers.1
        // in-projection: member 1 has the x rows in its LDS, member 0 gets them a hand-off (~1.5 us) later -- so member 1 takes up to
        // 8 column groups (one round of 16-chunk items) and member 0 the rest (5 of config 2's 13: 10 items of 8 chunks)
        static const int q0_forced = getenv("SBEV_CHAIN_PAIR_Q0") ? atoi(getenv("SBEV_CHAIN_PAIR_Q0")) : -1;      // A/B
        const int qcg = (c.attn_in_rows + 63) / 64;
        const int q0 = q0_forced >= 0 && q0_forced <= qcg ? q0_forced : (qcg > 8 ? qcg - 8 : qcg / 2);
        int n = 0;
        for (int mem = 0; mem < 2; ++mem) {
            Unit* u = mem ? a.units_b : a.units;
            n = 0;
            u[n++] = Unit{lin(pk + m.ffn0, o.x2, LDX, c.ffn, c.D, pieces(rg, half_cg, c.D), mem * half_cg, half_cg), kNone, EPI_FFN0, 0};
            u[n++] = Unit{lin_khalf(pk + m.ffn1, o.h, LDH, c.D, c.ffn, kh1, mem), kNone, EPI_FFN1, 0};
            const Lin br1 = lin(pk + (mem ? m.reg0 : m.cls0), o.x3, LDX, c.D, c.D, pieces(rg, dcg, c.D));
            const Lin br2 = lin(pk + (mem ? m.reg2 : m.cls3), mem ? o.r : o.c, LDX, c.D, c.D, pieces(rg, dcg, c.D));
            const Lin out = mem ? lin(pk + m.reg4, o.r, LDX, c.code_size, c.D, pieces(rg, 1, c.D)) : lin(pk + m.cls6, o.c, LDX, c.num_classes, c.D, pieces(rg, 1, c.D));
            u[n++] = mem ? Unit{kNone, br1, EPI_BR1, 0} : Unit{br1, kNone, EPI_BR1, 0};
            u[n++] = mem ? Unit{kNone, br2, EPI_BR2, 0} : Unit{br2, kNone, EPI_BR2, 0};
            u[n++] = mem ? Unit{kNone, out, EPI_OUT, 0} : Unit{out, kNone, EPI_OUT, 0};
            if (with_front) {
                u[n++] = Unit{mem ? lin(pk + m.pe3, o.c, LDX, c.D, c.D, pieces(rg, dcg, c.D)) : kNone, kNone, EPI_PE3, 0};
                const int cg0 = mem ? q0 : 0, ncg = mem ? qcg - q0 : q0;
                u[n++] = Unit{ncg > 0 ? lin(pk + m.attn_in, o.x2, LDX, c.attn_in_rows, c.D, pieces(rg, ncg, c.D), cg0, ncg) : kNone, kNone, EPI_QKV, cg0 * 64};
            }
        }
        a.n_units = n;
        a.n_pairs = (int)((a.M + 4 * rg - 1) / (4 * rg));
        a.pair_x = pair_x;
        a.pair_sync = pair_sync;
        a.debug_drop = g_chain_pair_drop.load(std::memory_order_relaxed);
        return rg == 2 ? launch_t<PRE_SLABS, 2, true>(a, s, "row chain (tail, pairs)") : launch_t<PRE_SLABS, 4, true>(a, s, "row chain (tail, pairs)");
    }
    int n = 0;
    const int rg = row_groups(a.M);
    const Offs o = offs(rg);
    a.units[n++] = Unit{lin(pk + m.ffn0, o.x2, LDX, c.ffn, c.D, pieces(rg, c.ffn / 64, c.D)), kNone, EPI_FFN0, 0};
    a.units[n++] = Unit{lin(pk + m.ffn1, o.h, LDH, c.D, c.ffn, pieces(rg, dcg, c.ffn)), kNone, EPI_FFN1, 0};
    a.units[n++] = Unit{lin(pk + m.cls0, o.x3, LDX, c.D, c.D, pieces(rg, 2 * dcg, c.D)), lin(pk + m.reg0, o.x3, LDX, c.D, c.D, pieces(rg, 2 * dcg, c.D)),
                        EPI_BR1, 0};
    a.units[n++] = Unit{lin(pk + m.cls3, o.c, LDX, c.D, c.D, pieces(rg, 2 * dcg, c.D)), lin(pk + m.reg2, o.r, LDX, c.D, c.D, pieces(rg, 2 * dcg, c.D)),
                        EPI_BR2, 0};
    a.units[n++] = Unit{lin(pk + m.cls6, o.c, LDX, c.num_classes, c.D, pieces(rg, 2, c.D)), lin(pk + m.reg4, o.r, LDX, c.code_size, c.D, pieces(rg, 2, c.D)),
                        EPI_OUT, 0};
    if (with_front) n = add_front(a, n, c, pk, m, rg);
    a.n_units = n;
    return launch(a, rg, s, "row chain (tail)");
}

}  // namespace sbev

// pair mode of the tail: on (default) / off; returns the previous setting (A/B measurements, tests)
extern "C" int sbev_decoder_chain_pair(int enable) { return g_chain_pair.exchange(enable != 0 ? 1 : 0); }

// TEST HOOK: with 1, member 1 of pair 0 of every pair-mode tail exits at once, so that its partner runs into the poll bound (the rows
// of that pair are garbage, every other row is unaffected, the launch ends); returns the previous setting
extern "C" int sbev_debug_chain_pair_drop(int enable) { return g_chain_pair_drop.exchange(enable != 0 ? 1 : 0); }

// partners that did not show up within the poll bound since the library was loaded (0 unless the GPU was shared with something
// that kept half of a pair from being scheduled for about a second); synchronises the device
extern "C" int64_t sbev_decoder_chain_pair_timeouts(void) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_chain_pair_timeouts), sizeof(v)) != hipSuccess) return -1;
    const long long f = sbev::out_fold_timeouts();          // (+ row tiles of the out-projection's in-launch fold that never completed)
    if (f < 0) return -1;
    return (int64_t)v + f;
}

// the sticky fault word: timed-out hand-offs since the last acknowledgement, read from pinned host memory (NO synchronisation; a fault
// becomes visible once the faulting launch has run, ~1 s after its partner went missing)
extern "C" int64_t sbev_decoder_chain_pair_faults(void) { return (int64_t)sbev::chain_pair_faults_pending(); }
// One-time per-device setup that must not happen inside a caller's stream capture: the pair tail's host-mapped fault word (hipHostMalloc +
// hipMemcpyToSymbol).  The Python binding calls it when it loads the library on a GPU box; C callers call it once per device before their
// first capture (sbev_decoder_workspace_bytes still does the same lazily, and a failed attempt is retried by later calls).
extern "C" int sbev_init(void) {
    if (g_chain_pair.load(std::memory_order_relaxed) == 0) return SBEV_OK;
    if (!pair_fault_install()) {
        sbev::set_error("sbev_init: could not install the pair-mode fault word on the current device (pair mode stays off until a later attempt succeeds)");
        return SBEV_ELAUNCH;
    }
    return SBEV_OK;
}
// acknowledge them (the caller has dealt with the invalid step): sbev_decoder_forward accepts calls again.  Returns how many there were.
extern "C" int64_t sbev_decoder_chain_pair_faults_ack(void) {
    volatile unsigned* h = pair_fault_host();
    if (!h) return 0;
    const unsigned now = *h;
    return (int64_t)(now - g_pair_fault_acked.exchange(now));
}

extern "C" int64_t sbev_decoder_chain_pack_floats(const sbev_decoder_config* cfg) {
    if (!cfg || !sbev::row_chain_supported(*cfg)) return 0;
    return pack_map(*cfg).total;
}

extern "C" int sbev_decoder_chain_pack(const sbev_decoder_config* cfg, const sbev_decoder_weights* w, float* out, sbev_stream_t stream) {
    SBEV_REQUIRE(cfg && w && out, "sbev_decoder_chain_pack: null pointer");
    SBEV_REQUIRE(sbev::row_chain_supported(*cfg), "sbev_decoder_chain_pack: config not covered by the row-chain kernels (embed_dims 256, ffn 512, code_size 10)");
    SBEV_REQUIRE((((uintptr_t)out) & 15) == 0, "sbev_decoder_chain_pack: output must be 16-byte aligned");
    const sbev_decoder_config& c = *cfg;
    const PackMap m = pack_map(c);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    SBEV_REQUIRE(sbev::chain_lds_ready(), "sbev_decoder_chain_pack: the device refuses %d bytes of LDS per workgroup", Lay<1>::LDS_TOTAL_FLOATS * 4);
    struct Job { const float* W; long long off; int N, K; };
    const Job jobs[12] = {{w->ffn0_w, m.ffn0, c.ffn, c.D}, {w->ffn1_w, m.ffn1, c.D, c.ffn}, {w->cls0_w, m.cls0, c.D, c.D},
                          {w->reg0_w, m.reg0, c.D, c.D}, {w->cls3_w, m.cls3, c.D, c.D}, {w->reg2_w, m.reg2, c.D, c.D},
                          {w->cls6_w, m.cls6, c.num_classes, c.D}, {w->reg4_w, m.reg4, c.code_size, c.D}, {w->pe3_w, m.pe3, c.D, c.D},
                          {w->attn_in_w, m.attn_in, c.attn_in_rows, c.D}, {w->attn_out_w, m.attn_out, c.D, c.D},
                          {w->samp_w, m.samp, c.G * c.P * (3 + c.L), c.D}};
    for (const Job& j : jobs) {
        SBEV_REQUIRE(j.W && (((uintptr_t)j.W) & 15) == 0, "sbev_decoder_chain_pack: null / unaligned weight");
        const long long n4 = (long long)((j.N + 63) / 64) * j.K * 16;
        hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, j.W, (long long)j.K, j.N, j.K, out + j.off);
    }
    // the small vectors, zero-padded to their slots
    float* pv = out + m.vec;
    struct Vec { const float* p; int off, n, slot; };
    const int soN = c.G * c.P * (3 + c.L);
    const Vec vecs[] = {
        {w->op_b, PV_OP_B, c.D, 256}, {w->norm2_g, PV_N2G, c.D, 256}, {w->norm2_b, PV_N2B, c.D, 256}, {w->ffn0_b, PV_FFN0_B, c.ffn, 512},
        {w->ffn1_b, PV_FFN1_B, c.D, 256}, {w->norm3_g, PV_N3G, c.D, 256}, {w->norm3_b, PV_N3B, c.D, 256},
        {w->cls0_b, PV_CLS0_B, c.D, 256}, {w->cls1_g, PV_CLS1G, c.D, 256}, {w->cls1_b, PV_CLS1B, c.D, 256},
        {w->cls3_b, PV_CLS3_B, c.D, 256}, {w->cls4_g, PV_CLS4G, c.D, 256}, {w->cls4_b, PV_CLS4B, c.D, 256},
        {w->cls6_b, PV_CLS6_B, c.num_classes, 64}, {w->reg0_b, PV_REG0_B, c.D, 256}, {w->reg2_b, PV_REG2_B, c.D, 256},
        {w->reg4_b, PV_REG4_B, c.code_size, 64},
        {w->pe0_w, PV_PE0_W, 3 * c.D, 768}, {w->pe0_b, PV_PE0_B, c.D, 256}, {w->pe1_g, PV_PE1G, c.D, 256}, {w->pe1_b, PV_PE1B, c.D, 256},
        {w->pe3_b, PV_PE3_B, c.D, 256}, {w->pe4_g, PV_PE4G, c.D, 256}, {w->pe4_b, PV_PE4B, c.D, 256},
        {w->attn_in_b, PV_QKV_B, c.attn_in_rows, PV_QKV_PAD},
        {w->attn_out_b, PV_FRONT_END + PV_AOUT_B, c.D, 256}, {w->norm1_g, PV_FRONT_END + PV_N1G, c.D, 256},
        {w->norm1_b, PV_FRONT_END + PV_N1B, c.D, 256}, {w->samp_b, PV_FRONT_END + PV_SAMP_B, soN, 256}};
    for (const Vec& v : vecs) {
        SBEV_REQUIRE(v.p, "sbev_decoder_chain_pack: null bias / LayerNorm pointer");
        hipLaunchKernelGGL(copy_pad_kernel, dim3((unsigned)((v.slot + 255) / 256)), dim3(256), 0, s, v.p, v.n, pv + v.off, v.slot);
    }
    return sbev::check_launch("sbev_decoder_chain_pack");
}
