// Row-local Linear chains on TEAMS of workgroups (round 4 EXPERIMENT, not part of libsbev_hip.so).
// Build + run: tools/exp/r4_run13.sh (hipcc -shared against libsbev_hip.so) and tools/exp/r4_team_proto.py.
// Result (DESIGN_HISTORY.md section 10): correct to 1.3e-6, deadlock-free, self-resetting -- and 7 - 9 us per stage, no faster than the
// row chains' 6.8 us: 1.7 - 3.7 us of VALU to LayerNorm / scale / split the gathered rows (repeated in all 8 slices), 2 - 3 us of
// exposed latency in the epilogue, 1 - 2 us per hand-off.  Kept as the measured record of that design.
//
// row_chain.hip gives every workgroup 4 rows and lets it stream ALL weights of the chain (3.25 MB for the tail): 225 workgroups x
// 3.25 MB through an L2 -> CU path that delivers ~60 GB/s per CU -- 54 us for 1.4 GFLOP, whatever the item order or the row count per
// workgroup (DESIGN_HISTORY.md section 10).  The only way to stream less per CU is to split the COLUMNS of every Linear over several CUs,
// which makes every stage an all-gather of the previous stage's output rows.  Here a team of 8 workgroups owns 32 rows: workgroup
// (team, slice) computes the column fragments slice, slice + 8, ... of each stage for those rows on v_mfma_f32_32x32x16_f16 with
// fp16 hi + lo operands (3 products, fp32-class: gemm_bf16s.hip), publishes them through global memory and meets its team at a
// counter before the next stage reads the full rows back.  A workgroup streams 1/8 of the weights (pre-packed MFMA fragments, straight
// into registers, requested BEFORE the team barrier: they do not depend on it).
//
// Hand-off (MI355X guide, "inter-workgroup visibility"): payload with sc0 sc1 stores (write-through to memory), every wave drains
// them (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds to the team's counter (agent-scope atomic); the consumer polls that
// counter with relaxed agent-scope loads from one lane (s_sleep between polls, bounded: a lost team member can never hang the GPU --
// the kernel raises an error word instead), then reads the payload with sc0 sc1 loads.  Correct for any workgroup -> XCD placement;
// `fast` (all 8 members on one XCD, which the block -> team map arranges under round-robin dispatch and the kernel VERIFIES from
// HW_REG_XCC_ID before using it) keeps the payload in that XCD's L2: plain stores, sc1 loads.
// The counters are self-resetting (the last workgroup to leave a team zeroes them): no memset node per launch.
#include "../../sparsebev_amd/csrc/sbev_common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TEAM = 8;              // workgroups per team = column slices
constexpr int ROWS = 32;             // rows per team (one MFMA row fragment)
constexpr int KMAX = 512;            // widest stage input (the ffn hidden layer)
constexpr int XS_BYTES = (KMAX / 16) * 2 * 1024;      // X as fp16 hi | lo MFMA fragments: [k-step][image][64 lanes][8 halfs]
constexpr int RED_BYTES = 4 * 16 * 64 * 4;            // k-split partial sums [wave][register][lane]
constexpr int LDS_BYTES = XS_BYTES + RED_BYTES + 2 * ROWS * 4 + 64;
constexpr unsigned POLL_LIMIT = 1u << 20;             // ~ a second: then the error word, never a hang
constexpr int WKS = 16;                               // k-steps of weights a wave may hold (128 registers)

// ---- exchange loads / stores -----------------------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ void xload8(const float* p, f32x4& a, f32x4& b) {       // 8 consecutive floats
    if constexpr (FAST)
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
template <bool FAST>
__device__ __forceinline__ float xload1(const float* p) {
    float v;
    if constexpr (FAST) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <bool FAST>
__device__ __forceinline__ void xstore1(float* p, float v) {
    if constexpr (FAST) *p = v;
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// phase stamps of workgroup 0 (tools/exp/r4_team_proto.py prints them): s_memtime ticks = shader cycles
#define TEAM_STAMP(i) do { if (team.trace && blockIdx.x == 0 && threadIdx.x == 0) team.trace[team.nstamp++] = (long long)__builtin_readcyclecounter(); (void)(i); } while (0)

struct Team {
    long long* trace;        // null: no stamps
    int nstamp;
    unsigned* counter;       // [2] per team: arrivals, departures
    unsigned* error;         // one word per launch target: set when a poll ran into POLL_LIMIT
    unsigned epoch;          // arrivals consumed so far by this workgroup (TEAM per barrier)
};

// every wave has drained its stores; one lane publishes, one lane polls
__device__ __forceinline__ void team_arrive(Team& t) {
    drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(t.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t.epoch += TEAM;
}
__device__ __forceinline__ void team_wait(Team& t) {
    if (threadIdx.x == 0) {
        unsigned n = 0;
        while (__hip_atomic_load(t.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++n > POLL_LIMIT) {
                __hip_atomic_store(t.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}
// the last member to leave resets both words: the next launch starts from zero without a memset node
__device__ __forceinline__ void team_leave(const Team& t) {
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(t.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == TEAM - 1) {
            __hip_atomic_store(t.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(t.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- one stage ---------------------------------------------------------------------------------------------------------------------
struct StageArgs {
    const float* X;              // [M, K] rows of the previous stage (exchange memory when `x_exchange`)
    int K, x_exchange;
    const float* ln_g;           // LayerNorm over K on load (null: none) ...
    const float* ln_b;
    int pro_relu;                // ... then ReLU
    float* xn_out;               // the prologue's output rows [M, K], written by slice 0 (null: not needed)
    const unsigned short* Wf;    // [N/32][K/16][2][64][8] fp16 hi | lo fragments (sbev_pack_f16s_frags, per-row scales)
    const float* wdown;          // [N] 2^-e of W's rows
    const float* bias;           // [N]
    int N;                       // multiple of 32
    int relu_from;               // output columns >= relu_from get a ReLU (N: none)
    const float* res;            // [M, N] residual added to the output (null: none); exchange memory when `res_exchange`
    int res_exchange;
    float* Y;                    // [M, N]
    int y_exchange;              // the next stage of this launch reads it
};

__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    return v;
}

template <bool FAST>
__device__ void run_stage(const StageArgs& s, long long M, int row0, int slice, float eps, unsigned char* lds, Team& team, bool wait_team) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = s.K, KS = K / 16;
    const int nfr_all = s.N / 32;
    const int NF = (nfr_all - slice + TEAM - 1) / TEAM;              // this workgroup's column fragments: slice, slice + 8, ...
    // waves per fragment (k split) and this wave's share
    const int ksplit = NF <= 1 ? 4 : NF == 2 ? 2 : 1;
    const int myf = NF <= 1 ? 0 : NF == 2 ? (wave >> 1) : wave;      // local fragment index (>= NF: idle wave)
    const int kq = NF <= 1 ? wave : NF == 2 ? (wave & 1) : 0;
    const int ks_n = KS / ksplit, ks0 = kq * ks_n;
    const bool busy = myf < NF && NF > 0;
    const int cf = slice + TEAM * myf;                                 // global column fragment

    // ---- this wave's weights: requested before the team barrier (they do not depend on it)
    bf16x8 wfr[WKS][2];
    if (busy) {
        const unsigned short* wb = s.Wf + (((long long)cf * KS + ks0) * 2 * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < WKS; ++i)
            if (i < ks_n) {
                wfr[i][0] = *reinterpret_cast<const bf16x8*>(wb + (long long)(i * 2) * 512);
                wfr[i][1] = *reinterpret_cast<const bf16x8*>(wb + (long long)(i * 2 + 1) * 512);
            }
    }
    TEAM_STAMP(0);
    if (wait_team) team_wait(team);
    TEAM_STAMP(1);

    // ---- X rows -> (LayerNorm, ReLU) -> fp16 hi | lo fragments in LDS, one power of two per row
    float* rowdown = reinterpret_cast<float*>(lds + XS_BYTES + RED_BYTES);
    {
        const int r = tid >> 3, sub = tid & 7;
        const long long gr = row0 + r;
        const long long grc = gr < M ? gr : M - 1;
        const float* xr = s.X + grc * K;
        const int nch = K / 64;                                         // chunks of 8 k per thread: chunk c = sub + 8 j
        f32x4 xa[KMAX / 64], xb[KMAX / 64];
#pragma unroll
        for (int j = 0; j < KMAX / 64; ++j)
            if (j < nch) {
                const float* p = xr + (sub + 8 * j) * 8;
                if (s.x_exchange) xload8<FAST>(p, xa[j], xb[j]);
                else { xa[j] = *reinterpret_cast<const f32x4*>(p); xb[j] = *reinterpret_cast<const f32x4*>(p + 4); }
            }
        if (s.x_exchange) {
#pragma unroll
            for (int j = 0; j < KMAX / 64; ++j)
                if (j < nch) asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa[j]), "+v"(xb[j])::"memory");
        }
        TEAM_STAMP(2);
        if (s.ln_g) {
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < KMAX / 64; ++j)
                if (j < nch)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sum += xa[j][e] + xb[j][e]; }
            sum = group8_sum(sum);
            const float mean = sum / (float)K;
#pragma unroll
            for (int j = 0; j < KMAX / 64; ++j)
                if (j < nch)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float a = xa[j][e] - mean, b = xb[j][e] - mean; sq += a * a + b * b; }
            sq = group8_sum(sq);
            const float rstd = rsqrtf(sq / (float)K + eps);
#pragma unroll
            for (int j = 0; j < KMAX / 64; ++j)
                if (j < nch) {
                    const int k0 = (sub + 8 * j) * 8;
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(s.ln_g + k0), g1 = *reinterpret_cast<const f32x4*>(s.ln_g + k0 + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(s.ln_b + k0), b1 = *reinterpret_cast<const f32x4*>(s.ln_b + k0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xa[j][e] = (xa[j][e] - mean) * rstd * g0[e] + b0[e];
                        xb[j][e] = (xb[j][e] - mean) * rstd * g1[e] + b1[e];
                    }
                }
        }
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < KMAX / 64; ++j)
            if (j < nch)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (s.pro_relu) { xa[j][e] = fmaxf(xa[j][e], 0.f); xb[j][e] = fmaxf(xb[j][e], 0.f); }
                    mx = fmaxf(mx, fmaxf(fabsf(xa[j][e]), fabsf(xb[j][e])));
                }
        mx = group8_max(mx);
        float up = 1.f, down = 1.f;
        if (mx > 0.f && mx < __builtin_inff()) {
            int ex;
            (void)frexpf(mx, &ex);
            const float e = fminf(fmaxf((float)(15 - ex), -126.f), 126.f);
            up = exp2f(e);
            down = exp2f(-e);
        }
        if (sub == 0) rowdown[r] = down;
        if (s.xn_out && slice == 0 && gr < M) {
#pragma unroll
            for (int j = 0; j < KMAX / 64; ++j)
                if (j < nch) {
                    float* o = s.xn_out + gr * K + (sub + 8 * j) * 8;
                    *reinterpret_cast<f32x4*>(o) = xa[j];
                    *reinterpret_cast<f32x4*>(o + 4) = xb[j];
                }
        }
#pragma unroll
        for (int j = 0; j < KMAX / 64; ++j)
            if (j < nch) {
                const int c = sub + 8 * j;                               // chunk: k-step c / 2, k half c & 1
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = (e < 2 ? xa[j][2 * e] : xb[j][2 * e - 4]) * up, v1 = (e < 2 ? xa[j][2 * e + 1] : xb[j][2 * e - 3]) * up;
                    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
                    const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
                    hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                }
                unsigned char* dst = lds + ((c >> 1) * 2) * 1024 + (r + 32 * (c & 1)) * 16;
                *reinterpret_cast<u32x4*>(dst) = (u32x4){hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4*>(dst + 1024) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
            }
    }
    TEAM_STAMP(3);
    __syncthreads();
    TEAM_STAMP(4);

    // ---- MFMAs: (x hi, w lo), (x lo, w hi), (x hi, w hi) per k-step, like the big GEMMs
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (busy) {
#pragma unroll
        for (int i = 0; i < WKS; ++i)
            if (i < ks_n) {
                const unsigned char* xs = lds + ((ks0 + i) * 2) * 1024 + lane * 16;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(xs), xl = *reinterpret_cast<const f16x8*>(xs + 1024);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, __builtin_bit_cast(f16x8, wfr[i][1]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, __builtin_bit_cast(f16x8, wfr[i][0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, __builtin_bit_cast(f16x8, wfr[i][0]), acc, 0, 0, 0);
            }
    }
    TEAM_STAMP(5);
    // ---- k-split partials through LDS; every wave of a fragment finishes 16 / ksplit of its registers
    float* red = reinterpret_cast<float*>(lds + XS_BYTES);
    if (ksplit > 1) {
        if (busy) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
    }
    if (busy) {
        const int l31 = lane & 31, lh = lane >> 5;
        const int n = cf * 32 + l31;
        const float cd = s.wdown[n], bv = s.bias ? s.bias[n] : 0.f;
        const int ne = 16 / ksplit, e0 = kq * ne;
        const int wbase = wave - kq;                                   // first wave of this fragment
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < ne) {
                const int e = e0 + i;
                float v;
                if (ksplit == 1) v = acc[e];
                else {
                    v = 0.f;
                    for (int q = 0; q < ksplit; ++q) v += red[((wbase + q) * 16 + e) * 64 + lane];
                }
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const long long gr = row0 + r;
                v = fmaf(v, rowdown[r] * cd, bv);
                if (n >= s.relu_from) v = fmaxf(v, 0.f);
                if (gr < M) {
                    if (s.res) v += s.res_exchange ? xload1<FAST>(s.res + gr * s.N + n) : s.res[gr * s.N + n];
                    float* o = s.Y + gr * s.N + n;
                    if (s.y_exchange) xstore1<FAST>(o, v);
                    else *o = v;
                }
            }
    }
    TEAM_STAMP(6);
    if (s.y_exchange) team_arrive(team);
    else __syncthreads();                                              // the LDS images are rewritten by the next stage
    TEAM_STAMP(7);
}

struct ProtoArgs {
    StageArgs st[4];
    int n_stages;
    long long M;
    float eps;
    unsigned* counters;      // [teams][2], zero
    unsigned* error;
    int n_teams;
    int force_safe;
    long long* trace;        // 2 + 8 per stage stamps of workgroup 0, or null
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ __launch_bounds__(256) void row_team_kernel(const ProtoArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // block -> (team, slice): team t lives on XCD t % 8 (block b runs on XCD b % 8 under round-robin dispatch): its members are the
    // blocks ((t / 8) * 8 + slice) * 8 + t % 8 -- a speed arrangement only, verified below before the fast path is taken
    const int b = blockIdx.x;
    const int xcd = b & 7, li = b >> 3;
    const int team_id = (li >> 3) * 8 + xcd, slice = li & 7;
    if (team_id >= a.n_teams) return;
    Team team{a.trace, 0, a.counters + 4 * team_id, a.error, 0u};
    TEAM_STAMP(-1);
    const int row0 = team_id * ROWS;
    // placement check: every member publishes its XCC id; one barrier later all of them know whether the team shares an L2
    bool fast = false;
    {
        unsigned* ids = a.counters + 4 * team_id + 2;                   // OR of (1 << xcc id) over the members
        if (threadIdx.x == 0) __hip_atomic_fetch_or(ids, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        team_arrive(team);
        team_wait(team);
        __shared__ unsigned mask_s;
        if (threadIdx.x == 0) mask_s = __hip_atomic_load(ids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned m = mask_s;
        fast = !a.force_safe && (m & (m - 1)) == 0;                     // one bit: one XCD
    }
    TEAM_STAMP(-2);
    for (int i = 0; i < a.n_stages; ++i) {
        const bool wait = i > 0 && a.st[i - 1].y_exchange;
        if (fast) run_stage<true>(a.st[i], a.M, row0, slice, a.eps, lds, team, wait);
        else run_stage<false>(a.st[i], a.M, row0, slice, a.eps, lds, team, wait);
    }
    // leave: the last member resets the team's words (arrivals, departures, XCC mask) for the next launch
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(team.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == TEAM - 1) {
            __hip_atomic_store(team.counter + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(team.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(team.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

// Prototype entry (tools/exp/r4_team_proto.py): ffn.0 (ReLU) -> ffn.1 + residual -> norm3 -> [cls_branch.0 | reg_branch.0 (ReLU)], the
// middle of the decoder layer's tail (models/sparsebev_transformer.py:175-181), for M rows of x2.  Weights as sbev_pack_f16s_frags
// images with their per-row down-scales.  counters: [ceil(M / 32)][4] zeroed words (self-resetting); error: one zeroed word.
extern "C" int sbev_row_team_proto(const float* x2, const uint16_t* w0f, const float* w0down, const float* b0, const uint16_t* w1f,
                                   const float* w1down, const float* b1, const float* g3, const float* be3, const uint16_t* w2f,
                                   const float* w2down, const float* b2, float* h, float* u, float* x3, float* y, uint32_t* counters,
                                   uint32_t* error, int64_t M, int force_safe, long long* trace, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 1 && M <= 32 * 32 * 8, "sbev_row_team_proto: 1 .. 8192 rows");
    SBEV_REQUIRE(x2 && w0f && w1f && w2f && h && u && x3 && y && counters && error, "sbev_row_team_proto: null pointer");
    ProtoArgs a{};
    a.M = M;
    a.eps = 1e-5f;
    a.counters = counters;
    a.error = error;
    a.n_teams = (int)((M + ROWS - 1) / ROWS);
    a.force_safe = force_safe;
    a.trace = trace;
    a.n_stages = 3;
    a.st[0] = StageArgs{x2, 256, 0, nullptr, nullptr, 0, nullptr, w0f, w0down, b0, 512, 0, nullptr, 0, h, 1};
    a.st[1] = StageArgs{h, 512, 1, nullptr, nullptr, 0, nullptr, w1f, w1down, b1, 256, 256, x2, 0, u, 1};
    a.st[2] = StageArgs{u, 256, 1, g3, be3, 0, x3, w2f, w2down, b2, 512, 256, nullptr, 0, y, 0};
    const int blocks = ((a.n_teams + 7) / 8) * 64;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(row_team_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
        sbev::set_error("sbev_row_team_proto: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e));
        return SBEV_ELAUNCH;
    }
    hipLaunchKernelGGL(row_team_kernel, dim3((unsigned)blocks), dim3(256), LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_row_team_proto");
}
