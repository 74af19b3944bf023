// Host-side decoder runtime (C++): enqueues the whole 6-layer SparseBEV decoder forward -- every kernel of
// every layer -- on one HIP stream from ONE C-ABI call, so no interpreter sits between launches.
//
// Replaces the Python control flow of SparseBEVTransformerDecoder.forward / SparseBEVTransformerDecoderLayer.
// forward (models/sparsebev_transformer.py:56-101,162-193) for inference.  All buffers come from a caller-
// provided workspace (size: sbev_decoder_workspace_bytes); nothing is allocated, nothing synchronises.
// The launch sequence is static for a given config, which also makes it capturable into a hipGraph by the
// caller (the stream may be in capture mode: no call in here is capture-illegal).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include "sbev_common.hpp"

#include <mutex>
#include <vector>

namespace {

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    float* take(size_t n_floats) {
        float* r = reinterpret_cast<float*>(base + off);
        off += (n_floats * sizeof(float) + 255) / 256 * 256;
        return r;
    }
};

struct Buffers {
    float *t0, *t1, *x, *x1, *x2, *x3, *qkvt, *att, *so, *wbp, *loc, *sampled, *params, *mixed, *slabs,
        *h, *c0, *c1, *r0, *r1, *reg, *bbox, *x1s, *xsc, *pair_x, *folded;
    uint32_t* pair_sync;
    int32_t* order;
    uint32_t *touch_need, *touch_done;      // on-demand relayout (sbev_decoder_forward_lazy): one word per feature tile each, or null
    size_t bytes;
};

// tiles of the config's dense pyramid [B*T*N, D, H_l * W_l]; false: the lazy relayout does not take it (ring, odd sizes, D != 256)
bool lazy_plan_of(const sbev_decoder_config& c, sbev::LazyPlan* p) {
    if (c.n_slots > 0 || c.G != 4 || c.D != 256) return false;
    int32_t s[SBEV_MAX_LEVELS];
    for (int l = 0; l < c.L; ++l) s[l] = c.hw[l][0] * c.hw[l][1];
    return sbev::lazy_plan(c.L, s, (long long)c.B * c.T * c.N, c.D, p);
}

int out_proj_splits(long long M, int N, int K) { return sbev_linear_splitk_plan(M, N, K); }

Buffers carve(const sbev_decoder_config& c, void* ws) {
    Carver k(ws);
    const size_t BQ = (size_t)c.B * c.Q, D = c.D;
    const int Cg = c.D / c.G, Pin = c.T * c.P;
    const size_t pgN = (size_t)c.G * (Cg * Cg + Pin * c.out_points);
    const size_t mixN = (size_t)c.G * c.out_points * Cg;
    Buffers b{};
    b.t0 = k.take(BQ * D); b.t1 = k.take(BQ * D);
    b.x = k.take(BQ * D); b.x1 = k.take(BQ * D); b.x2 = k.take(BQ * D); b.x3 = k.take(BQ * D);
    b.qkvt = k.take(BQ * (size_t)c.attn_in_rows);
    b.att = k.take(BQ * D);
    b.so = k.take(BQ * (size_t)(c.G * c.P * (3 + c.L)));
    b.wbp = k.take(BQ * c.T * c.G * c.P * c.L);
    b.loc = k.take(BQ * c.T * c.G * c.P * 3);
    b.sampled = k.take(BQ * c.G * Pin * Cg);
    b.params = k.take(BQ * pgN);
    b.mixed = k.take(BQ * mixN);
    {   // split-K slabs of the out-projection: the larger of the exact and the split-bf16 plans (the mode is a per-call choice)
        int sl = out_proj_splits(BQ, c.D, (int)mixN);
        const int sl2 = sbev_linear_bf16s_out_plan((int64_t)BQ, c.D, (int)mixN);
        sl = sl2 > sl ? sl2 : sl;
        b.slabs = k.take((size_t)sl * BQ * D);
    }
    b.h = k.take(BQ * c.ffn);
    b.c0 = k.take(BQ * D); b.c1 = k.take(BQ * D); b.r0 = k.take(BQ * D); b.r1 = k.take(BQ * D);
    b.reg = k.take(BQ * c.code_size);
    b.bbox = k.take(BQ * 10);
    b.x1s = k.take(2 * BQ * D + 64 * (size_t)D);   // x1 as bf16 image fragments (<= 3 images of 2 bytes, rows padded to 32): the
                                                   // generator's operand in the split-bf16 / fp16 modes
    b.xsc = k.take(64);                            // fp16 modes: {2^e, 2^-e} of x1 (written by the pack launch of every layer)
    b.pair_x = k.take((size_t)sbev::chain_pair_floats((long long)BQ));                   // tail chain in pair mode (row_chain.hip): exchange rows
    b.pair_sync = reinterpret_cast<uint32_t*>(k.take((size_t)sbev::chain_pair_sync_words((long long)BQ)));      // ... and arrival counters
    b.order = reinterpret_cast<int32_t*>(k.take(BQ));                                    // launch order of the gather items (sbev_query_order)
    b.folded = k.take(BQ * D);                                                           // the out-projection's slabs folded inside its launch (gemm_bf16s.hip)
    sbev::LazyPlan lp;
    if (lazy_plan_of(c, &lp)) {                                                          // (45 KB at config 2, 1.4 MB at config 4)
        b.touch_need = reinterpret_cast<uint32_t*>(k.take(lp.base[lp.n_levels]));
        b.touch_done = reinterpret_cast<uint32_t*>(k.take(lp.base[lp.n_levels]));
    }
    b.bytes = k.off;
    return b;
}

// gather + mixing in one launch (sbev_sample_mix_f32) where supported; sbev_decoder_fuse_sample_mix(0) restores the two
// launches (A/B measurements; results are bit-identical)
std::atomic<int> g_fuse_sample_mix{1};
// 5 fp32 levels: the fused instantiation needed 168 registers + spills for 3 waves per SIMD in round 2 (-1.8 % at config 4) and
// kept the two launches; with the lean chunk code (msmv_chunk.inc, round 3) it fits 168 without a spill.  SBEV_NO_FUSE_L5F32=1
// restores the two launches (A/B).
std::atomic<int> g_fuse_l5_f32{getenv("SBEV_NO_FUSE_L5F32") ? 0 : 1};

// the fused gather + mixing launch walks its items in the order of sbev_query_order (one group and one arc of the camera ring per XCD):
// 20 % fewer fabric reads for the launch at config 2 (PMC: 290 -> 232 MB, L2 hit 0.39 -> 0.47; tools/sampler_footprint.py predicts
// it) and NOT faster -- the launch is bound by a workgroup's chain of memory latencies at 4 workgroups per CU, not by fabric bytes
// (DESIGN_HISTORY.md section 10.8) -- and the sort is one more launch per layer: OFF by default, kept for A/B and for a chip whose HBM is
// shared.  sbev_decoder_query_order(1) / SBEV_QUERY_ORDER=1 switches it on (bit-identical results).
std::atomic<int> g_query_order{getenv("SBEV_QUERY_ORDER") ? (atoi(getenv("SBEV_QUERY_ORDER")) == 2 ? 2 : atoi(getenv("SBEV_QUERY_ORDER")) != 0) : 0};

// the row-local op chains of a layer as three launches (row_chain.hip) when the caller supplied packed weights
// (sbev_decoder_weights.chain_pack); sbev_decoder_row_chain(0) restores the op-by-op launches (A/B measurements)
std::atomic<int> g_row_chain{1};
// on-demand relayout: the scans of layers 1.. as launches of their own instead of riding in the generator GEMM's prologue (A/B)
std::atomic<int> g_lazy_scan_launch{getenv("SBEV_LAZY_SCAN_LAUNCH") ? 1 : 0};

int validate(const sbev_decoder_config* c) {
    SBEV_REQUIRE(c != nullptr, "sbev_decoder: null config");
    SBEV_REQUIRE(c->B >= 1 && c->Q >= 1 && c->T >= 1 && c->N >= 1 && c->G >= 1 && c->P >= 1, "sbev_decoder: bad sizes");
    SBEV_REQUIRE(c->L >= 1 && c->L <= SBEV_MAX_LEVELS, "sbev_decoder: num_levels %d", c->L);
    SBEV_REQUIRE(c->D % (4 * c->G) == 0 && c->D % c->H == 0 && c->D / c->H == 32, "sbev_decoder: embed_dims %d / heads %d (head_dim must be 32)", c->D, c->H);
    SBEV_REQUIRE(c->D / c->G == 64 && c->out_points == 128, "sbev_decoder: built for 64 channels per group and 128 out points");
    SBEV_REQUIRE(c->attn_in_rows >= 3 * c->D + c->H && c->attn_in_rows % 4 == 0, "sbev_decoder: attn_in_rows %d", c->attn_in_rows);
    // every box kernel (sasa, sampling_front, refine, linear3) reads query_bbox rows with a stride of 10 floats
    SBEV_REQUIRE(c->code_size == 10, "sbev_decoder: code_size %d (the box kernels are built for the 10-wide box code)", c->code_size);
    SBEV_REQUIRE(c->num_layers >= 1 && c->num_classes >= 1 && c->ffn % 4 == 0, "sbev_decoder: head sizes");
    SBEV_REQUIRE(c->gemm_mode >= SBEV_GEMM_F32 && c->gemm_mode <= SBEV_GEMM_F16X4, "sbev_decoder: gemm_mode %d", c->gemm_mode);
    return SBEV_OK;
}

// Second stream + events for the two independent sub-chains of a layer (created once per process, lazily):
//   * the parameter-generator GEMM (MFMA-bound) only needs x1, so it runs beside the sampling chain
//     (Linear -> sample points -> projection -> gather: memory/latency-bound);
//   * the classification branch only feeds the output, so it runs beside the regression branch / next layer.
struct Aux {
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    bool ok = false;
};
Aux& aux() {
    static Aux a;
    static std::once_flag once;
    std::call_once(once, [] {
        bool ok = hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking) == hipSuccess;
        for (auto& e : a.ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        a.ok = ok;
    });
    return a;
}
inline int hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return SBEV_OK;
    sbev::set_error("%s: %s", what, hipGetErrorString(e));
    return SBEV_ELAUNCH;
}

#define TRY(expr)                 \
    do {                          \
        int st__ = (expr);        \
        if (st__ != SBEV_OK) return st__; \
    } while (0)

}  // namespace

namespace sbev {
struct ProfCallScope {      // see sbev_profile_stride below
    bool prev;
    ProfCallScope();
    ~ProfCallScope();
};
}  // namespace sbev

// the fused gather + mixing kernel covers this config: shape (sbev_sample_mix_supported), the 5-level fp32 switch, and every level's
// per-(sample, frame) NHWC slab below 2 GiB (the fused kernel's taps are 31-bit buffer offsets: sbev_sample_mix_slabs_ok)
static bool sample_mix_fusable(const sbev_decoder_config& c) {
    if (sbev_sample_mix_supported(c.L, c.D / c.G, c.P, c.T, c.G, c.G) == 0) return false;
    if (c.L == 5 && c.feat_dtype == SBEV_F32 && g_fuse_l5_f32.load(std::memory_order_relaxed) == 0) return false;
    int32_t hw[2 * SBEV_MAX_LEVELS];
    int64_t sv[SBEV_MAX_LEVELS];
    for (int l = 0; l < c.L; ++l) {
        hw[2 * l] = c.hw[l][0];
        hw[2 * l + 1] = c.hw[l][1];
        sv[l] = (int64_t)c.hw[l][0] * c.hw[l][1] * c.D;
    }
    return sbev_sample_mix_slabs_ok(hw, c.L, c.feat_dtype, c.N, c.D / c.G, sv, c.D) != 0;
}

// Kernel launches sbev_decoder_forward enqueues per layer for this config / weight set under the current switches (what
// bench.py reports; the decision code is the forward's own)
extern "C" int sbev_decoder_launches_per_layer(const sbev_decoder_config* cfg, const sbev_decoder_weights* w) {
    if (validate(cfg) != SBEV_OK || !w) return -1;
    const sbev_decoder_config& c = *cfg;
    const int64_t BQ = (int64_t)c.B * c.Q;
    const bool fork = c.overlap != 0;
    const bool chain = g_row_chain.load(std::memory_order_relaxed) != 0 && w->chain_pack != nullptr && !fork && sbev::row_chain_supported(c) &&
                       sbev::row_chain_pays(BQ);
    const bool fused = g_fuse_sample_mix.load(std::memory_order_relaxed) != 0 &&
                       sample_mix_fusable(c);
    const int split = (c.gemm_mode == SBEV_GEMM_BF16X6 || c.gemm_mode == SBEV_GEMM_BF16X3S) ? 1      // x1 -> bf16 image fragments
                      : (c.gemm_mode == SBEV_GEMM_F16X3 || c.gemm_mode == SBEV_GEMM_F16X4) ? (chain ? 0 : 1)      // x1 -> fp16 image fragments (the attention chain writes them)
                      : (c.gemm_mode == SBEV_GEMM_BF16X3 && sbev_linear_bf16x3_strip_ok(BQ, c.G * ((c.D / c.G) * (c.D / c.G) + c.T * c.P * c.out_points), c.D)) ? 1 : 0;
    // chains: attention, attention chain, generator, gather + mixing, out-projection, tail (+ next front)
    // op by op: 17 with the fused gather + mixing (DESIGN_HISTORY.md section 4)
    const bool ordered = chain && fused && g_query_order.load(std::memory_order_relaxed) == 1 && c.Q <= sbev_query_order_max();      // (mode 2: one sort per STEP)
    return (chain ? 6 : 17) + (fused ? 0 : 1) + split + (ordered ? 1 : 0);
}

// fp16 modes: the out-projection's input -- relu(LayerNorm without affine over n = out_points * C / G elements), so |x| <= sqrt(n - 1)
// -- is multiplied by 2^e before its fp16 split: the largest e with sqrt(n) 2^e < 65504 (n = 8192: e = 9)
extern "C" int sbev_decoder_mixed_up_log2(const sbev_decoder_config* cfg) {
    if (!cfg || cfg->G < 1 || cfg->out_points < 1 || cfg->D < cfg->G) return 0;
    const double bound = std::sqrt((double)cfg->out_points * (cfg->D / cfg->G));
    int e = 0;
    while (std::ldexp(bound, e + 1) < 65504.0 && e < 15) ++e;
    return e;
}

extern "C" int64_t sbev_decoder_workspace_bytes(const sbev_decoder_config* cfg) {
    if (validate(cfg) != SBEV_OK) return -1;
    sbev::chain_pair_prepare();            // (the one call every user makes before a forward or a capture)
    return (int64_t)carve(*cfg, nullptr).bytes;
}

static int decoder_forward_impl(const sbev_decoder_config* cfg, const sbev_decoder_weights* w,
                                const void* const* feats_nhwc, const float* query_bbox, const float* query_feat,
                                const float* time_diff, const float* lidar2img, const float* vel_div,
                                const uint8_t* attn_mask, float* cls_out, float* bbox_out,
                                void* workspace, int64_t workspace_bytes, sbev_stream_t stream, const sbev_lazy_feats* lazy) {
    TRY(validate(cfg));
    const sbev_decoder_config& c = *cfg;
    SBEV_REQUIRE(w && feats_nhwc && query_bbox && query_feat && time_diff && lidar2img && cls_out && bbox_out && workspace,
                 "sbev_decoder_forward: null pointer");
    if (const unsigned faults = sbev::chain_pair_faults_pending()) {
        // an earlier step's pair tail lost a partner (the GPU was shared / preempted for ~1 s): that step's rows are wrong.  Sticky
        // until acknowledged; pair mode goes off so that the repeated step cannot fault again.
        sbev_decoder_chain_pair(0);
        sbev::set_error("sbev_decoder_forward: %u pair-mode hand-off(s) of an EARLIER decoder step timed out -- that step's outputs are invalid; "
                        "pair mode is now off: acknowledge with sbev_decoder_chain_pair_faults_ack() and repeat the step", faults);
        return SBEV_EFAULT;
    }
    const sbev::ProfCallScope prof_scope;     // with sbev_profile_stride(n): only every n-th call's launches are bracketed
    SBEV_REQUIRE((((uintptr_t)workspace) & 255) == 0, "sbev_decoder_forward: workspace must be 256-byte aligned");
    SBEV_REQUIRE(cfg->gemm_mode != SBEV_GEMM_BF16X3 || (w->pg_w2 && w->op_w2), "sbev_decoder_forward: gemm_mode bf16x3 needs pg_w2 / op_w2");
    // split-bf16 / fp16 kernels (gemm_bf16s.hip): their mode code 2 = bf16x3s, 3 = bf16x6, 4 = f16x3, 5 = f16x4
    const int nimg = cfg->gemm_mode == SBEV_GEMM_BF16X6 ? 3 : cfg->gemm_mode == SBEV_GEMM_BF16X3S ? 2 : cfg->gemm_mode == SBEV_GEMM_F16X3 ? 4
                     : cfg->gemm_mode == SBEV_GEMM_F16X4 ? 5 : 0;
    SBEV_REQUIRE(nimg == 0 || (w->pg_ws && w->op_wp), "sbev_decoder_forward: gemm_mode %d needs pg_ws / op_wp", cfg->gemm_mode);
    SBEV_REQUIRE(nimg < 4 || (w->pg_wdown && w->op_nscale && w->pg_xscale), "sbev_decoder_forward: gemm_mode %d needs pg_wdown / op_nscale / pg_xscale", cfg->gemm_mode);
    const int mixed_up = sbev_decoder_mixed_up_log2(cfg);
    const Buffers b = carve(c, workspace);
    SBEV_REQUIRE((int64_t)b.bytes <= workspace_bytes, "sbev_decoder_forward: workspace too small (%lld < %zu)", (long long)workspace_bytes, b.bytes);

    const int64_t BQ = (int64_t)c.B * c.Q;
    const int D = c.D, Cg = c.D / c.G, Pin = c.T * c.P;
    const int pgN = c.G * (Cg * Cg + Pin * c.out_points);
    const int mixN = c.G * c.out_points * Cg;
    const int soN = c.G * c.P * (3 + c.L);
    const int splits = out_proj_splits(BQ, D, mixN);
    const float eps = 1e-5f;

    // on-demand relayout (sbev_decoder_forward_lazy): feats_nhwc are DESTINATIONS; every layer's point selection marks the units its
    // points read and one launch behind it moves the marked units that this step has not moved yet (csrc/layout.hip)
    sbev::LazyPlan lplan{};
    if (lazy) {
        SBEV_REQUIRE(lazy_plan_of(c, &lplan) && b.touch_need, "sbev_decoder_forward_lazy: config not covered (dense pyramid, 4 groups of 64 channels)");
        for (int l = 0; l < c.L; ++l)
            SBEV_REQUIRE(feats_nhwc[l] && (((uintptr_t)feats_nhwc[l]) & 15) == 0 &&
                         (lazy->table ? lazy->index[l] >= 0 : (lazy->src[l] && (((uintptr_t)lazy->src[l]) & 15) == 0 && lazy->src[l] != feats_nhwc[l])),
                         "sbev_decoder_forward_lazy: level %d (16-byte aligned NCHW source and NHWC destination)", l);
        SBEV_REQUIRE(!lazy->table || (((uintptr_t)lazy->table) & 7) == 0, "sbev_decoder_forward_lazy: unaligned pointer table");
    }
    auto lazy_move = [&](int layer) -> int {
        if (!lazy) return SBEV_OK;
        return sbev::launch_lazy_relayout(lplan, lazy->table, lazy->index, lazy->src, const_cast<void* const*>(feats_nhwc), c.feat_dtype == SBEV_F32 ? 4 : 2,
                                          b.touch_need, b.touch_done, layer == 0, layer + 1 == c.num_layers, reinterpret_cast<hipStream_t>(stream));
    };

    // feature pyramid descriptors: zero-copy NHWC, group g = channel slice [g*Cg, (g+1)*Cg)
    int32_t hw[2 * SBEV_MAX_LEVELS];
    int64_t sbo[SBEV_MAX_LEVELS], sv[SBEV_MAX_LEVELS];
    for (int l = 0; l < c.L; ++l) {
        hw[2 * l] = c.hw[l][0];
        hw[2 * l + 1] = c.hw[l][1];
        sv[l] = (int64_t)c.hw[l][0] * c.hw[l][1] * D;
        sbo[l] = sv[l] * c.N;
    }

    hipStream_t s_main = reinterpret_cast<hipStream_t>(stream);
    Aux& ax = aux();
    const bool fork = c.overlap != 0 && ax.ok;        // classification branch on the aux stream
    const bool fork_pg = c.overlap == 1 && ax.ok;     // + parameter-generator GEMM beside the sampling chain
    sbev_stream_t s_aux = fork ? reinterpret_cast<sbev_stream_t>(ax.stream) : stream;
    int evi = 0;
    auto next_ev = [&]() { return ax.ev[(evi++) & 7]; };
    hipEvent_t ev_cls = nullptr;

    // parameter generator in the 3 x bf16 mode: x1 is split once per layer and streamed past W-stationary strips
    const bool pg_strip = c.gemm_mode == SBEV_GEMM_BF16X3 && sbev_linear_bf16x3_strip_ok(BQ, pgN, D) != 0;
    auto generator_bf16x3 = [&](sbev_stream_t st) -> int {
        if (!pg_strip) return sbev_linear_bf16x3(b.x1, w->pg_w2, w->pg_b, nullptr, b.params, BQ, pgN, D, D, pgN, 0, st);
        uint16_t* x2 = reinterpret_cast<uint16_t*>(b.x1s);
        int e = sbev_split_bf16x3_weights(b.x1, x2, BQ, D, st);
        if (e != SBEV_OK) return e;
        return sbev_linear_bf16x3_strip(x2, w->pg_w2, w->pg_b, b.params, BQ, pgN, D, pgN, 0, st);
    };

    SBEV_REQUIRE(nimg == 0 || (sbev_linear_bf16s_gen_ok(BQ, pgN, D) && sbev_linear_bf16s_out_ok(BQ, D, mixN)),
                 "sbev_decoder_forward: gemm_mode %d does not cover this shape (rows %lld, generator %d x %d, out-projection %d x %d)",
                 cfg->gemm_mode, (long long)BQ, pgN, D, D, mixN);
    // on-demand relayout, layers 1..5: the scan (find + move what this layer's points marked) rides in the generator GEMM's prologue
    // where that kernel is the weight-stationary one (fp16 modes) -- the only launch between the marks and the gather that does not touch
    // the features; sbev_decoder_lazy_scan_launch(1) / SBEV_LAZY_SCAN_LAUNCH=1 keeps it a launch of its own (A/B; bit-identical)
    const bool scan_own_launch = g_lazy_scan_launch.load(std::memory_order_relaxed) != 0;
    // (up to 1024 rows: measured at config 2 555 vs 541 samples/s; at 3200 / 3600 rows a layer adds tens of thousands of units and the
    // launch of its own, with one workgroup per 16 tiles, spreads them better: 1303-1321 vs 1312-1349 and 521 vs 523 -- neutral, kept apart)
    const bool scan_in_gen = lazy && nimg >= 4 && !scan_own_launch && BQ <= 1024 && sbev::linear_f16s_gen_takes_scan(BQ, pgN, D, pgN, nimg - 1);
    auto generator_bf16s = [&](sbev_stream_t st, bool packed = false, int scan_layer = -1) -> int {      // x1 -> image fragments (once per layer) -> Y = X W^T + b
        uint16_t* xs = reinterpret_cast<uint16_t*>(b.x1s);
        if (nimg >= 4 && scan_layer > 0) {                    // (fragments there already -- chain or pack launch --, this layer's scan inside)
            const sbev::LazyScan lz{&lplan, lazy->table, lazy->index, lazy->src, const_cast<void* const*>(feats_nhwc), c.feat_dtype == SBEV_F32 ? 4 : 2,
                                    b.touch_need, b.touch_done, scan_layer + 1 == c.num_layers};
            if (!packed) {
                int e = sbev_pack_f16s_frags(b.x1, D, xs, const_cast<float*>(w->pg_xscale), (int)BQ, D, 2, st);
                if (e != SBEV_OK) return e;
            }
            return sbev::linear_f16s_gen_scan(xs, w->pg_xscale, w->pg_ws, w->pg_wdown, w->pg_b, b.params, BQ, pgN, D, pgN, 0, nimg - 1, lz,
                                              reinterpret_cast<hipStream_t>(st));
        }
        if (nimg >= 4 && packed)                               // (the attention chain already wrote the fragments)
            return sbev_linear_f16s_gen(xs, w->pg_xscale, w->pg_ws, w->pg_wdown, w->pg_b, b.params, BQ, pgN, D, pgN, 0, nimg - 1, st);
        if (nimg >= 4) {                                       // fp16 hi + lo: x1 scaled by one power of two (its maximum -> [2^14, 2^15))
            // (the power of two comes with the weights: norm1's output is bounded by sqrt(D - 1) max|gamma| + max|beta| -- no pass for a maximum)
            int e = sbev_pack_f16s_frags(b.x1, D, xs, const_cast<float*>(w->pg_xscale), (int)BQ, D, 2, st);
            if (e != SBEV_OK) return e;
            return sbev_linear_f16s_gen(xs, w->pg_xscale, w->pg_ws, w->pg_wdown, w->pg_b, b.params, BQ, pgN, D, pgN, 0, nimg - 1, st);
        }
        int e = sbev_pack_bf16s_frags(b.x1, D, xs, (int)BQ, D, nimg, st);
        if (e != SBEV_OK) return e;
        return sbev_linear_bf16s_gen(xs, w->pg_ws, w->pg_b, b.params, BQ, pgN, D, pgN, 0, nimg, st);
    };

    // the mixing launches: in the fp16 GEMM modes their epilogue leaves `mixed` as (fp16 hi, fp16 lo) pairs of mixed 2^mixed_up -- the
    // out-projection's operand, split once per element where the VALU is idle instead of inside the GEMM
    auto mix_fused = [&](sbev_stream_t st, const int32_t* order = nullptr) -> int {
        if (nimg >= 4)
            return sbev_sample_mix_pairs_f16_ordered(feats_nhwc, hw, c.L, c.feat_dtype, c.B, c.N, c.Q, c.T, c.G, c.P, Cg, sbo, Cg, sv, D, b.loc, b.wbp,
                                                     c.n_slots > 0 ? c.frame_slots : nullptr, c.n_slots, b.params, b.mixed, c.out_points, eps, mixed_up,
                                                     order, st);
        return sbev_sample_mix_f32_ordered(feats_nhwc, hw, c.L, c.feat_dtype, c.B, c.N, c.Q, c.T, c.G, c.P, Cg, sbo, Cg, sv, D, b.loc, b.wbp,
                                           c.n_slots > 0 ? c.frame_slots : nullptr, c.n_slots, b.params, b.mixed, c.out_points, eps, order, st);
    };
    auto mix_plain = [&](sbev_stream_t st) -> int {
        if (nimg >= 4) return sbev_adaptive_mixing_pairs_f16(b.sampled, b.params, b.mixed, BQ, c.G, Pin, Cg, c.out_points, eps, mixed_up, st);
        return sbev_adaptive_mixing_f32(b.sampled, b.params, b.mixed, BQ, c.G, Pin, Cg, c.out_points, eps, st);
    };

    // read ONCE per call (a toggle 0 -> 2 between two layers' loads walked an unsorted b.order)
    const int query_order_mode = g_query_order.load(std::memory_order_relaxed);
    const float* bbox = query_bbox;
    const float* feat = query_feat;
    bool pe0_done = false;             // the previous layer's tail already ran this layer's first position-encoder stage
    // Row chains (row_chain.hip): 6 launches per layer instead of 17 -- everything between the out-projection GEMM and the self
    // attention, and between the self attention and the sampler, is row-local and runs with the rows in LDS.
    const bool chain = g_row_chain.load(std::memory_order_relaxed) != 0 && w->chain_pack != nullptr && !fork && sbev::row_chain_supported(c) &&
                       sbev::row_chain_pays(BQ);
    if (chain) TRY(sbev::launch_chain_front(c, *w, query_bbox, query_feat, b.x, b.qkvt, eps, s_main));
    for (int layer = 0; layer < c.num_layers; ++layer) {
        float* cls_l = cls_out + (int64_t)layer * BQ * c.num_classes;
        float* box_l = bbox_out + (int64_t)layer * BQ * c.code_size;
        if (chain) {
            const bool fused = g_fuse_sample_mix.load(std::memory_order_relaxed) != 0 &&
                               sample_mix_fusable(c);
            // launch order of this layer's gather items: sorted from the layer's input boxes (one workgroup per sample)
            const bool ordered = fused && query_order_mode != 0 && c.Q <= sbev_query_order_max();
            // (mode 2: sorted from the step's INPUT boxes only -- the refinements move a box by a fraction of its camera column)
            if (ordered && (layer == 0 || query_order_mode == 1)) TRY(sbev_query_order(bbox, c.code_size, c.pc_range, c.B, c.Q, b.order, stream));
            TRY(sbev_sasa_f32(b.qkvt, c.attn_in_rows, bbox, c.pc_range, attn_mask, b.att, c.B, c.Q, c.H, D / c.H, stream));
            // (fp16 GEMM modes: the chain also leaves x1 as the generator's fragment operand -- no pack launch)
            TRY(sbev::launch_chain_attn(c, *w, b.att, b.x, b.x1, bbox, time_diff, lidar2img, b.loc, b.wbp, eps, s_main,
                                        nimg >= 4 ? reinterpret_cast<uint16_t*>(b.x1s) : nullptr, nimg >= 4 ? w->pg_xscale : nullptr, b.pair_sync,
                                        lazy ? &lplan : nullptr, lazy ? b.touch_need : nullptr));
            const bool ride = scan_in_gen && layer > 0;
            if (!ride) TRY(lazy_move(layer));      // (layer 0's move behind the generator instead of in front of it: measured equal, 537-539 both ways)
            if (nimg)
                TRY(generator_bf16s(stream, true, ride ? layer : -1));
            else if (c.gemm_mode == SBEV_GEMM_BF16X3)
                TRY(generator_bf16x3(stream));
            else
                TRY(sbev_linear_f32(b.x1, w->pg_w, w->pg_b, nullptr, b.params, BQ, pgN, D, D, D, pgN, 0, stream));
            if (fused) {
                TRY(mix_fused(stream, ordered ? b.order : nullptr));
            } else {
                if (c.n_slots > 0)
                    TRY(sbev_msmv_fwd_ring(feats_nhwc, hw, c.L, c.feat_dtype, (int64_t)c.B * c.T * c.G, c.N, Cg, c.Q, c.P,
                                           c.G, sbo, Cg, sv, D, b.loc, b.wbp, b.sampled, SBEV_OUT_MIX, c.T, c.G, c.frame_slots, c.n_slots, stream));
                else
                    TRY(sbev_msmv_fwd(feats_nhwc, hw, c.L, c.feat_dtype, (int64_t)c.B * c.T * c.G, c.N, Cg, c.Q, c.P,
                                      c.G, sbo, Cg, sv, D, b.loc, b.wbp, b.sampled, SBEV_OUT_MIX, c.T, c.G, stream));
                TRY(mix_plain(stream));
            }
            int used = 0;
            // fp16 modes: the S slabs are folded inside the out-projection launch where all its workgroups are resident at once (<= ~1000
            // rows on 256 CUs) and the fault word is there to report a row tile that never completed; the tail then reads ONE row block
            const bool fold = nimg >= 4 && sbev::chain_pair_enabled() && sbev::out_fold_ok(BQ, mixN) && sbev::chain_fault_word_ready();
            if (nimg)
                TRY(sbev::launch_splitk_slabs_bf16s(b.mixed, w->op_wp, BQ, mixN, mixN, nimg, b.slabs, &used, s_main, mixed_up, w->op_nscale, nimg >= 4, nullptr,
                                                    fold ? b.pair_sync + sbev::chain_fold_sync_offset(BQ) : nullptr, fold ? b.folded : nullptr));
            else if (c.gemm_mode == SBEV_GEMM_BF16X3)
                TRY(sbev::launch_splitk_slabs_bf16x3(b.mixed, w->op_w2, BQ, D, mixN, mixN, splits, b.slabs, &used, s_main));
            else
                TRY(sbev::launch_splitk_slabs(b.mixed, w->op_w, BQ, D, mixN, mixN, mixN, splits, b.slabs, &used, s_main));
            TRY(sbev::launch_chain_tail(c, *w, (fold && used == 1) ? b.folded : b.slabs, used, b.x1, bbox, c.T > 1 ? vel_div : nullptr, b.x3, cls_l, box_l,
                                        layer + 1 < c.num_layers, b.x, b.qkvt, eps, s_main, b.pair_x, b.pair_sync));
            bbox = box_l;
            continue;
        }
        // position encoder -> x = feat + pos                                   (sparsebev_transformer.py:166-167)
        if (!pe0_done) TRY(sbev_linear3_ln_relu_f32(bbox, layer == 0 ? 10 : c.code_size, w->pe0_w, w->pe0_b, w->pe1_g, w->pe1_b, eps, b.t0, BQ, D, stream));
        pe0_done = false;
        TRY(sbev_linear_f32(b.t0, w->pe3_w, w->pe3_b, nullptr, b.t1, BQ, D, D, D, D, D, 0, stream));
        // Three of the layer's LayerNorms run as the PROLOGUE of the small-tile Linear that consumes them
        // (sbev_ln_linear_f32: one launch instead of two, the normalised rows are stored for the other readers):
        // here the position encoder's last norm (+ ReLU, + query_feat) -> x, with the attention in-projection
        TRY(sbev_ln_linear_f32(b.t1, w->pe4_g, w->pe4_b, eps, 1, feat, b.x, w->attn_in_w, w->attn_in_b, nullptr, b.qkvt,
                               BQ, c.attn_in_rows, D, D, c.attn_in_rows, 0, stream));
        // scale-adaptive self attention                                         (:169)
        TRY(sbev_sasa_f32(b.qkvt, c.attn_in_rows, bbox, c.pc_range, attn_mask, b.att, c.B, c.Q, c.H, D / c.H, stream));
        TRY(sbev_linear_f32(b.att, w->attn_out_w, w->attn_out_b, b.x, b.t1, BQ, D, D, D, D, D, 0, stream));
        // norm1 -> x1, with the Linear of the sampling offsets / level logits      (:169-170)
        TRY(sbev_ln_linear_f32(b.t1, w->norm1_g, w->norm1_b, eps, 0, nullptr, b.x1, w->samp_w, w->samp_b, nullptr, b.so,
                               BQ, soN, D, D, soN, 0, stream));
        // fork: parameter generator (needs only x1) on the aux stream, beside the sampling chain
        hipEvent_t ev_pg = nullptr;
        if (fork_pg) {
            hipEvent_t e = next_ev();
            TRY(hip_ok(hipEventRecord(e, s_main), "hipEventRecord"));
            TRY(hip_ok(hipStreamWaitEvent(ax.stream, e, 0), "hipStreamWaitEvent"));
        }
        if (nimg)
            TRY(generator_bf16s(fork_pg ? s_aux : stream));
        else if (c.gemm_mode == SBEV_GEMM_BF16X3)
            TRY(generator_bf16x3(fork_pg ? s_aux : stream));
        else
            TRY(sbev_linear_f32(b.x1, w->pg_w, w->pg_b, nullptr, b.params, BQ, pgN, D, D, D, pgN, 0, fork_pg ? s_aux : stream));
        if (fork_pg) {
            ev_pg = next_ev();
            TRY(hip_ok(hipEventRecord(ev_pg, ax.stream), "hipEventRecord"));
        }
        // adaptive spatio-temporal sampling                                     (:170)
        TRY(sbev::launch_sample_and_project(bbox, b.so, soN, b.so + c.G * c.P * 3, soN, time_diff, lidar2img, c.pc_range,
                                            c.B, c.Q, c.T, c.N, c.G, c.P, c.L, c.image_h, c.image_w, c.eps_homo, b.loc, b.wbp,
                                            lazy ? &lplan : nullptr, lazy ? b.touch_need : nullptr, c.hw, s_main));
        TRY(lazy_move(layer));
        // gather + adaptive mixing: ONE launch when the fused kernel covers the shape (the sampled features then never
        // touch HBM), else the sampler followed by the mixing kernel (same arithmetic, bit-identical results)
        // (round 2 kept two launches for 5 fp32 levels: 168 registers + spills, 272 vs 277 samples/s at config 4; the lean chunk code
        // of round 3 fits without spills -- g_fuse_l5_f32; 4 fp32 levels +1.6 % at config 2, 5 bf16 levels +4.3 % at config 5)
        const bool fused = g_fuse_sample_mix.load(std::memory_order_relaxed) != 0 &&
                           sample_mix_fusable(c);
        if (!fused) {
            if (c.n_slots > 0)
                TRY(sbev_msmv_fwd_ring(feats_nhwc, hw, c.L, c.feat_dtype, (int64_t)c.B * c.T * c.G, c.N, Cg, c.Q, c.P,
                                       c.G, sbo, Cg, sv, D, b.loc, b.wbp, b.sampled, SBEV_OUT_MIX, c.T, c.G, c.frame_slots, c.n_slots, stream));
            else
                TRY(sbev_msmv_fwd(feats_nhwc, hw, c.L, c.feat_dtype, (int64_t)c.B * c.T * c.G, c.N, Cg, c.Q, c.P,
                                  c.G, sbo, Cg, sv, D, b.loc, b.wbp, b.sampled, SBEV_OUT_MIX, c.T, c.G, stream));
        }
        // adaptive mixing + norm2 (join: the generator's output is needed now)  (:171)
        if (fork_pg) TRY(hip_ok(hipStreamWaitEvent(s_main, ev_pg, 0), "hipStreamWaitEvent"));
        if (fused)
            TRY(mix_fused(stream));
        else
            TRY(mix_plain(stream));
        if (nimg >= 4)
            TRY(sbev_linear_splitk_f16s(b.mixed, 1, mixed_up, w->op_wp, w->op_nscale, w->op_b, b.x1, w->norm2_g, w->norm2_b, eps, b.x2, BQ, D, mixN, mixN,
                                        0, nimg - 1, b.slabs, stream));
        else if (nimg)
            TRY(sbev_linear_splitk_bf16s(b.mixed, w->op_wp, w->op_b, b.x1, w->norm2_g, w->norm2_b, eps, b.x2, BQ, D, mixN, mixN,
                                         0, nimg, b.slabs, stream));
        else if (c.gemm_mode == SBEV_GEMM_BF16X3)
            TRY(sbev_linear_splitk_bf16x3(b.mixed, w->op_w2, w->op_b, b.x1, w->norm2_g, w->norm2_b, eps, b.x2, BQ, D, mixN, mixN,
                                          0, splits, b.slabs, stream));
        else
            TRY(sbev_linear_splitk_f32(b.mixed, w->op_w, w->op_b, b.x1, w->norm2_g, w->norm2_b, eps, b.x2, BQ, D, mixN, mixN, mixN,
                                       0, splits, b.slabs, stream));
        // (the previous layer's classification branch still reads x3 on the aux stream: join before x3 is rewritten)
        if (fork && ev_cls) TRY(hip_ok(hipStreamWaitEvent(s_main, ev_cls, 0), "hipStreamWaitEvent"));
        // FFN + norm3                                                           (:172)
        TRY(sbev_linear_f32(b.x2, w->ffn0_w, w->ffn0_b, nullptr, b.h, BQ, c.ffn, D, D, D, c.ffn, 1, stream));
        TRY(sbev_linear_f32(b.h, w->ffn1_w, w->ffn1_b, b.x2, b.t1, BQ, D, c.ffn, c.ffn, c.ffn, D, 0, stream));
        // norm3 -> x3 is the prologue of the branches' first Linear (below)
        // classification branch (output only) on the aux stream; regression branch + box refinement on the main one (:174-183)
        // the two branches are independent chains of small linears: each level of them shares one grouped launch
        // (with `fork` the classification branch goes to the aux stream instead)
        auto prob = [&](const float* X, const float* W, const float* bias, float* Y, int N, int relu) {
            return sbev_linear_problem{X, W, bias, nullptr, Y, BQ, N, D, D, D, N, relu};
        };
        // only while sbev_linear_f32 would pick the same small-tile kernel for each of them (keeps the results identical
        // to the op-by-op path); large batches have enough tiles per linear anyway
        const bool grouped = !fork && (D == 256 || D == 512) && ((BQ + 127) / 128) * ((D + 127) / 128) < 256;
        const sbev::LnPrologue norm3{w->norm3_g, w->norm3_b, eps, 0, nullptr, b.x3};
        if (grouped) {
            // 5 launches for norm3 + the 9 ops of the two branches + refine (+ the next layer's first position-encoder stage):
            // ops that do not depend on each other share a launch (gemm.hip: group / pair kernels, same arithmetic as alone)
            const sbev_linear_problem g1[2] = {prob(b.t1, w->cls0_w, w->cls0_b, b.c0, D, 0), prob(b.t1, w->reg0_w, w->reg0_b, b.r0, D, 1)};
            if (sbev::ln_linear_fusable(BQ, D, D)) {
                TRY(sbev::launch_linear_group(g1, 2, &norm3, s_main));
            } else {
                TRY(sbev_layer_norm_f32(b.t1, w->norm3_g, w->norm3_b, eps, nullptr, b.x3, BQ, D, 0, stream));
                const sbev_linear_problem g1x[2] = {prob(b.x3, w->cls0_w, w->cls0_b, b.c0, D, 0), prob(b.x3, w->reg0_w, w->reg0_b, b.r0, D, 1)};
                TRY(sbev_linear_group_f32(g1x, 2, stream));
            }
            TRY(sbev::launch_ln_and_linear(b.c0, w->cls1_g, w->cls1_b, eps, 1, b.c1, BQ, D,
                                           b.r0, w->reg2_w, w->reg2_b, b.r1, D, D, 1, s_main));
            const sbev_linear_problem g2[2] = {prob(b.c1, w->cls3_w, w->cls3_b, b.c0, D, 0),
                                               prob(b.r1, w->reg4_w, w->reg4_b, b.reg, c.code_size, 0)};
            TRY(sbev_linear_group_f32(g2, 2, stream));
            TRY(sbev::launch_ln_and_refine(b.c0, w->cls4_g, w->cls4_b, eps, 1, b.c1, BQ, D,
                                           bbox, b.reg, c.T > 1 ? vel_div : nullptr, box_l, c.Q, c.code_size, s_main));
            if (layer + 1 < c.num_layers) {   // the next layer's Linear(3->D)+LayerNorm+ReLU only needs box_l
                TRY(sbev::launch_linear_and_lin3(b.c1, w->cls6_w, w->cls6_b, cls_l, BQ, c.num_classes, D, 0,
                                                 box_l, c.code_size, w->pe0_w, w->pe0_b, w->pe1_g, w->pe1_b, eps, b.t0, D, s_main));
                pe0_done = true;
            } else {
                TRY(sbev_linear_f32(b.c1, w->cls6_w, w->cls6_b, nullptr, cls_l, BQ, c.num_classes, D, D, D, c.num_classes, 0, stream));
            }
        } else {   // s_aux == stream unless forked
            // norm3 + the classification branch's first Linear on the main stream (x3 is read by both branches) ...
            TRY(sbev_ln_linear_f32(b.t1, w->norm3_g, w->norm3_b, eps, 0, nullptr, b.x3, w->cls0_w, w->cls0_b, nullptr, b.c0,
                                   BQ, D, D, D, D, 0, stream));
            if (fork) {   // ... then the rest of it aside
                hipEvent_t e = next_ev();
                TRY(hip_ok(hipEventRecord(e, s_main), "hipEventRecord"));
                TRY(hip_ok(hipStreamWaitEvent(ax.stream, e, 0), "hipStreamWaitEvent"));
            }
            TRY(sbev_layer_norm_f32(b.c0, w->cls1_g, w->cls1_b, eps, nullptr, b.c1, BQ, D, 1, s_aux));
            TRY(sbev_linear_f32(b.c1, w->cls3_w, w->cls3_b, nullptr, b.c0, BQ, D, D, D, D, D, 0, s_aux));
            TRY(sbev_layer_norm_f32(b.c0, w->cls4_g, w->cls4_b, eps, nullptr, b.c1, BQ, D, 1, s_aux));
            TRY(sbev_linear_f32(b.c1, w->cls6_w, w->cls6_b, nullptr, cls_l, BQ, c.num_classes, D, D, D, c.num_classes, 0, s_aux));
            if (fork) {
                ev_cls = next_ev();
                TRY(hip_ok(hipEventRecord(ev_cls, ax.stream), "hipEventRecord"));
            }
            TRY(sbev_linear_f32(b.x3, w->reg0_w, w->reg0_b, nullptr, b.r0, BQ, D, D, D, D, D, 1, stream));
            TRY(sbev_linear_f32(b.r0, w->reg2_w, w->reg2_b, nullptr, b.r1, BQ, D, D, D, D, D, 1, stream));
            TRY(sbev_linear_f32(b.r1, w->reg4_w, w->reg4_b, nullptr, b.reg, BQ, c.code_size, D, D, D, c.code_size, 0, stream));
        }
        if (!grouped) TRY(sbev_refine_bbox(bbox, b.reg, c.T > 1 ? vel_div : nullptr, box_l, c.B, c.Q, c.code_size, stream));
        // next layer: query_bbox = bbox_pred.detach() (:93), query_feat = this layer's output
        bbox = box_l;
        // x3 is read by the next layer only as `feat` in its third launch and rewritten only by its norm3: no copy
        feat = b.x3;
    }
    if (fork && ev_cls) TRY(hip_ok(hipStreamWaitEvent(s_main, ev_cls, 0), "hipStreamWaitEvent"));   // final join
    return SBEV_OK;
}

extern "C" int sbev_decoder_forward(const sbev_decoder_config* cfg, const sbev_decoder_weights* w,
                                    const void* const* feats_nhwc, const float* query_bbox, const float* query_feat,
                                    const float* time_diff, const float* lidar2img, const float* vel_div,
                                    const uint8_t* attn_mask, float* cls_out, float* bbox_out,
                                    void* workspace, int64_t workspace_bytes, sbev_stream_t stream) {
    return decoder_forward_impl(cfg, w, feats_nhwc, query_bbox, query_feat, time_diff, lidar2img, vel_div, attn_mask, cls_out, bbox_out, workspace,
                                workspace_bytes, stream, nullptr);
}

// The same step from the reference's NCHW feature maps WITHOUT a dense relayout: feats_nhwc[l] are channels-last buffers of the step's
// own (any contents), `lazy` names the NCHW sources; only the 64-pixel x 64-channel units a sample point reads are moved, layer by
// layer (models/sparsebev_transformer.py:73-85 regroups everything; sparsebev_sampling.py:88-109 reads under half of it).
extern "C" int sbev_decoder_forward_lazy(const sbev_decoder_config* cfg, const sbev_decoder_weights* w, void* const* feats_nhwc,
                                         const sbev_lazy_feats* lazy, const float* query_bbox, const float* query_feat,
                                         const float* time_diff, const float* lidar2img, const float* vel_div,
                                         const uint8_t* attn_mask, float* cls_out, float* bbox_out,
                                         void* workspace, int64_t workspace_bytes, sbev_stream_t stream) {
    SBEV_REQUIRE(lazy, "sbev_decoder_forward_lazy: null source descriptor");
    return decoder_forward_impl(cfg, w, const_cast<const void* const*>(feats_nhwc), query_bbox, query_feat, time_diff, lidar2img, vel_div, attn_mask,
                                cls_out, bbox_out, workspace, workspace_bytes, stream, lazy);
}

extern "C" int sbev_decoder_lazy_supported(const sbev_decoder_config* cfg) {
    sbev::LazyPlan p;
    return cfg && validate(cfg) == SBEV_OK && lazy_plan_of(*cfg, &p) ? 1 : 0;
}

// ---- kernel launch timing (HIP events on the launch stream), used by bench.py for the roofline figures --------
// kind 0 = sampler (msmv_fwd_kernel), 1 = parameter-generator GEMM (strip kernel), 2 = out-projection GEMM (register tile)
namespace sbev {
static std::mutex g_prof_mu;
static int g_prof_mask = 0;          // bit k: bracket launches of kind k
static int g_prof_stride = 1;        // inside sbev_decoder_forward: bracket only every n-th call (the events are not free: two
static long long g_prof_calls = 0;   // records around a launch leave ~5.6 us of idle stream each, 2 % of a step for the sampler)
static thread_local bool tl_prof_skip = false;
struct ProfEvent { hipEvent_t e0, e1; int kind; };
static std::vector<ProfEvent> g_prof_events;

bool profile_begin(hipStream_t s, hipEvent_t* e0, hipEvent_t* e1, int kind) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!((g_prof_mask >> kind) & 1) || tl_prof_skip) return false;
    if (hipEventCreate(e0) != hipSuccess || hipEventCreate(e1) != hipSuccess) return false;
    (void)hipEventRecord(*e0, s);
    return true;
}
void profile_end(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int kind) {
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_events.push_back({e0, e1, kind});
}
}  // namespace sbev

// decides at the top of a decoder call whether its launches are bracketed; restores the flag on every exit path
namespace sbev {
ProfCallScope::ProfCallScope() : prev(tl_prof_skip) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_mask != 0 && g_prof_stride > 1) tl_prof_skip = (g_prof_calls++ % g_prof_stride) != 0;
}
ProfCallScope::~ProfCallScope() { tl_prof_skip = prev; }
}  // namespace sbev

extern "C" int sbev_profile_stride(int every_n_calls) {
    SBEV_REQUIRE(every_n_calls >= 1, "sbev_profile_stride: need n >= 1 (got %d)", every_n_calls);
    std::lock_guard<std::mutex> lk(sbev::g_prof_mu);
    sbev::g_prof_stride = every_n_calls;
    sbev::g_prof_calls = 0;
    return SBEV_OK;
}

// returns the previous setting
extern "C" int sbev_decoder_lazy_scan_launch(int enable) { return g_lazy_scan_launch.exchange(enable ? 1 : 0, std::memory_order_relaxed); }

extern "C" int sbev_decoder_row_chain(int enable) {
    g_row_chain.store(enable != 0, std::memory_order_relaxed);
    return SBEV_OK;
}

extern "C" int sbev_decoder_fuse_sample_mix(int enable) {
    g_fuse_sample_mix.store(enable ? 1 : 0, std::memory_order_relaxed);
    return SBEV_OK;
}

// returns the previous setting
extern "C" int sbev_decoder_query_order(int enable) {
    return g_query_order.exchange(enable == 2 ? 2 : enable ? 1 : 0, std::memory_order_relaxed);      // 1: sorted every layer; 2: once per step (layer 0)
}

extern "C" int sbev_profile_sampler(int enable) {
    std::lock_guard<std::mutex> lk(sbev::g_prof_mu);
    sbev::g_prof_mask = enable;          // 0 off, 1 sampler only, bit 1 / 2: the two mixing GEMMs
    return SBEV_OK;
}

extern "C" int sbev_profile_read(int kind, float* ms, int max_n) {
    std::lock_guard<std::mutex> lk(sbev::g_prof_mu);
    int n = 0;
    std::vector<sbev::ProfEvent> keep;
    for (auto& ev : sbev::g_prof_events) {
        if (ev.kind != kind) {
            keep.push_back(ev);
            continue;
        }
        float t = 0.f;
        if (hipEventSynchronize(ev.e1) == hipSuccess && hipEventElapsedTime(&t, ev.e0, ev.e1) == hipSuccess && ms && n < max_n)
            ms[n++] = t;
        (void)hipEventDestroy(ev.e0);
        (void)hipEventDestroy(ev.e1);
    }
    sbev::g_prof_events.swap(keep);
    return n;
}

extern "C" int sbev_profile_sampler_read(float* ms, int max_n) { return sbev_profile_read(0, ms, max_n); }

// ---- hipGraph capture of one decoder step -----------------------------------------------------------------------
// The launch sequence is static per (config, pointers): capture it once, replay it per sample.  Inputs are read
// through the captured device pointers, so the caller refreshes them in place (new queries / matrices / features,
// or -- with the frame ring -- a new slot table means a new capture per ring phase).
struct sbev_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0;
};

extern "C" int sbev_decoder_capture(const sbev_decoder_config* cfg, const sbev_decoder_weights* w,
                                    const void* const* feats_nhwc, const float* query_bbox, const float* query_feat,
                                    const float* time_diff, const float* lidar2img, const float* vel_div,
                                    const uint8_t* attn_mask, float* cls_out, float* bbox_out,
                                    void* workspace, int64_t workspace_bytes, sbev_stream_t stream, sbev_graph** out) {
    SBEV_REQUIRE(out, "sbev_decoder_capture: null output handle");
    SBEV_REQUIRE(stream, "sbev_decoder_capture: needs an explicit (non-default) stream to capture on");
    {
        std::lock_guard<std::mutex> lk(sbev::g_prof_mu);
        SBEV_REQUIRE(sbev::g_prof_mask == 0, "sbev_decoder_capture: sampler profiling is on (events cannot be read back from a captured graph)");
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    (void)aux();   // create the side stream / events outside the capture
    TRY(hip_ok(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture"));
    const int st = sbev_decoder_forward(cfg, w, feats_nhwc, query_bbox, query_feat, time_diff, lidar2img, vel_div, attn_mask,
                                        cls_out, bbox_out, workspace, workspace_bytes, stream);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);   // always end the capture, also after a failed step
    if (st != SBEV_OK) {
        if (g) (void)hipGraphDestroy(g);
        return st;
    }
    TRY(hip_ok(e, "hipStreamEndCapture"));
    sbev_graph* h = new sbev_graph();
    h->graph = g;
    (void)hipGraphGetNodes(g, nullptr, &h->nodes);
    const hipError_t ei = hipGraphInstantiate(&h->exec, g, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        (void)hipGraphDestroy(g);
        delete h;
        return hip_ok(ei, "hipGraphInstantiate");
    }
    *out = h;
    return SBEV_OK;
}

// Generic capture of ANY sequence of this library's launches (e.g. the per-level feature relayout followed by the decoder step)
extern "C" int sbev_capture_begin(sbev_stream_t stream) {
    SBEV_REQUIRE(stream, "sbev_capture_begin: needs an explicit (non-default) stream to capture on");
    {
        std::lock_guard<std::mutex> lk(sbev::g_prof_mu);
        SBEV_REQUIRE(sbev::g_prof_mask == 0, "sbev_capture_begin: launch profiling is on (events cannot be read back from a captured graph)");
    }
    (void)aux();   // create the side stream / events outside the capture
    return hip_ok(hipStreamBeginCapture(reinterpret_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
}

extern "C" int sbev_capture_end(sbev_stream_t stream, sbev_graph** out) {
    SBEV_REQUIRE(stream, "sbev_capture_end: null stream");
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(reinterpret_cast<hipStream_t>(stream), &g);   // always end the capture
    if (!out) {                                     // abort: the caller's sequence failed, drop whatever was recorded
        if (g) (void)hipGraphDestroy(g);
        return SBEV_OK;
    }
    TRY(hip_ok(e, "hipStreamEndCapture"));
    sbev_graph* h = new sbev_graph();
    h->graph = g;
    (void)hipGraphGetNodes(g, nullptr, &h->nodes);
    const hipError_t ei = hipGraphInstantiate(&h->exec, g, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        (void)hipGraphDestroy(g);
        delete h;
        return hip_ok(ei, "hipGraphInstantiate");
    }
    *out = h;
    return SBEV_OK;
}

extern "C" int sbev_graph_launch(sbev_graph* g, sbev_stream_t stream) {
    SBEV_REQUIRE(g && g->exec, "sbev_graph_launch: null graph");
    if (const unsigned faults = sbev::chain_pair_faults_pending()) {      // (see sbev_decoder_forward: sticky until acknowledged)
        sbev_decoder_chain_pair(0);
        sbev::set_error("sbev_graph_launch: %u pair-mode hand-off(s) of an EARLIER step timed out -- that step's outputs are invalid; pair mode "
                        "is now off: acknowledge with sbev_decoder_chain_pair_faults_ack(), re-capture and repeat the step", faults);
        return SBEV_EFAULT;
    }
    return hip_ok(hipGraphLaunch(g->exec, reinterpret_cast<hipStream_t>(stream)), "hipGraphLaunch");
}

extern "C" int64_t sbev_graph_num_nodes(const sbev_graph* g) { return g ? (int64_t)g->nodes : -1; }

extern "C" int sbev_graph_destroy(sbev_graph* g) {
    if (!g) return SBEV_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return SBEV_OK;
}
