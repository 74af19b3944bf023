// Shared host-side helpers of libsbev_hip.so (error slot, launch checks).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/sbev_hip.h"

namespace sbev {

void set_error(const char* fmt, ...);
int box_convention();    // SBEV_BOX_* (process-wide, like the reference's VERSION global)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SBEV_ELAUNCH;
    }
    return SBEV_OK;
}

#define SBEV_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::sbev::set_error(__VA_ARGS__);     \
            return SBEV_EINVAL;                 \
        }                                       \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// gemm_regtile.hip: the split-K partial-slab GEMM of sbev_linear_splitk_f32 for N % 128 == 0, K % 32 == 0
// (writes *slabs_written <= splits partial slabs: pairs of K splits are summed inside the kernel when splits is even)
int launch_splitk_regtile(const float* X, const float* W, float* slabs, int64_t M, int N, int K, int64_t ldx, int64_t ldw,
                          int splits, int* slabs_written, hipStream_t stream);

int regtile_plan(int64_t M, int N, int K);

// gemm.hip: the GEMM half of sbev_linear_splitk_f32 (*used partial slabs [used, M, N], not reduced)
int launch_splitk_slabs(const float* X, const float* W, int64_t M, int N, int K, int64_t ldx, int64_t ldw, int splits,
                        float* workspace, int* used, hipStream_t s);

int launch_splitk_slabs_bf16x3(const float* X, const uint16_t* W2, int64_t M, int N, int K, int64_t ldx, int splits,
                               float* workspace, int* used, hipStream_t s);

// gemm_bf16s.hip: the GEMM half of sbev_linear_splitk_bf16s (*used partial slabs [used, M, 256], not reduced)
// fold_sync + folded (pre-split fp16 operand only): the S chunk-workgroups of a row tile fold their slabs INSIDE the launch into
// `folded` [M, 256] and *used = 1 -- taken only where out_fold_ok(M, K) (every workgroup of the launch resident at once); fold_sync: one
// zeroed word per row tile (<= 64).  A row tile that never completes within the poll bound raises the decoder's fault word.
int launch_splitk_slabs_bf16s(const float* X, const uint16_t* Wp, int64_t M, int K, int64_t ldx, int nimg, float* slabs, int* used,
                              hipStream_t s, int x_up_log2 = 0, const float* nscale = nullptr, bool x_pairs = false, const float* xdev = nullptr,
                              unsigned* fold_sync = nullptr, float* folded = nullptr);
bool out_fold_ok(long long M, int K);
bool out_fold_install(void* host_word_dev);
long long out_fold_timeouts();
int out_fold_switch(int enable);
int out8_min_rows(int rows);
int out_fold_drop(int enable);

// layout.hip: on-demand relayout of the units the sample points mark (sample_point.hpp::TouchMap); need / done: one 4-byte word per tile
struct LazyPlan {
    int n_levels, R;
    long long n_images;
    int S[SBEV_MAX_LEVELS];
    unsigned tiles[SBEV_MAX_LEVELS];
    unsigned base[SBEV_MAX_LEVELS + 1];      // base[n_levels] = tiles of the pyramid
};
bool lazy_plan(int n_levels, const int32_t* hw, long long n_images, int channels, LazyPlan* p);
// the scan of layers 1..5 riding in the generator GEMM's prologue (gemm_bf16s.hip): what the stand-alone launch would have been given
struct LazyScan {
    const LazyPlan* plan;
    const void* const* table;
    const int32_t* index;
    const void* const* src;
    void* const* out;
    int esize;
    uint32_t *need, *done;
    bool last;
};
bool linear_f16s_gen_takes_scan(int64_t M, int N, int K, int64_t ldy, int nprod);
int linear_f16s_gen_scan(const uint16_t* Xs, const float* xscale, const uint16_t* Ws, const float* wdown, const float* bias, float* Y, int64_t M,
                         int N, int K, int64_t ldy, int relu, int nprod, const LazyScan& lz, hipStream_t stream);
int launch_lazy_relayout(const LazyPlan& p, const void* const* table, const int32_t* index, const void* const* src, void* const* out,
                         int esize, uint32_t* need, uint32_t* done, bool first, bool last, hipStream_t s);

// row_chain.hip: the row-local op chains of a decoder layer as single launches (weights pre-packed: sbev_decoder_chain_pack)
bool row_chain_supported(const sbev_decoder_config& c);
bool row_chain_pays(long long rows);
int launch_chain_front(const sbev_decoder_config& c, const sbev_decoder_weights& w, const float* bbox, const float* feat, float* x,
                       float* qkvt, float eps, hipStream_t s);
int launch_chain_attn(const sbev_decoder_config& c, const sbev_decoder_weights& w, const float* att, const float* x, float* x1,
                      const float* bbox, const float* time_diff, const float* lidar2img, float* loc_bp, float* w_bp, float eps,
                      hipStream_t s, uint16_t* x1_frag = nullptr, const float* x1_scale = nullptr, uint32_t* pair_sync = nullptr,
                      const LazyPlan* touch = nullptr, uint32_t* touch_need = nullptr);
// project.hip: sbev_sample_and_project that also marks the relayout units its points read (touch != null)
int launch_sample_and_project(const float* query_bbox, const float* offset, int64_t ld_offset, const float* scale_logits, int64_t ld_logits,
                              const float* time_diff, const float* lidar2img, const double* pc_range, int B, int Q, int T, int N, int G, int P,
                              int L, float image_h, float image_w, float eps, float* loc_bp, float* weights_bp, const LazyPlan* touch,
                              uint32_t* touch_need, const int32_t (*hw)[2], hipStream_t stream);
int launch_chain_tail(const sbev_decoder_config& c, const sbev_decoder_weights& w, const float* slabs, int splits, const float* x1,
                      const float* bbox, const float* vel_div, float* x3, float* cls_out, float* box_out, int with_front, float* x,
                      float* qkvt, float eps, hipStream_t s, float* pair_x = nullptr, uint32_t* pair_sync = nullptr);
long long chain_pair_floats(long long rows);         // pair mode of the tail: exchange rows / arrival counters for `rows` rows
long long chain_pair_sync_words(long long rows);
long long chain_fold_sync_offset(long long rows);    // the out-projection's fold counters inside the same zeroed block (64 words)
bool chain_pair_enabled();                           // sbev_decoder_chain_pair's current setting (the in-launch hand-offs' master switch)
bool chain_fault_word_ready();                       // the host-mapped fault word is installed on the current device (never under capture: sbev_init)
void chain_pair_prepare();                          // install the pair tail's host-mapped fault word for the current device (outside any capture)
unsigned chain_pair_faults_pending();               // pair hand-offs that timed out and were not acknowledged (host word, no sync)

// gemm.hip: pairs of independent small ops of a decoder layer's tail in ONE launch (see pair_kernel)
bool small_linear_shape(int64_t M, int N, int K);
int launch_ln_and_linear(const float* X, const float* ln_w, const float* ln_b, float eps, int ln_relu, float* Yln, int64_t M, int N,
                         const float* Xg, const float* W, const float* bias, float* Yg, int Ng, int K, int relu, hipStream_t s);
int launch_ln_and_refine(const float* X, const float* ln_w, const float* ln_b, float eps, int ln_relu, float* Yln, int64_t M, int N,
                         const float* bbox, const float* reg, const float* vel_div, float* out, int Q, int code, hipStream_t s);
int launch_linear_and_lin3(const float* Xg, const float* W, const float* bias, float* Yg, int64_t M, int Ng, int K, int relu,
                           const float* x3, int64_t ldx3, const float* w3, const float* b3, const float* ln_w, const float* ln_b,
                           float eps, float* y3, int N3, hipStream_t s);

// gemm.hip: LayerNorm as the PROLOGUE of the small-tile linears that consume it (sbev_ln_linear_f32); a group launch
// shares one prologue between its problems (they all read the same X)
struct LnPrologue {
    const float* g;
    const float* b;
    float eps;
    int relu;
    const float* add;   // [M, 256] or null: added after LayerNorm (+ ReLU)
    float* xn;          // [M, 256]: the normalised rows, written once (problem 0's column-tile-0 workgroups)
};
int launch_linear_group(const sbev_linear_problem* probs, int n, const LnPrologue* ln, hipStream_t s);
bool ln_linear_fusable(int64_t M, int N, int K);

// attention_bwd_mfma.hip: the MFMA flash backward behind sbev_sasa_bwd_f32 (lse / dvec: [B, H, Q] scratch each)
int launch_sasa_bwd_mfma(const float* qkvt, int64_t ld, const float* bbox, const float* lo, const float* span, const uint8_t* mask,
                         const float* out, const float* grad_out, float* grad_qkvt, float* lse, float* dvec,
                         int B, int Q, int H, float scale, float p_drop, uint64_t seed, const uint64_t* seed_dev, hipStream_t s);

// optional HIP-event bracket around sampler launches (decoder.hip; switched by sbev_profile_sampler)
bool profile_begin(hipStream_t s, hipEvent_t* e0, hipEvent_t* e1, int kind = 0);
void profile_end(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int kind = 0);   // kind: 0 sampler, 1 generator GEMM, 2 out-projection GEMM

// Sum over the 64 lanes of a wave without LDS traffic (ds_bpermute-based __shfl_xor costs an LDS round trip per
// step, which is pure exposed latency when only a few waves share a SIMD): 4 DPP steps reduce each 16-lane row
// (row_mirror, row_half_mirror, two quad permutes), then the 4 row sums are combined with v_readlane.
// Every lane receives the total.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define SBEV_DPP_F(x, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), ctrl, 0xf, 0xf, true))
    v += SBEV_DPP_F(v, 0x140);
    v += SBEV_DPP_F(v, 0x141);
    v += SBEV_DPP_F(v, 0x4e);
    v += SBEV_DPP_F(v, 0xb1);
#undef SBEV_DPP_F
    const int b = (int)__float_as_uint(v);
    const float r0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 0));
    const float r1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 16));
    const float r2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 32));
    const float r3 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

}  // namespace sbev
