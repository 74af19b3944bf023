/*
 * sbev_hip.h -- C ABI of libsbev_hip.so: the MI355X (gfx950) implementation of SparseBEV's
 * adaptive spatio-temporal sampling + adaptive mixing decoder hot path.
 *
 * This is the drop-in boundary.  Every entry point is plain C:
 * raw DEVICE pointers, sizes, an explicit hipStream_t (passed as void*), an int status.
 *   - 0 on success, a negative SBEV_E* code on failure; sbev_last_error() returns the message of the
 *     last failure on the calling thread.  No C++ exception crosses this boundary.
 *   - No hidden allocation and no synchronisation: outputs and workspaces are caller-allocated, all
 *     work is enqueued on `stream` (NULL = the legacy default stream, which is what the reference's
 *     `<<<grid, block>>>` launches use: models/csrc/msmv_sampling/msmv_sampling_forward.cu:290).
 *   - Stateless and thread-safe for distinct streams.
 *   - Feature addressing is 64-bit at the sample-batch level (base = b' * stride_bo, so pyramids beyond 2^31 elements
 *     work; the reference's 32-bit offsets overflow at B'*N*H*W*C >= 2^31: msmv_sampling_forward.cu:127).  The
 *     offset of a tap INSIDE one sample-batch slab, (N-1)*stride_v + (H*W-1)*stride_px + C, is 32-bit and
 *     host-checked (SBEV_EINVAL beyond 2^31 - 1), as are item / row counts (B'*Q < 2^31).
 * Reference interfaces replaced are cited per function as file:line inside the reference checkout.
 */
#ifndef SBEV_HIP_H
#define SBEV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBEV_ABI_VERSION 1
#define SBEV_MAX_LEVELS 5   /* c2345 and c23456 variants: models/csrc/msmv_sampling/msmv_sampling.cpp:362-369 */
#define SBEV_MAX_POINTS 32  /* MAX_POINT: models/csrc/msmv_sampling/msmv_sampling.cpp:3,125 */
#define SBEV_MAX_FRAMES 16  /* frames per sample in the online feature ring (the reference evicts its cache at 16: models/sparsebev.py:286-292) */

typedef void* sbev_stream_t; /* hipStream_t */

enum sbev_status {
    SBEV_OK = 0,
    SBEV_EINVAL = -1,  /* bad argument (the reference raises AT_ASSERTM -> RuntimeError: msmv_sampling.cpp:106-125) */
    SBEV_ELAUNCH = -2, /* hipLaunch / hipGetLastError failure (the reference only printf's: msmv_sampling_forward.cu:295-298) */
    SBEV_ENODEV = -3,  /* no gfx950 device visible */
    SBEV_EFAULT = -4   /* an EARLIER decoder step's result is invalid (a pair-mode hand-off timed out: sbev_decoder_chain_pair_faults) */
};

enum sbev_dtype { SBEV_F32 = 0, SBEV_BF16 = 1, SBEV_F16 = 2 };   /* feature STORAGE types of the sampler (fp32 math throughout): bf16 and fp16 taps are widened exactly.
                                                                   * fp16 is what the reference's eval mode produces before its out_fp32 cast (val.py:115, models/sparsebev.py:46) */
enum sbev_gemm_mode {
    SBEV_GEMM_F32 = 0,      /* exact: f32-input MFMA */
    SBEV_GEMM_BF16X3 = 1,   /* opt-in 3 x bf16 split (rounds 1-2 kernels, gemm_bf16x3.hip) */
    SBEV_GEMM_BF16X6 = 2,   /* fp32-class: hi + mid + lo bf16 images, 6 products, fp32 accumulate (gemm_bf16s.hip) */
    SBEV_GEMM_BF16X3S = 3,  /* 3 x bf16 split on the gemm_bf16s.hip kernels */
    SBEV_GEMM_F16X3 = 4,    /* fp32-class: scaled fp16 hi + lo images, 3 products (hl, lh, hh), fp32 accumulate (gemm_bf16s.hip) */
    SBEV_GEMM_F16X4 = 5     /* the same with all 4 products */
};

/* Output layouts of the sampler. */
enum sbev_out_layout {
    SBEV_OUT_REF = 0, /* [B', Q, C, P]  -- what _ms_deform_attn_cuda_c2345_forward returns (msmv_sampling.cpp:136) */
    SBEV_OUT_MIX = 1  /* [B, Q, G, T*P, C] with b' = (b*T + t)*G + g, point index t*P + p -- the layout
                         sampling_4d hands to AdaptiveMixing (models/sparsebev_sampling.py:125-128), written
                         directly so the [B',Q,C,P] -> [B,Q,G,T*P,C] permute never touches HBM */
};

int sbev_abi_version(void);
const char* sbev_last_error(void);
/* Number of visible HIP devices whose arch is gfx950 (0 if none / no driver). */
int sbev_device_count(void);

/*
 * Box / rotation convention of the checkpoint, a process-wide switch like the reference's `VERSION` global
 * (models/utils.py:320-325, set from checkpoint['version'] in val.py:128-129): it changes the rotation sign of the sample
 * offsets (rotation_3d_in_axis, models/utils.py:66-77) and the box layout SparseBEVHead.get_bboxes returns
 * (models/sparsebev_head.py:472-476).  Read at launch time by sbev_sampling_front, sbev_sample_and_project,
 * sbev_decoder_forward and sbev_nms_free_decode (bottom_center != 0).  Default SBEV_BOX_V1_0_0.
 */
enum { SBEV_BOX_V1_0_0 = 0, SBEV_BOX_V0_17_1 = 1 };
int sbev_set_box_convention(int convention);
int sbev_get_box_convention(void);

/*
 * Multi-scale multi-view bilinear sampling, forward.
 * Replaces: _ms_deform_attn_cuda_c2345_forward / _ms_deform_attn_cuda_c23456_forward
 *           (models/csrc/msmv_sampling/msmv_sampling.cpp:98-210, kernels msmv_sampling_forward.cu:75-267)
 *           and their Python entry msmv_sampling() (models/csrc/wrapper.py:87-93).
 *
 *   out[b',q,c,p] = sum_l w[b',q,p,l] * bilinear(feat_l[b', view, :, :, c]; y*(H_l-1), x*(W_l-1))
 *   view = round(loc.z*(N-1)); align_corners=True; a level contributes only if -1 < h_im < H_l and
 *   -1 < w_im < W_l; each of the 4 corners is zero outside the map.
 *
 * feats[l]      device pointer of level l (fp32 or bf16 per feat_dtype), channel-last.  Element offset of
 *               (b', view, h, w, c) = (b'/gdiv)*stride_bo[l] + (b'%gdiv)*stride_g + view*stride_v[l]
 *                                     + (h*W_l + w)*stride_px + c.
 *               The reference layout [B',N,H,W,C] is gdiv=1, stride_bo=N*H*W*C, stride_v=H*W*C, stride_px=C.
 *               A zero-copy NHWC pyramid [B*T, N, H, W, G*C] (no regroup copy) is
 *               gdiv=G, stride_bo=N*H*W*G*C, stride_g=C, stride_v=H*W*G*C, stride_px=G*C.
 * hw            host int32 [L][2] = (H_l, W_l)
 * loc           device fp32 [B',Q,P,3] = (x, y in [0,1], view/(N-1));  weights device fp32 [B',Q,P,L]
 * out           device fp32, layout per out_layout; T and G are only read for SBEV_OUT_MIX (B' = B*T*G)
 * Constraints: 1 <= L <= 5, 1 <= P <= 32, C % 4 == 0.
 */
int sbev_msmv_fwd(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                  int64_t Bp, int N, int C, int Q, int P,
                  int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                  const float* loc, const float* weights, float* out,
                  int out_layout, int T, int G, sbev_stream_t stream);

/*
 * sbev_msmv_fwd over an ONLINE FRAME RING (streaming inference).
 * Replaces: the per-call `torch.cat` of cached per-frame features (models/sparsebev.py:297-303) plus the decoder's
 *           regroup copy (models/sparsebev_transformer.py:73-85): features of every frame stay where they were
 *           written -- level l is ONE resident NHWC buffer [B, n_slots, N, H_l, W_l, G*C] -- and logical frame t
 *           (0 = current, T-1 = oldest) of sample batch b' = (b*T + t)*G + g is read from slot frame_slots[t].
 *           Per step only the newest frame's 6 images are relayouted into the slot of the evicted frame.
 * Same arguments as sbev_msmv_fwd except: stride_slot[l] = elements between consecutive slots (N*H_l*W_l*G*C);
 * gdiv must equal G; frame_slots = host int32 [T], values in [0, n_slots); 1 <= T <= SBEV_MAX_FRAMES.
 */
int sbev_msmv_fwd_ring(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                       int64_t Bp, int N, int C, int Q, int P,
                       int gdiv, const int64_t* stride_slot, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                       const float* loc, const float* weights, float* out,
                       int out_layout, int T, int G, const int32_t* frame_slots, int n_slots, sbev_stream_t stream);

/* sbev_msmv_fwd / _ring gather through raw BUFFER loads whenever every level's (sample-batch) slab is below 2 GiB: an out-of-map
 * bilinear corner is an out-of-range buffer offset, answered with zeros by the hardware and never read -- the reference's semantics
 * (msmv_sampling_forward.cu:47-66) also for Inf / NaN border pixels.  Larger slabs take 64-bit global loads + a select (same
 * results).  sbev_msmv_buffer_taps(0) forces that path (tests, A/B; env SBEV_MSMV_NO_BUF=1); returns the previous setting. */
int sbev_msmv_buffer_taps(int enable);


/*
 * Multi-scale multi-view bilinear sampling, backward (fp32 features).
 * Replaces: _ms_deform_attn_cuda_c2345_backward / _c23456_backward (models/csrc/msmv_sampling/msmv_sampling.cpp:212-360,
 *           kernels msmv_sampling_backward.cu:29-361) = MSMVSamplingC2345.backward (models/csrc/wrapper.py:51-61).
 * Same feature addressing as sbev_msmv_fwd.  grad_feats[l] (fp32, same layout as feats[l]) is ACCUMULATED into with
 * float atomics -- the caller zero-fills it (the reference's host code does `zeros_like`, :244-249).
 * grad_out [B',Q,C,P]; grad_loc [B',Q,P,3] (x, y scaled by (W_l-1), (H_l-1) like :102-104; the view component is
 * written as 0) and grad_weights [B',Q,P,L] are fully overwritten, without atomics.
 */
int sbev_msmv_bwd(const void* const* feats, void* const* grad_feats, const int32_t* hw, int L,
                  int64_t Bp, int N, int C, int Q, int P,
                  int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                  const float* loc, const float* weights, const float* grad_out,
                  float* grad_loc, float* grad_weights, sbev_stream_t stream);

/*
 * Projection of 3-D sample points into all T*N cameras, camera-hit mask, first-hit view selection.
 * Replaces: the front half of sampling_4d (models/sparsebev_sampling.py:49-114) and its DUMP taps (:82-86).
 *
 * BIT-EXACT contract (vs. the reference's native-PyTorch path on CPU): homogeneous coordinates are
 * ((m0*x + m1*y) + m2*z) + m3 with separately rounded fp32 multiplies and adds (no FMA), followed by two
 * IEEE divisions (by max(homo, eps), then by image_w / image_h);
 * valid = homo > eps && 0 < v < 1 && 0 < u < 1; view = first valid view, 0 if none.
 *
 * sample_points device fp32 [B,Q,T,G*P,3] (metres);  lidar2img device fp32 [B,T*N,4,4] (image index t*N+n)
 * loc_bp        device fp32 [B*T*G, Q, P, 3] = (u, v, view/(N-1)) -- the sampler's `loc` operand
 * dump_uvh      optional (may be NULL) device fp32 [B,T,N,Q,G*P,3] = (u, v, max(homo,eps))   (DUMP tap)
 * dump_valid    optional device uint8 [B,T,N,Q,G*P]                                          (DUMP tap)
 * i_view        optional device int32 [B,T,Q,G*P]
 */
int sbev_project_select(const float* sample_points, const float* lidar2img,
                        int B, int Q, int T, int N, int G, int P,
                        float image_h, float image_w, float eps,
                        float* loc_bp, float* dump_uvh, uint8_t* dump_valid, int32_t* i_view,
                        sbev_stream_t stream);

/*
 * Sample-point generation and scale-weight softmax (everything in SparseBEVSampling.inner_forward between
 * the two Linear layers and sampling_4d).
 * Replaces: make_sample_points (models/sparsebev_sampling.py:8-24), decode_bbox (models/bbox/utils.py:63-77),
 *           rotation_3d_in_axis v1.0.0 (models/utils.py:49-84), the velocity warp and the level softmax
 *           (models/sparsebev_transformer.py:279-300), and the weight reorder of sampling_4d (:117-119)
 *           INCLUDING its (b,g,t)-vs-(b,t,g) flattening quirk.
 *
 * query_bbox    device fp32 [B,Q,10] (cx,cy,cz in [0,1], log w,l,h, sin, cos, vx, vy)
 * offset        device fp32 [B*Q, ld_offset]: first G*P*3 columns = output of the sampling_offset Linear
 * scale_logits  device fp32 [B*Q, ld_logits]: first G*P*L columns = output of the scale_weights Linear (pre-softmax)
 *               (row strides let both live in ONE packed GEMM output: the two Linears are fused into N = G*P*(3+L))
 * time_diff     device fp32 [B,T];  pc_range host double [6] (x0,y0,z0,x1,y1,z1)
 * sample_points device fp32 [B,Q,T,G*P,3] (out; may be NULL)
 * weights_bp    device fp32 [B*G*T, Q, P, L] (out; may be NULL): row b' holds softmax(scale_logits)[b,q,g',p,:]
 *               with g' = ((b' % (T*G))) / T  -- the weights the reference actually applies to sample batch b'.
 */
int sbev_sampling_front(const float* query_bbox, const float* offset, int64_t ld_offset,
                        const float* scale_logits, int64_t ld_logits,
                        const float* time_diff, const double* pc_range,
                        int B, int Q, int T, int G, int P, int L,
                        float* sample_points, float* weights_bp, sbev_stream_t stream);

/*
 * nn.Linear on the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate).
 *   Y[m,n] = act( sum_k X[m,k] * W[n,k] + bias[n] ) + residual[m,n]
 * Replaces: every torch.nn.Linear / F.linear of the decoder layer (models/sparsebev_transformer.py:116-153,
 *           203,262-263,343-344,358,378; mmcv MultiheadAttention in/out projections and FFN, :7,125,202).
 * X [M,ldx], W [N,ldw] (both K-contiguous, 16-byte aligned, ld % 4 == 0), bias [N] or NULL,
 * residual [M,ldy] or NULL (added AFTER the activation, the `identity + out` form of mmcv FFN / MHA),
 * relu != 0 applies max(.,0) before the residual.
 */
int sbev_linear_f32(const float* X, const float* W, const float* bias, const float* residual, float* Y,
                    int64_t M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldy, int relu,
                    sbev_stream_t stream);

/*
 * Up to 3 independent small Linear layers (K = 256 or 512, same K) in ONE launch, e.g. the classification and the
 * regression branch of a decoder layer (models/sparsebev_transformer.py:174-175), which both consume the layer output.
 * Each problem has exactly the semantics of sbev_linear_f32.
 */
typedef struct sbev_linear_problem {
    const float* X; const float* W; const float* bias; const float* residual; float* Y;
    int64_t M; int32_t N, K; int64_t ldx, ldw, ldy; int32_t relu;
} sbev_linear_problem;
int sbev_linear_group_f32(const sbev_linear_problem* probs, int n, sbev_stream_t stream);

/*
 * Split-K variant for long reductions with a small output (AdaptiveMixing.out_proj: K = 32768, N = 256,
 * models/sparsebev_transformer.py:344,378), with the whole epilogue of that call site fused into the slab
 * reduction: + bias, optional ReLU, + residual (`query + out`, :379), optional LayerNorm(N) (norm2, :171).
 * workspace: device buffer of sbev_linear_splitk_workspace(M, N, splits) bytes.  N % 4 == 0, N <= 1024.
 */
/* Recommended number of K splits for sbev_linear_splitk_f32 at this shape (fills the GPU once). */
int sbev_linear_splitk_plan(int64_t M, int N, int K);
int64_t sbev_linear_splitk_workspace(int64_t M, int N, int splits);
int sbev_linear_splitk_f32(const float* X, const float* W, const float* bias, const float* residual,
                           const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                           int64_t M, int N, int K, int64_t ldx, int64_t ldw, int relu,
                           int splits, float* workspace, sbev_stream_t stream);

/* Row LayerNorm (+ optional ReLU) (+ optional add_after [M,N], e.g. query_feat + position encoding, :167) of X [M,N],
 * N % 4 == 0, N <= 1024.
 * Replaces: nn.LayerNorm(embed_dims) (+ nn.ReLU) at models/sparsebev_transformer.py:117-121,127-129,132-136,169-172. */
int sbev_layer_norm_f32(const float* X, const float* ln_w, const float* ln_b, float eps,
                        const float* add_after, float* Y, int64_t M, int N, int relu, sbev_stream_t stream);

/*
 * LayerNorm as the prologue of the Linear that consumes it:
 *   Xn = relu?(LayerNorm(X; ln_w, ln_b, eps)) (+ ln_add),   Y = act(Xn W^T + bias) (+ residual);  Xn is also stored.
 * Replaces: a norm immediately followed by a Linear on its output -- the position encoder's last LayerNorm + ReLU, the
 *           `query_feat + query_pos` add and the attention in_proj (models/sparsebev_transformer.py:117-121,166-169;
 *           mmcv MultiheadAttention in_proj), norm1 + the sampling offset / scale-weight Linear (:169-170,262-268), norm3 +
 *           the first Linear of the cls / reg branches (:172-175,132-147).
 * X, Xn [M, K] dense; ln_add [M, K] or NULL; W [N, ldw]; Y / residual [M, ldy].  One launch when K == 256, M <= 2048 and
 * the shape is one sbev_linear_f32 runs on its 32 x 32 small-tile kernel (each column tile re-normalises its rows, so the
 * prologue only pays while rows are few); otherwise sbev_layer_norm_f32 + sbev_linear_f32.  Pointers 16-byte aligned.
 */
int sbev_ln_linear_f32(const float* X, const float* ln_w, const float* ln_b, float ln_eps, int ln_relu,
                       const float* ln_add, float* Xn, const float* W, const float* bias, const float* residual,
                       float* Y, int64_t M, int N, int K, int64_t ldw, int64_t ldy, int relu, sbev_stream_t stream);

/*
 * sbev_sampling_front + sbev_project_select fused into one launch (what the decoder runtime uses): the intermediate
 * [B,Q,T,G*P,3] sample-point tensor never exists.  Same arithmetic in the same no-FMA translation unit, so loc_bp and
 * weights_bp are bit-identical to the two-call path (tested).  No DUMP outputs: use the two calls for the debug taps.
 */
int sbev_sample_and_project(const float* query_bbox, const float* offset, int64_t ld_offset,
                            const float* scale_logits, int64_t ld_logits,
                            const float* time_diff, const float* lidar2img, const double* pc_range,
                            int B, int Q, int T, int N, int G, int P, int L,
                            float image_h, float image_w, float eps,
                            float* loc_bp, float* weights_bp, sbev_stream_t stream);

/*
 * Adaptive mixing core: per (query, group)  y = relu(LN_[Pout,C]( S @ relu(LN_[Pin,C]( x @ M )) )).
 * Replaces: the two dynamic matmuls + F.layer_norm + ReLU of AdaptiveMixing.inner_forward
 *           (models/sparsebev_transformer.py:362-374).  The parameter generator (:358) and the out-projection
 *           (:377-379) around it are sbev_linear_f32 / sbev_linear_splitk_f32.
 * x       device fp32 [BQ, G, Pin, C]          (= the sampler's SBEV_OUT_MIX output)
 * params  device fp32 [BQ, G, C*C + Pout*Pin]  (per (q,g): M[C_in,C_out] then S[Pout,Pin]; the generator's output)
 * y       device fp32 [BQ, G, Pout, C]         (flattened [g][o][c] it is the out-projection's input row)
 * Built for C = 64, Pout = 128 (hard-coded in the reference: sparsebev_transformer.py:124); Pin % 4 == 0, Pin <= 120.
 */
int sbev_adaptive_mixing_f32(const float* x, const float* params, float* y,
                             int64_t BQ, int G, int Pin, int C, int Pout, float eps, sbev_stream_t stream);

/*
 * Scale-adaptive self attention core (flash-style; the [B,H,Q,Q] bias is never materialised).
 * Replaces: SparseBEVSelfAttention.inner_forward's distance/tau mask construction
 *           (models/sparsebev_transformer.py:215-227,236-248) and the softmax(QK^T/sqrt(d) + mask)V core of
 *           the mmcv/torch MultiheadAttention it calls (:228).
 * qkvt    device fp32 [B,Q,ld]: columns [0,HD*H) = q, [HD*H,2HD*H) = k, [2HD*H,3HD*H) = v (the packed in_proj
 *         output), [3HD*H, 3HD*H+H) = tau (gen_tau output; one GEMM with the 8 tau rows appended to in_proj)
 * query_bbox device fp32 [B,Q,10]; the box centres in metres, c = bbox[:, 0:2] * (range_max - range_min) + range_min
 *         (decode_bbox, models/bbox/utils.py:63-71), are formed inside the kernel from pc_range (host double [6])
 * mask    optional device uint8 [Q,Q], 1 = masked (-inf), the query-denoising mask (:224-225); NULL at inference
 * out     device fp32 [B,Q,H*HD] (input of the out-projection)
 * Built for head_dim = 32.
 */
int sbev_sasa_f32(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                  const uint8_t* mask, float* out, int B, int Q, int H, int head_dim, sbev_stream_t stream);

/*
 * Box refinement of one decoder layer.
 * Replaces: refine_bbox (models/sparsebev_transformer.py:155-160) with inverse_sigmoid (models/utils.py:87-102)
 *           and the velocity / time_diff division (:179-183).
 * out[:, 0:3] = sigmoid(reg[:, 0:3] + logit(clamp(query_bbox[:, 0:3]))), out[:, 3:] = reg[:, 3:],
 * out[:, 8:] /= vel_div[b] when vel_div != NULL (vel_div[b] = time_diff[b,1], values < 1e-5 replaced by 1).
 */
int sbev_refine_bbox(const float* query_bbox, const float* reg, const float* vel_div, float* out,
                     int B, int Q, int code_size, sbev_stream_t stream);

/* ---- detection head pre / post-processing -------------------------------------------------- */

/*
 * Decoder inputs at inference (no query denoising).
 * Replaces: SparseBEVHead.forward's `init_query_bbox.weight.clone()` + the eval branch of prepare_for_dn_input
 *           (models/sparsebev_head.py:70,123-126,209-211).
 * query_bbox[b] = init_query_bbox [Q,10];  query_feat[b,q] = [label_row (D-1 values = label_enc.weight[num_classes]), 0].
 */
int sbev_head_prepare(const float* init_query_bbox, const float* label_row, float* query_bbox, float* query_feat,
                      int B, int Q, int D, sbev_stream_t stream);

/*
 * Decoder boxes -> head output format.
 * Replaces: models/sparsebev_head.py:85-95 -- xyz * (pc_max - pc_min) + pc_min (two fp32 roundings each, as the
 *           reference's separate mul and add) and the column reorder to (cx, cy, w, l, cz, h, sin, cos, vx, vy).
 * bbox_norm / out: [n, 10] rows (n = num_layers * B * Q); in place (out == bbox_norm) is allowed.
 */
int sbev_head_denorm(const float* bbox_norm, const double* pc_range, float* out, int64_t n, sbev_stream_t stream);

/*
 * NMS-free box decoding of the last decoder layer.
 * Replaces: NMSFreeCoder.decode_single (models/bbox/coders/nms_free_coder.py:37-88) with denormalize_bbox
 *           (models/bbox/utils.py:26-47), per sample of NMSFreeCoder.decode (:90-111); with bottom_center != 0 also
 *           the gravity-centre -> bottom-centre shift of SparseBEVHead.get_bboxes (models/sparsebev_head.py:471).
 * cls_scores [B,Q,num_classes] logits, bbox_preds [B,Q,10] in the head format above.
 * Top max_num of the Q*num_classes sigmoid scores (score descending; equal scores: lower flat index first -- torch.topk
 * leaves that order unspecified), label = index % num_classes, box = bbox_preds[index / num_classes] denormalised to
 * (cx, cy, cz, w, l, h, rot, vx, vy); kept iff the centre lies inside post_center_range [6] (required, like the
 * reference) and, when use_score_threshold != 0, score > score_threshold.  Kept rows are compacted in order into
 * boxes [B,max_num,9] / scores [B,max_num] / labels [B,max_num] (int32), the rest zero-filled; count[b] = rows kept.
 * Limits: Q * num_classes <= 16384 (the per-sample sort lives in one CU's LDS), max_num <= 1024.
 */
int sbev_nms_free_decode(const float* cls_scores, const float* bbox_preds, int B, int Q, int num_classes, int max_num,
                         float score_threshold, int use_score_threshold, const double* post_center_range,
                         int bottom_center, float* boxes, float* scores, int32_t* labels, int32_t* count,
                         sbev_stream_t stream);


/*
 * Batched NCHW -> NHWC relayout of one pyramid level: in [n_images, channels, hw] -> out [n_images, hw, channels].
 * Replaces: the permute + contiguous feature regroup of the decoder (models/sparsebev_transformer.py:73-85); the
 *           per-group split of that regroup is not needed (sbev_msmv_fwd addresses group g as a channel slice).
 */
int sbev_nchw_to_nhwc_f32(const float* in, float* out, int64_t n_images, int channels, int hw, sbev_stream_t stream);

/* Staging launches of a REPLAYABLE step (sparsebev_amd/runtime.py StepGraphs): the source pointer is table[index], read on the
 * device when the kernel starts.  `table` is a device array of 64-bit pointers that the host refreshes before every launch of the
 * captured graph, so a caller that passes newly allocated tensors of the same shape every step -- how the reference's loops feed
 * model(...) (timing.py:77-96, val.py) -- replays ONE graph.  Sources must be 16-byte aligned.
 * sbev_nchw_to_nhwc_f32_indirect = sbev_nchw_to_nhwc_f32 with in = table[index]; sbev_copy_indirect: nseg <= 4 contiguous copies in one
 * launch, segment k = nbytes[k] bytes from table[index[k]] to dst[k] (index, dst, nbytes: host arrays). */
int sbev_nchw_to_nhwc_f32_indirect(const void* const* table, int index, float* out, int64_t n_images, int channels, int hw,
                                   sbev_stream_t stream);
/* All levels of one pyramid in ONE launch (round 5): level l = [n_images, channels, hw[l]] from table[index[l]] to out[l]; the coarse
 * levels' few tiles run in the tail of the finest level's instead of in short launches of their own.  Needs hw[l] % 4 == 0, channels % 4
 * == 0 and 16-byte aligned buffers (SBEV_EINVAL otherwise: launch the per-level form). */
int sbev_nchw_to_nhwc_f32_multi_indirect(const void* const* table, int n_levels, const int32_t* index, float* const* out,
                                         int64_t n_images, int channels, const int32_t* hw, sbev_stream_t stream);
/* The same relayouts for 2-byte channels -- bf16 or fp16 feature STORAGE (enum sbev_dtype; bytes are moved, never interpreted): what
 * an fp16 backbone (the reference's eval mode, val.py:115, before the out_fp32 cast of models/sparsebev.py:46) or a bf16 neck emits
 * goes to the sampler's layout at half the traffic of the fp32 relayout.  128-channel x 64-pixel tiles; any sizes (channels % 8,
 * hw % 4 and 8- / 16-byte alignment select the vector path). */
int sbev_nchw_to_nhwc_b16(const void* in, void* out, int64_t n_images, int channels, int hw, sbev_stream_t stream);
int sbev_nchw_to_nhwc_b16_indirect(const void* const* table, int index, void* out, int64_t n_images, int channels, int hw,
                                   sbev_stream_t stream);
int sbev_copy_indirect(const void* const* table, int nseg, const int32_t* index, void* const* dst, const int64_t* nbytes,
                       sbev_stream_t stream);

/* The decoder step's LAST launch (round 5): the stacked outputs cls [n_cls floats] / bbox [n_box floats] copied out of the runtime's
 * buffers into the caller's, nan_to_num'ed on the way (NaN -> 0, +-Inf -> +-FLT_MAX, every other value bit for bit) -- both tensors
 * in one launch.  Replaces: the two torch.nan_to_num launches of SparseBEVTransformer.forward (models/sparsebev_transformer.py:35-36)
 * and, for a replayed step graph, the clones out of the graph's buffers.  _indirect: destinations = table[idx_cls] / table[idx_box]
 * (device pointer table, see above), so that a captured step writes straight into tensors allocated per call.  Sources and
 * destinations must not overlap. */
int sbev_finish_outputs(const float* cls_src, const float* box_src, float* cls_dst, float* box_dst, int64_t n_cls, int64_t n_box,
                        sbev_stream_t stream);
int sbev_finish_outputs_indirect(const void* const* table, int idx_cls, int idx_box, const float* cls_src, const float* box_src,
                                 int64_t n_cls, int64_t n_box, sbev_stream_t stream);


/*
 * y = relu(LayerNorm(x[:, 0:3] @ w^T + b)): the first half of the position encoder
 * (nn.Linear(3, D), nn.LayerNorm(D), nn.ReLU: models/sparsebev_transformer.py:116-119).  x [M, ldx], w [N,3].
 */
int sbev_linear3_ln_relu_f32(const float* x, int64_t ldx, const float* w, const float* b,
                             const float* ln_w, const float* ln_b, float eps, float* y,
                             int64_t M, int N, sbev_stream_t stream);

/* Split-K slab reducer: Y = LayerNorm?( relu?( sum_z slabs[z] + bias ) + residual ), slabs [splits, M, N]. */
int sbev_splitk_reduce_f32(const float* slabs, int splits, const float* bias, const float* residual,
                           const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                           int64_t M, int N, int relu, sbev_stream_t stream);

/*
 * Opt-in "3 x bf16" Linear for the two large GEMMs of AdaptiveMixing (models/sparsebev_transformer.py:358,378):
 * both fp32 operands are split into bf16 (hi, lo) pairs and Y = Xhi.Whi + Xhi.Wlo + Xlo.Whi is accumulated in
 * fp32 on v_mfma_f32_32x32x16_bf16.  fp32-class accuracy (about 2^-16 relative per product; 1.2e-5 max abs error on
 * the reference's AdaptiveMixing fixture), ~3-5x the speed of the exact-fp32 MFMA path; NOT bit-equal to fp32 math,
 * hence never the default.  sbev_split_bf16x3_weights converts W [N,K] fp32 once into W2 [N, K/8, 2, 8] bf16
 * (same byte count); K % 32 == 0.
 */
int sbev_split_bf16x3_weights(const float* W, uint16_t* W2, int64_t N, int K, sbev_stream_t stream);
int sbev_linear_bf16x3(const float* X, const uint16_t* W2, const float* bias, const float* residual, float* Y,
                       int64_t M, int N, int K, int64_t ldx, int64_t ldy, int relu, sbev_stream_t stream);
/* The parameter generator's shape (K = 256, N % 128 == 0, N >= 1024; sbev_linear_bf16x3_strip_ok) with the activation split
 * ONCE by the caller (X2 = sbev_split_bf16x3_weights(X [M, 256])): W-stationary strips in registers, X2 streamed from L2. */
int sbev_linear_bf16x3_strip_ok(int64_t M, int N, int K);
int sbev_linear_bf16x3_strip(const uint16_t* X2, const uint16_t* W2, const float* bias, float* Y, int64_t M, int N, int K,
                             int64_t ldy, int relu, sbev_stream_t stream);
int sbev_linear_splitk_bf16x3(const float* X, const uint16_t* W2, const float* bias, const float* residual,
                              const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                              int64_t M, int N, int K, int64_t ldx, int relu, int splits, float* workspace,
                              sbev_stream_t stream);

/*
 * Split-bf16 Linears on the bf16 matrix core, round 3 (gemm_bf16s.hip).  nimg = 3 ("bf16x6"): every fp32 operand is the exact
 * sum of three RNE bf16 images (hi + mid + lo) and Y accumulates, in fp32 on v_mfma_f32_32x32x16_bf16, the six image products of
 * weight >= 2^-16 (hh, hm, mh, mm, hl, lh); the dropped ones are <= 2^-23 |a b| per product, i.e. below what an fp32 fma chain
 * rounds away per step -- fp32-class, not bit-equal to fp32 math.  nimg = 2 ("bf16x3"): hi + lo, three products, 2^-16 class.
 * Replaces: torch.nn.Linear of AdaptiveMixing.parameter_generator / out_proj (models/sparsebev_transformer.py:358,378).
 *
 *   sbev_split_bf16s_rows   X [rows, ldx] fp32 -> out [nimg][rows][K] bf16 row-major planes (rows * K * nimg elements): the
 *                           images themselves (image i = RNE_bf16 of what images < i left over).  K % 8 == 0.
 *   sbev_pack_bf16s_frags   W [N, ldw] fp32 -> out [ceil(N/32)][K/16][nimg][64][8] bf16 (sbev_bf16s_image_elems(N, K, nimg)
 *                           elements): the same images in v_mfma_f32_32x32x16_bf16 operand order -- lane l of fragment (nf, ks)
 *                           holds row 32 nf + (l & 31), k = 16 ks + 8 (l >> 5) + 0..7; a ragged last block repeats row N - 1.
 *                           The operand format of both kernels (weights once per weight update, the generator's X once per
 *                           call); K % 16 == 0.
 *   sbev_linear_bf16s_gen   Y [M, ldy] = X W^T + bias (ReLU optional) from the fragment images Xs / Ws.  N % 256 == 0,
 *                           K % 32 == 0, K <= 4096 (sbev_linear_bf16s_gen_ok).
 *   sbev_linear_splitk_bf16s  Y = LayerNorm?(relu?(X W^T + bias) + residual) for N == 256, K % 32 == 0 (sbev_linear_bf16s_out_ok):
 *                           X stays fp32 [M, ldx] and is split inside the kernel; workspace = sbev_linear_bf16s_out_plan(M, N, K)
 *                           partial slabs [plan, M, 256] fp32, summed in a fixed order (bit-reproducible) by sbev_splitk_reduce_f32.
 */
int64_t sbev_bf16s_image_elems(int64_t rows, int K, int nimg);
int sbev_split_bf16s_rows(const float* X, int64_t ldx, uint16_t* out, int64_t rows, int K, int nimg, sbev_stream_t stream);
int sbev_pack_bf16s_frags(const float* W, int64_t ldw, uint16_t* out, int N, int K, int nimg, sbev_stream_t stream);
int sbev_linear_bf16s_gen_ok(int64_t M, int N, int K);
int sbev_linear_bf16s_gen(const uint16_t* Xs, const uint16_t* Ws, const float* bias, float* Y, int64_t M, int N, int K,
                          int64_t ldy, int relu, int nimg, sbev_stream_t stream);
int sbev_linear_bf16s_out_ok(int64_t M, int N, int K);
int sbev_linear_bf16s_out_plan(int64_t M, int N, int K);
int sbev_linear_splitk_bf16s(const float* X, const uint16_t* Wp, const float* bias, const float* residual,
                             const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                             int64_t M, int N, int K, int64_t ldx, int relu, int nimg, float* workspace,
                             sbev_stream_t stream);

/*
 * fp16 hi + lo Linears on the same kernels (gemm_mode f16x3 / f16x4, round 3): an operand row (or the whole X) is multiplied by
 * 2^e -- a power of two, exact -- so that its largest |value| lands in [2^14, 2^15), then split into hi = RNE_fp16 and
 * lo = RNE_fp16(remainder): 11 + 11 significand bits + lo's sign = the fp32 value to <= 2^-23 relative for elements within 2^-17
 * of that maximum, to 2^-40 of the maximum below.  Products hl + lh + hh (nprod = 3; the dropped lo x lo is <= 2^-24 |a b|) or all
 * four (nprod = 4) on v_mfma_f32_32x32x16_f16, fp32 accumulation; the result is multiplied by 2^-(ex + ew) (exact).
 * tests/test_gpu_bf16s.py: max and rms error against fp64 <= the exact f32-MFMA kernels' at both AdaptiveMixing shapes.
 *   sbev_pack_f16s_frags     W [N, ldw] fp32 -> out [ceil(N/32)][K/16][2][64][8] fp16 (sbev_bf16s_image_elems(N, K, 2) elements) and
 *                            scales: per row [2][N] (2^e, then 2^-e) or, per_tensor = 1, [2] for the whole matrix (one extra pass over
 *                            it); per_tensor = 2: scales [2] is an input (a power of two from the caller's bound on |W|)
 *   sbev_linear_f16s_gen     as sbev_linear_bf16s_gen; xscale = X's [2] scales, wdown = W's 2^-e row [N]
 *   sbev_linear_splitk_f16s  as sbev_linear_splitk_bf16s; X fp32 is multiplied by 2^x_up_log2 inside the kernel (x_is_pairs = 0) -- the caller's bound
 *                            (|X| 2^x_up_log2 < 65504; an overflow shows as Inf / NaN, never silently); nscale = sbev_f16s_out_scale
 *   sbev_f16s_out_scale      nscale[n] = wdown[n] 2^-x_up_log2
 */
int sbev_pack_f16s_frags(const float* W, int64_t ldw, uint16_t* out, float* scales, int N, int K, int per_tensor, sbev_stream_t stream);
int sbev_linear_f16s_gen(const uint16_t* Xs, const float* xscale, const uint16_t* Ws, const float* wdown, const float* bias, float* Y,
                         int64_t M, int N, int K, int64_t ldy, int relu, int nprod, sbev_stream_t stream);

/* K = 256 with two images per operand (f16x3 / f16x4 / bf16x3s) runs on the WEIGHT-STATIONARY generator kernel: a wave keeps the
 * weights of its 32 output columns in registers for the whole launch, only X streams (LDS-DMA ring), persistent workgroups walk
 * (column tile, row split) tasks -- half the operand bytes per MFMA of the tiled kernel, bit-identical results.  0 restores the tiled
 * ping-pong kernel everywhere (A/B, tests; env SBEV_NO_GEN_WS=1); returns the previous setting. */
int sbev_linear_gen_weight_stationary(int enable);

int sbev_linear_splitk_f16s(const float* X, int x_is_pairs, int x_up_log2, const uint16_t* Wp, const float* nscale, const float* bias, const float* residual,
                            const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                            int64_t M, int N, int K, int64_t ldx, int relu, int nprod, float* workspace, sbev_stream_t stream);
int sbev_f16s_out_scale(const float* wdown, int x_up_log2, float* nscale, int N, sbev_stream_t stream);
/* sbev_linear_splitk_f16s with X's scale in device memory (x_scale = {2^e, 2^-e}, e.g. sbev_f16s_tensor_scale) instead of a host-side
 * bound; wdown = W's [N] down-scales (sbev_pack_f16s_frags scales + N).  X fp32 only. */
int sbev_linear_splitk_f16s_xdev(const float* X, const float* x_scale, const uint16_t* Wp, const float* wdown, const float* bias,
                                 const float* residual, const float* ln_w, const float* ln_b, float ln_eps, float* Y,
                                 int64_t M, int N, int K, int64_t ldx, int relu, int nprod, float* workspace, sbev_stream_t stream);
/* The pre-split operand: out[i] = (fp16 hi, fp16 lo) of X[i] 2^up_log2 packed in one 32-bit slot (hi in the low half).  With
 * x_is_pairs = 1 sbev_linear_splitk_f16s takes X in this format (same [M, ldx] geometry) and only de-interleaves it.  The mixing
 * launches can emit it directly: sbev_adaptive_mixing_pairs_f16 / sbev_sample_mix_pairs_f16 = sbev_adaptive_mixing_f32 /
 * sbev_sample_mix_f32 with y written as pairs of y 2^up_log2 (what sbev_decoder_forward does in the fp16 GEMM modes). */
int sbev_f16s_pairs(const float* X, void* out, int64_t n, int up_log2, sbev_stream_t stream);
/* updown[0..1] = {2^e, 2^-e} with max |X| 2^e in [2^14, 2^15) over X [rows, ldx] (K columns): the operand scale of sbev_gemm_tn_f16s */
int sbev_f16s_tensor_scale(const float* X, int64_t ldx, int64_t rows, int K, float* updown, sbev_stream_t stream);
int sbev_adaptive_mixing_pairs_f16(const float* x, const float* params, void* y, int64_t BQ, int G, int Pin, int Cg, int Pout, float eps,
                                   int up_log2, sbev_stream_t stream);
int sbev_sample_mix_pairs_f16(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                              int64_t B, int N, int Q, int T, int G, int P, int Cg,
                              const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                              const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                              const float* params, void* y, int Pout, float eps, int up_log2, sbev_stream_t stream);

/* dst[i] = (float)src[i] over a contiguous run; src_dtype: 0 fp32, 1 bf16, 2 fp16.  How the online frame ring takes
 * channels-last frames (sparsebev_amd/cache.py); replaces the torch.cat of cached frames, models/sparsebev.py:297-303. */
int sbev_copy_widen_f32(const void* src, int src_dtype, float* dst, int64_t n, sbev_stream_t stream);

/*
 * Gather + adaptive mixing in ONE launch: the workgroup of item (b*Q + q, g) samples its own x[T*P, 64] (the arithmetic of
 * sbev_msmv_fwd / sbev_msmv_fwd_ring with out_layout SBEV_OUT_MIX, frames spread over its four waves) into LDS and runs
 * sbev_adaptive_mixing_f32 on it -- bit-identical to the two launches, without the [B,Q,G,T*P,C] round trip through HBM.
 * Replaces: sampling_4d's gather + the middle of AdaptiveMixing.inner_forward (models/sparsebev_sampling.py:122-128,
 *           models/sparsebev_transformer.py:362-374).
 * Shapes covered: sbev_sample_mix_supported(L, C, P, T, gdiv, G) != 0  (L in {4,5}, C = 64, P = 4, gdiv = G, T*P in
 * {16, 32, 48, 64}); feats / strides / loc / weights as for sbev_msmv_fwd with B' = B*T*G; frame_slots NULL = dense
 * pyramid, else the online ring (n_slots, see sbev_msmv_fwd_ring); params [B*Q, G, C*C + Pout*T*P]; y [B*Q, G, Pout, C].
 */
int sbev_sample_mix_supported(int L, int C, int P, int T, int gdiv, int G);
/* The fused kernel's taps are 31-bit BUFFER byte offsets (an out-of-map bilinear corner is a hardware-zeroed out-of-range load, never a
 * read: msmv_sampling_forward.cu:47-66), so every level's (sample-batch) slab -- N views of H_l x W_l pixels -- must stay below 2 GiB;
 * 1 if so.  hw = {H_0, W_0, H_1, W_1, ...}, strides in elements.  sbev_msmv_fwd itself has a 64-bit path for larger slabs. */
int sbev_sample_mix_slabs_ok(const int32_t* hw, int L, int feat_dtype, int N, int C, const int64_t* stride_v, int64_t stride_px);
int sbev_sample_mix_f32(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                        int64_t B, int N, int Q, int T, int G, int P, int C,
                        const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                        const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                        const float* params, float* y, int Pout, float eps, sbev_stream_t stream);
/* sbev_decoder_forward uses the fused launch where supported (default 1); 0 restores sampler + mixing as two launches. */
int sbev_decoder_fuse_sample_mix(int enable);

/*
 * Launch order of the fused gather + mixing items (round 4).  The gather's fabric traffic is set by WHICH items share an XCD's
 * private L2 and WHEN they run: in launch order (block b = item (row b / G, group b % G) on XCD b % 8) every group is split
 * over two XCDs by row parity and both fetch the whole group's coarse levels, and rows far apart on the BEV grid evict each
 * other's lines (tools/sampler_footprint.py: 172 MB of fabric reads per launch at config 2 against 124 MB of distinct tap
 * segments; 800 against 503 MB at the 1600-query config).  sbev_query_order sorts each sample's rows by the direction of the
 * box centre around the ego origin -- order [B*Q] int32, sample b's rows b*Q + q in slots [b*Q, (b+1)*Q), one workgroup per
 * sample, Q <= sbev_query_order_max() -- and sbev_sample_mix_*_ordered walk the GROUP-major list (g, position) in eight
 * contiguous pieces, one per XCD: one group and one arc of the camera ring per L2.  A placement hint only: results are
 * bit-identical for ANY permutation (order = NULL: launch order).  query_bbox rows: ld floats apart, columns 0, 1 = normalised
 * centre (decode_bbox, models/bbox/utils.py:69-70; the reference has no counterpart: its CUDA op takes the launch order).
 * Measured (round 4, config 2): fabric reads of the launch 290 -> 232 MB, L2 hit 0.39 -> 0.47, duration unchanged (85.6 -> 87.7 us:
 * the launch is bound by each workgroup's chain of memory latencies, not by fabric bytes), plus 17 us for the sort -- so
 * sbev_decoder_forward keeps the launch order unless sbev_decoder_query_order(1) (env SBEV_QUERY_ORDER=1; returns the previous
 * setting): then every layer's input boxes are sorted in front of its self attention and the fused launch walks that order;
 * sbev_decoder_query_order(2) (SBEV_QUERY_ORDER=2) sorts the step's INPUT boxes once and every layer walks that order.
 */
int sbev_query_order_max(void);
int sbev_query_order(const float* query_bbox, int64_t ld, const double* pc_range, int B, int Q, int32_t* order, sbev_stream_t stream);
int sbev_sample_mix_f32_ordered(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                                int64_t B, int N, int Q, int T, int G, int P, int C,
                                const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                                const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                                const float* params, float* y, int Pout, float eps, const int32_t* order, sbev_stream_t stream);
int sbev_sample_mix_pairs_f16_ordered(const void* const* feats, const int32_t* hw, int L, int feat_dtype,
                                      int64_t B, int N, int Q, int T, int G, int P, int Cg,
                                      const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                                      const float* loc, const float* weights, const int32_t* frame_slots, int n_slots,
                                      const float* params, void* y, int Pout, float eps, int up_log2, const int32_t* order,
                                      sbev_stream_t stream);
int sbev_decoder_query_order(int enable);

/* ------------------------------------------------------------------------------------------------------------
 * Training: backward passes of the decoder layer.
 * Replaces: torch autograd through SparseBEVTransformerDecoderLayer.forward and the activation-checkpointed
 *           inner_forward()s (models/sparsebev_transformer.py:162-193,231-234,313-317,383-387).  Every entry point is the
 *           hand-written backward of one forward entry point above; sparsebev_amd/autograd.py chains them.
 * ---------------------------------------------------------------------------------------------------------- */

/* Layout-generic fp32 MFMA GEMM:  C[M,N] (+)= sum_k A(m,k) B(k,n),  A(m,k) = a_kmajor ? A[k*lda+m] : A[m*lda+k],
 * B(k,n) = b_kmajor ? B[k*ldb+n] : B[n*ldb+k].  For nn.Linear y = x W^T:  grad_x = sbev_gemm_f32(grad_y, 0, ., W, 1, .),
 * grad_W = sbev_gemm_f32(grad_y, 1, ., x, 1, .).  `workspace` (sbev_gemm_f32_workspace bytes, may be NULL when that is 0)
 * holds split-K slabs for few-tile / long-K shapes; results are bit-reproducible (no atomics). */
int64_t sbev_gemm_f32_workspace(int64_t M, int N, int64_t K);
int sbev_gemm_f32(const float* A, int a_kmajor, int64_t lda, const float* B, int b_kmajor, int64_t ldb,
                  float* C, int64_t ldc, int64_t M, int N, int64_t K, int accumulate,
                  float* workspace, sbev_stream_t stream);
/* The same reduction over up to 8 operand pairs of identical layout and shape in ONE launch (+ the slab sum): C (+)= sum_s A_s B_s --
 * the weight gradient of a Linear shared by several decoder layers.  A / B: host arrays of nseg device pointers;
 * workspace: sbev_gemm_f32_multi_workspace(M, N, K, nseg) bytes (always needed).  Bit-reproducible. */
int64_t sbev_gemm_f32_multi_workspace(int64_t M, int N, int64_t K, int nseg);
int sbev_gemm_f32_multi(const float* const* A, int a_kmajor, int64_t lda, const float* const* B, int b_kmajor, int64_t ldb,
                        int nseg, float* C, int64_t ldc, int64_t M, int N, int64_t K, int accumulate,
                        float* workspace, sbev_stream_t stream);

/* grad_W-shaped product on the fp16 matrix core (gemm_tn_f16s.hip): C[M,N] (+)= sum_k A[k*lda + m] B[k*ldb + n], both operands
 * k-major (grad_W = grad_y^T . x of a Linear, reduced over the B*Q rows; replaces sbev_gemm_f32(a_kmajor = b_kmajor = 1) where
 * sbev_gemm_tn_f16s_ok: M, N multiples of 4 and >= 256 tiles of 128 x 128).  Operands are multiplied by a_scale[0] / b_scale[0]
 * (device, {2^e, 2^-e}: sbev_f16s_tensor_scale or a caller's bound), split into fp16 hi + lo inside the kernel, 3 products
 * (hl, lh, hh), fp32 accumulation, result times a_scale[1] b_scale[1].  Error vs fp64 <= the exact f32-MFMA kernel's. */
int sbev_gemm_tn_f16s_ok(int64_t M, int64_t N, int64_t K);
int sbev_gemm_tn_f16s(const float* A, int64_t lda, const float* a_scale, const float* B, int64_t ldb, const float* b_scale,
                      float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, sbev_stream_t stream);

/* dZ = dY * (Y > 0) (Y = a ReLU's forward output, NULL: dZ = dY; dZ may be NULL or alias dY) and db[n] = sum_m dZ[m,n]
 * (NULL: skipped).  Rows have stride ld.  workspace: sbev_colsum_workspace(M, N) bytes (needed when db != NULL);
 * column sums are added in a fixed order (bit-reproducible). */
int64_t sbev_colsum_workspace(int64_t M, int N);
int sbev_bias_relu_bwd(const float* dY, const float* Y, float* dZ, float* db, int64_t M, int N, int64_t ld,
                       float* workspace, sbev_stream_t stream);

/* LayerNorm(+ReLU) backward over the last dim (N % 4 == 0, N <= 1024): X is the LayerNorm INPUT;
 * workspace: sbev_layer_norm_bwd_workspace(M, N) bytes. */
int64_t sbev_layer_norm_bwd_workspace(int64_t M, int N);
int sbev_layer_norm_bwd(const float* dY, const float* X, const float* gamma, const float* beta, float eps, int relu,
                        float* dX, float* dgamma, float* dbeta, float* workspace, int64_t M, int N, sbev_stream_t stream);
/* sbev_bias_relu_bwd / sbev_layer_norm_bwd with the parameter gradients (db; dgamma, dbeta) ADDED to their buffers when accumulate != 0:
 * a parameter shared by the layers of a decoder call collects its gradient in one buffer, no separate summation launches. */
/* LayerNorm backward in two halves for parameters shared by several layers: sbev_layer_norm_bwd_rows = dX and the [M, 2] (mean, rstd)
 * rows (workspace >= 2 M floats); sbev_layer_norm_param_group = dgamma / dbeta of ng <= 8 LayerNorms, each summed over <= 8
 * (dY, X, stats) triples, in ONE launch.  Pointer tables are host arrays [ng][8]; the per-LayerNorm arrays [ng]. */
int sbev_layer_norm_bwd_rows(const float* dY, const float* X, const float* gamma, const float* beta, float eps, int relu,
                             float* dX, float* workspace, int64_t M, int N, sbev_stream_t stream);
int sbev_layer_norm_param_group(const float* const* dYs, const float* const* Xs, const float* const* stats,
                                const float* const* gammas, const float* const* betas, float* const* dgammas, float* const* dbetas,
                                const int32_t* Ns, const int32_t* nsegs, const int32_t* relus, const int32_t* accumulate,
                                int ng, int64_t M, sbev_stream_t stream);
/* Grouped bias gradients: out_b[n] (+)= sum_s colsum(seg_b[s] [M, N_b]) for ng <= 16 biases with <= 8 gradient matrices each (the
 * layers of a call sharing the bias) in ONE launch.  segs: host array [ng][8] of device pointers; outs / Ns / nsegs / accumulate: [ng]. */
int sbev_colsum_group(const float* const* segs, float* const* outs, const int32_t* Ns, const int32_t* nsegs,
                      const int32_t* accumulate, int ng, int64_t M, sbev_stream_t stream);
int sbev_bias_relu_bwd_acc(const float* dY, const float* Y, float* dZ, float* db, int64_t M, int N, int64_t ld,
                           float* workspace, int accumulate, sbev_stream_t stream);
int sbev_layer_norm_bwd_acc(const float* dY, const float* X, const float* gamma, const float* beta, float eps, int relu,
                            float* dX, float* dgamma, float* dbeta, float* workspace, int64_t M, int N, int accumulate, sbev_stream_t stream);

/* sbev_linear3_ln_relu_f32 that also stores the Linear's pre-LayerNorm output `pre` [M,N] (may be NULL). */
int sbev_linear3_ln_relu_ex_f32(const float* x, int64_t ldx, const float* w, const float* b,
                                const float* ln_w, const float* ln_b, float eps, float* y, float* pre,
                                int64_t M, int N, sbev_stream_t stream);

/* Backward of sbev_adaptive_mixing_f32 (the forward is recomputed inside): grad_y [BQ,G,Pout,C] ->
 * grad_x [BQ,G,Pin,C], grad_params [BQ,G,C*C+Pout*Pin]. */
int sbev_adaptive_mixing_bwd_f32(const float* x, const float* params, const float* grad_y, float* grad_x, float* grad_params,
                                 int64_t BQ, int G, int Pin, int C, int Pout, float eps, sbev_stream_t stream);
/* the same, also writing item_max[BQ*G][4]: four partial maxima of |grad_params| per (row, group), their maximum = that item's
 * maximum: sbev_f16s_tensor_scale of the array is the fp16 scale of grad_params without another pass over its 118 MB */
int sbev_adaptive_mixing_bwd_max_f32(const float* x, const float* params, const float* grad_y, float* grad_x, float* grad_params,
                                     float* item_max, int64_t BQ, int G, int Pin, int C, int Pout, float eps, sbev_stream_t stream);

/* Training forward of sbev_sasa_f32 with attention dropout (mmcv MultiheadAttention attn_drop; keep decisions are a
 * hash of (seed, b, h, i, j)), and the backward of either forward: grad_out [B,Q,H*32] -> grad_qkvt [B,Q,ld]
 * (q | k | v | tau columns, padding zeroed).  `out` is the forward output; workspace = 2*B*H*Q floats. */
int sbev_sasa_train_fwd_f32(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                            const uint8_t* mask, float* out, int B, int Q, int H, int head_dim,
                            float attn_drop, uint64_t seed, sbev_stream_t stream);
int sbev_sasa_bwd_f32(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range,
                      const uint8_t* mask, const float* out, const float* grad_out, float* grad_qkvt,
                      float* workspace, int B, int Q, int H, int head_dim, float attn_drop, uint64_t seed,
                      sbev_stream_t stream);

/* sbev_msmv_bwd with (a) grad_out in either output layout of sbev_msmv_fwd (SBEV_OUT_REF / SBEV_OUT_MIX with T, G) and
 * (b) grad_feats == NULL when the features need no gradient (no atomics are issued at all). */
int sbev_msmv_bwd_ex(const void* const* feats, void* const* grad_feats, const int32_t* hw, int L,
                     int64_t Bp, int N, int C, int Q, int P,
                     int gdiv, const int64_t* stride_bo, int64_t stride_g, const int64_t* stride_v, int64_t stride_px,
                     const float* loc, const float* weights, const float* grad_out, int grad_out_layout, int T, int G,
                     float* grad_loc, float* grad_weights, sbev_stream_t stream);

/* Backward of sbev_project_select: grad_loc [B*T*G,Q,P,3] -> grad_points [B,Q,T,G*P,3] through the camera the forward
 * selected (re-selected bit-exactly).  models/sparsebev_sampling.py:49-114. */
int sbev_project_select_bwd(const float* sample_points, const float* lidar2img, const float* grad_loc,
                            int B, int Q, int T, int N, int G, int P, float image_h, float image_w, float eps,
                            float* grad_points, sbev_stream_t stream);

/* Backward of sbev_sampling_front: grad_points [B,Q,T,G*P,3] and grad_weights_bp [B*G*T,Q,P,L] (either may be NULL) ->
 * grad_offset (first G*P*3 columns) and grad_logits (first G*P*L columns) of rows with stride ld_grad -- e.g. the two
 * column blocks of one packed [B*Q, G*P*(3+L)] gradient --, grad_bbox [B*Q,10] (or NULL; velocity is detached). */
int sbev_sampling_front_bwd(const float* query_bbox, const float* offset, int64_t ld_off, const float* logits, int64_t ld_logit,
                            const double* pc_range, int B, int Q, int T, int G, int P, int L,
                            const float* grad_points, const float* grad_weights_bp,
                            float* grad_offset, float* grad_logits, int64_t ld_grad, float* grad_bbox, sbev_stream_t stream);

/* Backward of sbev_refine_bbox (code_size 10): grad_out, the forward's `out` and the proposal -> grad_reg, grad_bbox (NULL ok). */
int sbev_refine_bbox_bwd(const float* grad_out, const float* out, const float* query_bbox, const float* vel_div,
                         float* grad_reg, float* grad_bbox, int B, int Q, sbev_stream_t stream);

/* y = x * keep / (1 - p), keep_i = hash(seed, i) >= p: the mmcv FFN dropouts; the backward is the same call on grad_y. */
int sbev_dropout_f32(const float* x, float* y, int64_t n, uint64_t seed, float p, sbev_stream_t stream);

/* The three dropout launches with a part of the seed in DEVICE memory (round 4): the keep decisions hash `seed + *seed_dev` (seed_dev
 * null: `seed` alone, i.e. the calls above).  A training step captured as a hipGraph freezes every by-value argument, so the host
 * seeds of its dropout sites stay what they were at capture; the word behind `seed_dev` is what the caller changes between replays
 * (sparsebev_amd/train_graph.py refreshes it with one small launch before each replay).  Forward and backward of a site must see the
 * same word: change it only between steps. */
int sbev_sasa_train_fwd_f32_ds(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range, const uint8_t* mask,
                               float* out, int B, int Q, int H, int head_dim, float attn_drop, uint64_t seed, const uint64_t* seed_dev,
                               sbev_stream_t stream);
int sbev_sasa_bwd_f32_ds(const float* qkvt, int64_t ld, const float* query_bbox, const double* pc_range, const uint8_t* mask,
                         const float* out, const float* grad_out, float* grad_qkvt, float* workspace, int B, int Q, int H, int head_dim,
                         float attn_drop, uint64_t seed, const uint64_t* seed_dev, sbev_stream_t stream);
int sbev_dropout_f32_ds(const float* x, float* y, int64_t n, uint64_t seed, const uint64_t* seed_dev, float p, sbev_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Whole-decoder runtime: ONE call enqueues every kernel of every layer (18 launches per layer, DESIGN_HISTORY.md section 4) on `stream`.
 * Replaces: the Python control flow of SparseBEVTransformerDecoder.forward / ...DecoderLayer.forward
 *           (models/sparsebev_transformer.py:56-101,162-193) at inference.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct sbev_decoder_config {
    int32_t B, Q, T, N, G, P, L;        /* batch, queries, frames, views (6), groups (4), points, levels */
    int32_t D, H, ffn;                  /* embed dims (256), heads (8), FFN hidden (512) */
    int32_t num_classes, code_size, num_layers, out_points;
    int32_t attn_in_rows;               /* rows of attn_in_w: 3*D + H rounded up to a multiple of 4 */
    int32_t feat_dtype;                 /* enum sbev_dtype of the pyramid */
    int32_t hw[SBEV_MAX_LEVELS][2];     /* (H_l, W_l) */
    float image_h, image_w, eps_homo;   /* img_shape and the 1e-5 of sampling_4d */
    int32_t gemm_mode;                  /* enum sbev_gemm_mode for the two large mixing GEMMs (0 = exact fp32) */
    int32_t n_slots;                    /* 0: feats_nhwc[l] = [B*T*N, H, W, D] (dense); > 0: online frame ring [B, n_slots, N, H, W, D] */
    int32_t frame_slots[SBEV_MAX_FRAMES]; /* ring only: physical slot of logical frame t (see sbev_msmv_fwd_ring) */
    int32_t overlap;                    /* != 0: run the parameter-generator GEMM and the classification branch on an
                                           internal second stream (created once per process) beside the sampling chain /
                                           regression branch, joined back into `stream` with events.  The side stream and
                                           its events are process-wide: calls with overlap != 0 must not run concurrently
                                           from several host threads (overlap == 0, the default, has no shared state) */
    double pc_range[6];
} sbev_decoder_config;

/* Device pointers of the shared decoder layer's parameters (reference state-dict names in comments, prefix
 * decoder.decoder_layer.).  Linear weights are [out, in] row-major exactly as nn.Linear stores them. */
typedef struct sbev_decoder_weights {
    const float *pe0_w, *pe0_b, *pe1_g, *pe1_b, *pe3_w, *pe3_b, *pe4_g, *pe4_b;  /* position_encoder.{0,1,3,4} */
    const float *attn_in_w, *attn_in_b;   /* rows: self_attn.attention.attn.in_proj_{weight,bias}, then self_attn.gen_tau.*, then zero padding */
    const float *attn_out_w, *attn_out_b; /* self_attn.attention.attn.out_proj.* */
    const float *samp_w, *samp_b;         /* rows: sampling.sampling_offset.*, then sampling.scale_weights.* */
    const float *pg_w, *pg_b, *op_w, *op_b;                       /* mixing.parameter_generator.*, mixing.out_proj.* */
    const uint16_t *pg_w2, *op_w2;                                /* their sbev_split_bf16x3_weights images (gemm_mode 1; else NULL) */
    const float *ffn0_w, *ffn0_b, *ffn1_w, *ffn1_b;               /* ffn.layers.0.0.*, ffn.layers.1.* */
    const float *norm1_g, *norm1_b, *norm2_g, *norm2_b, *norm3_g, *norm3_b;
    const float *cls0_w, *cls0_b, *cls1_g, *cls1_b, *cls3_w, *cls3_b, *cls4_g, *cls4_b, *cls6_w, *cls6_b;  /* cls_branch.* */
    const float *reg0_w, *reg0_b, *reg2_w, *reg2_b, *reg4_w, *reg4_b;                                        /* reg_branch.* */
    const float *chain_pack;   /* sbev_decoder_chain_pack image of the small Linears' weights, or NULL (op-by-op launches) */
    const uint16_t *pg_ws;     /* gemm_mode 2 / 3: sbev_pack_bf16s_frags image of pg_w (3 / 2 images); else NULL */
    const uint16_t *op_wp;     /* gemm_mode 2 / 3: sbev_pack_bf16s_frags image of op_w (3 / 2 images); else NULL */
    /* gemm_mode 4 / 5 (fp16 hi + lo): pg_ws / op_wp are the sbev_pack_f16s_frags images (per-row scales) and */
    const float *pg_wdown;     /*   the 2^-e row of pg_w's scales [G * (C*C + out_points * T*P)] */
    const float *op_nscale;    /*   sbev_f16s_out_scale(op_w's 2^-e row, sbev_decoder_mixed_up_log2(cfg)) [embed_dims] */
    const float *pg_xscale;    /*   {2^e, 2^-e} (device memory) for the generator's input, norm1's output: the largest e with
                                *   (sqrt(embed_dims - 1) max|norm1_g| + max|norm1_b|) 2^e < 65504 */
} sbev_decoder_weights;

/*
 * Row chains.  Every op of a decoder layer except the self attention, the sampler and the two big mixing GEMMs is
 * row-local (position_encoder, attention in/out projections, norm1-3, sampling Linear, ffn, cls_branch, reg_branch,
 * refine_bbox: models/sparsebev_transformer.py:166-183), so with `chain_pack` set sbev_decoder_forward runs them as three
 * launches per layer with the rows in LDS (6 launches per layer instead of 17; the sample-point projection runs inside too).  The kernels stream the weights in a
 * lane-ordered layout: sbev_decoder_chain_pack writes that image (sbev_decoder_chain_pack_floats floats, 16-byte aligned,
 * 0 = config not covered: needs embed_dims 256, ffn 512, code_size 10) from the weight pointers of `weights`; re-pack when
 * the weights change.  Results agree with the op-by-op launches to fp32 round-off (different summation order), not bit for bit.
 * sbev_decoder_row_chain(0) makes sbev_decoder_forward ignore chain_pack (default 1).
 */
int64_t sbev_decoder_chain_pack_floats(const sbev_decoder_config* cfg);
int sbev_decoder_chain_pack(const sbev_decoder_config* cfg, const sbev_decoder_weights* weights, float* out, sbev_stream_t stream);
int sbev_decoder_row_chain(int enable);

/* Pair mode of the tail chain (round 4).  A chain workgroup streams every weight of its chain through its CU's ~60 GB/s L2 path, so
 * where two workgroups per row block fit the device in ONE round (<= ~1000 rows at 8 rows per pair, <= ~2000 at 16, on 256 CUs) the
 * tail runs on PAIRS: the members split ffn.layers.0 by columns and ffn.layers.1 by k (partial sums exchanged), take one branch each
 * (classification | regression + refine_bbox + the next position encoder) and split the attention in-projection by columns -- half
 * the weight stream per CU, two hand-offs through write-through stores and one arrival counter per pair in the decoder workspace
 * (zeroed by the layer's attention chain; bounded polls).  Results equal the single-workgroup chain to fp32 round-off.
 * sbev_decoder_chain_pair(0 / 1) switches it (returns the previous setting; default 1, SBEV_NO_CHAIN_PAIR=1 in the environment
 * starts with 0).  sbev_decoder_chain_pair_timeouts: hand-offs whose partner did not arrive within the poll bound (~1 s) since the
 * library was loaded -- 0 unless something kept half of a pair off the GPU; the results of such a launch are undefined; synchronises
 * the device; -1 on a HIP error. */
int sbev_decoder_chain_pair(int enable);
int64_t sbev_decoder_chain_pair_timeouts(void);
/* The same event on the NORMAL path (round 5): a timed-out hand-off also bumps a sticky word in pinned host memory, read without
 * synchronising anything.  While sbev_decoder_chain_pair_faults() > 0, sbev_decoder_forward (and sbev_graph_launch of a captured
 * step) refuses with SBEV_EFAULT: the step that faulted -- the PREVIOUS one or an earlier one, its launches were asynchronous -- holds
 * wrong rows and must be repeated.  The refusing call switches pair mode off (single-workgroup tail from here on: no hand-offs, no
 * faults); the caller acknowledges with sbev_decoder_chain_pair_faults_ack() (returns the count) and re-runs.  A caller that
 * synchronises on its own can ask sbev_decoder_chain_pair_faults() right after its synchronisation and learn about the CURRENT step.
 * Pair mode is only used when the word could be installed (hipHostMalloc). */
int64_t sbev_decoder_chain_pair_faults(void);
int64_t sbev_decoder_chain_pair_faults_ack(void);
/* In-launch fold of the out-projection's split-K slabs (round 6; fp16 GEMM modes, row chains on, every workgroup of the out-projection
 * resident at once: <= ~1000 rows on 256 CUs).  The S chunk-workgroups of a row tile write their slab through, meet at a counter in the
 * decoder workspace (zeroed by the layer's attention chain, like the pair counters) and each sums ceil(rows / S) rows of the tile over all
 * slabs in slab order -- bit-identical to the tail's own sum -- so the tail reads ONE row block instead of S (32 x 8 rows x 1 KB per
 * workgroup, twice per pair: 59 MB per layer at 900 rows).  Polls are bounded like the pair hand-off's; a row tile that never completes
 * raises the same fault word (sbev_decoder_chain_pair_faults) and counts in sbev_decoder_chain_pair_timeouts.  Follows the pair switch
 * (sbev_decoder_chain_pair(0) turns both off); sbev_decoder_out_fold(0 / 1) switches it alone (returns the previous setting).  DEFAULT 0:
 * measured at config 2 the fold costs the out-projection 12.6 us and saves the tail 4 -- an A/B switch, not the
 * product path; SBEV_OUT_FOLD=1 in the environment starts with 1.  sbev_debug_out_fold_drop: test hook, never set in production. */
int sbev_decoder_out_fold(int enable);
/* Rows from which the fp16-mode out-projection with the pre-split operand (the decoder's path) runs on 256-row tiles
 * (gemm_bf16s_out8_kernel: two row halves as phase groups sharing one W ring -- a third less operand delivery per product than the 128-row
 * kernel; default 1024: the batch configs and the 1600-query config; env SBEV_OUT8_MIN_ROWS; 0 = never, other values are raised to 1024).  Returns the previous threshold. */
int sbev_linear_out8_min_rows(int rows);
int sbev_debug_out_fold_drop(int enable);
/* Per-device setup that is illegal under stream capture (hipHostMalloc + hipMemcpyToSymbol of the fault word above): call once per device
 * before the first capture that may contain a decoder step -- the ctypes binding does when it loads the library in a process that already has its device context.  Without it
 * sbev_decoder_workspace_bytes attempts the same; an attempt that fails (e.g. made inside a torch.cuda.graph capture) is retried by later
 * calls, with one message on stderr / sbev_last_error(), and pair mode stays off meanwhile.  The word is shared by all devices of a process. */
int sbev_init(void);
/* Test hook for the poll bound: with 1, one member of the first pair of every pair-mode tail exits at once; its partner must time out
 * (sbev_decoder_chain_pair_timeouts grows, the launch ends after about a second per hand-off, only that pair's 8 rows are wrong).
 * Returns the previous setting.  Never set in production. */
int sbev_debug_chain_pair_drop(int enable);

/* Kernel launches per layer sbev_decoder_forward enqueues for this config and weight set under the current process-wide switches
 * (row chains, gather + mixing fusion), -1 on an invalid config: 6 with the row chains, 17 op by op, + 1 for the two-launch gather
 * + mixing, + 1 for the activation split of the split-bf16 GEMM modes. */
int sbev_decoder_launches_per_layer(const sbev_decoder_config* cfg, const sbev_decoder_weights* weights);

/* fp16 GEMM modes: log2 of the power of two the out-projection's input (relu(LayerNorm) over out_points * embed_dims / G elements:
 * |x| <= sqrt(n - 1)) is multiplied by before its fp16 split inside sbev_decoder_forward: 9 for n = 8192. */
int sbev_decoder_mixed_up_log2(const sbev_decoder_config* cfg);

/* Bytes of scratch sbev_decoder_forward needs for this config (-1 on an invalid config). */
int64_t sbev_decoder_workspace_bytes(const sbev_decoder_config* cfg);

/*
 * feats_nhwc  host array [L] of device pointers, level l = [B*T*N, H_l, W_l, D] channels-last (fp32 / bf16)
 * query_bbox  [B,Q,10], query_feat [B,Q,D], time_diff [B,T], lidar2img [B,T*N,4,4], vel_div [B] (or NULL if T == 1),
 * attn_mask   optional uint8 [Q,Q] (query denoising), NULL at inference
 * cls_out     [num_layers,B,Q,num_classes], bbox_out [num_layers,B,Q,code_size]  (NOT nan_to_num'ed)
 * workspace   256-byte aligned device scratch of >= sbev_decoder_workspace_bytes(cfg) bytes
 */
int sbev_decoder_forward(const sbev_decoder_config* cfg, const sbev_decoder_weights* weights,
                         const void* const* feats_nhwc, const float* query_bbox, const float* query_feat,
                         const float* time_diff, const float* lidar2img, const float* vel_div,
                         const uint8_t* attn_mask, float* cls_out, float* bbox_out,
                         void* workspace, int64_t workspace_bytes, sbev_stream_t stream);

/*
 * On-demand relayout (round 6).  The reference regroups EVERY pixel of every level on every call (models/sparsebev_transformer.py:73-85);
 * the gather reads the 4 bilinear corners of each sample point per level (models/csrc/msmv_sampling/msmv_sampling_forward.cu:41-66,
 * sparsebev_sampling.py:88-109) -- under half of the pyramid.  sbev_decoder_forward_lazy takes the reference's NCHW maps
 * [B*T*N, 256, H_l, W_l] as SOURCES (direct pointers, or entries of a device pointer table read at launch time: the replayable form)
 * and channels-last buffers of the same sizes as destinations (feats_nhwc; contents arbitrary): every layer's camera selection marks
 * the 64-pixel x 64-channel units its points read, one launch behind it moves the marked units this step has not moved yet.  Outputs are
 * bit-identical to sbev_nchw_to_nhwc_* followed by sbev_decoder_forward.  Dense pyramids of 4 groups x 64 channels
 * (sbev_decoder_lazy_supported); fp32 / bf16 / fp16 storage; 16-byte aligned sources and destinations.  Every argument after `lazy` as in
 * sbev_decoder_forward.
 */
typedef struct sbev_lazy_feats {
    const void* const* table;              /* device array of source pointers, or NULL */
    int32_t index[SBEV_MAX_LEVELS];        /* table != NULL: level l's NCHW source is table[index[l]] */
    const void* src[SBEV_MAX_LEVELS];      /* table == NULL: level l's NCHW source */
} sbev_lazy_feats;
int sbev_decoder_lazy_supported(const sbev_decoder_config* cfg);
/* Layers 1.. find and move the units their points marked and no earlier launch moved ("scan").  Up to 1024 rows in the fp16 GEMM modes that
 * scan rides in the generator GEMM's prologue (every generator workgroup takes its share before its first request: five launches per step
 * less at config 2); sbev_decoder_lazy_scan_launch(1) (env SBEV_LAZY_SCAN_LAUNCH=1) keeps it a launch of its own.  Bit-identical; returns
 * the previous setting.  Read when a step is enqueued / captured. */
int sbev_decoder_lazy_scan_launch(int enable);
int sbev_decoder_forward_lazy(const sbev_decoder_config* cfg, const sbev_decoder_weights* weights, void* const* feats_nhwc,
                              const sbev_lazy_feats* lazy, const float* query_bbox, const float* query_feat,
                              const float* time_diff, const float* lidar2img, const float* vel_div,
                              const uint8_t* attn_mask, float* cls_out, float* bbox_out,
                              void* workspace, int64_t workspace_bytes, sbev_stream_t stream);
/* The pieces, for callers that run the layers themselves: tiles of a pyramid (one 4-byte word of `need` and of `done` each; -1: shape not
 * covered); the marking form of sbev_sample_and_project (hw: [L][2] = (H_l, W_l); byte g of a tile's `need` word := 1 when a point of
 * group g reads one of its pixels); the move (first != 0: the step's first launch -- one workgroup per tile, `done` rebuilt from `need`;
 * else only need & ~done; last != 0 additionally clears `need` for the next step).  A stale `need` only moves more units. */
int64_t sbev_lazy_relayout_tiles(int n_levels, const int32_t* hw_pixels, int64_t n_images, int channels);
int sbev_sample_and_project_touch(const float* query_bbox, const float* offset, int64_t ld_offset,
                                  const float* scale_logits, int64_t ld_logits,
                                  const float* time_diff, const float* lidar2img, const double* pc_range,
                                  int B, int Q, int T, int N, int G, int P, int L,
                                  float image_h, float image_w, float eps,
                                  float* loc_bp, float* weights_bp, const int32_t* hw, uint32_t* need, sbev_stream_t stream);
int sbev_nchw_to_nhwc_lazy(const void* const* table, const int32_t* index, const void* const* src, void* const* out, int n_levels,
                           const int32_t* hw_pixels, int64_t n_images, int channels, int dtype, uint32_t* need, uint32_t* done, int first,
                           int last, sbev_stream_t stream);

/*
 * hipGraph capture of one decoder step: the launch sequence of sbev_decoder_forward is static per (config, pointer
 * set), so it can be recorded once on `stream` (explicit, non-default; nothing executes during the capture) and
 * replayed per sample.  The graph reads its inputs through the captured device pointers: refresh them in place.
 * The workspace and every input / output buffer must outlive the graph.  Replaces per-call Python/launch overhead
 * of the reference's eager module chain (models/sparsebev_transformer.py:86-97).
 */
typedef struct sbev_graph sbev_graph;
int sbev_decoder_capture(const sbev_decoder_config* cfg, const sbev_decoder_weights* weights,
                         const void* const* feats_nhwc, const float* query_bbox, const float* query_feat,
                         const float* time_diff, const float* lidar2img, const float* vel_div,
                         const uint8_t* attn_mask, float* cls_out, float* bbox_out,
                         void* workspace, int64_t workspace_bytes, sbev_stream_t stream, sbev_graph** out);
/* Capture of ANY sequence of this library's launches on `stream` (explicit, non-default) between the two calls -- e.g. the per-level
 * NCHW -> NHWC relayout followed by sbev_decoder_forward: the whole per-sample step as one replayable graph.  Nothing executes
 * during the capture; every buffer the recorded launches read or write must outlive the graph.  sbev_capture_end(stream, NULL)
 * abandons a capture (after a failed call inside it). */
int sbev_capture_begin(sbev_stream_t stream);
int sbev_capture_end(sbev_stream_t stream, sbev_graph** out);
int sbev_graph_launch(sbev_graph* graph, sbev_stream_t stream);
int64_t sbev_graph_num_nodes(const sbev_graph* graph);   /* kernel nodes recorded (-1 for NULL) */
int sbev_graph_destroy(sbev_graph* graph);

/* Bracket launches with HIP events on their stream -- `enable` is a bit mask of kinds: 1 = every sbev_msmv_fwd launch,
 * 2 = the parameter-generator GEMM, 4 = the out-projection GEMM, 8 = the fused gather + mixing launch, 0 = off -- and read back + clear the elapsed times in ms
 * (blocks until those launches finished).  Measurement aid for bench.py's roofline figures. */
int sbev_profile_sampler(int enable);
int sbev_profile_sampler_read(float* ms, int max_n);
/* The events are not free (two records around a launch leave ~5.6 us of idle stream each -- 2 % of a decoder step for the
 * six sampler launches): bracket the launches of only every n-th sbev_decoder_forward call (n = 1: every call, the default).
 * Resets the call counter, so the first call after it is a bracketed one. */
int sbev_profile_stride(int every_n_calls);
/* Same for the other bracketed launches: kind 0 = sampler, 1 = parameter-generator GEMM, 2 = out-projection GEMM, 3 = fused gather + mixing. */
int sbev_profile_read(int kind, float* ms, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* SBEV_HIP_H */
