// Detection-head pre / post-processing on the device: everything between the FPN features
// and the decoder input, and between the decoder output and the list of boxes, that the reference does with a chain
// of small torch ops on [B, 900, *] tensors (models/sparsebev_head.py:69-95,205-213,463-482 and
// models/bbox/coders/nms_free_coder.py:37-88, models/bbox/utils.py:26-47).
//
//   head_prepare_kernel      query_bbox = init_query_bbox.repeat(B), query_feat = [label_enc.weight[num_classes], 0].repeat(B, Q)
//   head_denorm_kernel       xyz back to metres + the (cx, cy, w, l, cz, h, sin, cos, vx, vy) column order of the head output
//   nms_free_decode_kernel   sigmoid, top-k over all Q x num_classes scores, gather, denormalize_bbox (atan2 / exp),
//                            centre-range + score-threshold mask, order-preserving compaction, optional bottom-centre z
//
// The top-k is a bitonic sort of (score key, flat index) pairs of ONE sample in the LDS of ONE 1024-thread workgroup:
// 900 x 10 scores pad to 16384 x 8 B = 128 KiB, which is exactly what a CDNA4 CU's 160 KiB LDS is for -- no global
// scratch, no second launch, and the order is total (score descending, flat index ascending on ties), where
// torch.topk leaves ties unspecified.
#include "sbev_common.hpp"

namespace {

struct PrepArgs {
    const float* init_bbox;   // [Q, 10]
    const float* label_row;   // [D - 1]
    float* qbbox;             // [B, Q, 10]
    float* qfeat;             // [B, Q, D]
    int B, Q, D;
};

__global__ void head_prepare_kernel(const PrepArgs a) {
    const long long per = (long long)a.Q * (10 + a.D);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * a.B) return;
    const int b = (int)(i / per);
    const long long r = i % per;
    const long long nb = (long long)a.Q * 10;
    if (r < nb) {
        a.qbbox[(long long)b * nb + r] = a.init_bbox[r];
    } else {
        const long long e = r - nb;
        const int d = (int)(e % a.D);
        a.qfeat[(long long)b * a.Q * a.D + e] = d < a.D - 1 ? a.label_row[d] : 0.f;     // trailing 0 = the DN indicator
    }
}

struct DenormArgs {
    const float* in;   // [n, 10] (x, y, z normalised, w, l, h, sin, cos, vx, vy)
    float* out;        // [n, 10] (cx, cy, w, l, cz, h, sin, cos, vx, vy), centre in metres
    long long n;
    float lo[3], span[3];
};

__global__ void head_denorm_kernel(const DenormArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float* p = a.in + i * 10;
    float* o = a.out + i * 10;
    // x * (max - min) + min with the two roundings of the reference's separate mul and add (sparsebev_head.py:85-87)
    const float cx = __fadd_rn(__fmul_rn(p[0], a.span[0]), a.lo[0]);
    const float cy = __fadd_rn(__fmul_rn(p[1], a.span[1]), a.lo[1]);
    const float cz = __fadd_rn(__fmul_rn(p[2], a.span[2]), a.lo[2]);
    const float w = p[3], l = p[4], h = p[5], s = p[6], c = p[7], vx = p[8], vy = p[9];
    o[0] = cx; o[1] = cy; o[2] = w; o[3] = l; o[4] = cz; o[5] = h; o[6] = s; o[7] = c; o[8] = vx; o[9] = vy;
}

struct DecodeArgs {
    const float* cls;    // [B, Q, NC] logits
    const float* bbox;   // [B, Q, 10] head format (cx, cy, w, l, cz, h, sin, cos, vx, vy)
    float* boxes;        // [B, max_num, 9] (cx, cy, cz, w, l, h, rot, vx, vy)
    float* scores;       // [B, max_num]
    int* labels;         // [B, max_num]
    int* count;          // [B]
    int Q, NC, max_num;
    int use_thr, bottom;
    float thr;
    float lim[6];
};

// ascending u32 order == descending float order (NaN logits sort first, as "largest", like torch.topk)
__device__ __forceinline__ unsigned desc_key(float x) {
    const unsigned u = __float_as_uint(x);
    const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

template <int NP>
__global__ __launch_bounds__(1024) void nms_free_decode_kernel(const DecodeArgs a) {
    __shared__ unsigned long long keys[NP];
    __shared__ int wave_cnt[16];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int n = a.Q * a.NC;
    const float* cls = a.cls + (long long)b * n;
    for (int i = tid; i < NP; i += 1024)
        keys[i] = i < n ? (((unsigned long long)desc_key(cls[i]) << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < NP / 2; t += 1024) {
                const int i = 2 * t - (t & (j - 1));      // the pair member with bit j clear
                const int l = i + j;
                const bool up = (i & k) == 0;
                const unsigned long long x = keys[i], y = keys[l];
                if ((x > y) == up) {
                    keys[i] = y;
                    keys[l] = x;
                }
            }
            __syncthreads();
        }
    }
    // ---- the first max_num entries: gather, denormalise, mask ------------------------------------------------------
    const int kmax = a.max_num < n ? a.max_num : n;
    bool keep = false;
    float bx[9], score = 0.f;
    int label = 0;
    if (tid < kmax) {
        const unsigned idx = (unsigned)(keys[tid] & 0xffffffffu);
        const int q = idx / a.NC;
        label = idx % a.NC;
        score = 1.f / (1.f + expf(-cls[idx]));
        const float* p = a.bbox + ((long long)b * a.Q + q) * 10;
        bx[0] = p[0]; bx[1] = p[1]; bx[2] = p[4];                        // cx, cy, cz
        bx[3] = expf(p[2]); bx[4] = expf(p[3]); bx[5] = expf(p[5]);      // w, l, h
        bx[6] = atan2f(p[6], p[7]);
        bx[7] = p[8]; bx[8] = p[9];
        keep = bx[0] >= a.lim[0] && bx[1] >= a.lim[1] && bx[2] >= a.lim[2] && bx[0] <= a.lim[3] && bx[1] <= a.lim[4] && bx[2] <= a.lim[5];
        if (a.use_thr) keep = keep && score > a.thr;
        if (a.bottom) bx[2] = __fsub_rn(bx[2], __fmul_rn(bx[5], 0.5f));  // gravity centre -> bottom centre (sparsebev_head.py:471)
        if (a.bottom == 2) {                                              // 'v0.17.1' layout (sparsebev_head.py:472-476)
            const float w = bx[3];
            bx[3] = bx[4];
            bx[4] = w;
            bx[6] = __fsub_rn(-bx[6], 1.5707963267948966f);
        }
    }
    // ---- order-preserving compaction (boxes3d[mask]) ---------------------------------------------------------------
    const unsigned long long ballot = __ballot(keep);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(ballot);
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        const int c = wave_cnt[w];
        base += w < wave ? c : 0;
        total += c;
    }
    const int pos = base + __popcll(ballot & ((1ull << lane) - 1ull));
    float* ob = a.boxes + (long long)b * a.max_num * 9;
    float* os = a.scores + (long long)b * a.max_num;
    int* ol = a.labels + (long long)b * a.max_num;
    if (keep) {
#pragma unroll
        for (int d = 0; d < 9; ++d) ob[pos * 9 + d] = bx[d];
        os[pos] = score;
        ol[pos] = label;
    }
    if (tid >= total && tid < a.max_num) {          // rows past the count are zero-filled
#pragma unroll
        for (int d = 0; d < 9; ++d) ob[tid * 9 + d] = 0.f;
        os[tid] = 0.f;
        ol[tid] = 0;
    }
    if (tid == 0) a.count[b] = total;
}

}  // namespace

extern "C" int sbev_head_prepare(const float* init_query_bbox, const float* label_row, float* query_bbox, float* query_feat,
                                 int B, int Q, int D, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && D >= 2, "sbev_head_prepare: bad sizes B=%d Q=%d D=%d", B, Q, D);
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(init_query_bbox && label_row && query_bbox && query_feat, "sbev_head_prepare: null pointer");
    PrepArgs a{init_query_bbox, label_row, query_bbox, query_feat, B, Q, D};
    const long long total = (long long)B * Q * (10 + D);
    SBEV_REQUIRE((total + 255) / 256 <= 0x7fffffffLL, "sbev_head_prepare: too many elements");
    hipLaunchKernelGGL(head_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_head_prepare");
}

extern "C" int sbev_head_denorm(const float* bbox_norm, const double* pc_range, float* out, int64_t n, sbev_stream_t stream) {
    SBEV_REQUIRE(n >= 0, "sbev_head_denorm: bad size");
    if (n == 0) return SBEV_OK;
    SBEV_REQUIRE(bbox_norm && pc_range && out, "sbev_head_denorm: null pointer");
    DenormArgs a{};
    a.in = bbox_norm; a.out = out; a.n = n;
    for (int i = 0; i < 3; ++i) {     // python-float scalars cast to fp32 by the tensor op (sparsebev_head.py:85-87)
        a.lo[i] = (float)pc_range[i];
        a.span[i] = (float)(pc_range[3 + i] - pc_range[i]);
    }
    SBEV_REQUIRE((n + 255) / 256 <= 0x7fffffffLL, "sbev_head_denorm: too many rows");
    hipLaunchKernelGGL(head_denorm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_head_denorm");
}

extern "C" int sbev_nms_free_decode(const float* cls_scores, const float* bbox_preds, int B, int Q, int num_classes, int max_num,
                                    float score_threshold, int use_score_threshold, const double* post_center_range,
                                    int bottom_center, float* boxes, float* scores, int32_t* labels, int32_t* count,
                                    sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 1 && num_classes >= 1 && max_num >= 1, "sbev_nms_free_decode: bad sizes");
    SBEV_REQUIRE(max_num <= 1024, "sbev_nms_free_decode: max_num %d > 1024 (one thread per kept box)", max_num);
    const long long n = (long long)Q * num_classes;
    // torch.topk raises "selected index k out of range" (nms_free_coder.py:52)
    SBEV_REQUIRE(max_num <= n, "sbev_nms_free_decode: max_num %d > Q * num_classes = %lld (selected index k out of range)", max_num, n);
    SBEV_REQUIRE(n <= 16384, "sbev_nms_free_decode: Q * num_classes = %lld > 16384 (the per-sample sort lives in one CU's LDS)", n);
    // the reference raises NotImplementedError without a centre range (nms_free_coder.py:80-84)
    SBEV_REQUIRE(post_center_range != nullptr, "sbev_nms_free_decode: post_center_range is required (the reference supports nothing else)");
    if (B == 0) return SBEV_OK;
    SBEV_REQUIRE(cls_scores && bbox_preds && boxes && scores && labels && count, "sbev_nms_free_decode: null pointer");
    DecodeArgs a{};
    a.cls = cls_scores; a.bbox = bbox_preds; a.boxes = boxes; a.scores = scores; a.labels = labels; a.count = count;
    a.Q = Q; a.NC = num_classes; a.max_num = max_num;
    a.use_thr = use_score_threshold != 0; a.thr = score_threshold;
    a.bottom = bottom_center == 0 ? 0 : (sbev::box_convention() == SBEV_BOX_V0_17_1 ? 2 : 1);
    for (int i = 0; i < 6; ++i) a.lim[i] = (float)post_center_range[i];     // torch.tensor(list) is fp32
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n <= 2048)
        hipLaunchKernelGGL(nms_free_decode_kernel<2048>, dim3(B), dim3(1024), 0, s, a);
    else if (n <= 4096)
        hipLaunchKernelGGL(nms_free_decode_kernel<4096>, dim3(B), dim3(1024), 0, s, a);
    else if (n <= 8192)
        hipLaunchKernelGGL(nms_free_decode_kernel<8192>, dim3(B), dim3(1024), 0, s, a);
    else
        hipLaunchKernelGGL(nms_free_decode_kernel<16384>, dim3(B), dim3(1024), 0, s, a);
    return sbev::check_launch("sbev_nms_free_decode");
}
