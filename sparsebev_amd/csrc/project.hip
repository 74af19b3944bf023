// Sample-point generation, projection into the T*N cameras, camera-hit mask and view selection
// (gfx950).  Behind sbev_sampling_front / sbev_project_select (include/sbev_hip.h).
//
// This file is compiled with -ffp-contract=off AND spells the projection with explicitly rounded
// __fmul_rn / __fadd_rn / __fdiv_rn so that no FMA can be formed: the reference's batched fp32 4x4
// matmul on CPU is ((m0*x + m1*y) + m2*z) + m3 with separate roundings, and the camera-hit mask has to
// match it bit for bit ("Bit-exact hit mask"; models/sparsebev_sampling.py:49-79).
#include "sbev_common.hpp"
#include "sample_point.hpp"

namespace {

struct ProjArgs {
    const float* pts;       // [B,Q,T,GP,3]
    const float* l2i;       // [B,T*N,4,4]
    float* loc_bp;          // [B*T*G,Q,P,3]
    float* dump_uvh;        // [B,T,N,Q,GP,3] or null
    unsigned char* dump_valid;  // [B,T,N,Q,GP] or null
    int* i_view;            // [B,T,Q,GP] or null
    int B, Q, T, N, G, P;
    float image_h, image_w, eps;
};

// one thread per (b, t, q, gp); consecutive threads walk gp then q so the 12-byte point reads and the
// loc writes are as dense as the layouts allow; the 6 camera matrices of (b, t) are wave-mostly-uniform.
__global__ __launch_bounds__(256) void project_select_kernel(const ProjArgs a) {
    const int GP = a.G * a.P;
    const long long total = (long long)a.B * a.T * a.Q * GP;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // 32-bit index split (total < 2^31, host-checked): 64-bit per-thread divisions are ~100 VALU instructions each
    const unsigned ui = (unsigned)idx;
    unsigned r = ui / (unsigned)GP;
    const int gp = (int)(ui - r * (unsigned)GP);
    const unsigned r2 = r / (unsigned)a.Q;
    const int q = (int)(r - r2 * (unsigned)a.Q);
    const int b = (int)(r2 / (unsigned)a.T);
    const int t = (int)(r2 - (unsigned)b * (unsigned)a.T);

    const float* pt = a.pts + ((((long long)b * a.Q + q) * a.T + t) * GP + gp) * 3;
    const float x = pt[0], y = pt[1], z = pt[2];

    int view = 0;
    bool found = false;
    float su = 0.f, sv = 0.f;
    for (int n = 0; n < a.N; ++n) {
        const float* m = a.l2i + (((long long)b * a.T + t) * a.N + n) * 16;
        // ((m0*x + m1*y) + m2*z) + m3*1  -- every product and sum individually rounded
        const float uh = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z)), m[3]);
        const float vh = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[4], x), __fmul_rn(m[5], y)), __fmul_rn(m[6], z)), m[7]);
        const float hm = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[8], x), __fmul_rn(m[9], y)), __fmul_rn(m[10], z)), m[11]);
        const float hn = fmaxf(hm, a.eps);                                     // torch.maximum(homo, eps)
        const float u = __fdiv_rn(__fdiv_rn(uh, hn), a.image_w);               // two IEEE divides each
        const float v = __fdiv_rn(__fdiv_rn(vh, hn), a.image_h);
        const bool valid = (hm > a.eps) && (v > 0.f) && (v < 1.f) && (u > 0.f) && (u < 1.f);
        if (a.dump_uvh) {
            float* d = a.dump_uvh + (((((long long)b * a.T + t) * a.N + n) * a.Q + q) * GP + gp) * 3;
            d[0] = u; d[1] = v; d[2] = hn;
        }
        if (a.dump_valid)
            a.dump_valid[((((long long)b * a.T + t) * a.N + n) * a.Q + q) * GP + gp] = valid ? 1 : 0;
        // argmax over the 0/1 mask = first hit; with no hit argmax returns view 0 (coordinates of view 0)
        if (n == 0 || (valid && !found)) { su = u; sv = v; view = n; }
        found = found || valid;
    }
    if (a.i_view) a.i_view[(((long long)b * a.T + t) * a.Q + q) * GP + gp] = view;
    const int g = gp / a.P, p = gp - g * a.P;
    float* o = a.loc_bp + (((((long long)b * a.T + t) * a.G + g) * a.Q + q) * a.P + p) * 3;
    o[0] = su;
    o[1] = sv;
    o[2] = __fdiv_rn((float)view, (float)(a.N - 1));                           // i_view.float() / (N - 1)
}

struct FrontArgs {
    const float* bbox;      // [B,Q,10]
    const float* offset;    // [B*Q, ld_off]   (first GP*3 columns)
    const float* logits;    // [B*Q, ld_logit] (first GP*L columns)
    long long ld_off, ld_logit;
    const float* time_diff; // [B,T]
    float* pts;             // [B,Q,T,GP,3] or null
    float* w_bp;            // [B*G*T,Q,P,L] or null
    float pc_lo[3], pc_span[3];
    int B, Q, T, G, P, L;
    float rot_sign;         // +1: 'v1.0.0' rotation (x cos - y sin, x sin + y cos); -1: 'v0.17.1' (models/utils.py:66-77)
};

// one thread per (b, q, gp): decodes the box once, emits the T warped copies of its sample point and the
// level softmax of (g, p), replicated into every sample batch b' that the reference's reorder maps to g.
__global__ __launch_bounds__(256) void sampling_front_kernel(const FrontArgs a) {
    const int GP = a.G * a.P;
    const long long total = (long long)a.B * a.Q * GP;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const unsigned ui = (unsigned)idx;                    // total < 2^31 (host-checked)
    const unsigned ubq = ui / (unsigned)GP;
    const int gp = (int)(ui - ubq * (unsigned)GP);
    const long long bq = ubq;
    const int b = (int)(ubq / (unsigned)a.Q);
    const int q = (int)(ubq - (unsigned)b * (unsigned)a.Q);
    const float* bb = a.bbox + bq * 10;

    if (a.pts) {
        // decode_bbox: xyz = c * span + lo ; wlh = exp(log-dims) ; yaw = atan2(sin, cos)
        const float cx = bb[0] * a.pc_span[0] + a.pc_lo[0];
        const float cy = bb[1] * a.pc_span[1] + a.pc_lo[1];
        const float cz = bb[2] * a.pc_span[2] + a.pc_lo[2];
        const float yaw = atan2f(bb[6], bb[7]);
        const float cs = cosf(yaw), sn = a.rot_sign * sinf(yaw);      // x (+-1) is exact
        const float* of = a.offset + bq * a.ld_off + gp * 3;
        const float dx = expf(bb[3]) * of[0], dy = expf(bb[4]) * of[1], dz = expf(bb[5]) * of[2];
        // rotation about z, v1.0.0 convention: x' = x cos - y sin ; y' = x sin + y cos
        const float px = cx + (dx * cs + dy * (-sn));
        const float py = cy + (dx * sn + dy * cs);
        const float pz = cz + dz;
        const float vx = bb[8], vy = bb[9];
        for (int t = 0; t < a.T; ++t) {
            const float td = a.time_diff[b * a.T + t];
            float* o = a.pts + ((bq * a.T + t) * GP + gp) * 3;
            o[0] = px - vx * td;
            o[1] = py - vy * td;
            o[2] = pz;
        }
    }
    if (a.w_bp) {
        const int g = gp / a.P, p = gp - g * a.P;
        const float* lg = a.logits + bq * a.ld_logit + gp * a.L;
        float mx = lg[0];
        for (int l = 1; l < a.L; ++l) mx = fmaxf(mx, lg[l]);
        float e[SBEV_MAX_LEVELS];
        float sum = 0.f;
        for (int l = 0; l < a.L; ++l) {
            e[l] = expf(lg[l] - mx);
            sum += e[l];
        }
        // Reference reorder (models/sparsebev_sampling.py:117-119): weight row r = (b*G + g)*T + t' holds group g
        // for every t'; the sampler reads row b' = (b*T + t)*G + g_sample.  So group g's softmax lands in the
        // T rows r = b*G*T + g*T + t', t' = 0..T-1 (quirk q1 reproduced by construction).
        for (int tp = 0; tp < a.T; ++tp) {
            const long long row = ((long long)b * a.G + g) * a.T + tp;
            float* o = a.w_bp + ((row * a.Q + q) * a.P + p) * a.L;
            for (int l = 0; l < a.L; ++l) o[l] = e[l] / sum;
        }
    }
}

using FusedArgs = sbev_ops::SamplePointArgs;      // sample_point.hpp (shared with the row-chain kernel)

// sampling_front_kernel + project_select_kernel in one launch (the decoder runtime's path): one thread per
// (b, t, q, gp) rebuilds its 3-D sample point from the box and the offset (the same expressions, individually rounded,
// so the point -- and therefore the hit mask -- is bit-identical to the two-kernel path) and projects it; the
// [B,Q,T,GP,3] point tensor never exists.  Every thread also writes the level softmax of its (g, p) into row t of
// group g's T weight rows.
__global__ __launch_bounds__(256) void sample_project_kernel(const FusedArgs a, const float* offset, const float* logits,
                                                             long long ld_off, long long ld_logit) {
    const int GP = a.G * a.P;
    const long long total = (long long)a.B * a.T * a.Q * GP;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // 32-bit index split (total < 2^31, host-checked): 64-bit per-thread divisions are ~100 VALU instructions each
    const unsigned ui = (unsigned)idx;
    unsigned r = ui / (unsigned)GP;
    const int gp = (int)(ui - r * (unsigned)GP);
    const unsigned r2 = r / (unsigned)a.Q;
    const int q = (int)(r - r2 * (unsigned)a.Q);
    const int b = (int)(r2 / (unsigned)a.T);
    const int t = (int)(r2 - (unsigned)b * (unsigned)a.T);
    const long long bq = (long long)b * a.Q + q;
    sbev_ops::sample_point(a, b, t, q, gp, a.bbox + bq * 10, offset + bq * ld_off + gp * 3, logits + bq * ld_logit + gp * a.L);
}

}  // namespace

extern "C" int sbev_project_select(const float* sample_points, const float* lidar2img,
                                   int B, int Q, int T, int N, int G, int P,
                                   float image_h, float image_w, float eps,
                                   float* loc_bp, float* dump_uvh, uint8_t* dump_valid, int32_t* i_view,
                                   sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && T >= 1 && N >= 1 && G >= 1 && P >= 1, "sbev_project_select: bad sizes");
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(sample_points && lidar2img && loc_bp, "sbev_project_select: null pointer");
    ProjArgs a{sample_points, lidar2img, loc_bp, dump_uvh, dump_valid, i_view, B, Q, T, N, G, P, image_h, image_w, eps};
    const long long total = (long long)B * T * Q * G * P;
    const long long blocks = (total + 255) / 256;
    SBEV_REQUIRE(total <= 0x7fffffffLL, "sbev_project_select: too many points");
    hipLaunchKernelGGL(project_select_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_project_select");
}

extern "C" int sbev_sampling_front(const float* query_bbox, const float* offset, int64_t ld_offset,
                                   const float* scale_logits, int64_t ld_logits,
                                   const float* time_diff, const double* pc_range,
                                   int B, int Q, int T, int G, int P, int L,
                                   float* sample_points, float* weights_bp, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && T >= 1 && G >= 1 && P >= 1, "sbev_sampling_front: bad sizes");
    SBEV_REQUIRE(L >= 1 && L <= SBEV_MAX_LEVELS, "sbev_sampling_front: L=%d not in 1..%d", L, SBEV_MAX_LEVELS);
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(query_bbox && pc_range, "sbev_sampling_front: null pointer");
    SBEV_REQUIRE(!sample_points || (offset && time_diff), "sbev_sampling_front: sample_points needs offset and time_diff");
    SBEV_REQUIRE(!weights_bp || scale_logits, "sbev_sampling_front: weights_bp needs scale_logits");
    SBEV_REQUIRE((!sample_points || ld_offset >= (int64_t)G * P * 3) && (!weights_bp || ld_logits >= (int64_t)G * P * L),
                 "sbev_sampling_front: row strides smaller than the rows");
    FrontArgs a{};
    a.bbox = query_bbox; a.offset = offset; a.logits = scale_logits; a.time_diff = time_diff;
    a.ld_off = ld_offset; a.ld_logit = ld_logits;
    a.pts = sample_points; a.w_bp = weights_bp;
    for (int i = 0; i < 3; ++i) {
        a.pc_lo[i] = (float)pc_range[i];                          // python float -> fp32 scalar, as torch does
        a.pc_span[i] = (float)(pc_range[3 + i] - pc_range[i]);    // difference taken in double first (models/bbox/utils.py:69-71)
    }
    a.B = B; a.Q = Q; a.T = T; a.G = G; a.P = P; a.L = L;
    a.rot_sign = sbev::box_convention() == SBEV_BOX_V0_17_1 ? -1.f : 1.f;
    const long long total = (long long)B * Q * G * P;
    SBEV_REQUIRE(total <= 0x7fffffffLL, "sbev_sampling_front: too many points");
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(sampling_front_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_sampling_front");
}

namespace sbev {
int launch_sample_and_project(const float* query_bbox, const float* offset, int64_t ld_offset, const float* scale_logits, int64_t ld_logits,
                              const float* time_diff, const float* lidar2img, const double* pc_range, int B, int Q, int T, int N, int G, int P,
                              int L, float image_h, float image_w, float eps, float* loc_bp, float* weights_bp, const LazyPlan* touch,
                              uint32_t* touch_need, const int32_t (*hw)[2], hipStream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && T >= 1 && N >= 1 && G >= 1 && P >= 1, "sbev_sample_and_project: bad sizes");
    SBEV_REQUIRE(L >= 1 && L <= SBEV_MAX_LEVELS, "sbev_sample_and_project: L=%d not in 1..%d", L, SBEV_MAX_LEVELS);
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(query_bbox && offset && scale_logits && time_diff && lidar2img && pc_range && loc_bp && weights_bp,
                 "sbev_sample_and_project: null pointer");
    SBEV_REQUIRE(ld_offset >= (int64_t)G * P * 3 && ld_logits >= (int64_t)G * P * L, "sbev_sample_and_project: row strides smaller than the rows");
    FusedArgs a = sbev_ops::sample_point_args(query_bbox, time_diff, lidar2img, pc_range, B, Q, T, N, G, P, L, image_h, image_w,
                                              eps, loc_bp, weights_bp);
    if (touch && touch_need) {
        SBEV_REQUIRE(hw && touch->n_levels == L && G <= 4 && touch->n_images == (long long)B * T * N, "sbev_sample_and_project: touch map does not match the pyramid");
        sbev_ops::sample_point_touch(a, *touch, hw, touch_need);
    }
    const long long total = (long long)B * T * Q * G * P;
    const long long blocks = (total + 255) / 256;
    SBEV_REQUIRE(total <= 0x7fffffffLL, "sbev_sample_and_project: too many points");
    hipLaunchKernelGGL(sample_project_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, offset,
                       scale_logits, (long long)ld_offset, (long long)ld_logits);
    return sbev::check_launch("sbev_sample_and_project");
}
}  // namespace sbev

extern "C" int sbev_sample_and_project(const float* query_bbox, const float* offset, int64_t ld_offset,
                                       const float* scale_logits, int64_t ld_logits,
                                       const float* time_diff, const float* lidar2img, const double* pc_range,
                                       int B, int Q, int T, int N, int G, int P, int L,
                                       float image_h, float image_w, float eps,
                                       float* loc_bp, float* weights_bp, sbev_stream_t stream) {
    return sbev::launch_sample_and_project(query_bbox, offset, ld_offset, scale_logits, ld_logits, time_diff, lidar2img, pc_range, B, Q, T, N, G, P, L,
                                           image_h, image_w, eps, loc_bp, weights_bp, nullptr, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream));
}

// the same launch, additionally marking the relayout units the selected points read (on-demand relayout, csrc/layout.hip): need = one
// 4-byte word per tile of the pyramid [B*T*N, 256, hw_l] (sbev_lazy_relayout_tiles words), byte g of a tile's word is set to 1 when a
// point of group g reads one of the tile's 64 pixels
extern "C" int sbev_sample_and_project_touch(const float* query_bbox, const float* offset, int64_t ld_offset,
                                             const float* scale_logits, int64_t ld_logits,
                                             const float* time_diff, const float* lidar2img, const double* pc_range,
                                             int B, int Q, int T, int N, int G, int P, int L,
                                             float image_h, float image_w, float eps,
                                             float* loc_bp, float* weights_bp, const int32_t* hw, uint32_t* need, sbev_stream_t stream) {
    SBEV_REQUIRE(hw && need && L >= 1 && L <= SBEV_MAX_LEVELS, "sbev_sample_and_project_touch: null pointer / bad level count");
    int32_t hw2[SBEV_MAX_LEVELS][2], s[SBEV_MAX_LEVELS];
    for (int l = 0; l < L; ++l) {
        hw2[l][0] = hw[2 * l]; hw2[l][1] = hw[2 * l + 1];
        s[l] = hw[2 * l] * hw[2 * l + 1];
    }
    sbev::LazyPlan plan;
    SBEV_REQUIRE(sbev::lazy_plan(L, s, (long long)B * T * N, 256, &plan), "sbev_sample_and_project_touch: pyramid not covered by the lazy relayout");
    return sbev::launch_sample_and_project(query_bbox, offset, ld_offset, scale_logits, ld_logits, time_diff, lidar2img, pc_range, B, Q, T, N, G, P, L,
                                           image_h, image_w, eps, loc_bp, weights_bp, &plan, need, hw2, reinterpret_cast<hipStream_t>(stream));
}
