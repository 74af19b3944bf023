// One sample point of the adaptive spatio-temporal sampling, shared by sample_project_kernel (project.hip) and the
// row-chain kernel (row_chain.hip), so that both produce the same bits from the same inputs:
//   box + learned offset -> 3-D point (models/sparsebev_sampling.py:22-40 / sparsebev_transformer.py:327-345, rotation per
//   models/utils.py:66-77), temporal warp by the box velocity (:346-352), projection into the N cameras of frame t,
//   first-hit camera selection (sparsebev_sampling.py:57-88), level softmax of the point's scale logits.
// Every product and sum of the projection is individually rounded like the reference's fp32 matmul chain (SURVEY.md
// section 7): contraction is switched off for this function body whatever the translation unit's flags are.
#pragma once
#include <cmath>

#include "sbev_common.hpp"

namespace sbev_ops {

// On-demand relayout (round 6, csrc/layout.hip "lazy relayout"): the thread that selects a point's camera also marks the relayout units
// the gather will read for it -- per level the (up to) 4 bilinear corners under the sampler's own rule (msmv_chunk.inc phase 1 ==
// msmv_sampling_forward.cu:41-66).  A unit = 64 consecutive pixels of one image of one level x the 64 channels of the point's group;
// one BYTE per unit, set with a plain store (idempotent: no atomics, any number of writers), word = the 4 groups of a tile.
struct TouchMap {
    unsigned char* need;                   // null: off
    int H[SBEV_MAX_LEVELS], W[SBEV_MAX_LEVELS];
    unsigned tiles[SBEV_MAX_LEVELS];       // ceil(H * W / 64) per image
    unsigned base[SBEV_MAX_LEVELS];        // first tile of level l (levels stacked, images inside a level)
};

struct SamplePointArgs {
    const float* bbox;      // [B,Q,10]
    const float* time_diff; // [B,T]
    const float* l2i;       // [B,T*N,4,4]
    float* loc_bp;          // [B*T*G,Q,P,3]
    float* w_bp;            // [B*G*T,Q,P,L]
    float pc_lo[3], pc_span[3];
    int B, Q, T, N, G, P, L;
    float image_h, image_w, eps;
    float rot_sign;         // +1: 'v1.0.0' rotation (x cos - y sin, x sin + y cos); -1: 'v0.17.1' (models/utils.py:66-77)
    TouchMap touch;
};

// the units (b, t, view, group g) reads at normalised image coordinates (u, v): exactly the corners msmv_chunk.inc loads
__device__ __forceinline__ void touch_units(const SamplePointArgs& a, int b, int t, int g, float u, float v, float zview) {
#pragma clang fp contract(off)
    int view = (int)roundf(zview * (float)(a.N - 1));         // the sampler's own reading of loc.z
    view = min(max(view, 0), a.N - 1);
    const unsigned img = (unsigned)((b * a.T + t) * a.N + view);
    for (int l = 0; l < a.L; ++l) {
        const int H = a.touch.H[l], W = a.touch.W[l];
        const float h_im = v * (float)(H - 1), w_im = u * (float)(W - 1);
        if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) continue;      // (NaN: nothing is read)
        const int h0 = (int)fminf(fmaxf(floorf(h_im), -1.f), (float)H), w0 = (int)fminf(fmaxf(floorf(w_im), -1.f), (float)W);
        unsigned char* row = a.touch.need + ((size_t)a.touch.base[l] + (size_t)img * a.touch.tiles[l]) * 4 + g;
        // the two corners of a row share a tile unless they straddle a 64-pixel boundary: one store for both then
        const int wa = max(w0, 0), wb = min(w0 + 1, W - 1);           // (w0 in [-1, W - 1] here: at least one corner column is inside)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            const int hc = h0 + dh;
            if (hc < 0 || hc > H - 1) continue;
            const unsigned ta = (unsigned)(hc * W + wa) >> 6, tb = (unsigned)(hc * W + wb) >> 6;
            // (plain idempotent stores; read-before-write -- most units are marked already -- measured slower: attention chain 22.6 vs 21.0 us)
            row[(size_t)ta * 4] = 1;
            if (tb != ta) row[(size_t)tb * 4] = 1;
        }
    }
}

inline SamplePointArgs sample_point_args(const float* bbox, const float* time_diff, const float* l2i, const double* pc_range, int B, int Q,
                                         int T, int N, int G, int P, int L, float image_h, float image_w, float eps, float* loc_bp,
                                         float* w_bp) {
    SamplePointArgs a{};
    a.bbox = bbox; a.time_diff = time_diff; a.l2i = l2i; a.loc_bp = loc_bp; a.w_bp = w_bp;
    for (int i = 0; i < 3; ++i) {
        a.pc_lo[i] = (float)pc_range[i];
        a.pc_span[i] = (float)(pc_range[3 + i] - pc_range[i]);
    }
    a.B = B; a.Q = Q; a.T = T; a.N = N; a.G = G; a.P = P; a.L = L;
    a.image_h = image_h; a.image_w = image_w; a.eps = eps;
    a.rot_sign = sbev::box_convention() == SBEV_BOX_V0_17_1 ? -1.f : 1.f;
    return a;
}

// host: switch the marking on for a launch (plan: layout.hip::lazy_plan of the same pyramid; hw: the config's (H_l, W_l))
inline void sample_point_touch(SamplePointArgs& a, const sbev::LazyPlan& plan, const int32_t (*hw)[2], uint32_t* need) {
    a.touch.need = reinterpret_cast<unsigned char*>(need);
    for (int l = 0; l < plan.n_levels; ++l) {
        a.touch.H[l] = hw[l][0]; a.touch.W[l] = hw[l][1];
        a.touch.tiles[l] = plan.tiles[l]; a.touch.base[l] = plan.base[l];
    }
}

// (b, t, q, gp): bb = the query's box row, of = its 3 offsets of point gp, lg = its L level logits of point gp
__device__ __forceinline__ void sample_point(const SamplePointArgs& a, int b, int t, int q, int gp, const float* bb, const float* of,
                                             const float* lg) {
#pragma clang fp contract(off)
    const int g = gp / a.P, p = gp - g * a.P;
    const float cx = bb[0] * a.pc_span[0] + a.pc_lo[0];
    const float cy = bb[1] * a.pc_span[1] + a.pc_lo[1];
    const float cz = bb[2] * a.pc_span[2] + a.pc_lo[2];
    const float yaw = atan2f(bb[6], bb[7]);
    const float cs = cosf(yaw), sn = a.rot_sign * sinf(yaw);
    const float dx = expf(bb[3]) * of[0], dy = expf(bb[4]) * of[1], dz = expf(bb[5]) * of[2];
    const float px = cx + (dx * cs + dy * (-sn));
    const float py = cy + (dx * sn + dy * cs);
    const float pz = cz + dz;
    const float td = a.time_diff[b * a.T + t];
    const float x = px - bb[8] * td, y = py - bb[9] * td, z = pz;

    int view = 0;
    bool found = false;
    float su = 0.f, sv = 0.f;
    for (int n = 0; n < a.N; ++n) {
        const float* m = a.l2i + (((long long)b * a.T + t) * a.N + n) * 16;
        const float uh = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z)), m[3]);
        const float vh = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[4], x), __fmul_rn(m[5], y)), __fmul_rn(m[6], z)), m[7]);
        const float hm = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[8], x), __fmul_rn(m[9], y)), __fmul_rn(m[10], z)), m[11]);
        const float hn = fmaxf(hm, a.eps);
        const float u = __fdiv_rn(__fdiv_rn(uh, hn), a.image_w);
        const float v = __fdiv_rn(__fdiv_rn(vh, hn), a.image_h);
        const bool valid = (hm > a.eps) && (v > 0.f) && (v < 1.f) && (u > 0.f) && (u < 1.f);
        if (n == 0 || (valid && !found)) { su = u; sv = v; view = n; }
        found = found || valid;
    }
    float* o = a.loc_bp + (((((long long)b * a.T + t) * a.G + g) * a.Q + q) * a.P + p) * 3;
    o[0] = su;
    o[1] = sv;
    const float zv = __fdiv_rn((float)view, (float)(a.N - 1));
    o[2] = zv;
    if (a.touch.need) touch_units(a, b, t, g, su, sv, zv);

    // level softmax of (g, p): every frame's thread recomputes it (L expf) and writes ITS row t of group g's T weight
    // rows, instead of the t == 0 threads writing all T rows (their 56 workgroups were the kernel's tail)
    float mx = lg[0];
    for (int l = 1; l < a.L; ++l) mx = fmaxf(mx, lg[l]);
    float e[SBEV_MAX_LEVELS];
    float sum = 0.f;
    for (int l = 0; l < a.L; ++l) {
        e[l] = expf(lg[l] - mx);
        sum += e[l];
    }
    const long long row = ((long long)b * a.G + g) * a.T + t;
    float* ow = a.w_bp + ((row * a.Q + q) * a.P + p) * a.L;
    for (int l = 0; l < a.L; ++l) ow[l] = e[l] / sum;
}

}  // namespace sbev_ops
