/*
 * Scalar C restatement of the two device-side pieces of the hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle/sparsebev_oracle.py for the rules); compiled by oracle/Makefile with -ffp-contract=off.
 *
 *  oracle_msmv_fwd   follows the reference CUDA kernel's loop structure one (b', q, channel) at a time:
 *                    models/csrc/msmv_sampling/msmv_sampling_forward.cu:75-164 (4 levels) / :166-267 (5),
 *                    bilinear helper :27-73.  The CUDA source itself cannot be built here (no nvcc, and it
 *                    includes THC/THCAtomics.cuh which modern torch no longer ships) -- this is a restatement.
 *  oracle_project    models/sparsebev_sampling.py:49-79,102 in the association order the reference's CPU
 *                    matmul uses: ((m0*x + m1*y) + m2*z) + m3, no FMA, IEEE divides.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float bilinear(const float* f, int H, int W, int C, float h, float w, int c) {
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h - h_low, lw = w - w_low;
    const float hh = 1 - lh, hw = 1 - lw;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = f[((int64_t)h_low * W + w_low) * C + c];
    if (h_low >= 0 && w_high <= W - 1) v2 = f[((int64_t)h_low * W + w_high) * C + c];
    if (h_high <= H - 1 && w_low >= 0) v3 = f[((int64_t)h_high * W + w_low) * C + c];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = f[((int64_t)h_high * W + w_high) * C + c];
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* feats[l]: [B',N,H_l,W_l,C] contiguous; loc [B',Q,P,3]; w [B',Q,P,L]; out [B',Q,C,P] */
void oracle_msmv_fwd(const float* const* feats, const int32_t* hw, int L, int64_t Bp, int N, int C, int Q, int P,
                     const float* loc, const float* w, float* out) {
    for (int64_t b = 0; b < Bp; ++b)
        for (int q = 0; q < Q; ++q) {
            const int64_t si = b * Q + q;
            for (int c = 0; c < C; ++c)
                for (int p = 0; p < P; ++p) {
                    const float lw_ = loc[(si * P + p) * 3 + 0];
                    const float lh_ = loc[(si * P + p) * 3 + 1];
                    const int lv = (int)roundf(loc[(si * P + p) * 3 + 2] * (N - 1));
                    float res = 0;
                    for (int l = 0; l < L; ++l) {
                        const int H = hw[2 * l], W = hw[2 * l + 1];
                        const float h_im = lh_ * (H - 1), w_im = lw_ * (W - 1);
                        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                            const float* f = feats[l] + ((int64_t)b * N + lv) * H * W * C;
                            res += bilinear(f, H, W, C, h_im, w_im, c) * w[(si * P + p) * L + l];
                        }
                    }
                    out[(si * C + c) * P + p] = res;
                }
        }
}

/* pts [B,Q,T,GP,3]; l2i [B,T*N,4,4]; uvh [B,T,N,Q,GP,3]; valid [B,T,N,Q,GP]; iview [B,T,Q,GP] */
void oracle_project(const float* pts, const float* l2i, int B, int Q, int T, int N, int GP,
                    float image_h, float image_w, float eps, float* uvh, uint8_t* valid, int32_t* iview) {
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int q = 0; q < Q; ++q)
                for (int g = 0; g < GP; ++g) {
                    const float* p = pts + ((((int64_t)b * Q + q) * T + t) * GP + g) * 3;
                    int first = 0, found = 0;
                    for (int n = 0; n < N; ++n) {
                        const float* m = l2i + (((int64_t)b * T + t) * N + n) * 16;
                        const float uh = ((m[0] * p[0] + m[1] * p[1]) + m[2] * p[2]) + m[3] * 1.0f;
                        const float vh = ((m[4] * p[0] + m[5] * p[1]) + m[6] * p[2]) + m[7] * 1.0f;
                        const float hm = ((m[8] * p[0] + m[9] * p[1]) + m[10] * p[2]) + m[11] * 1.0f;
                        const float hn = hm > eps ? hm : eps;
                        const float u = (uh / hn) / image_w, v = (vh / hn) / image_h;
                        const int ok = (hm > eps) && (v > 0.0f) && (v < 1.0f) && (u > 0.0f) && (u < 1.0f);
                        const int64_t o = ((((int64_t)b * T + t) * N + n) * Q + q) * GP + g;
                        uvh[o * 3] = u; uvh[o * 3 + 1] = v; uvh[o * 3 + 2] = hn;
                        valid[o] = (uint8_t)ok;
                        if (ok && !found) { first = n; found = 1; }
                    }
                    iview[(((int64_t)b * T + t) * Q + q) * GP + g] = first;
                }
}
