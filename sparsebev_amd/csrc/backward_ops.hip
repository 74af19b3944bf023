// Backward of the decoder layer's row-wise ops (gfx950): bias / ReLU, LayerNorm, box refinement, sample-point
// generation, camera projection + view selection, dropout.  SURVEY.md section 8f rank 4 -- the reference differentiates
// these through autograd (models/sparsebev_transformer.py:162-193,270-311; models/sparsebev_sampling.py:8-24,49-114);
// here each has one hand-written kernel behind the C ABI, and sparsebev_amd/autograd.py wires them as
// torch.autograd.Function backward passes.  All of them are HBM / latency bound and tiny next to the GEMMs.
//
// This file is compiled with -ffp-contract=off: the projection backward must re-select exactly the camera the forward
// selected, so it re-runs the forward's individually rounded expressions (project.hip) bit for bit.
#include "sbev_common.hpp"

namespace {

// ---- column sums (bias gradients) with an optional ReLU mask -------------------------------------------------------
// dZ = dY * (Y > 0)  (Y = the forward OUTPUT of a ReLU; null -> dZ = dY);  db[n] = sum_m dZ[m, n].
// Two deterministic passes: block (column block of 64, row chunk of ROW_CHUNK rows) = 64 columns x 4 row lanes writes one
// partial row of sums, a second tiny kernel adds the chunks in order.  (One block per 64 columns over ALL rows was 4
// workgroups for a 256-wide Linear: 73 us for 0.9 MB.)
constexpr int ROW_CHUNK = 32;     // rows per partial-sum block: 8 dependent iterations per thread (128 measured 13.7 us for a 900 x 256 gradient)
struct ColArgs {
    const float* dY;     // [M, ld]
    const float* Y;      // [M, ld] or null
    float* dZ;           // [M, ld] or null (may alias dY)
    float* part;         // [chunks, N] or null (no column sums wanted)
    long long M, ld;
    int N;
    int accumulate;      // column sums are ADDED to db (the shared parameters' gradients of a decoder call: autograd.ParamTap)
};

__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const ColArgs a) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long long n = (long long)blockIdx.x * 64 + c;
    const long long m0 = (long long)blockIdx.y * ROW_CHUNK, m1 = m0 + ROW_CHUNK < a.M ? m0 + ROW_CHUNK : a.M;
    float s = 0.f;
    if (n < a.N) {
        for (long long m = m0 + rg; m < m1; m += 4) {
            float g = a.dY[m * a.ld + n];
            if (a.Y && !(a.Y[m * a.ld + n] > 0.f)) g = 0.f;
            if (a.dZ) a.dZ[m * a.ld + n] = g;
            s += g;
        }
    }
    if (!a.part) return;
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < a.N) a.part[(long long)blockIdx.y * a.N + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// Few rows (<= ONE_PASS_ROWS: the decoder's B*Q = 900): the whole column sum in ONE launch -- block = 64 columns x 16 row lanes, every
// lane walks its rows 4 at a time (independent loads), the 16 partial sums are added in lane order through LDS: deterministic, and
// one 8 us launch instead of two of 7.9 + 7.6 (round 3: 222 of a training step's ~1100 launches were this pair).
constexpr int ONE_PASS_ROWS = 2048;
__global__ __launch_bounds__(1024) void bias_relu_bwd_onepass_kernel(const ColArgs a, float* __restrict__ db) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long long n = (long long)blockIdx.x * 64 + c;
    float s = 0.f;
    if (n < a.N) {
        long long m = rg;
        for (; m + 48 < a.M; m += 64) {
            float g[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = a.dY[(m + 16 * i) * a.ld + n];
            if (a.Y) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (!(a.Y[(m + 16 * i) * a.ld + n] > 0.f)) g[i] = 0.f;
            }
            if (a.dZ) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a.dZ[(m + 16 * i) * a.ld + n] = g[i];
            }
            s += (g[0] + g[1]) + (g[2] + g[3]);
        }
        for (; m < a.M; m += 16) {
            float g = a.dY[m * a.ld + n];
            if (a.Y && !(a.Y[m * a.ld + n] > 0.f)) g = 0.f;
            if (a.dZ) a.dZ[m * a.ld + n] = g;
            s += g;
        }
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < a.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][c];
        db[n] = a.accumulate ? db[n] + t : t;
    }
}

// Grouped column sums: the bias gradients of up to 16 Linears, each the sum over up to 8 [M, N_b] gradient matrices (the layers of a
// decoder call that share the bias), in ONE launch -- block = 64 columns of one bias x 16 row lanes, walking all segments' rows 8 at a
// time; lane-ordered LDS sum: deterministic.  (Per layer and bias this was one 10 us launch: 66 of a training step's launches.)
struct ColGroup {
    const float* seg[8];
    float* out;
    int N, nseg, blk0, accumulate;
};
struct ColGroupArgs {
    ColGroup g[16];
    int ng;
    long long M;
};
__global__ __launch_bounds__(1024) void colsum_group_kernel(const ColGroupArgs a) {
    __shared__ float red[16][64];
    int gi = 0;
    for (int i = 1; i < a.ng; ++i)
        if ((int)blockIdx.x >= a.g[i].blk0) gi = i;
    const ColGroup& g = a.g[gi];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = ((int)blockIdx.x - g.blk0) * 64 + c;
    float s = 0.f;
    if (n < g.N) {
        for (int sgi = 0; sgi < g.nseg; ++sgi) {
            const float* p = g.seg[sgi] + n;
            long long m = rg;
            for (; m + 112 < a.M; m += 128) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = p[(m + 16 * i) * g.N];
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; m < a.M; m += 16) s += p[m * g.N];
        }
    }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < g.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][c];
        g.out[n] = g.accumulate ? g.out[n] + t : t;
    }
}

// out[k][n] = sum_chunks part[k][chunk][n]  for K stacked partial sets (K = 1: bias; K = 2: dgamma, dbeta)
__global__ __launch_bounds__(256) void chunk_sum_kernel(const float* __restrict__ part, float* __restrict__ out0, float* __restrict__ out1,
                                                        int chunks, int N, int accumulate = 0) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float* out = blockIdx.y == 0 ? out0 : out1;
    const float* p = part + (long long)blockIdx.y * chunks * N;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += p[(long long)c * N + n];
    out[n] = accumulate ? out[n] + s : s;
}

// ---- LayerNorm backward ---------------------------------------------------------------------------------------------
// y = relu?(xhat * gamma + beta), xhat = (x - mean) * rstd.   One wave per row (N <= 1024, N % 4 == 0):
//   g  = dY * (y_pre_relu > 0)                         dxhat = g * gamma
//   dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
// and the row's (mean, rstd) are left in `stats` for the column kernel: dgamma[n] = sum_m g * xhat, dbeta[n] = sum_m g.
struct LnBwdArgs {
    const float* dY;     // [M, N]
    const float* X;      // [M, N]   (the LayerNorm INPUT)
    const float* gamma;  // [N]
    const float* beta;   // [N]      (only read when relu)
    float* dX;           // [M, N]
    float* stats;        // workspace: [M, 2] (mean, rstd) followed by [2][chunks][N] partial column sums
    float* dgamma;       // [N]
    float* dbeta;        // [N]
    long long M;
    int N, relu;
    float eps;
    int accumulate;      // dgamma / dbeta are added to
};

__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const LnBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    constexpr int MAXV = 4;
    const int nvec = a.N / 4;
    float4 x[MAXV];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int i4 = lane + 64 * c;
        x[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < nvec) {
            x[c] = *reinterpret_cast<const float4*>(a.X + row * a.N + i4 * 4);
            s += (x[c].x + x[c].y) + (x[c].z + x[c].w);
        }
    }
    s = sbev::wave_sum_dpp(s);
    const float mean = s / (float)a.N;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c)
        if (lane + 64 * c < nvec) {
            const float dx = x[c].x - mean, dy = x[c].y - mean, dz = x[c].z - mean, dw = x[c].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    q = sbev::wave_sum_dpp(q);
    const float rstd = rsqrtf(q / (float)a.N + a.eps);
    float dh[MAXV][4], xh[MAXV][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int i4 = lane + 64 * c;
        if (i4 < nvec) {
            const float4 g4 = *reinterpret_cast<const float4*>(a.dY + row * a.N + i4 * 4);
            const float4 w4 = *reinterpret_cast<const float4*>(a.gamma + i4 * 4);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w};
            const float xv[4] = {x[c].x, x[c].y, x[c].z, x[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (xv[e] - mean) * rstd;
                float g = gv[e];
                if (a.relu && !(h * wv[e] + a.beta[i4 * 4 + e] > 0.f)) g = 0.f;
                xh[c][e] = h;
                dh[c][e] = g * wv[e];
                s1 += dh[c][e];
                s2 += dh[c][e] * h;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[c][e] = dh[c][e] = 0.f;
        }
    }
    s1 = sbev::wave_sum_dpp(s1) / (float)a.N;
    s2 = sbev::wave_sum_dpp(s2) / (float)a.N;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int i4 = lane + 64 * c;
        if (i4 < nvec) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (dh[c][e] - s1 - xh[c][e] * s2);
            *reinterpret_cast<float4*>(a.dX + row * a.N + i4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (lane == 0) {
        a.stats[row * 2] = mean;
        a.stats[row * 2 + 1] = rstd;
    }
}

__global__ __launch_bounds__(256) void ln_bwd_cols_kernel(const LnBwdArgs a) {
    __shared__ float red[2][4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const long long m0 = (long long)blockIdx.y * ROW_CHUNK, m1 = m0 + ROW_CHUNK < a.M ? m0 + ROW_CHUNK : a.M;
    const int chunks = gridDim.y;
    float sg = 0.f, sb = 0.f;
    if (n < a.N) {
        const float w = a.gamma[n], bt = a.relu ? a.beta[n] : 0.f;
        for (long long m = m0 + rg; m < m1; m += 4) {
            const float h = (a.X[m * a.N + n] - a.stats[m * 2]) * a.stats[m * 2 + 1];
            float g = a.dY[m * a.N + n];
            if (a.relu && !(h * w + bt > 0.f)) g = 0.f;
            sg += g * h;
            sb += g;
        }
    }
    red[0][rg][c] = sg;
    red[1][rg][c] = sb;
    __syncthreads();
    if (rg == 0 && n < a.N) {
        float* part = a.stats + 2 * a.M;                   // [2][chunks][N] behind the row statistics
        part[(long long)blockIdx.y * a.N + n] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        part[((long long)chunks + blockIdx.y) * a.N + n] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// few rows: dgamma / dbeta in one launch (see bias_relu_bwd_onepass_kernel): block = 64 columns x 16 row lanes
__global__ __launch_bounds__(1024) void ln_bwd_cols_onepass_kernel(const LnBwdArgs a) {
    __shared__ float red[2][16][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float sg = 0.f, sb = 0.f;
    if (n < a.N) {
        const float w = a.gamma[n], bt = a.relu ? a.beta[n] : 0.f;
        long long m = rg;
        for (; m + 16 < a.M; m += 32) {                       // two independent rows per trip
            const float h0 = (a.X[m * a.N + n] - a.stats[m * 2]) * a.stats[m * 2 + 1];
            const float h1 = (a.X[(m + 16) * a.N + n] - a.stats[(m + 16) * 2]) * a.stats[(m + 16) * 2 + 1];
            float g0 = a.dY[m * a.N + n], g1 = a.dY[(m + 16) * a.N + n];
            if (a.relu && !(h0 * w + bt > 0.f)) g0 = 0.f;
            if (a.relu && !(h1 * w + bt > 0.f)) g1 = 0.f;
            sg += g0 * h0; sb += g0;
            sg += g1 * h1; sb += g1;
        }
        for (; m < a.M; m += 16) {
            const float h = (a.X[m * a.N + n] - a.stats[m * 2]) * a.stats[m * 2 + 1];
            float g = a.dY[m * a.N + n];
            if (a.relu && !(h * w + bt > 0.f)) g = 0.f;
            sg += g * h; sb += g;
        }
    }
    red[0][rg][c] = sg;
    red[1][rg][c] = sb;
    __syncthreads();
    if (rg < 2 && n < a.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[rg][k][c];
        float* o = (rg == 0 ? a.dgamma : a.dbeta) + n;
        *o = a.accumulate ? *o + t : t;
    }
}

// Grouped LayerNorm parameter gradients: dgamma / dbeta of up to 8 LayerNorms, each summed over up to 8 (dY, X, row statistics)
// triples -- the layers of a decoder call sharing the LayerNorm -- in ONE launch (block = 64 columns of one LayerNorm x 16 row lanes;
// the statistics are the [M, 2] (mean, rstd) rows sbev_layer_norm_bwd_acc leaves at the start of its workspace).
struct LnGroup {
    const float* dY[8];
    const float* X[8];
    const float* stats[8];
    const float* gamma;
    const float* beta;
    float* dgamma;
    float* dbeta;
    int N, nseg, blk0, relu, accumulate;
};
struct LnGroupArgs {
    LnGroup g[8];
    int ng;
    long long M;
};
__global__ __launch_bounds__(1024) void ln_param_group_kernel(const LnGroupArgs a) {
    __shared__ float red[2][16][64];
    int gi = 0;
    for (int i = 1; i < a.ng; ++i)
        if ((int)blockIdx.x >= a.g[i].blk0) gi = i;
    const LnGroup& g = a.g[gi];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = ((int)blockIdx.x - g.blk0) * 64 + c;
    float sg = 0.f, sb = 0.f;
    if (n < g.N) {
        const float w = g.gamma[n], bt = g.relu ? g.beta[n] : 0.f;
        for (int sgi = 0; sgi < g.nseg; ++sgi) {
            const float* dY = g.dY[sgi] + n;
            const float* X = g.X[sgi] + n;
            const float* st = g.stats[sgi];
            long long m = rg;
            for (; m + 48 < a.M; m += 64) {
                float h[4], gr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long long r = m + 16 * i;
                    h[i] = (X[r * g.N] - st[r * 2]) * st[r * 2 + 1];
                    gr[i] = dY[r * g.N];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (g.relu && !(h[i] * w + bt > 0.f)) gr[i] = 0.f;
                    sg += gr[i] * h[i];
                    sb += gr[i];
                }
            }
            for (; m < a.M; m += 16) {
                const float hh = (X[m * g.N] - st[m * 2]) * st[m * 2 + 1];
                float gg = dY[m * g.N];
                if (g.relu && !(hh * w + bt > 0.f)) gg = 0.f;
                sg += gg * hh;
                sb += gg;
            }
        }
    }
    red[0][rg][c] = sg;
    red[1][rg][c] = sb;
    __syncthreads();
    if (rg < 2 && n < g.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[rg][k][c];
        float* o = (rg == 0 ? g.dgamma : g.dbeta) + n;
        *o = g.accumulate ? *o + t : t;
    }
}

// ---- refine_bbox backward (models/sparsebev_transformer.py:155-160,179-183; models/utils.py:87-102) ----------------
struct RefineBwdArgs {
    const float* gout;     // [BQ,10]  grad of the refined box
    const float* out;      // [BQ,10]  the refined box (forward output: xyz = the sigmoid)
    const float* bbox;     // [BQ,10]  the proposal
    const float* vel_div;  // [B] or null
    float* greg;           // [BQ,10]
    float* gbbox;          // [BQ,10] or null (grad of the proposal: only xyz is non-zero)
    long long BQ;
    int Q;
};

__global__ __launch_bounds__(256) void refine_bwd_kernel(const RefineBwdArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.BQ) return;
    const float* go = a.gout + i * 10;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float s = a.out[i * 10 + d];
        const float gz = go[d] * s * (1.f - s);
        a.greg[i * 10 + d] = gz;
        if (a.gbbox) {
            const float p = a.bbox[i * 10 + d];
            float g = 0.f;
            if (p >= 0.f && p <= 1.f) {                       // clamp(0, 1) passes the gradient inside the range only
                if (p >= 1e-5f) g += 1.f / p;                  // d log(max(p, eps)) / dp
                if (1.f - p >= 1e-5f) g += 1.f / (1.f - p);    // - d log(max(1 - p, eps)) / dp
            }
            a.gbbox[i * 10 + d] = gz * g;
        }
    }
    for (int d = 3; d < 10; ++d) {
        float g = go[d];
        if (d >= 8 && a.vel_div) g = g / a.vel_div[(unsigned)i / (unsigned)a.Q];
        a.greg[i * 10 + d] = g;
        if (a.gbbox) a.gbbox[i * 10 + d] = 0.f;
    }
}

// ---- projection + view selection backward (models/sparsebev_sampling.py:49-114) ------------------------------------
// One thread per (b, t, q, gp): re-run the forward's projection into the N cameras (same individually rounded
// expressions as project.hip, so the same first-hit camera is selected), then
//   u = (uh / hn) / W,  hn = max(hm, eps):   d/duh = 1 / (hn W);   d/dhm = -(uh / hn^2) / W  if hm > eps else 0
// (torch.maximum passes the gradient to the larger argument), chained through the selected camera's 4x4 matrix.
struct ProjBwdArgs {
    const float* pts;       // [B,Q,T,GP,3]
    const float* l2i;       // [B,T*N,4,4]
    const float* gloc;      // [B*T*G,Q,P,3]  (x, y used)
    float* gpts;            // [B,Q,T,GP,3]
    int B, Q, T, N, G, P;
    float image_h, image_w, eps;
};

__global__ __launch_bounds__(256) void project_bwd_kernel(const ProjBwdArgs a) {
    const int GP = a.G * a.P;
    const long long total = (long long)a.B * a.T * a.Q * GP;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const unsigned ui = (unsigned)idx;
    unsigned r = ui / (unsigned)GP;
    const int gp = (int)(ui - r * (unsigned)GP);
    const unsigned r2 = r / (unsigned)a.Q;
    const int q = (int)(r - r2 * (unsigned)a.Q);
    const int b = (int)(r2 / (unsigned)a.T);
    const int t = (int)(r2 - (unsigned)b * (unsigned)a.T);
    const long long pi = ((((long long)b * a.Q + q) * a.T + t) * GP + gp) * 3;
    const float x = a.pts[pi], y = a.pts[pi + 1], z = a.pts[pi + 2];
    int view = 0;
    bool found = false;
    float s_uh = 0.f, s_vh = 0.f, s_hm = 0.f;
    for (int n = 0; n < a.N; ++n) {
        const float* m = a.l2i + (((long long)b * a.T + t) * a.N + n) * 16;
        const float uh = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z)), m[3]);
        const float vh = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[4], x), __fmul_rn(m[5], y)), __fmul_rn(m[6], z)), m[7]);
        const float hm = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[8], x), __fmul_rn(m[9], y)), __fmul_rn(m[10], z)), m[11]);
        const float hn = fmaxf(hm, a.eps);
        const float u = __fdiv_rn(__fdiv_rn(uh, hn), a.image_w);
        const float v = __fdiv_rn(__fdiv_rn(vh, hn), a.image_h);
        const bool valid = (hm > a.eps) && (v > 0.f) && (v < 1.f) && (u > 0.f) && (u < 1.f);
        if (n == 0 || (valid && !found)) { s_uh = uh; s_vh = vh; s_hm = hm; view = n; }
        found = found || valid;
    }
    const int g = gp / a.P, p = gp - g * a.P;
    const float* gl = a.gloc + (((((long long)b * a.T + t) * a.G + g) * a.Q + q) * a.P + p) * 3;
    const float gu = gl[0], gv = gl[1];
    const float hn = fmaxf(s_hm, a.eps);
    const float g_uh = gu / (hn * a.image_w), g_vh = gv / (hn * a.image_h);
    const float g_hm = s_hm > a.eps ? -(g_uh * s_uh + g_vh * s_vh) / hn : 0.f;
    const float* m = a.l2i + (((long long)b * a.T + t) * a.N + view) * 16;
    a.gpts[pi] = g_uh * m[0] + g_vh * m[4] + g_hm * m[8];
    a.gpts[pi + 1] = g_uh * m[1] + g_vh * m[5] + g_hm * m[9];
    a.gpts[pi + 2] = g_uh * m[2] + g_vh * m[6] + g_hm * m[10];
}

// ---- sample-point generation + level softmax backward (models/sparsebev_transformer.py:270-311) ---------------------
// Per query: each of its G*P points sums its gradient over the T warped copies (which share the un-warped point) and emits
// its offset / logit gradients; the 8 box gradients (centre, log-dims, sin, cos) are summed over the points.
// Velocity is detached in the reference (:288) -> no gradient to columns 8, 9.
struct FrontBwdArgs {
    const float* bbox;      // [B,Q,10]
    const float* offset;    // [B*Q, ld_off]
    const float* logits;    // [B*Q, ld_logit]
    long long ld_off, ld_logit;
    const float* gpts;      // [B,Q,T,GP,3] or null
    const float* gw_bp;     // [B*G*T,Q,P,L] or null
    float* goffset;         // [B*Q, ld_g] (first GP*3 columns)
    float* glogits;         // [B*Q, ld_g] (first GP*L columns)
    long long ld_g;
    float* gbbox;           // [B*Q, 10] or null
    float pc_span[3];
    int B, Q, T, G, P, L;
    float rot_sign;
};

// one wave per (b, q), lane = sample point gp (G*P <= 64): per-point work in parallel, the 8 box gradients reduced over the
// wave with DPP sums (one THREAD per query walking its 16 points x T frames serially took 139 us at config 2)
__global__ __launch_bounds__(256) void front_bwd_kernel(const FrontBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const long long bq = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bq >= (long long)a.B * a.Q) return;
    const int GP = a.G * a.P;
    const int gp = lane;
    const bool act = gp < GP;
    const int b = (int)((unsigned)bq / (unsigned)a.Q), q = (int)((unsigned)bq - (unsigned)b * (unsigned)a.Q);
    const float* bb = a.bbox + bq * 10;
    float gb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.gpts) {
        const float s = bb[6], c = bb[7];
        const float yaw = atan2f(s, c);
        const float cs = cosf(yaw), sn_raw = sinf(yaw), sn = a.rot_sign * sn_raw;
        const float ew = expf(bb[3]), el = expf(bb[4]), eh = expf(bb[5]);
        float g_yaw = 0.f;
        if (act) {
            float gx = 0.f, gy = 0.f, gz = 0.f;
            for (int t = 0; t < a.T; ++t) {
                const float* gpt = a.gpts + ((bq * a.T + t) * GP + gp) * 3;
                gx += gpt[0]; gy += gpt[1]; gz += gpt[2];
            }
            const float* of = a.offset + bq * a.ld_off + gp * 3;
            const float dx = ew * of[0], dy = el * of[1], dz = eh * of[2];
            // px = cx + (dx cs - dy sn), py = cy + (dx sn + dy cs), pz = cz + dz
            const float g_dx = gx * cs + gy * sn, g_dy = -gx * sn + gy * cs;
            float* go = a.goffset + bq * a.ld_g + gp * 3;
            go[0] = g_dx * ew; go[1] = g_dy * el; go[2] = gz * eh;
            gb[0] = gx; gb[1] = gy; gb[2] = gz;
            gb[3] = g_dx * dx; gb[4] = g_dy * dy; gb[5] = gz * dz;
            const float g_cs = gx * dx + gy * dy, g_sn = -gx * dy + gy * dx;
            g_yaw = g_cs * (-sn_raw) + g_sn * a.rot_sign * cs;
        }
#pragma unroll
        for (int d = 0; d < 6; ++d) gb[d] = sbev::wave_sum_dpp(gb[d]);
        g_yaw = sbev::wave_sum_dpp(g_yaw);
        const float r2 = s * s + c * c;
        gb[6] = g_yaw * c / r2;
        gb[7] = -g_yaw * s / r2;
        gb[0] *= a.pc_span[0]; gb[1] *= a.pc_span[1]; gb[2] *= a.pc_span[2];
    } else if (act) {
        for (int i = 0; i < 3; ++i) a.goffset[bq * a.ld_g + gp * 3 + i] = 0.f;
    }
    if (a.gbbox && lane == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) a.gbbox[bq * 10 + d] = gb[d];
        a.gbbox[bq * 10 + 8] = 0.f;
        a.gbbox[bq * 10 + 9] = 0.f;
    }
    if (!act) return;
    float* gl = a.glogits + bq * a.ld_g + gp * a.L;
    if (!a.gw_bp) {
        for (int l = 0; l < a.L; ++l) gl[l] = 0.f;
        return;
    }
    const int g = gp / a.P, p = gp - g * a.P;
    const float* lg = a.logits + bq * a.ld_logit + gp * a.L;
    float mx = lg[0];
    for (int l = 1; l < a.L; ++l) mx = fmaxf(mx, lg[l]);
    float e[SBEV_MAX_LEVELS], gs[SBEV_MAX_LEVELS];
    float sum = 0.f;
    for (int l = 0; l < a.L; ++l) {
        e[l] = expf(lg[l] - mx);
        sum += e[l];
        gs[l] = 0.f;
    }
    // the softmax of (g, p) was replicated into the T weight rows r = (b*G + g)*T + t' (forward kernel): sum their grads
    for (int tp = 0; tp < a.T; ++tp) {
        const long long row = ((long long)b * a.G + g) * a.T + tp;
        const float* gw = a.gw_bp + ((row * a.Q + q) * a.P + p) * a.L;
        for (int l = 0; l < a.L; ++l) gs[l] += gw[l];
    }
    float dot = 0.f;
    for (int l = 0; l < a.L; ++l) dot += (e[l] / sum) * gs[l];
    for (int l = 0; l < a.L; ++l) gl[l] = (e[l] / sum) * (gs[l] - dot);
}

// ---- dropout (training only; mmcv MultiheadAttention / FFN: models/sparsebev_transformer.py:125,202) ----------------
// Counter-based: the keep decision of element i is a hash of (seed, i), so the backward regenerates the mask instead of
// storing it.  y = x * keep / (1 - p).
__device__ __forceinline__ unsigned mix32(unsigned long long z) {       // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                      unsigned long long seed, const unsigned long long* seed_dev, float p, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned thr = (unsigned)((double)p * 4294967296.0);
    if (seed_dev) seed += *seed_dev;        // captured training steps: the part of the seed that changes from replay to replay
    y[i] = mix32(seed * 0x100000001b3ull + (unsigned long long)i) >= thr ? x[i] * scale : 0.f;
}

}  // namespace

extern "C" int64_t sbev_colsum_workspace(int64_t M, int N) {
    if (M < 0 || N < 0) return -1;
    return ((M + ROW_CHUNK - 1) / ROW_CHUNK) * (int64_t)N * (int64_t)sizeof(float);
}

static int bias_relu_bwd_impl(const float* dY, const float* Y, float* dZ, float* db, int64_t M, int N, int64_t ld,
                              float* workspace, int accumulate, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 0 && ld >= N, "sbev_bias_relu_bwd: bad sizes");
    if (N == 0) return SBEV_OK;
    SBEV_REQUIRE(dY != nullptr && (!db || workspace), "sbev_bias_relu_bwd: null grad / workspace");
    const int chunks = (int)((M + ROW_CHUNK - 1) / ROW_CHUNK);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (db && M > 0 && M <= ONE_PASS_ROWS) {
        ColArgs a{dY, Y, dZ, nullptr, M, ld, N, accumulate};
        hipLaunchKernelGGL(bias_relu_bwd_onepass_kernel, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, s, a, db);
        return sbev::check_launch("sbev_bias_relu_bwd");
    }
    if (chunks > 0) {
        SBEV_REQUIRE(chunks <= 65535, "sbev_bias_relu_bwd: too many rows");
        ColArgs a{dY, Y, dZ, db ? workspace : nullptr, M, ld, N, 0};
        hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)chunks), dim3(256), 0, s, a);
        int st = sbev::check_launch("sbev_bias_relu_bwd");
        if (st != SBEV_OK) return st;
    }
    if (db) {
        hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((N + 255) / 256), 1), dim3(256), 0, s, workspace, db, db, chunks, N, accumulate);
        return sbev::check_launch("sbev_bias_relu_bwd (sum)");
    }
    return SBEV_OK;
}

extern "C" int sbev_bias_relu_bwd(const float* dY, const float* Y, float* dZ, float* db, int64_t M, int N, int64_t ld,
                                  float* workspace, sbev_stream_t stream) {
    return bias_relu_bwd_impl(dY, Y, dZ, db, M, N, ld, workspace, 0, stream);
}
// the same with the column sums ADDED to db (accumulate != 0): one parameter used by several layers of a call
extern "C" int sbev_bias_relu_bwd_acc(const float* dY, const float* Y, float* dZ, float* db, int64_t M, int N, int64_t ld,
                                      float* workspace, int accumulate, sbev_stream_t stream) {
    return bias_relu_bwd_impl(dY, Y, dZ, db, M, N, ld, workspace, accumulate, stream);
}

// dX and the row statistics only (no dgamma / dbeta): the first half of sbev_layer_norm_bwd, for callers that sum the parameter
// gradients later with sbev_layer_norm_param_group.  workspace: >= 2 M floats, receives [M, 2] (mean, rstd).
extern "C" int sbev_layer_norm_bwd_rows(const float* dY, const float* X, const float* gamma, const float* beta, float eps, int relu,
                                        float* dX, float* workspace, int64_t M, int N, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024, "sbev_layer_norm_bwd_rows: N=%d must be a multiple of 4 in 4..1024", N);
    if (M == 0) return SBEV_OK;
    SBEV_REQUIRE(dY && X && gamma && dX && workspace && (!relu || beta), "sbev_layer_norm_bwd_rows: null pointer");
    LnBwdArgs a{dY, X, gamma, beta, dX, workspace, nullptr, nullptr, M, N, relu, eps, 0};
    hipLaunchKernelGGL(ln_bwd_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_layer_norm_bwd_rows");
}

// dgamma_b / dbeta_b (+)= sums over the nseg_b (dY, X, stats) triples of LayerNorm b, ng <= 8 LayerNorms in one launch.
// dYs / Xs / stats: host arrays [ng][8] of device pointers; gammas / betas / dgammas / dbetas: [ng]; Ns / nsegs / relus / accumulate: [ng]
extern "C" int sbev_layer_norm_param_group(const float* const* dYs, const float* const* Xs, const float* const* stats,
                                           const float* const* gammas, const float* const* betas, float* const* dgammas, float* const* dbetas,
                                           const int32_t* Ns, const int32_t* nsegs, const int32_t* relus, const int32_t* accumulate,
                                           int ng, int64_t M, sbev_stream_t stream) {
    SBEV_REQUIRE(ng >= 0 && ng <= 8 && M >= 0, "sbev_layer_norm_param_group: at most 8 groups");
    if (ng == 0) return SBEV_OK;
    SBEV_REQUIRE(dYs && Xs && stats && gammas && betas && dgammas && dbetas && Ns && nsegs && relus && accumulate, "sbev_layer_norm_param_group: null pointer");
    LnGroupArgs a{};
    a.ng = ng; a.M = M;
    int blk = 0;
    for (int b = 0; b < ng; ++b) {
        SBEV_REQUIRE(Ns[b] >= 1 && nsegs[b] >= 1 && nsegs[b] <= 8 && gammas[b] && dgammas[b] && dbetas[b] && (!relus[b] || betas[b]),
                     "sbev_layer_norm_param_group: group %d", b);
        LnGroup& g = a.g[b];
        g.N = Ns[b]; g.nseg = nsegs[b]; g.blk0 = blk; g.relu = relus[b]; g.accumulate = accumulate[b];
        g.gamma = gammas[b]; g.beta = betas[b]; g.dgamma = dgammas[b]; g.dbeta = dbetas[b];
        for (int s = 0; s < nsegs[b]; ++s) {
            SBEV_REQUIRE(dYs[b * 8 + s] && Xs[b * 8 + s] && stats[b * 8 + s], "sbev_layer_norm_param_group: null segment");
            g.dY[s] = dYs[b * 8 + s]; g.X[s] = Xs[b * 8 + s]; g.stats[s] = stats[b * 8 + s];
        }
        blk += (Ns[b] + 63) / 64;
    }
    hipLaunchKernelGGL(ln_param_group_kernel, dim3((unsigned)blk), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_layer_norm_param_group");
}

// out_b[n] (+)= sum over the nseg_b matrices seg_b[s] [M, N_b] (row-major, contiguous) of their column sums, for ng <= 16 groups b:
// segs = host array of ng * 8 device pointers (row b: its nseg_b pointers first), outs / Ns / nsegs / accumulate = host arrays [ng]
extern "C" int sbev_colsum_group(const float* const* segs, float* const* outs, const int32_t* Ns, const int32_t* nsegs,
                                 const int32_t* accumulate, int ng, int64_t M, sbev_stream_t stream) {
    SBEV_REQUIRE(ng >= 0 && ng <= 16 && M >= 0, "sbev_colsum_group: at most 16 groups");
    if (ng == 0) return SBEV_OK;
    SBEV_REQUIRE(segs && outs && Ns && nsegs && accumulate, "sbev_colsum_group: null pointer");
    ColGroupArgs a{};
    a.ng = ng; a.M = M;
    int blk = 0;
    for (int b = 0; b < ng; ++b) {
        SBEV_REQUIRE(Ns[b] >= 1 && nsegs[b] >= 1 && nsegs[b] <= 8 && outs[b], "sbev_colsum_group: group %d", b);
        a.g[b].N = Ns[b]; a.g[b].nseg = nsegs[b]; a.g[b].out = outs[b]; a.g[b].blk0 = blk; a.g[b].accumulate = accumulate[b];
        for (int s = 0; s < nsegs[b]; ++s) {
            SBEV_REQUIRE(segs[b * 8 + s], "sbev_colsum_group: null segment");
            a.g[b].seg[s] = segs[b * 8 + s];
        }
        blk += (Ns[b] + 63) / 64;
    }
    hipLaunchKernelGGL(colsum_group_kernel, dim3((unsigned)blk), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_colsum_group");
}

extern "C" int64_t sbev_layer_norm_bwd_workspace(int64_t M, int N) {
    if (M < 0 || N < 0) return -1;
    return (2 * M + 2 * ((M + ROW_CHUNK - 1) / ROW_CHUNK) * (int64_t)N) * (int64_t)sizeof(float);
}

static int layer_norm_bwd_impl(const float* dY, const float* X, const float* gamma, const float* beta, float eps, int relu,
                               float* dX, float* dgamma, float* dbeta, float* workspace, int64_t M, int N, int accumulate, sbev_stream_t stream) {
    SBEV_REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && N <= 1024, "sbev_layer_norm_bwd: N=%d must be a multiple of 4 in 4..1024", N);
    SBEV_REQUIRE(dY && X && gamma && dX && dgamma && dbeta && workspace && (!relu || beta), "sbev_layer_norm_bwd: null pointer");
    LnBwdArgs a{dY, X, gamma, beta, dX, workspace, dgamma, dbeta, M, N, relu, eps, accumulate};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int chunks = (int)((M + ROW_CHUNK - 1) / ROW_CHUNK);
    SBEV_REQUIRE(chunks <= 65535, "sbev_layer_norm_bwd: too many rows");
    if (M > 0) {
        hipLaunchKernelGGL(ln_bwd_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, a);
        int st = sbev::check_launch("sbev_layer_norm_bwd (rows)");
        if (st != SBEV_OK) return st;
        if (M <= ONE_PASS_ROWS) {
            hipLaunchKernelGGL(ln_bwd_cols_onepass_kernel, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, s, a);
            return sbev::check_launch("sbev_layer_norm_bwd (columns)");
        }
        hipLaunchKernelGGL(ln_bwd_cols_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)chunks), dim3(256), 0, s, a);
        st = sbev::check_launch("sbev_layer_norm_bwd (columns)");
        if (st != SBEV_OK) return st;
    }
    hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((N + 255) / 256), 2), dim3(256), 0, s, workspace + 2 * M, dgamma, dbeta, chunks, N, accumulate);
    return sbev::check_launch("sbev_layer_norm_bwd (sum)");
}

extern "C" int sbev_layer_norm_bwd(const float* dY, const float* X, const float* gamma, const float* beta, float eps, int relu,
                                   float* dX, float* dgamma, float* dbeta, float* workspace, int64_t M, int N, sbev_stream_t stream) {
    return layer_norm_bwd_impl(dY, X, gamma, beta, eps, relu, dX, dgamma, dbeta, workspace, M, N, 0, stream);
}
// the same with dgamma / dbeta ADDED to (accumulate != 0)
extern "C" int sbev_layer_norm_bwd_acc(const float* dY, const float* X, const float* gamma, const float* beta, float eps, int relu,
                                       float* dX, float* dgamma, float* dbeta, float* workspace, int64_t M, int N, int accumulate,
                                       sbev_stream_t stream) {
    return layer_norm_bwd_impl(dY, X, gamma, beta, eps, relu, dX, dgamma, dbeta, workspace, M, N, accumulate, stream);
}

extern "C" int sbev_refine_bbox_bwd(const float* grad_out, const float* out, const float* query_bbox, const float* vel_div,
                                    float* grad_reg, float* grad_bbox, int B, int Q, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && (int64_t)B * Q < 0x7fffffffLL, "sbev_refine_bbox_bwd: bad sizes");
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE(grad_out && out && query_bbox && grad_reg, "sbev_refine_bbox_bwd: null pointer");
    RefineBwdArgs a{grad_out, out, query_bbox, vel_div, grad_reg, grad_bbox, (long long)B * Q, Q};
    hipLaunchKernelGGL(refine_bwd_kernel, dim3((unsigned)((a.BQ + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_refine_bbox_bwd");
}

extern "C" int sbev_project_select_bwd(const float* sample_points, const float* lidar2img, const float* grad_loc,
                                       int B, int Q, int T, int N, int G, int P, float image_h, float image_w, float eps,
                                       float* grad_points, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && T >= 1 && N >= 1 && G >= 1 && P >= 1, "sbev_project_select_bwd: bad sizes");
    const long long total = (long long)B * T * Q * G * P;
    SBEV_REQUIRE(total < 0x7fffffffLL, "sbev_project_select_bwd: too many points");
    if (total == 0) return SBEV_OK;
    SBEV_REQUIRE(sample_points && lidar2img && grad_loc && grad_points, "sbev_project_select_bwd: null pointer");
    ProjBwdArgs a{sample_points, lidar2img, grad_loc, grad_points, B, Q, T, N, G, P, image_h, image_w, eps};
    hipLaunchKernelGGL(project_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_project_select_bwd");
}

extern "C" int sbev_sampling_front_bwd(const float* query_bbox, const float* offset, int64_t ld_off, const float* logits, int64_t ld_logit,
                                       const double* pc_range, int B, int Q, int T, int G, int P, int L,
                                       const float* grad_points, const float* grad_weights_bp,
                                       float* grad_offset, float* grad_logits, int64_t ld_grad, float* grad_bbox, sbev_stream_t stream) {
    SBEV_REQUIRE(B >= 0 && Q >= 0 && T >= 1 && G >= 1 && P >= 1 && L >= 1 && L <= SBEV_MAX_LEVELS, "sbev_sampling_front_bwd: bad sizes");
    if (B == 0 || Q == 0) return SBEV_OK;
    SBEV_REQUIRE((int64_t)B * Q < 0x7fffffffLL, "sbev_sampling_front_bwd: too many queries");
    SBEV_REQUIRE(query_bbox && offset && logits && pc_range && grad_offset && grad_logits, "sbev_sampling_front_bwd: null pointer");
    FrontBwdArgs a{};
    a.bbox = query_bbox; a.offset = offset; a.logits = logits; a.ld_off = ld_off; a.ld_logit = ld_logit;
    SBEV_REQUIRE(ld_grad >= (int64_t)G * P * 3 && ld_grad >= (int64_t)G * P * L, "sbev_sampling_front_bwd: ld_grad too small");
    a.gpts = grad_points; a.gw_bp = grad_weights_bp; a.goffset = grad_offset; a.glogits = grad_logits; a.ld_g = ld_grad; a.gbbox = grad_bbox;
    for (int d = 0; d < 3; ++d) a.pc_span[d] = (float)(pc_range[3 + d] - pc_range[d]);
    a.B = B; a.Q = Q; a.T = T; a.G = G; a.P = P; a.L = L;
    a.rot_sign = sbev::box_convention() == SBEV_BOX_V0_17_1 ? -1.f : 1.f;
    SBEV_REQUIRE(G * P <= 64, "sbev_sampling_front_bwd: G*P = %d sample points per query exceed one wave", G * P);
    const long long n = (long long)B * Q;
    hipLaunchKernelGGL(front_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return sbev::check_launch("sbev_sampling_front_bwd");
}

extern "C" int sbev_dropout_f32(const float* x, float* y, int64_t n, uint64_t seed, float p, sbev_stream_t stream) {
    return sbev_dropout_f32_ds(x, y, n, seed, nullptr, p, stream);
}

extern "C" int sbev_dropout_f32_ds(const float* x, float* y, int64_t n, uint64_t seed, const uint64_t* seed_dev, float p, sbev_stream_t stream) {
    SBEV_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, "sbev_dropout_f32: need 0 <= p < 1");
    if (n == 0) return SBEV_OK;
    SBEV_REQUIRE(x && y, "sbev_dropout_f32: null pointer");
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                       (long long)n, (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(seed_dev), p, 1.f / (1.f - p));
    return sbev::check_launch("sbev_dropout_f32");
}
