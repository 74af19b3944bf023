// Scale-adaptive self attention BACKWARD on the matrix cores (gfx950) -- the flash-attention-style counterpart of
// attention.hip's forward kernel, behind sbev_sasa_bwd_f32 (attention_bwd.hip keeps the entry point, the dropout training
// forward and a plain-VALU version of both passes that this file replaced: 0.35 ms per layer at config 2).
//
//   S_ij = (q_i . k_j)/sqrt(d) - dist_ij tau_ih (-inf under the DN mask)   P = softmax_j S   Pd = dropout(P)   O = Pd V
//   D_i = dO_i . O_i      dPd = dO V^T      dS = P o (keep dPd/(1-p) - D)
//   dq = dS K / sqrt(d)   dtau_ih = -sum_j dS_ij dist_ij        (ROW kernel, also leaves lse_i and D_i for the column kernel)
//   dk = dS^T q / sqrt(d) dv = Pd^T dO                           (COLUMN kernel, works on S^T = K Q^T tiles directly)
// Nothing of size Q x Q is materialised.  Workgroup = 8 waves = 2 groups of 16 rows (queries resp. keys) x 4 splits of the
// other axis, operand tiles of 64 staged in LDS as [row][36] so that every MFMA operand row is a 16-byte read (k order
// 16 blk + 4 fk + j, as in the forward kernel), products on v_mfma_f32_16x16x4_f32 (exact fp32), C-layout -> A-layout
// hand-overs through a per-wave LDS patch, split partials merged through LDS in a fixed order (no atomics: bit-reproducible).
#include "sbev_common.hpp"

namespace sbev_attn_bwd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 32, KT = 64, NG = 2, NS = 4, NWAVES = NG * NS;
constexpr int LDT = HD + 4;    // staged [row][d] tiles
constexpr int LDP = KT + 4;    // per-wave patch rows

struct Args {
    const float* qkvt;          // [B, Q, ld]: q | k | v | tau
    const float* bbox;          // [B, Q, 10]
    float lo[2], span[2];
    const unsigned char* mask;  // [Q, Q] or null
    const float* O;             // [B, Q, D]
    const float* dO;            // [B, Q, D]
    float* dqkvt;               // [B, Q, ld]
    float* lse;                 // [B, H, Q]
    float* dvec;                // [B, H, Q]
    int B, Q, H, ld;
    float scale, p_drop, inv_keep;
    unsigned long long seed;
    const unsigned long long* seed_dev;     // null, or a device word added to `seed` at run time (attention_bwd.hip)
};

#define SBEV_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl, 0xf, 0xf, true))
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, SBEV_DPP(v, 0x140));
    v = fmaxf(v, SBEV_DPP(v, 0x141));
    v = fmaxf(v, SBEV_DPP(v, 0x4e));
    v = fmaxf(v, SBEV_DPP(v, 0xb1));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += SBEV_DPP(v, 0x140);
    v += SBEV_DPP(v, 0x141);
    v += SBEV_DPP(v, 0x4e);
    v += SBEV_DPP(v, 0xb1);
    return v;
}
#undef SBEV_DPP
#define MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

__device__ __forceinline__ unsigned mix32(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ bool keep_of(const Args& a, unsigned thr, int b, int h, int i, int j) {
    const unsigned long long idx = (((unsigned long long)b * a.H + h) * a.Q + i) * a.Q + j;
    return mix32((a.seed + (a.seed_dev ? *a.seed_dev : 0ull)) * 0x100000001b3ull + idx) >= thr;
}

// stage 64 rows x 32 floats of `src` (row r at src + rows[r] * ld) into tile[64][LDT], scaled; 64 lanes, one row each
__device__ __forceinline__ void stage_rows(float* tile, const float* src, long long ld, int row, int lane, float scale) {
    const float* p = src + (long long)row * ld;
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * d4);
        v *= scale;
        *reinterpret_cast<f32x4*>(&tile[lane * LDT + 4 * d4]) = v;
    }
}

// acc[c] (16 rows of A x 16 columns c*16 + fi of the staged tile) = sum_d A[row][d] * tile[col][d]
__device__ __forceinline__ void rows_times_tile_t(const f32x4 (&af)[HD / 16], const float* tile, int fi, int fk, f32x4 (&acc)[KT / 16]) {
#pragma unroll
    for (int c = 0; c < KT / 16; ++c) {
        acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < HD / 16; ++blk) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&tile[(c * 16 + fi) * LDT + 16 * blk + 4 * fk]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c] = MFMA16(af[blk][j], bq[j], acc[c]);
        }
    }
}

// out[t] (16 rows x dims 16 t + fi) += patch[16 rows][64] . tile[64][d]   (A from the wave's patch, B walks the tile's rows)
__device__ __forceinline__ void patch_times_tile(const float* patch, const float* tile, int fi, int fk, f32x4 (&out)[2]) {
#pragma unroll
    for (int blk = 0; blk < KT / 16; ++blk) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(&patch[fi * LDP + 16 * blk + 4 * fk]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* row = tile + (16 * blk + 4 * fk + j) * LDT;
            out[0] = MFMA16(pa[j], row[fi], out[0]);
            out[1] = MFMA16(pa[j], row[16 + fi], out[1]);
        }
    }
}

// ---- ROW kernel: dq, dtau, lse, D -------------------------------------------------------------------------------------
template <bool MASK, bool DROP>
__global__ __launch_bounds__(64 * NWAVES) void sasa_bwd_rows_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) float Ks[NS * KT * LDT];
    __shared__ __attribute__((aligned(16))) float Vs[NS * KT * LDT];
    __shared__ __attribute__((aligned(16))) float Cs[NS * KT * 2];
    __shared__ __attribute__((aligned(16))) float Ps[NWAVES * 16 * LDP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave % NG, ks = wave / NG;
    const int fi = lane & 15, fk = lane >> 4;
    const int qtiles = (a.Q + 16 * NG - 1) / (16 * NG);
    const int qt = blockIdx.x % qtiles;
    const int h = (blockIdx.x / qtiles) % a.H;
    const int b = blockIdx.x / (qtiles * a.H);
    const int D = a.H * HD;
    const float* base = a.qkvt + (long long)b * a.Q * a.ld;
    const int q0 = qt * 16 * NG + qg * 16;
    const unsigned thr = (unsigned)((double)a.p_drop * 4294967296.0);

    f32x4 qf[HD / 16], gf[HD / 16];
    {
        const int qi = min(q0 + fi, a.Q - 1);
#pragma unroll
        for (int blk = 0; blk < HD / 16; ++blk) {
            qf[blk] = *reinterpret_cast<const f32x4*>(base + (long long)qi * a.ld + h * HD + 16 * blk + 4 * fk) * a.scale;
            gf[blk] = *reinterpret_cast<const f32x4*>(a.dO + ((long long)b * a.Q + qi) * D + h * HD + 16 * blk + 4 * fk);
        }
    }
    float cx[4], cy[4], tau[4], drow[4];
    int qrow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qi = min(q0 + fk * 4 + e, a.Q - 1);
        qrow[e] = qi;
        cx[e] = a.bbox[((long long)b * a.Q + qi) * 10 + 0] * a.span[0] + a.lo[0];
        cy[e] = a.bbox[((long long)b * a.Q + qi) * 10 + 1] * a.span[1] + a.lo[1];
        tau[e] = base[(long long)qi * a.ld + 3 * D + h];
        const float* go = a.dO + ((long long)b * a.Q + qi) * D + h * HD;
        const float* oo = a.O + ((long long)b * a.Q + qi) * D + h * HD;
        drow[e] = row16_sum(go[fi] * oo[fi] + go[16 + fi] * oo[16 + fi]);
    }
    float* Pw = Ps + wave * 16 * LDP;
    float* Kw = Ks + ks * KT * LDT;
    float* Vw = Vs + ks * KT * LDT;
    float* Cw = Cs + ks * KT * 2;

    // one wave stages its own split's tile (64 keys: lane = key); all waves of a split see the same tile, so the two row groups
    // of a split share the staging work: row group 0 stages K (+ centres), row group 1 stages V
    auto stage = [&](int kbase, bool want_v) {
        const int kj = min(kbase + lane, a.Q - 1);
        if (qg == 0) {
            stage_rows(Kw, base + D + h * HD, a.ld, kj, lane, 1.f);
            Cw[2 * lane] = a.bbox[((long long)b * a.Q + kj) * 10] * a.span[0] + a.lo[0];
            Cw[2 * lane + 1] = a.bbox[((long long)b * a.Q + kj) * 10 + 1] * a.span[1] + a.lo[1];
        } else if (want_v) {
            stage_rows(Vw, base + 2 * D + h * HD, a.ld, kj, lane, 1.f);
        }
    };
    // S tile (+ bias, masks) of this wave: s[c][e] for key c*16 + fi, row fk*4 + e; dist kept for the dtau term
    auto s_tile = [&](int kbase, f32x4 (&s)[KT / 16], f32x4 (&dist)[KT / 16]) {
        rows_times_tile_t(qf, Kw, fi, fk, s);
#pragma unroll
        for (int c = 0; c < KT / 16; ++c) {
            const int kj = kbase + c * 16 + fi;
            const float kx = Cw[2 * (c * 16 + fi)], ky = Cw[2 * (c * 16 + fi) + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx = cx[e] - kx, dy = cy[e] - ky;
                const float di = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
                dist[c][e] = di;
                float v = s[c][e] - di * tau[e];
                bool dead = kj >= a.Q;
                if (MASK) dead = dead || a.mask[(long long)qrow[e] * a.Q + min(kj, a.Q - 1)] != 0;
                s[c][e] = dead ? -INFINITY : v;
            }
        }
    };

    // ---- pass 1: softmax statistics of this wave's key split ----
    float m_run[4], l_run[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m_run[e] = -INFINITY; l_run[e] = 0.f; }
    for (int k0 = 0; k0 < a.Q; k0 += NS * KT) {
        const int kbase = k0 + ks * KT;
        __syncthreads();
        if (kbase < a.Q) stage(kbase, false);
        __syncthreads();
        if (kbase < a.Q) {
            f32x4 s[KT / 16], dist[KT / 16];
            s_tile(kbase, s, dist);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float tmax = -INFINITY;
#pragma unroll
                for (int c = 0; c < KT / 16; ++c) tmax = fmaxf(tmax, s[c][e]);
                const float m_new = fmaxf(m_run[e], row16_max(tmax));
                const float m_use = m_new == -INFINITY ? 0.f : m_new;
                float psum = 0.f;
#pragma unroll
                for (int c = 0; c < KT / 16; ++c) psum += __expf(s[c][e] - m_use);
                l_run[e] = l_run[e] * __expf(m_run[e] - m_use) + row16_sum(psum);
                m_run[e] = m_new;
            }
        }
    }
    // merge (max, sum) over the NS key splits of each row group -> lse per row, known to every wave of the group
    __syncthreads();
    float* mg = Ps;                                         // [NG][NS][16 rows][2]
    if (fi == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mg[((qg * NS + ks) * 16 + fk * 4 + e) * 2] = m_run[e];
            mg[((qg * NS + ks) * 16 + fk * 4 + e) * 2 + 1] = l_run[e];
        }
    }
    __syncthreads();
    float lse[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float m = -INFINITY;
#pragma unroll
        for (int z = 0; z < NS; ++z) m = fmaxf(m, mg[((qg * NS + z) * 16 + fk * 4 + e) * 2]);
        float l = 0.f;
#pragma unroll
        for (int z = 0; z < NS; ++z) {
            const float mz = mg[((qg * NS + z) * 16 + fk * 4 + e) * 2];
            l += mz == -INFINITY ? 0.f : mg[((qg * NS + z) * 16 + fk * 4 + e) * 2 + 1] * __expf(mz - m);
        }
        lse[e] = l > 0.f ? m + __logf(l) : INFINITY;       // a fully masked row: every p below becomes exp(-inf) = 0
    }

    // ---- pass 2: dS tiles -> dq, dtau ----
    f32x4 dq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float dtau[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.Q; k0 += NS * KT) {
        const int kbase = k0 + ks * KT;
        __syncthreads();
        if (kbase < a.Q) stage(kbase, true);
        __syncthreads();
        if (kbase < a.Q) {
            f32x4 s[KT / 16], dist[KT / 16], dp[KT / 16];
            s_tile(kbase, s, dist);
            rows_times_tile_t(gf, Vw, fi, fk, dp);
#pragma unroll
            for (int c = 0; c < KT / 16; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __expf(s[c][e] - lse[e]);               // exp(-inf) = 0 for masked / padded keys
                    float g = dp[c][e];
                    if (DROP) g = keep_of(a, thr, b, h, qrow[e], min(kbase + c * 16 + fi, a.Q - 1)) ? g * a.inv_keep : 0.f;
                    const float ds = p * (g - drow[e]);
                    dtau[e] -= ds * dist[c][e];
                    Pw[(fk * 4 + e) * LDP + c * 16 + fi] = ds;
                }
            __builtin_amdgcn_wave_barrier();                // the patch is private to this wave (DS operations execute in order)
            patch_times_tile(Pw, Kw, fi, fk, dq);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dtau[e] = row16_sum(dtau[e]);
    // merge dq / dtau over the key splits (fixed order), split 0 of each row group writes
    __syncthreads();
    float* pr = Ks;                                         // [NS - 1][NG][12][64]
    if (ks > 0) {
        float* d = pr + ((ks - 1) * NG + qg) * 12 * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[e * 64 + lane] = dq[0][e];
            d[(4 + e) * 64 + lane] = dq[1][e];
            d[(8 + e) * 64 + lane] = dtau[e];
        }
    }
    __syncthreads();
    if (ks != 0) return;
#pragma unroll
    for (int z = 1; z < NS; ++z) {
        const float* d = pr + ((z - 1) * NG + qg) * 12 * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dq[0][e] += d[e * 64 + lane];
            dq[1][e] += d[(4 + e) * 64 + lane];
            dtau[e] += d[(8 + e) * 64 + lane];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qi = q0 + fk * 4 + e;
        if (qi >= a.Q) continue;
        float* g = a.dqkvt + ((long long)b * a.Q + qi) * a.ld;
        g[h * HD + fi] = dq[0][e] * a.scale;                 // C layout: column = dim fi (+16), row = query
        g[h * HD + 16 + fi] = dq[1][e] * a.scale;
        if (fi == 0) {
            g[3 * D + h] = dtau[e];
            a.lse[((long long)b * a.H + h) * a.Q + qi] = lse[e];
            a.dvec[((long long)b * a.H + h) * a.Q + qi] = drow[e];
        }
        if (h == 0 && 3 * D + a.H + fi < a.ld) g[3 * D + a.H + fi] = 0.f;   // padding columns (at most 3)
    }
}

// ---- COLUMN kernel: dk, dv (works on S^T = K Q^T tiles: rows = this wave's 16 keys, columns = queries) ------------------
template <bool MASK, bool DROP>
__global__ __launch_bounds__(64 * NWAVES) void sasa_bwd_cols_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) float Qs[NS * KT * LDT];
    __shared__ __attribute__((aligned(16))) float Gs[NS * KT * LDT];
    __shared__ __attribute__((aligned(16))) float Rs[NS * KT * 5];      // per query: cx, cy, tau, lse, D
    __shared__ __attribute__((aligned(16))) float Ps[NWAVES * 2 * 16 * LDP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave % NG, qs = wave / NG;
    const int fi = lane & 15, fk = lane >> 4;
    const int ktiles = (a.Q + 16 * NG - 1) / (16 * NG);
    const int kt = blockIdx.x % ktiles;
    const int h = (blockIdx.x / ktiles) % a.H;
    const int b = blockIdx.x / (ktiles * a.H);
    const int D = a.H * HD;
    const float* base = a.qkvt + (long long)b * a.Q * a.ld;
    const int j0 = kt * 16 * NG + kg * 16;                  // this wave's first key
    const unsigned thr = (unsigned)((double)a.p_drop * 4294967296.0);

    f32x4 kf[HD / 16], vf[HD / 16];
    {
        const int kj = min(j0 + fi, a.Q - 1);
#pragma unroll
        for (int blk = 0; blk < HD / 16; ++blk) {
            kf[blk] = *reinterpret_cast<const f32x4*>(base + (long long)kj * a.ld + D + h * HD + 16 * blk + 4 * fk);
            vf[blk] = *reinterpret_cast<const f32x4*>(base + (long long)kj * a.ld + 2 * D + h * HD + 16 * blk + 4 * fk);
        }
    }
    float kx[4], ky[4];
    int krow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int kj = min(j0 + fk * 4 + e, a.Q - 1);
        krow[e] = kj;
        kx[e] = a.bbox[((long long)b * a.Q + kj) * 10 + 0] * a.span[0] + a.lo[0];
        ky[e] = a.bbox[((long long)b * a.Q + kj) * 10 + 1] * a.span[1] + a.lo[1];
    }
    float* Pd = Ps + wave * 2 * 16 * LDP;                   // Pd^T patch
    float* Pg = Pd + 16 * LDP;                              // dS^T patch
    float* Qw = Qs + qs * KT * LDT;
    float* Gw = Gs + qs * KT * LDT;
    float* Rw = Rs + qs * KT * 5;
    f32x4 dk[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 dv[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

    for (int i0 = 0; i0 < a.Q; i0 += NS * KT) {
        const int ibase = i0 + qs * KT;
        __syncthreads();
        if (ibase < a.Q) {                                  // key group 0 stages Q (pre-scaled) + per-query scalars, group 1 stages dO
            const int qi = min(ibase + lane, a.Q - 1);
            if (kg == 0) {
                stage_rows(Qw, base + h * HD, a.ld, qi, lane, a.scale);
                Rw[lane * 5 + 0] = a.bbox[((long long)b * a.Q + qi) * 10] * a.span[0] + a.lo[0];
                Rw[lane * 5 + 1] = a.bbox[((long long)b * a.Q + qi) * 10 + 1] * a.span[1] + a.lo[1];
                Rw[lane * 5 + 2] = base[(long long)qi * a.ld + 3 * D + h];
                Rw[lane * 5 + 3] = a.lse[((long long)b * a.H + h) * a.Q + qi];
                Rw[lane * 5 + 4] = a.dvec[((long long)b * a.H + h) * a.Q + qi];
            } else {
                stage_rows(Gw, a.dO + (long long)b * a.Q * D + h * HD, D, qi, lane, 1.f);
            }
        }
        __syncthreads();
        if (ibase >= a.Q) continue;
        f32x4 st[KT / 16], dp[KT / 16];
        rows_times_tile_t(kf, Qw, fi, fk, st);              // S^T[key row][query col] (q pre-scaled)
        rows_times_tile_t(vf, Gw, fi, fk, dp);              // dPd^T = V dO^T
#pragma unroll
        for (int c = 0; c < KT / 16; ++c) {
            const int col = c * 16 + fi;
            const int qi = ibase + col;
            const float qx = Rw[col * 5], qy = Rw[col * 5 + 1], tq = Rw[col * 5 + 2], lq = Rw[col * 5 + 3], dq_ = Rw[col * 5 + 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx = qx - kx[e], dy = qy - ky[e];
                const float s = st[c][e] - __builtin_amdgcn_sqrtf(dx * dx + dy * dy) * tq;
                bool dead = qi >= a.Q || (j0 + fk * 4 + e) >= a.Q;
                if (MASK) dead = dead || a.mask[(long long)min(qi, a.Q - 1) * a.Q + krow[e]] != 0;
                const float p = dead ? 0.f : __expf(s - lq);
                bool keep = true;
                if (DROP) keep = keep_of(a, thr, b, h, min(qi, a.Q - 1), krow[e]);
                const float g = keep ? dp[c][e] * a.inv_keep : 0.f;
                Pd[(fk * 4 + e) * LDP + col] = keep ? p * a.inv_keep : 0.f;
                Pg[(fk * 4 + e) * LDP + col] = p * (g - dq_);
            }
        }
        __builtin_amdgcn_wave_barrier();
        patch_times_tile(Pd, Gw, fi, fk, dv);               // dv += Pd^T dO
        patch_times_tile(Pg, Qw, fi, fk, dk);               // dk += dS^T (q / sqrt(d))
    }
    // merge over the query splits, split 0 of each key group writes
    __syncthreads();
    float* pr = Qs;                                         // [NS - 1][NG][16][64]
    if (qs > 0) {
        float* d = pr + ((qs - 1) * NG + kg) * 16 * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[e * 64 + lane] = dk[0][e];
            d[(4 + e) * 64 + lane] = dk[1][e];
            d[(8 + e) * 64 + lane] = dv[0][e];
            d[(12 + e) * 64 + lane] = dv[1][e];
        }
    }
    __syncthreads();
    if (qs != 0) return;
#pragma unroll
    for (int z = 1; z < NS; ++z) {
        const float* d = pr + ((z - 1) * NG + kg) * 16 * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dk[0][e] += d[e * 64 + lane];
            dk[1][e] += d[(4 + e) * 64 + lane];
            dv[0][e] += d[(8 + e) * 64 + lane];
            dv[1][e] += d[(12 + e) * 64 + lane];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int kj = j0 + fk * 4 + e;
        if (kj >= a.Q) continue;
        float* g = a.dqkvt + ((long long)b * a.Q + kj) * a.ld;
        g[D + h * HD + fi] = dk[0][e];
        g[D + h * HD + 16 + fi] = dk[1][e];
        g[2 * D + h * HD + fi] = dv[0][e];
        g[2 * D + h * HD + 16 + fi] = dv[1][e];
    }
}

template <bool MASK, bool DROP>
int launch_both(const Args& a, hipStream_t s) {
    const long long blocks = (long long)a.B * a.H * ((a.Q + 16 * NG - 1) / (16 * NG));
    if (blocks > 0x7fffffffLL) {
        sbev::set_error("sbev_sasa_bwd_f32: too many blocks");
        return SBEV_EINVAL;
    }
    hipLaunchKernelGGL((sasa_bwd_rows_kernel<MASK, DROP>), dim3((unsigned)blocks), dim3(64 * NWAVES), 0, s, a);
    int st = sbev::check_launch("sbev_sasa_bwd_f32 (rows)");
    if (st != SBEV_OK) return st;
    hipLaunchKernelGGL((sasa_bwd_cols_kernel<MASK, DROP>), dim3((unsigned)blocks), dim3(64 * NWAVES), 0, s, a);
    return sbev::check_launch("sbev_sasa_bwd_f32 (columns)");
}

}  // namespace sbev_attn_bwd

namespace sbev {
// called by sbev_sasa_bwd_f32 (attention_bwd.hip) after argument validation
int launch_sasa_bwd_mfma(const float* qkvt, int64_t ld, const float* bbox, const float* lo, const float* span, const uint8_t* mask,
                         const float* out, const float* grad_out, float* grad_qkvt, float* lse, float* dvec,
                         int B, int Q, int H, float scale, float p_drop, uint64_t seed, const uint64_t* seed_dev, hipStream_t s) {
    sbev_attn_bwd::Args a{};
    a.qkvt = qkvt; a.bbox = bbox; a.mask = mask; a.O = out; a.dO = grad_out; a.dqkvt = grad_qkvt; a.lse = lse; a.dvec = dvec;
    a.B = B; a.Q = Q; a.H = H; a.ld = (int)ld; a.scale = scale; a.p_drop = p_drop; a.inv_keep = 1.f / (1.f - p_drop); a.seed = seed;
    a.seed_dev = reinterpret_cast<const unsigned long long*>(seed_dev);
    for (int i = 0; i < 2; ++i) { a.lo[i] = lo[i]; a.span[i] = span[i]; }
    const bool drop = p_drop > 0.f;
    if (mask) return drop ? sbev_attn_bwd::launch_both<true, true>(a, s) : sbev_attn_bwd::launch_both<true, false>(a, s);
    return drop ? sbev_attn_bwd::launch_both<false, true>(a, s) : sbev_attn_bwd::launch_both<false, false>(a, s);
}
}  // namespace sbev
