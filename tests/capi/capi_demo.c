/*
 * The C boundary used from plain C: no Python, no torch.  Allocates device buffers with the HIP runtime, calls
 * sbev_msmv_fwd (include/sbev_hip.h) on a seeded 4-level pyramid in the reference layout [B',N,H,W,C] and checks the
 * result against the scalar C oracle (oracle/msmv_oracle.c, linked in as the CHECKER), then checks the error contract
 * (negative status + sbev_last_error()).  Built by tests/capi/Makefile; run by tests/test_gpu_capi_c.py.
 * Exit status 0 = parity within 1e-5 and errors reported as documented.
 */
#include <hip/hip_runtime_api.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sbev_hip.h"

void oracle_msmv_fwd(const float* const* feats, const int32_t* hw, int L, int64_t Bp, int N, int C, int Q, int P,
                     const float* loc, const float* w, float* out);

static uint32_t lcg_state = 12345u;
static float frand(void) { /* uniform [0, 1) */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (1.0f / 16777216.0f);
}

#define CHECK_HIP(x)                                                              \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));        \
            return 2;                                                             \
        }                                                                         \
    } while (0)

int main(void) {
    enum { L = 4, BP = 8, N = 6, C = 64, Q = 40, P = 4 };
    const int32_t hw[2 * L] = {16, 44, 8, 22, 4, 11, 2, 6};
    if (sbev_abi_version() != 1) { fprintf(stderr, "unexpected ABI version %d\n", sbev_abi_version()); return 1; }

    float* h_feat[L];
    void* d_feat[L];
    int64_t stride_bo[L], stride_v[L];
    for (int l = 0; l < L; ++l) {
        const size_t n = (size_t)BP * N * hw[2 * l] * hw[2 * l + 1] * C;
        h_feat[l] = (float*)malloc(n * sizeof(float));
        for (size_t i = 0; i < n; ++i) h_feat[l][i] = 2.f * frand() - 1.f;
        CHECK_HIP(hipMalloc(&d_feat[l], n * sizeof(float)));
        CHECK_HIP(hipMemcpy(d_feat[l], h_feat[l], n * sizeof(float), hipMemcpyHostToDevice));
        stride_v[l] = (int64_t)hw[2 * l] * hw[2 * l + 1] * C;
        stride_bo[l] = stride_v[l] * N;
    }
    const size_t n_loc = (size_t)BP * Q * P * 3, n_w = (size_t)BP * Q * P * L, n_out = (size_t)BP * Q * C * P;
    float* h_loc = (float*)malloc(n_loc * sizeof(float));
    float* h_w = (float*)malloc(n_w * sizeof(float));
    for (size_t i = 0; i < n_loc; i += 3) {
        h_loc[i + 0] = 1.3f * frand() - 0.15f;                 /* some points just outside the maps */
        h_loc[i + 1] = 1.3f * frand() - 0.15f;
        h_loc[i + 2] = (float)((int)(frand() * N) % N) / (float)(N - 1);
    }
    h_loc[0] = 0.f; h_loc[1] = 1.f;                            /* exact corners */
    h_loc[3] = 1.f; h_loc[4] = 0.f;
    for (size_t i = 0; i < n_w; ++i) h_w[i] = frand();
    float *d_loc, *d_w, *d_out;
    CHECK_HIP(hipMalloc((void**)&d_loc, n_loc * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_w, n_w * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_out, n_out * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_loc, h_loc, n_loc * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_w, h_w, n_w * sizeof(float), hipMemcpyHostToDevice));

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    int rc = sbev_msmv_fwd((const void* const*)d_feat, hw, L, SBEV_F32, BP, N, C, Q, P, 1, stride_bo, 0, stride_v, C,
                           d_loc, d_w, d_out, SBEV_OUT_REF, 1, 1, (sbev_stream_t)stream);
    if (rc != SBEV_OK) { fprintf(stderr, "sbev_msmv_fwd: %d (%s)\n", rc, sbev_last_error()); return 1; }
    CHECK_HIP(hipStreamSynchronize(stream));
    float* h_out = (float*)malloc(n_out * sizeof(float));
    float* h_ref = (float*)malloc(n_out * sizeof(float));
    CHECK_HIP(hipMemcpy(h_out, d_out, n_out * sizeof(float), hipMemcpyDeviceToHost));
    oracle_msmv_fwd((const float* const*)h_feat, hw, L, BP, N, C, Q, P, h_loc, h_w, h_ref);
    double worst = 0.0, ref_abs = 0.0;
    for (size_t i = 0; i < n_out; ++i) {
        const double d = fabs((double)h_out[i] - (double)h_ref[i]);
        if (d > worst) worst = d;
        ref_abs += fabs((double)h_ref[i]);
    }
    printf("sbev_msmv_fwd from C: %zu outputs, mean |ref| %.3f, max |hip - oracle| %.3e\n", n_out, ref_abs / (double)n_out, worst);
    if (!(worst < 1e-5) || !(ref_abs > 0.0)) return 1;

    /* error contract: unsupported level count -> negative status, message available, nothing launched */
    rc = sbev_msmv_fwd((const void* const*)d_feat, hw, 6, SBEV_F32, BP, N, C, Q, P, 1, stride_bo, 0, stride_v, C,
                       d_loc, d_w, d_out, SBEV_OUT_REF, 1, 1, (sbev_stream_t)stream);
    if (rc >= 0 || strlen(sbev_last_error()) == 0) { fprintf(stderr, "L = 6 was not rejected (rc %d)\n", rc); return 1; }
    printf("L = 6 rejected: %d (%s)\n", rc, sbev_last_error());
    rc = sbev_msmv_fwd((const void* const*)d_feat, hw, L, SBEV_F32, BP, N, C, Q, 33, 1, stride_bo, 0, stride_v, C,
                       d_loc, d_w, d_out, SBEV_OUT_REF, 1, 1, (sbev_stream_t)stream);
    if (rc >= 0) { fprintf(stderr, "P = 33 was not rejected\n"); return 1; }
    printf("P = 33 rejected: %d (%s)\n", rc, sbev_last_error());

    /* round 5: the decoder step's last launch from C -- nan_to_num (models/sparsebev_transformer.py:35-36) of two buffers in one launch,
       checked against the scalar rule; and the pair tail's fault word, which a C caller polls without synchronising anything */
    {
        enum { NA = 1027, NB = 38 };                              /* odd sizes: the scalar tail of the vector path */
        float h_a[NA], h_b[NB], r_a[NA], r_b[NB];
        const float specials[8] = {NAN, INFINITY, -INFINITY, -0.0f, 1e-42f, -1e-42f, FLT_MAX, -FLT_MAX};
        for (int i = 0; i < NA; ++i) h_a[i] = i < 8 ? specials[i] : 4.f * frand() - 2.f;
        for (int i = 0; i < NB; ++i) h_b[i] = i % 5 == 0 ? specials[(i / 5) % 8] : frand();
        float *d_a, *d_b, *d_ao, *d_bo;
        CHECK_HIP(hipMalloc((void**)&d_a, sizeof h_a)); CHECK_HIP(hipMalloc((void**)&d_b, sizeof h_b));
        CHECK_HIP(hipMalloc((void**)&d_ao, sizeof h_a)); CHECK_HIP(hipMalloc((void**)&d_bo, sizeof h_b));
        CHECK_HIP(hipMemcpy(d_a, h_a, sizeof h_a, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_b, h_b, sizeof h_b, hipMemcpyHostToDevice));
        rc = sbev_finish_outputs(d_a, d_b, d_ao, d_bo, NA, NB, (sbev_stream_t)stream);
        if (rc != SBEV_OK) { fprintf(stderr, "sbev_finish_outputs: %d (%s)\n", rc, sbev_last_error()); return 1; }
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(r_a, d_ao, sizeof h_a, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(r_b, d_bo, sizeof h_b, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < NA + NB; ++i) {
            const float x = i < NA ? h_a[i] : h_b[i - NA], got = i < NA ? r_a[i] : r_b[i - NA];
            const float want = isnan(x) ? 0.f : isinf(x) ? (x > 0 ? FLT_MAX : -FLT_MAX) : x;
            bad += memcmp(&got, &want, sizeof got) != 0;          /* bit for bit: -0, denormals, FLT_MAX unchanged */
        }
        printf("sbev_finish_outputs from C: %d + %d values, %d differ from the scalar nan_to_num\n", NA, NB, bad);
        if (bad) return 1;
        if (sbev_decoder_chain_pair_faults() != 0 || sbev_decoder_chain_pair_faults_ack() != 0) { fprintf(stderr, "fault word not clean\n"); return 1; }
        printf("pair fault word: clean\n");
    }
    return 0;
}
