/*
 * oracle/selftest.c — sanitizer run of the CPU checker (TEST INFRASTRUCTURE ONLY, like pq_oracle.c).
 *
 *   gcc -std=c11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off \
 *       -o selftest selftest.c pq_oracle.c -lm && ./selftest
 *
 * Exercises every exported function of pq_oracle.c on small random inputs at the edge shapes the tests use (T = 0, r = 0,
 * one vector, ragged split counts, uint16 codes) under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU
 * sanitizers are not available on the pool), and checks three invariants: decode(encode(x)) is the nearest centroid,
 * the split / LSE-merge form agrees with the fp64 form, and the gap variant returns the same codes.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void pq_encode_direct(const float *X, const float *cents, uint8_t *codes, int64_t n_vec, int d, int M, int C);
void pq_encode_direct_gap(const float *X, const float *cents, uint8_t *codes, float *gap, int64_t n_vec, int d, int M, int C);
void pq_decode(const uint8_t *codes, const float *cents, float *out, int64_t n_vec, int d, int M, int C);
void pq_encode_direct_u16(const float *X, const float *cents, uint16_t *codes, int64_t n_vec, int d, int M, int C);
void pq_decode_u16(const uint16_t *codes, const float *cents, float *out, int64_t n_vec, int d, int M, int C);
void decode_attn_f64(const float *q, const uint8_t *k_codes, const uint8_t *v_codes, const float *k_cents, const float *v_cents,
                     const float *k_res, const float *v_res, double *out, double *lse, int bs, int nh, int nh_k, int64_t T,
                     int r, int Lt, int d, int M, int C);
void decode_attn_split_f32(const float *q, const uint8_t *k_codes, const uint8_t *v_codes, const float *k_cents,
                           const float *v_cents, const float *k_res, const float *v_res, float *partial_out,
                           float *partial_lse, float *out, int bs, int nh, int nh_k, int64_t T, int r, int Lt, int d, int M,
                           int C, int Ns);

static unsigned long long rng = 88172645463325252ull;
static float frand(void) {      /* xorshift, roughly N(0,1) by summing uniforms */
    float s = 0.f;
    for (int i = 0; i < 4; ++i) {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        s += (float)(rng >> 40) / 16777216.0f - 0.5f;
    }
    return s * 1.7320508f;
}
static float *fvec(size_t n) {
    float *p = (float *)malloc(sizeof(float) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) p[i] = frand();
    return p;
}

static int check_attn(int bs, int nh, int nh_k, int64_t T, int r, int Lt, int d, int M, int C, int Ns) {
    float *q = fvec((size_t)bs * nh * d), *kc = fvec((size_t)M * C * (d / M)), *vc = fvec((size_t)M * C * (d / M));
    float *kr = fvec((size_t)bs * nh_k * Lt * d), *vr = fvec((size_t)bs * nh_k * Lt * d);
    size_t nc = (size_t)bs * nh_k * (size_t)T * M;
    uint8_t *kk = (uint8_t *)malloc(nc ? nc : 1), *vv = (uint8_t *)malloc(nc ? nc : 1);
    for (size_t i = 0; i < nc; ++i) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; kk[i] = (uint8_t)(rng % C); vv[i] = (uint8_t)((rng >> 20) % C); }
    double *o64 = (double *)malloc(sizeof(double) * bs * nh * d), *lse = (double *)malloc(sizeof(double) * bs * nh);
    float *po = (float *)malloc(sizeof(float) * bs * nh * (Ns + 1) * d), *pl = (float *)malloc(sizeof(float) * bs * nh * (Ns + 1));
    float *o32 = (float *)malloc(sizeof(float) * bs * nh * d);
    decode_attn_f64(q, kk, vv, kc, vc, kr, vr, o64, lse, bs, nh, nh_k, T, r, Lt, d, M, C);
    decode_attn_f64(q, kk, vv, kc, vc, kr, vr, o64, NULL, bs, nh, nh_k, T, r, Lt, d, M, C);
    decode_attn_split_f32(q, kk, vv, kc, vc, kr, vr, po, pl, o32, bs, nh, nh_k, T, r, Lt, d, M, C, Ns);
    double num = 0, den = 0;
    for (int i = 0; i < bs * nh * d; ++i) { num += (o64[i] - o32[i]) * (o64[i] - o32[i]); den += o64[i] * o64[i]; }
    int bad = (T + r > 0) && !(sqrt(num) <= 1e-4 * sqrt(den) + 1e-9);
    if (bad) printf("attn mismatch bs=%d nh=%d T=%lld r=%d Ns=%d: %g / %g\n", bs, nh, (long long)T, r, Ns, sqrt(num), sqrt(den));
    free(q); free(kc); free(vc); free(kr); free(vr); free(kk); free(vv); free(o64); free(lse); free(po); free(pl); free(o32);
    return bad;
}

int main(void) {
    int bad = 0;
    /* encode / decode: the decoded vector must be at least as close as every other centroid */
    const int shapes[][3] = {{128, 64, 256}, {128, 32, 256}, {64, 16, 128}, {64, 64, 7}, {128, 16, 1}};
    for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s) {
        const int d = shapes[s][0], M = shapes[s][1], C = shapes[s][2], dm = d / M;
        for (int64_t n = 0; n <= 33; n += 11) {
            float *X = fvec((size_t)n * d), *cents = fvec((size_t)M * C * dm), *dec = fvec((size_t)n * d), *gap = fvec((size_t)n * M);
            uint8_t *codes = (uint8_t *)malloc((size_t)n * M + 1), *codes2 = (uint8_t *)malloc((size_t)n * M + 1);
            pq_encode_direct(X, cents, codes, n, d, M, C);
            pq_encode_direct_gap(X, cents, codes2, gap, n, d, M, C);
            pq_decode(codes, cents, dec, n, d, M, C);
            for (int64_t i = 0; i < n; ++i)
                for (int m = 0; m < M; ++m) {
                    if (codes[i * M + m] != codes2[i * M + m] || codes[i * M + m] >= C) { ++bad; continue; }
                    float best = 0.f;
                    for (int k = 0; k < dm; ++k) { float e = X[i * d + m * dm + k] - dec[i * d + m * dm + k]; best += e * e; }
                    for (int c = 0; c < C; ++c) {
                        float acc = 0.f;
                        for (int k = 0; k < dm; ++k) { float e = X[i * d + m * dm + k] - cents[((size_t)m * C + c) * dm + k]; acc += e * e; }
                        if (acc < best * (1.f - 1e-5f)) ++bad;
                    }
                }
            free(X); free(cents); free(dec); free(gap); free(codes); free(codes2);
        }
    }
    {   /* uint16 codes */
        const int d = 128, M = 32, C = 700, dm = d / M;
        const int64_t n = 9;
        float *X = fvec((size_t)n * d), *cents = fvec((size_t)M * C * dm), *dec = fvec((size_t)n * d);
        uint16_t *codes = (uint16_t *)malloc(sizeof(uint16_t) * n * M);
        pq_encode_direct_u16(X, cents, codes, n, d, M, C);
        pq_decode_u16(codes, cents, dec, n, d, M, C);
        for (int64_t i = 0; i < n * M; ++i) bad += codes[i] >= C;
        free(X); free(cents); free(dec); free(codes);
    }
    /* attention: empty store, empty window, both, ragged splits, GQA and MHA */
    bad += check_attn(1, 8, 2, 0, 17, 128, 128, 64, 256, 2);
    bad += check_attn(1, 8, 2, 129, 0, 128, 128, 64, 256, 4);
    bad += check_attn(1, 4, 4, 0, 0, 128, 128, 64, 256, 1);
    bad += check_attn(2, 4, 4, 1000, 128, 128, 128, 64, 256, 16);
    bad += check_attn(1, 8, 1, 37, 5, 64, 64, 16, 128, 32);
    bad += check_attn(1, 6, 2, 300, 33, 256, 128, 32, 256, 8);
    printf(bad ? "selftest FAILED (%d)\n" : "selftest ok\n", bad);
    return bad ? 1 : 0;
}
