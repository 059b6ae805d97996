/*
 * oracle/pq_oracle.c — CPU restatement of MILLION's PQ-KV hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker.  The product path (million_amd/, bindings/) never links or calls it.
 *
 * Build: `make -C oracle` (gcc -O2 -ffp-contract=off: every fp32 operation below is a single IEEE
 * round-to-nearest operation, no FMA contraction, so the HIP encode kernel — which uses the same
 * operations in the same order — can be compared bit for bit).
 *
 * Parity status:
 *   - pq_encode_direct: restates the formula of reference scripts/utils/pq_utils.py:483-494
 *     (sa_encode_4d_keops: fp32 upcast, ((x-c)**2).sum(-1), argmin over c).  The arithmetic of the
 *     reference lives in third-party pykeops/keopscore 2.2.3 (requirements.txt), absent from
 *     /root/reference: tie rule and FMA contraction of KeOps are "parity unpinned".  Pinned instead
 *     against the importable reference sa_encode_4d (pq_utils.py:410-449, torch.cdist form) through
 *     tests/golden (agreement except documented near-tie flips).
 *   - pq_decode: reference pq_utils.py:501-540 (sa_decode_4d) — pinned bit-exactly by tests/golden.
 *   - decode_attn_*: the oracle formula of reference pq_utils.py:360-368 (non-causal softmax over
 *     [K_hat ; K_resid[:r]]) with scale 1/sqrt(d) (Kernel.cuh:48) and the GQA map hk = h/(nh/nh_k)
 *     (Kernel.cuh:52) — pinned by tests/golden (torch SDPA over sa_decode_4d output).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- PQ encode: direct form, fp32, sequential k, strict '<' scan in increasing c -------------
 * X      (n_vec, d) fp32   (fp16 values upcast exactly by the caller; pq_utils.py:483)
 * cents  (M, C, d_m) fp32  (pq_utils.py:484)
 * codes  (n_vec, M) u8     (row-major per vector; the reference's (bs,nh_k,n,M) contiguous layout,
 *                           pq_utils.py:497-499)
 * Lowest centroid index wins on exact ties (torch.argmin behaviour, pq_utils.py:447). */
void pq_encode_direct(const float *X, const float *cents, uint8_t *codes,
                      int64_t n_vec, int d, int M, int C)
{
    const int dm = d / M;
    for (int64_t i = 0; i < n_vec; ++i) {
        const float *x = X + i * (int64_t)d;
        for (int m = 0; m < M; ++m) {
            const float *xm = x + m * dm;
            const float *cm = cents + (int64_t)m * C * dm;
            float best = INFINITY;
            int best_c = 0;
            for (int c = 0; c < C; ++c) {
                const float *cc = cm + c * dm;
                float acc = 0.0f;
                for (int k = 0; k < dm; ++k) {
                    volatile float e = xm[k] - cc[k];   /* one rounding */
                    volatile float sq = e * e;          /* one rounding */
                    acc = acc + sq;                     /* one rounding, never fused */
                }
                if (acc < best) { best = acc; best_c = c; }
            }
            codes[i * (int64_t)M + m] = (uint8_t)best_c;
        }
    }
}

/* Same, but also returns the best and the second-best distance gap so tests can list near-ties. */
void pq_encode_direct_gap(const float *X, const float *cents, uint8_t *codes, float *gap,
                          int64_t n_vec, int d, int M, int C)
{
    const int dm = d / M;
    for (int64_t i = 0; i < n_vec; ++i) {
        const float *x = X + i * (int64_t)d;
        for (int m = 0; m < M; ++m) {
            const float *xm = x + m * dm;
            const float *cm = cents + (int64_t)m * C * dm;
            float best = INFINITY, second = INFINITY;
            int best_c = 0;
            for (int c = 0; c < C; ++c) {
                const float *cc = cm + c * dm;
                float acc = 0.0f;
                for (int k = 0; k < dm; ++k) {
                    volatile float e = xm[k] - cc[k];
                    volatile float sq = e * e;
                    acc = acc + sq;
                }
                if (acc < best) { second = best; best = acc; best_c = c; }
                else if (acc < second) { second = acc; }
            }
            codes[i * (int64_t)M + m] = (uint8_t)best_c;
            gap[i * (int64_t)M + m] = second - best;
        }
    }
}

/* ---- PQ decode (sa_decode_4d, pq_utils.py:501-540): x_hat[i, m*dm + k] = cents[m, code, k] ---- */
void pq_decode(const uint8_t *codes, const float *cents, float *out,
               int64_t n_vec, int d, int M, int C)
{
    const int dm = d / M;
    for (int64_t i = 0; i < n_vec; ++i)
        for (int m = 0; m < M; ++m) {
            const float *cc = cents + ((int64_t)m * C + codes[i * (int64_t)M + m]) * dm;
            for (int k = 0; k < dm; ++k) out[i * (int64_t)d + m * dm + k] = cc[k];
        }
}

/* ---- nbits 9..16: uint16 codes (reference nbits2dtype, pq_utils.py:542-552; sa_encode_4d(_keops) casts the argmin
 * indices with `.to(target_dtype)`, :449/:499; sa_decode_4d widens any code dtype to long, :525).  Same arithmetic and
 * tie rule as pq_encode_direct, C up to 65536. */
void pq_encode_direct_u16(const float *X, const float *cents, uint16_t *codes,
                          int64_t n_vec, int d, int M, int C)
{
    const int dm = d / M;
    for (int64_t i = 0; i < n_vec; ++i) {
        const float *x = X + i * (int64_t)d;
        for (int m = 0; m < M; ++m) {
            const float *xm = x + m * dm;
            const float *cm = cents + (int64_t)m * C * dm;
            float best = INFINITY;
            int best_c = 0;
            for (int c = 0; c < C; ++c) {
                const float *cc = cm + c * dm;
                float acc = 0.0f;
                for (int k = 0; k < dm; ++k) {
                    volatile float e = xm[k] - cc[k];
                    volatile float sq = e * e;
                    acc = acc + sq;
                }
                if (acc < best) { best = acc; best_c = c; }
            }
            codes[i * (int64_t)M + m] = (uint16_t)best_c;
        }
    }
}

void pq_decode_u16(const uint16_t *codes, const float *cents, float *out,
                   int64_t n_vec, int d, int M, int C)
{
    const int dm = d / M;
    for (int64_t i = 0; i < n_vec; ++i)
        for (int m = 0; m < M; ++m) {
            const float *cc = cents + ((int64_t)m * C + codes[i * (int64_t)M + m]) * dm;
            for (int k = 0; k < dm; ++k) out[i * (int64_t)d + m * dm + k] = cc[k];
        }
}

/* ---- decode-step attention, fp64 accumulation (the gold) --------------------------------------
 * q        (bs, nh, d) fp32
 * k_codes  (bs, nh_k, T, M) u8 ; v_codes (bs, nh_k, T, M) u8   (row-major, Interface.template.cu:29-30)
 * k_cents, v_cents (M, C, dm) fp32
 * k_res, v_res (bs, nh_k, Lt, d) fp32, first r rows valid (Interface.template.cu:33-35)
 * out (bs, nh, d) fp64 ; lse (bs, nh) fp64 (natural log; may be NULL)
 * Formula: reference pq_utils.py:360-368; scale Kernel.cuh:48; GQA Kernel.cuh:52. */
void decode_attn_f64(const float *q, const uint8_t *k_codes, const uint8_t *v_codes,
                     const float *k_cents, const float *v_cents,
                     const float *k_res, const float *v_res,
                     double *out, double *lse,
                     int bs, int nh, int nh_k, int64_t T, int r, int Lt, int d, int M, int C)
{
    const int dm = d / M;
    const int G = nh / nh_k;
    const double scale = 1.0 / sqrt((double)d);
    const int64_t N = T + r;
    double *s = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    double *lut = (double *)malloc(sizeof(double) * (size_t)M * C);
    for (int b = 0; b < bs; ++b)
        for (int h = 0; h < nh; ++h) {
            const int hk = h / G;
            const float *qv = q + ((int64_t)b * nh + h) * d;
            /* LUT (Interface.template.cu:49-50): lut[m][c] = q[m,:] . k_cents[m,c,:] */
            for (int m = 0; m < M; ++m)
                for (int c = 0; c < C; ++c) {
                    double a = 0.0;
                    for (int k = 0; k < dm; ++k)
                        a += (double)qv[m * dm + k] * (double)k_cents[((int64_t)m * C + c) * dm + k];
                    lut[m * C + c] = a;
                }
            const uint8_t *kc = k_codes + ((int64_t)b * nh_k + hk) * T * M;
            const uint8_t *vc = v_codes + ((int64_t)b * nh_k + hk) * T * M;
            const float *kr = k_res + ((int64_t)b * nh_k + hk) * (int64_t)Lt * d;
            const float *vr = v_res + ((int64_t)b * nh_k + hk) * (int64_t)Lt * d;
            double mx = -INFINITY;
            for (int64_t t = 0; t < T; ++t) {
                double a = 0.0;
                for (int m = 0; m < M; ++m) a += lut[m * C + kc[t * M + m]];
                s[t] = a * scale;
                if (s[t] > mx) mx = s[t];
            }
            for (int j = 0; j < r; ++j) {
                double a = 0.0;
                for (int k = 0; k < d; ++k) a += (double)qv[k] * (double)kr[(int64_t)j * d + k];
                s[T + j] = a * scale;
                if (s[T + j] > mx) mx = s[T + j];
            }
            double *o = out + ((int64_t)b * nh + h) * d;
            for (int k = 0; k < d; ++k) o[k] = 0.0;
            double l = 0.0;
            for (int64_t t = 0; t < T; ++t) {
                const double p = exp(s[t] - mx);
                l += p;
                for (int m = 0; m < M; ++m) {
                    const float *cc = v_cents + ((int64_t)m * C + vc[t * M + m]) * dm;
                    for (int k = 0; k < dm; ++k) o[m * dm + k] += p * (double)cc[k];
                }
            }
            for (int j = 0; j < r; ++j) {
                const double p = exp(s[T + j] - mx);
                l += p;
                for (int k = 0; k < d; ++k) o[k] += p * (double)vr[(int64_t)j * d + k];
            }
            if (N > 0) {
                for (int k = 0; k < d; ++k) o[k] /= l;
                if (lse) lse[(int64_t)b * nh + h] = log(l) + mx;
            } else if (lse) {
                lse[(int64_t)b * nh + h] = -INFINITY;
            }
        }
    free(s);
    free(lut);
}

/* ---- decode-step attention in the reference's split structure, fp32 ---------------------------
 * Follows the launch structure of flash_decoding_allocated_buffer (Interface.template.cu:26-120):
 *   Ls = ceil(T/Ns) (:45); per split sid, tokens [sid*Ls, min((sid+1)*Ls, T)) normalised partial
 *   out/sum and lse = log(sum)+max (Kernel.cuh:161-165); residual partial in slot Ns
 *   (Kernel.cuh:1163,1204); LSE merge w_i = exp(lse_i - L)/sum (Kernel.cuh:1249-1269).
 * All arithmetic fp32 (the reference's is fp16; see SURVEY.md 7 "Reference numerics").
 * [QUIRK not reproduced] an empty split writes 0/0 = NaN in the reference; here lse=-inf, out=0.
 * partial_out (bs, nh, Ns+1, d) fp32 ; partial_lse (bs, nh, Ns+1) fp32 ; out (bs, nh, d) fp32. */
void decode_attn_split_f32(const float *q, const uint8_t *k_codes, const uint8_t *v_codes,
                           const float *k_cents, const float *v_cents,
                           const float *k_res, const float *v_res,
                           float *partial_out, float *partial_lse, float *out,
                           int bs, int nh, int nh_k, int64_t T, int r, int Lt, int d, int M, int C,
                           int Ns)
{
    const int dm = d / M;
    const int G = nh / nh_k;
    const float scale = 1.0f / sqrtf((float)d);
    const int64_t Ls = (T + Ns - 1) / Ns;
    float *lut = (float *)malloc(sizeof(float) * (size_t)M * C);
    float *s = (float *)malloc(sizeof(float) * (size_t)((Ls > Lt ? Ls : Lt) + 1));
    for (int b = 0; b < bs; ++b)
        for (int h = 0; h < nh; ++h) {
            const int hk = h / G;
            const float *qv = q + ((int64_t)b * nh + h) * d;
            for (int m = 0; m < M; ++m)
                for (int c = 0; c < C; ++c) {
                    float a = 0.0f;
                    for (int k = 0; k < dm; ++k)
                        a += qv[m * dm + k] * k_cents[((int64_t)m * C + c) * dm + k];
                    lut[m * C + c] = a;
                }
            const uint8_t *kc = k_codes + ((int64_t)b * nh_k + hk) * T * M;
            const uint8_t *vc = v_codes + ((int64_t)b * nh_k + hk) * T * M;
            float *po = partial_out + ((int64_t)b * nh + h) * (Ns + 1) * d;
            float *pl = partial_lse + ((int64_t)b * nh + h) * (Ns + 1);
            for (int sid = 0; sid <= Ns; ++sid) {
                float *o = po + (int64_t)sid * d;
                for (int k = 0; k < d; ++k) o[k] = 0.0f;
                int64_t n = 0;
                float mx = -INFINITY;
                if (sid < Ns) {
                    const int64_t j0 = sid * Ls;
                    const int64_t j1 = (sid + 1) * Ls < T ? (sid + 1) * Ls : T;
                    for (int64_t t = j0; t < j1; ++t) {
                        float a = 0.0f;
                        for (int m = 0; m < M; ++m) a += lut[m * C + kc[t * M + m]];
                        s[n] = a * scale;
                        if (s[n] > mx) mx = s[n];
                        ++n;
                    }
                    float l = 0.0f;
                    for (int64_t i = 0; i < n; ++i) {
                        const float p = expf(s[i] - mx);
                        l += p;
                        const int64_t t = j0 + i;
                        for (int m = 0; m < M; ++m) {
                            const float *cc = v_cents + ((int64_t)m * C + vc[t * M + m]) * dm;
                            for (int k = 0; k < dm; ++k) o[m * dm + k] += p * cc[k];
                        }
                    }
                    if (n > 0) { for (int k = 0; k < d; ++k) o[k] /= l; pl[sid] = logf(l) + mx; }
                    else pl[sid] = -INFINITY;
                } else {
                    const float *kr = k_res + ((int64_t)b * nh_k + hk) * (int64_t)Lt * d;
                    const float *vr = v_res + ((int64_t)b * nh_k + hk) * (int64_t)Lt * d;
                    for (int j = 0; j < r; ++j) {
                        float a = 0.0f;
                        for (int k = 0; k < d; ++k) a += qv[k] * kr[(int64_t)j * d + k];
                        s[j] = a * scale;
                        if (s[j] > mx) mx = s[j];
                    }
                    float l = 0.0f;
                    for (int j = 0; j < r; ++j) {
                        const float p = expf(s[j] - mx);
                        l += p;
                        for (int k = 0; k < d; ++k) o[k] += p * vr[(int64_t)j * d + k];
                    }
                    if (r > 0) { for (int k = 0; k < d; ++k) o[k] /= l; pl[sid] = logf(l) + mx; }
                    else pl[sid] = -INFINITY;
                }
            }
            /* merge (Kernel.cuh:1249-1269) */
            float L = -INFINITY;
            for (int i = 0; i <= Ns; ++i) if (pl[i] > L) L = pl[i];
            float *oo = out + ((int64_t)b * nh + h) * d;
            for (int k = 0; k < d; ++k) oo[k] = 0.0f;
            if (L > -INFINITY) {
                float denom = 0.0f;
                for (int i = 0; i <= Ns; ++i) denom += expf(pl[i] - L);
                for (int i = 0; i <= Ns; ++i) {
                    const float w = expf(pl[i] - L) / denom;
                    if (w > 0.0f)
                        for (int k = 0; k < d; ++k) oo[k] += po[(int64_t)i * d + k] * w;
                }
            }
        }
    free(lut);
    free(s);
}
