/*
 * cabi_bench.c - a plain-C caller of libmillion_hip.so (include/million_hip.h): no Python, no torch, HIP runtime API only.
 *
 * What the reference ships in this role: scripts/modeldb/bindings/Kernel_Test/main.cu:59-226, a C++ harness that fills
 * random codes / centroids / residuals, launches the decode kernels directly and times them.  This program does the same
 * THROUGH THE C ABI, i.e. exactly as a non-Python host would drive the path:
 *   1. a random fp16 codebook pair            -> million_prepare_cents
 *   2. a random fp16 prompt (K and V rows)    -> million_pq_encode into K pages / transposed V pages (per layer)
 *   3. N decode launches                      -> million_pq_decode_attn_append (fused window append + attention), captured
 *                                                 into ONE hipGraph over rotating layers and replayed (how tools/ab_bench.py
 *                                                 and bench.py's roofline region time the launch), and eagerly for comparison
 *   4. a host check, self-contained (no oracle/, no Python): sampled codes against a direct fp32 argmin on the host, and the
 *      attention output of the query heads of kv head 0 against a double-precision softmax over the dequantised codes.
 *
 * Build:  make cabi-bench      (gcc -std=c99; links libmillion_hip.so and the HIP runtime)
 * Run:    build/cabi_bench [--bs 1] [--ctx 32768] [--M 64] [--layers 32] [--launches 96] [--reps 5] [--random-codes 1]
 *         --random-codes 1: after the encode, overwrite the pages with uniformly random bytes - the data tools/ab_bench.py and the
 *         reference's micro-benchmark use (test_kernel.py:59-65) - so that the two programs time the same launch on the same kind of
 *         data (codes of encoded Gaussian rows are not uniform: fewer distinct centroids per LDS gather, ~2-4 % shorter launches)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "million_hip.h"

#define HIP_OK(x)                                                                                        \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) {                                                                          \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
            exit(2);                                                                                     \
        }                                                                                                \
    } while (0)
#define MIL_OK(x)                                                                                        \
    do {                                                                                                 \
        int rc_ = (x);                                                                                   \
        if (rc_ != MILLION_OK) {                                                                         \
            fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, million_last_error()); \
            exit(3);                                                                                     \
        }                                                                                                \
    } while (0)

/* ---- host-side fp16 <-> fp32 (IEEE binary16, round to nearest even) and a small generator ---- */
static uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));      /* overflow / inf / nan */
    if (x < 0x38800000u) {                                                                       /* subnormal or zero */
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 113 - (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t half = 1u << (shift + 12), rest = m & ((half << 1) - 1);
        m >>= shift + 13;
        if (rest > half || (rest == half && (m & 1))) ++m;
        return (uint16_t)(sign | m);
    }
    uint32_t m = x - 0x38000000u;                                                                /* rebias 127 -> 15 */
    const uint32_t rest = m & 0x1fffu;
    m >>= 13;
    if (rest > 0x1000u || (rest == 0x1000u && (m & 1))) ++m;
    return (uint16_t)(sign | m);
}
static float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int s = 0;
            while (!(m & 0x400u)) { m <<= 1; ++s; }
            x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
static uint64_t g_rng = 0x9e3779b97f4a7c15ull;
static uint32_t rnd32(void) {      /* xorshift64* */
    g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
    return (uint32_t)((g_rng * 0x2545f4914f6cdd1dull) >> 32);
}
static float rnd_normal(void) {    /* sum of 4 uniforms, variance 1: close enough to N(0,1) for synthetic K/V rows */
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += (float)(rnd32() >> 8) * (1.0f / 16777216.0f);
    return (s - 2.0f) * 1.7320508f;
}
static void fill_normal_f16(uint16_t *p, size_t n) { for (size_t i = 0; i < n; ++i) p[i] = f2h(rnd_normal()); }

static void *dev_alloc(size_t bytes) { void *p = NULL; HIP_OK(hipMalloc(&p, bytes ? bytes : 16)); return p; }
static void *dev_upload(const void *src, size_t bytes) { void *p = dev_alloc(bytes); HIP_OK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice)); return p; }

static int arg_int(int argc, char **argv, const char *name, int dflt) {
    for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], name)) return atoi(argv[i + 1]);
    return dflt;
}
static int cmp_float(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }

int main(int argc, char **argv) {
    const int bs = arg_int(argc, argv, "--bs", 1), T = arg_int(argc, argv, "--ctx", 32768) / 64 * 64, M = arg_int(argc, argv, "--M", 64);
    const int layers = arg_int(argc, argv, "--layers", 32), launches = arg_int(argc, argv, "--launches", 96), reps = arg_int(argc, argv, "--reps", 5);
    const int nh = arg_int(argc, argv, "--nh", 32), nhk = arg_int(argc, argv, "--nh-k", 8), d = 128, C = 256, ps = 64, cap = 128, r = 99;
    const int dm = d / M, G = nh / nhk, n_pages = T / ps, random_codes = arg_int(argc, argv, "--random-codes", 0);
    if (bs < 1 || T < 64 || layers < 1 || launches < 1 || reps < 1 || d % M || nh % nhk) { fprintf(stderr, "bad arguments\n"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "cabi_bench: no GPU (this program has no CPU path)\n"); return 1; }
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    printf("cabi_bench: libmillion_hip ABI v%d; bs=%d nh=%d nh_k=%d d=%d M=%d C=%d T=%d page=%d window=%d r=%d+1, %d layers, %d launches per graph\n",
           million_version(), bs, nh, nhk, d, M, C, T, ps, cap, r, layers, launches);

    /* 1. codebooks */
    const size_t tab = (size_t)M * C * dm;
    uint16_t *h_kc = malloc(tab * 2), *h_vc = malloc(tab * 2);
    fill_normal_f16(h_kc, tab);
    fill_normal_f16(h_vc, tab);
    void *d_kc = dev_upload(h_kc, tab * 2), *d_vc = dev_upload(h_vc, tab * 2);
    const size_t prep_bytes = million_prepared_cents_bytes(M, C, dm);
    void *d_kprep = dev_alloc(prep_bytes), *d_vprep = dev_alloc(prep_bytes);
    MIL_OK(million_prepare_cents(d_kc, M, C, dm, d_kprep, stream));
    MIL_OK(million_prepare_cents(d_vc, M, C, dm, d_vprep, stream));

    /* 2. the prompt: K and V rows (bs, nh_k, T, d), encoded per layer from a different ring start (x_row_start / x_row_mod:
     *    the encoder's residual-ring addressing), so every layer's pages hold different bytes and the layers' pools cannot
     *    share cache lines */
    const size_t x_elems = (size_t)bs * nhk * T * d;
    uint16_t *h_xk = malloc(x_elems * 2), *h_xv = malloc(x_elems * 2);
    fill_normal_f16(h_xk, x_elems);
    fill_normal_f16(h_xv, x_elems);
    void *d_xk = dev_upload(h_xk, x_elems * 2), *d_xv = dev_upload(h_xv, x_elems * 2);
    const size_t pool_pages = (size_t)bs * nhk * n_pages, pool_bytes = pool_pages * ps * M;
    int32_t *h_ids = malloc(pool_pages * sizeof(int32_t));
    for (size_t i = 0; i < pool_pages; ++i) h_ids[i] = (int32_t)i;      /* allocation order, as PagedPQCache hands pages out */
    void *d_ids = dev_upload(h_ids, pool_pages * sizeof(int32_t));
    void **d_kpool = malloc(layers * sizeof(void *)), **d_vpool = malloc(layers * sizeof(void *));
    void **d_kres = malloc(layers * sizeof(void *)), **d_vres = malloc(layers * sizeof(void *));
    const size_t res_elems = (size_t)bs * nhk * cap * d;
    uint16_t *h_kres = malloc(res_elems * 2), *h_vres = malloc(res_elems * 2);
    fill_normal_f16(h_kres, res_elems);
    fill_normal_f16(h_vres, res_elems);
    million_encode_desc ed;
    memset(&ed, 0, sizeof(ed));
    ed.struct_size = sizeof(ed);
    ed.bs = bs; ed.nh_k = nhk; ed.n = T; ed.d = d; ed.M = M; ed.C = C;
    ed.x_stride_b = (int64_t)nhk * T * d; ed.x_stride_h = (int64_t)T * d; ed.x_stride_n = d;
    ed.x_row_mod = T;
    ed.page_size = ps; ed.n_pages_cap = n_pages;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, stream));
    for (int l = 0; l < layers; ++l) {
        d_kpool[l] = dev_alloc(pool_bytes);
        d_vpool[l] = dev_alloc(pool_bytes);
        d_kres[l] = dev_upload(h_kres, res_elems * 2);
        d_vres[l] = dev_upload(h_vres, res_elems * 2);
        ed.x_row_start = (int32_t)(((int64_t)l * 997) % T);
        ed.dst_layout = MILLION_CODES_KPAGES; ed.cents_prepared = d_kprep;
        MIL_OK(million_pq_encode(&ed, d_xk, d_kc, d_kpool[l], (const int32_t *)d_ids, stream));
        ed.dst_layout = MILLION_CODES_VPAGES; ed.cents_prepared = d_vprep;
        MIL_OK(million_pq_encode(&ed, d_xv, d_vc, d_vpool[l], (const int32_t *)d_ids, stream));
    }
    HIP_OK(hipEventRecord(e1, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (random_codes) {
        uint8_t *h_rand = malloc(2 * pool_bytes + 4096 * (size_t)layers);
        uint32_t *w = (uint32_t *)h_rand;
        for (size_t i = 0; i < (2 * pool_bytes + 4096 * (size_t)layers) / 4; ++i) w[i] = rnd32();
        for (int l = 0; l < layers; ++l) {
            HIP_OK(hipMemcpy(d_kpool[l], h_rand + 4096 * (size_t)l, pool_bytes, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(d_vpool[l], h_rand + pool_bytes + 4096 * (size_t)l, pool_bytes, hipMemcpyHostToDevice));
        }
        free(h_rand);
        printf("pages overwritten with uniformly random code bytes (--random-codes): the code check below is skipped\n");
    }
    float enc_ms = 0.f;
    HIP_OK(hipEventElapsedTime(&enc_ms, e0, e1));
    printf("encode: %d layers x (K + V) x %d rows x %d kv heads into pages: %.2f ms (allocations and window uploads included)\n",
           layers, T, bs * nhk, enc_ms);

    /* 3. decode launches */
    const size_t q_elems = (size_t)bs * nh * d, new_elems = (size_t)bs * nhk * d;
    uint16_t *h_q = malloc(q_elems * 2), *h_kn = malloc(new_elems * 2), *h_vn = malloc(new_elems * 2);
    fill_normal_f16(h_q, q_elems);
    fill_normal_f16(h_kn, new_elems);
    fill_normal_f16(h_vn, new_elems);
    void *d_q = dev_upload(h_q, q_elems * 2), *d_kn = dev_upload(h_kn, new_elems * 2), *d_vn = dev_upload(h_vn, new_elems * 2);
    void **d_out = malloc(layers * sizeof(void *));
    for (int l = 0; l < layers; ++l) d_out[l] = dev_alloc(q_elems * 2);
    million_attn_desc ad;
    memset(&ad, 0, sizeof(ad));
    ad.struct_size = sizeof(ad);
    ad.bs = bs; ad.nh = nh; ad.nh_k = nhk; ad.d = d; ad.M = M; ad.C = C;
    ad.n_tokens = T; ad.r = r; ad.resid_start = 0; ad.resid_cap = cap;
    ad.resid_stride_b = (int64_t)nhk * cap * d; ad.resid_stride_h = (int64_t)cap * d;
    ad.k_layout = MILLION_KV_PAGED; ad.v_layout = MILLION_KV_PAGED; ad.page_size = ps; ad.n_pages_cap = n_pages;
    ad.k_pool_pages = (int32_t)pool_pages; ad.v_pool_pages = (int32_t)pool_pages;
    const size_t ws_bytes = million_attn_workspace_bytes(&ad);
    void *d_ws = dev_alloc(ws_bytes);
    MIL_OK(million_workspace_init(d_ws, ws_bytes, stream));
    printf("decode: kernel kind %d (1 = streaming MFMA kernel), workspace %.1f KiB\n", million_attn_kernel_kind(&ad), ws_bytes / 1024.0);
#define LAUNCH(l)                                                                                                        \
    MIL_OK(million_pq_decode_attn_append(&ad, d_q, d_kn, d_vn, d_kpool[l], d_vpool[l], d_ids, d_ids, d_kprep, d_vprep,   \
                                         d_kres[l], d_vres[l], d_out[l], d_ws, ws_bytes, stream))
    for (int i = 0; i < 2 * layers; ++i) LAUNCH(i % layers);      /* warm-up; also parks row r of every window */
    HIP_OK(hipStreamSynchronize(stream));

    const double alg = 2.0 * bs * nhk * T * M + 2.0 * bs * nhk * (r + 1) * d * 2 + 2.0 * M * C * dm * 2 + 2.0 * bs * nh * d * 2;
    float *us = malloc(reps * sizeof(float));
    /* (a) ONE captured hipGraph of the launches, replayed: the launch period tools/ab_bench.py and bench.py report */
    hipGraph_t graph;
    hipGraphExec_t gexec;
    HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    for (int i = 0; i < launches; ++i) LAUNCH(i % layers);
    HIP_OK(hipStreamEndCapture(stream, &graph));
    HIP_OK(hipGraphInstantiate(&gexec, graph, NULL, NULL, 0));
    HIP_OK(hipGraphLaunch(gexec, stream));                         /* warm replay; the timed ones follow it directly */
    HIP_OK(hipStreamSynchronize(stream));
    for (int k = 0; k < reps; ++k) {
        float ms = 0.f;
        HIP_OK(hipEventRecord(e0, stream));
        HIP_OK(hipGraphLaunch(gexec, stream));
        HIP_OK(hipEventRecord(e1, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        us[k] = ms * 1e3f / launches;
    }
    qsort(us, reps, sizeof(float), cmp_float);
    const float g_best = us[0], g_med = us[reps / 2];
    /* (b) the same launches enqueued eagerly (host-bound below ~3 us per launch; here the kernel is longer than the enqueue) */
    for (int k = 0; k < reps; ++k) {
        float ms = 0.f;
        HIP_OK(hipEventRecord(e0, stream));
        for (int i = 0; i < launches; ++i) LAUNCH(i % layers);
        HIP_OK(hipEventRecord(e1, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        us[k] = ms * 1e3f / launches;
    }
    qsort(us, reps, sizeof(float), cmp_float);
    const float e_best = us[0], e_med = us[reps / 2];
    printf("decode launch (million_pq_decode_attn_append), %.2f MB algorithmic per launch:\n", alg / 1e6);
    printf("  hipGraph replay : best %.2f us, median %.2f us per launch  = %.0f GB/s = %.1f %% of 8 TB/s (best)\n", g_best, g_med,
           alg / g_best / 1e3, alg / g_best / 8e6 * 100.0);
    printf("  eager enqueue   : best %.2f us, median %.2f us per launch\n", e_best, e_med);

    /* 4. host check (layer 0, request 0, kv head 0): sampled codes bit-exact; attention of its G query heads */
    const size_t head_pages = (size_t)n_pages, head_bytes = head_pages * ps * M;
    uint8_t *h_kp = malloc(head_bytes), *h_vp = malloc(head_bytes);
    HIP_OK(hipMemcpy(h_kp, d_kpool[0], head_bytes, hipMemcpyDeviceToHost));      /* pages 0 .. n_pages-1 = (b 0, kv head 0) */
    HIP_OK(hipMemcpy(h_vp, d_vpool[0], head_bytes, hipMemcpyDeviceToHost));
    int bad_codes = 0, checked = 0;
    for (int s = 0; s < (random_codes ? 0 : 64); ++s) {
        const int t = (int)(rnd32() % (uint32_t)T);
        for (int side = 0; side < 2; ++side) {
            const uint16_t *x = (side ? h_xv : h_xk) + (size_t)t * d, *cb = side ? h_vc : h_kc;      /* layer 0: ring start 0 */
            for (int m = 0; m < M; ++m) {
                int best = 0;
                float bd = INFINITY;
                for (int c = 0; c < C; ++c) {
                    float acc = 0.f;
                    for (int k = 0; k < dm; ++k) {
                        const volatile float e = h2f(x[m * dm + k]) - h2f(cb[((size_t)m * C + c) * dm + k]);      /* no contraction */
                        const volatile float sq = e * e;
                        acc += sq;
                    }
                    if (acc < bd) { bd = acc; best = c; }
                }
                const int page = t / ps, off = t % ps;
                const int got = side ? h_vp[((size_t)page * M + m) * ps + off] : h_kp[((size_t)page * ps + off) * M + m];
                bad_codes += got != best;
                ++checked;
            }
        }
    }
    uint16_t *h_out = malloc(q_elems * 2);
    HIP_OK(hipMemcpy(h_out, d_out[0], q_elems * 2, hipMemcpyDeviceToHost));
    double worst_rel = 0.0;
    const int n_rows = T + r + 1;
    double *sc = malloc((size_t)n_rows * sizeof(double));
    for (int g = 0; g < G; ++g) {
        const uint16_t *qh = h_q + (size_t)g * d;      /* request 0, query head g (kv head 0) */
        double mx = -1e300;
        for (int j = 0; j < n_rows; ++j) {
            double s = 0.0;
            if (j < T) {
                const uint8_t *row = h_kp + ((size_t)(j / ps) * ps + j % ps) * M;
                for (int m = 0; m < M; ++m)
                    for (int k = 0; k < dm; ++k) s += (double)h2f(qh[m * dm + k]) * h2f(h_kc[((size_t)m * C + row[m]) * dm + k]);
            } else {
                const uint16_t *row = (j - T < r) ? h_kres + (size_t)(j - T) * d : h_kn;      /* window rows, then the appended row */
                for (int k = 0; k < d; ++k) s += (double)h2f(qh[k]) * h2f(row[k]);
            }
            sc[j] = s / sqrt((double)d);
            if (sc[j] > mx) mx = sc[j];
        }
        double den = 0.0, o[128];
        memset(o, 0, sizeof(o));
        for (int j = 0; j < n_rows; ++j) {
            const double p = exp(sc[j] - mx);
            den += p;
            if (j < T) {
                const uint8_t *pg = h_vp + (size_t)(j / ps) * M * ps;
                for (int m = 0; m < M; ++m)
                    for (int k = 0; k < dm; ++k) o[m * dm + k] += p * h2f(h_vc[((size_t)m * C + pg[(size_t)m * ps + j % ps]) * dm + k]);
            } else {
                const uint16_t *row = (j - T < r) ? h_vres + (size_t)(j - T) * d : h_vn;
                for (int k = 0; k < d; ++k) o[k] += p * h2f(row[k]);
            }
        }
        double num = 0.0, ref = 0.0;
        for (int k = 0; k < d; ++k) {
            const double want = o[k] / den, got = h2f(h_out[(size_t)g * d + k]);
            num += (got - want) * (got - want);
            ref += want * want;
        }
        const double rel = sqrt(num / ref);
        if (rel > worst_rel) worst_rel = rel;
    }
    const int faults = million_debug_tail_faults();
    printf("host check: %d of %d sampled codes differ from the direct fp32 argmin; attention of %d heads: worst rel-L2 %.2e (bar 1e-3); tail faults %d\n",
           bad_codes, checked, G, worst_rel, faults);
    const int ok = bad_codes == 0 && worst_rel < 1e-3 && faults == 0;
    printf("cabi_bench: %s  graph_us_per_launch=%.2f\n", ok ? "PASS" : "FAIL", g_best);
    HIP_OK(hipGraphExecDestroy(gexec));
    HIP_OK(hipGraphDestroy(graph));
    return ok ? 0 : 4;
}
