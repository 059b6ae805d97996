// million_api.hip — C-ABI entry points of libmillion_hip.so (see include/million_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace million {

static thread_local char g_err[512] = "";
// Diagnostic knobs (million_set_force_generic, million_debug_set_stamp_buffer): process-wide, plain words, meant to be
// set from one thread before the calls they affect (A/B runs and the stamp profiler); not part of the product path.
static int g_force_generic = 0;
static unsigned long long *g_dbg = nullptr;

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-device state: the only mutable globals of the library besides the diagnostic knobs below ----
constexpr int kMaxDevices = 64;
static std::mutex g_dev_mutex;
static int g_dev_cus[kMaxDevices];            // 0 = not queried yet
static unsigned g_dev_once[kMaxDevices];      // bit f: family f has been set up on this device

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}
int device_cus() {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    if (g_dev_cus[dev] == 0) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        g_dev_cus[dev] = n > 0 ? n : 256;
    }
    return g_dev_cus[dev];
}
bool device_once(int family) {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    const unsigned bit = 1u << family;
    if (g_dev_once[dev] & bit) return false;
    g_dev_once[dev] |= bit;
    return true;
}
static int num_cus() { return device_cus(); }

// ---- prepare_cents: (M, C, dm) -> fp16 row image [m][c][dm], fp16 col image [c][m][dm], fp32 row image ----
__global__ void prepare_cents_kernel(const f16 *__restrict__ src, f16 *__restrict__ dst, int M, int C, int dm) {
    const int n = M * C * dm;
    float *dst32 = (float *)(dst + 2 * (long long)n);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const f16 v = src[i];
        const int k = i % dm, c = (i / dm) % C, m = i / (dm * C);
        dst[i] = v;
        dst[n + ((long long)c * M + m) * dm + k] = v;
        dst32[i] = (float)v;
    }
}

// ---- residual append: one workgroup; lane-contiguous 2-byte copies of 2*bs*nh_k rows of d halfs ----
__global__ void residual_append_kernel(const f16 *__restrict__ k_new, const f16 *__restrict__ v_new,
                                       f16 *__restrict__ k_res, f16 *__restrict__ v_res,
                                       int bs, int nh_k, int d, int cap, long long sb, long long sh,
                                       int r, int rstart, int *dev_lengths) {
    const int n = bs * nh_k * d;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int k = i % d, hk = (i / d) % nh_k, b = i / (d * nh_k);
        int rr = r, rs = rstart;
        if (dev_lengths) { rr = dev_lengths[b * 4 + 1]; rs = dev_lengths[b * 4 + 2]; }
        const int row = (rs + rr) % cap;
        const long long o = b * sb + hk * sh + (long long)row * d + k;
        k_res[o] = k_new[i];
        v_res[o] = v_new[i];
    }
    if (dev_lengths) {
        __syncthreads();
        for (int b = threadIdx.x; b < bs; b += blockDim.x) dev_lengths[b * 4 + 1] += 1;
    }
}

__global__ void lengths_advance_kernel(int *dl, int bs, int n, int cap) {
    for (int b = threadIdx.x; b < bs; b += blockDim.x) {
        dl[b * 4 + 0] += n;
        dl[b * 4 + 1] -= n;
        dl[b * 4 + 2] = (dl[b * 4 + 2] + n) % cap;
    }
}

// ---- row-major V codes (bs, nh_k, T, M) -> dense transposed pages (bh * n_pages + page, M, 64) ----
// The reference's production call passes V row-major (Interface.template.cu:30); the MFMA kernel wants
// 16 consecutive tokens of one subspace in one 16-byte load.  One workgroup per page: 64 x M tile
// through LDS.  Tokens >= T of the last page are zero-filled.
__global__ __launch_bounds__(256) void codes_transpose_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                              int nh_k, int T, int M, long long sb, long long sh,
                                                              int n_pages) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * (64 + 16)];      // M <= 64, rows padded
    const int page = blockIdx.x, bh = blockIdx.y;
    const int b = bh / nh_k, hk = bh % nh_k;
    const uint8_t *s0 = src + b * sb + hk * sh;
    const int chunks = M / 16;                                                 // 16-byte chunks per row
    for (int i = threadIdx.x; i < 64 * chunks; i += 256) {
        const int tok = i / chunks, ch = i % chunks;
        const int t = page * 64 + tok;
        uint4 v = {0, 0, 0, 0};
        if (t < T) v = *(const uint4 *)(s0 + (long long)t * M + ch * 16);
        *(uint4 *)(tile + tok * 80 + ch * 16) = v;
    }
    __syncthreads();
    uint8_t *d0 = dst + ((long long)bh * n_pages + page) * M * 64;
    for (int i = threadIdx.x; i < M * 4; i += 256) {
        const int m = i >> 2, tq = i & 3;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) x |= (uint32_t)tile[(tq * 16 + j * 4 + k) * 80 + m] << (8 * k);
            w[j] = x;
        }
        *(uint4 *)(d0 + m * 64 + tq * 16) = uint4{w[0], w[1], w[2], w[3]};
    }
}

// Workspace layout: [records: P x 128 B | second-level tickets: P x int] [flags + census: P x 2 x kFlagWords x 4 B] [partials], with
// P = max(kHeadPairs, bs * nh_k): the head does not move with the shape, so calls of different shapes may share one
// (once-zeroed) workspace - every call leaves the head at rest and nobody ever reads another call's partials.  (Round 2
// sized the counter block by bs * nh_k: a larger shape found an earlier call's partials where it expected zeros.)  A
// shape with more than kHeadPairs (b, kv head) pairs needs a workspace of its own (million_hip.h).
constexpr size_t kHeadPairs = 2048;
static size_t head_pairs(int bs, int nh_k) { const size_t n = (size_t)bs * nh_k; return n > kHeadPairs ? n : kHeadPairs; }
static size_t slot_floats_for(int G, int d) { return (size_t)(G * d + 2 * G + 31) / 32 * 32; }      // whole 128-byte lines
static size_t attn_cnt_bytes(int bs, int nh_k) {
    const size_t cnt = head_pairs(bs, nh_k) * (kRecWords * sizeof(unsigned) + sizeof(int));
    return (cnt + kCntBytes - 1) / kCntBytes * kCntBytes;
}
static size_t attn_flag_bytes(int bs, int nh_k) {
    const size_t f = head_pairs(bs, nh_k) * 2 * kFlagWords * sizeof(unsigned);      // flags + XCD census line
    return (f + kCntBytes - 1) / kCntBytes * kCntBytes;
}
// floats of one (b, kv head, split) slot as the workspace is SIZED: the lean kernel may run 5 .. 16 heads as ceil(G / 4) virtual kv
// heads of ceil(G / parts) heads (attn_mfma.hip mfma_virtual) - that many slots of the smaller group, each rounded up by itself
static size_t slot_floats_sized(int G, int d) {
    size_t n = slot_floats_for(G, d);
    if (G > 4 && G <= 16) {
        const int P = (G + 3) / 4;
        const size_t v = (size_t)P * slot_floats_for((G + P - 1) / P, d);
        if (v > n) n = v;
    }
    return n;
}
static size_t attn_partial_bytes(int bs, int nh_k, int G, int d) {
    const size_t b = attn_cnt_bytes(bs, nh_k) + attn_flag_bytes(bs, nh_k) +
                     (size_t)bs * nh_k * (kMaxSplits + 1) * slot_floats_sized(G, d) * sizeof(float);
    return (b + 255) / 256 * 256;
}

static int fill_attn_params(const million_attn_desc *desc, AttnParams &p) {
    if (!desc || desc->struct_size != sizeof(million_attn_desc)) { set_error("attn: bad desc / struct_size"); return MILLION_ERR_ARG; }
    memset(&p, 0, sizeof(p));
    p.bs = desc->bs; p.nh = desc->nh; p.nh_k = desc->nh_k; p.d = desc->d; p.M = desc->M; p.C = desc->C;
    if (p.bs <= 0 || p.nh <= 0 || p.nh_k <= 0 || p.nh % p.nh_k) { set_error("attn: bs=%d nh=%d nh_k=%d", p.bs, p.nh, p.nh_k); return MILLION_ERR_SHAPE; }
    p.Gt = p.nh / p.nh_k;                     // any group size: the MFMA kernels serve up to kMaxGMfma = 16 query heads per kv head
    p.G = p.Gt < kMaxGMfma ? p.Gt : kMaxGMfma;      // in one launch, the tile / scalar kernels kMaxG = 8; bigger groups run as several
                                              // launches (attn_impl).  G = heads of one launch, sized here for the largest
    p.g0 = 0;
    p.nhk_real = 1 << 28; p.nhk_mul = 0; p.hparts_m1 = 0; p.G_all = 0;      // no query-head parts (attn_mfma.hip mfma_virtual sets them)
    if (p.M <= 0 || p.d <= 0 || p.d % p.M || p.M % 4) { set_error("attn: d=%d M=%d", p.d, p.M); return MILLION_ERR_SHAPE; }
    p.dm = p.d / p.M;
    if (p.C < 2 || p.C > 256) { set_error("attn: C=%d (uint8 codes)", p.C); return MILLION_ERR_SHAPE; }
    p.T = desc->n_tokens; p.r = desc->r; p.rstart = desc->resid_start; p.rcap = desc->resid_cap;
    if (p.T < 0 || p.rcap <= 0 || p.r < 0 || p.r > p.rcap || p.rstart < 0 || p.rstart >= p.rcap) {
        set_error("attn: T=%d r=%d resid_start=%d resid_cap=%d", p.T, p.r, p.rstart, p.rcap);
        return MILLION_ERR_ARG;
    }
    p.res_sb = desc->resid_stride_b; p.res_sh = desc->resid_stride_h;
    p.k_sb = desc->k_stride_b; p.k_sh = desc->k_stride_h; p.v_sb = desc->v_stride_b; p.v_sh = desc->v_stride_h;
    p.k_paged = desc->k_layout == MILLION_KV_PAGED;
    p.v_paged = desc->v_layout == MILLION_KV_PAGED;
    if ((desc->k_layout != MILLION_KV_PAGED && desc->k_layout != MILLION_KV_ROWMAJOR) ||
        (desc->v_layout != MILLION_KV_PAGED && desc->v_layout != MILLION_KV_ROWMAJOR)) { set_error("attn: k_layout=%d v_layout=%d", desc->k_layout, desc->v_layout); return MILLION_ERR_ARG; }
    p.page_size = desc->page_size; p.n_pages_cap = desc->n_pages_cap; p.ids64 = desc->page_ids_i64;
    p.v_identity = desc->v_pages_dense != 0;
    if (p.v_identity && !p.v_paged) { set_error("attn: v_pages_dense needs v_layout = PAGED"); return MILLION_ERR_ARG; }
    p.ps_shift = p.page_size == 32 ? 5 : p.page_size == 64 ? 6 : 7;
    if (p.k_paged || p.v_paged) {
        if (p.page_size != 32 && p.page_size != 64 && p.page_size != 128) { set_error("attn: page_size=%d (32, 64, 128)", p.page_size); return MILLION_ERR_SHAPE; }
        if ((long long)p.n_pages_cap * p.page_size < p.T) { set_error("attn: n_pages_cap*page_size < n_tokens"); return MILLION_ERR_ARG; }
        // the kernels index the table with 32-bit arithmetic (int64 ids: two dwords per entry)
        if (p.n_pages_cap < 0 || (long long)p.bs * p.nh_k * p.n_pages_cap > (p.ids64 ? 0x3fffffffLL : 0x7fffffffLL)) { set_error("attn: page table of %lld entries", (long long)p.bs * p.nh_k * p.n_pages_cap); return MILLION_ERR_ARG; }
    }
    p.dev_lengths = desc->dev_lengths;
    p.scale_log2e = 1.4426950408889634f / sqrtf((float)p.d);
    p.slot_floats = (int)slot_floats_for(p.G, p.d);
    return MILLION_OK;
}

// Split policy: about one workgroup per CU over all (b, hk); a split is a multiple of 64 tokens and at
// least 256 tokens long (the reference picks Ns from the binding name, pq_utils.py:8-22; here the
// split count is internal and the Ns of the name is ignored).
static void choose_splits(AttnParams &p, int min_tokens) {
    const int bh = p.bs * p.nh_k;
    int ns = (num_cus() + bh - 1) / bh;
    if (ns > kMaxSplits) ns = kMaxSplits;
    int by_len = (p.T + min_tokens - 1) / min_tokens;
    if (by_len < 1) by_len = 1;
    if (ns > by_len) ns = by_len;
    if (ns < 1) ns = 1;
    int len = (p.T + ns - 1) / ns;
    len = (len + 63) / 64 * 64;
    if (len < 64) len = 64;
    ns = p.T > 0 ? (p.T + len - 1) / len : 1;
    p.nsplit = ns;
    p.split_len = len;
}

}  // namespace million

using namespace million;

extern "C" {

int million_version(void) { return MILLION_HIP_VERSION; }
const char *million_last_error(void) { return g_err; }
void million_set_force_generic(int on) {
    g_force_generic = (on == 1);
    million::set_mfma_policy(on == 2 ? 1 : on == 4 ? 2 : on == 8 ? 4 : on == 16 ? 8 : 0);
    million::set_prefill_policy(on == 64 ? 1 : on == 128 ? 2 : 0);      // (128: a development build's experimental form; the product build ignores it)
}
void million_debug_set_stamp_buffer(void *buf) { g_dbg = (unsigned long long *)buf; }

#ifdef MILLION_DEBUG_CHECK_IDS
// device counter of out-of-range page ids (one per process; a diagnostic build is not for concurrent devices)
static int *g_bad_ids = nullptr;
static int *bad_ids_counter() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (hipMalloc((void **)&g_bad_ids, sizeof(int)) != hipSuccess) { g_bad_ids = nullptr; return; }
        (void)hipMemset(g_bad_ids, 0, sizeof(int));
    });
    return g_bad_ids;
}
int million_debug_bad_page_ids(void) {
    int *c = bad_ids_counter();
    if (!c) return -1;
    int n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&n, c, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipMemset(c, 0, sizeof(int));
    return n;
}
#else
int million_debug_bad_page_ids(void) { return -1; }
#endif
int million_debug_tail_faults(void) { return million::read_tail_faults(); }
int million_debug_rows_reduce(const float *in64, float *out_max64, float *out_sum64, million_stream_t stream) {
    return launch_rows_reduce_check(in64, out_max64, out_sum64, (hipStream_t)stream);
}

size_t million_prepared_cents_bytes(int M, int C, int d_m) {
    return (size_t)M * C * d_m * (2 * sizeof(f16) + sizeof(float));      // two fp16 images + the fp32 image
}

int million_prepare_cents(const void *cents, int M, int C, int d_m, void *prepared, million_stream_t stream) {
    if (!cents || !prepared) { set_error("prepare_cents: null pointer"); return MILLION_ERR_ARG; }
    if (M <= 0 || C <= 0 || C > 256 || d_m <= 0) { set_error("prepare_cents: M=%d C=%d d_m=%d", M, C, d_m); return MILLION_ERR_SHAPE; }
    const int n = M * C * d_m;
    hipLaunchKernelGGL(prepare_cents_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const f16 *)cents, (f16 *)prepared, M, C, d_m);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("prepare_cents launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

static int fill_enc_params(const char *who, const million_encode_desc *desc, const void *x, const void *cents, void *dst,
                           const int32_t *page_ids, EncParams &p) {
    if (!desc || desc->struct_size != sizeof(million_encode_desc)) { set_error("%s: bad desc / struct_size", who); return MILLION_ERR_ARG; }
    if (!x || !cents || !dst) { set_error("%s: null pointer", who); return MILLION_ERR_ARG; }
    memset(&p, 0, sizeof(p));
    p.x = (const f16 *)x; p.cents = (const f16 *)cents; p.dst = (uint8_t *)dst; p.page_ids = page_ids;
    if (desc->cents_prepared && desc->C <= 256)
        p.cents32 = (const float *)((const f16 *)desc->cents_prepared + 2 * (size_t)desc->M * desc->C * (desc->d / (desc->M > 0 ? desc->M : 1)));
    p.bs = desc->bs; p.nh_k = desc->nh_k; p.n = desc->n; p.d = desc->d; p.M = desc->M; p.C = desc->C;
    if (p.bs <= 0 || p.nh_k <= 0 || p.n < 0) { set_error("%s: bs=%d nh_k=%d n=%d", who, p.bs, p.nh_k, p.n); return MILLION_ERR_SHAPE; }
    if (p.M <= 0 || p.d <= 0 || p.d % p.M) { set_error("%s: d=%d M=%d", who, p.d, p.M); return MILLION_ERR_SHAPE; }
    if (p.C < 1 || p.C > 65536) { set_error("%s: C=%d (uint8 codes up to 256, uint16 codes up to 65536)", who, p.C); return MILLION_ERR_SHAPE; }
    p.dm = p.d / p.M;
    p.xsb = desc->x_stride_b; p.xsh = desc->x_stride_h; p.xsn = desc->x_stride_n;
    p.xrow_start = desc->x_row_start; p.xrow_mod = desc->x_row_mod;
    p.layout = desc->dst_layout; p.tok0 = desc->dst_token_start;
    p.dsb = desc->dst_stride_b; p.dsh = desc->dst_stride_h;
    p.page_size = desc->page_size; p.n_pages_cap = desc->n_pages_cap;
    p.dev_lengths = desc->dev_lengths;
    if (p.layout != MILLION_CODES_ROWMAJOR) {
        if (p.layout != MILLION_CODES_KPAGES && p.layout != MILLION_CODES_VPAGES) { set_error("%s: dst_layout=%d", who, p.layout); return MILLION_ERR_ARG; }
        if (!page_ids || p.page_size <= 0) { set_error("%s: paged destination needs page_ids and page_size", who); return MILLION_ERR_ARG; }
        if (!p.dev_lengths && (long long)p.n_pages_cap * p.page_size < (long long)p.tok0 + p.n) { set_error("%s: page table too short", who); return MILLION_ERR_ARG; }
    }
    if (p.tok0 < 0) { set_error("%s: dst_token_start=%d", who, p.tok0); return MILLION_ERR_ARG; }
    return MILLION_OK;
}

int million_pq_encode(const million_encode_desc *desc, const void *x, const void *cents, void *dst,
                      const int32_t *page_ids, million_stream_t stream) {
    EncParams p;
    const int rc = fill_enc_params("encode", desc, x, cents, dst, page_ids, p);
    if (rc != MILLION_OK) return rc;
    return launch_encode(p, (hipStream_t)stream);
}

int million_pq_flush(const million_encode_desc *desc, const void *k_rows, const void *v_rows, const void *k_cents,
                     const void *v_cents, void *k_pool, void *v_pool, const int32_t *page_ids, int32_t *dev_lengths,
                     int resid_cap, int min_r, million_stream_t stream) {
    return million_pq_flush_layers(desc, k_rows, v_rows, k_cents, v_cents, k_pool, v_pool, page_ids, dev_lengths, resid_cap, min_r,
                                   1, 0, 0, 0, 1, stream);
}

int million_pq_flush_layers(const million_encode_desc *desc, const void *k_rows, const void *v_rows, const void *k_cents,
                            const void *v_cents, void *k_pool, void *v_pool, const int32_t *page_ids, int32_t *dev_lengths,
                            int resid_cap, int min_r, int n_layers, int64_t rows_layer_stride, int64_t ids_layer_stride,
                            int64_t lengths_layer_stride, int advance, million_stream_t stream) {
    if (n_layers < 1 || rows_layer_stride < 0 || ids_layer_stride < 0 || lengths_layer_stride < 0 || (rows_layer_stride & 7)) {
        set_error("flush: n_layers=%d, layer strides %lld / %lld / %lld (rows: multiple of 8 elements)", n_layers,
                  (long long)rows_layer_stride, (long long)ids_layer_stride, (long long)lengths_layer_stride);
        return MILLION_ERR_ARG;
    }
    if (n_layers > 1 && desc && desc->dev_lengths && lengths_layer_stride < 4LL * desc->bs) { set_error("flush: lengths_layer_stride < 4 bs"); return MILLION_ERR_ARG; }
    EncParams k, v;
    int rc = fill_enc_params("flush", desc, k_rows, k_cents, k_pool, page_ids, k);
    if (rc != MILLION_OK) return rc;
    rc = fill_enc_params("flush", desc, v_rows, v_cents, v_pool, page_ids, v);
    if (rc != MILLION_OK) return rc;
    if (k.C > 256) { set_error("flush: uint8 codes only (C=%d)", k.C); return MILLION_ERR_SHAPE; }
    if (desc->dst_layout != MILLION_CODES_KPAGES) { set_error("flush: dst_layout must be MILLION_CODES_KPAGES (the V side is written as VPAGES)"); return MILLION_ERR_ARG; }
    if (resid_cap <= 0) { set_error("flush: resid_cap=%d", resid_cap); return MILLION_ERR_ARG; }
    if ((const int32_t *)dev_lengths != desc->dev_lengths) { set_error("flush: dev_lengths must equal desc->dev_lengths (or both null)"); return MILLION_ERR_ARG; }
    v.layout = MILLION_CODES_VPAGES;
    if (min_r < 0 || min_r > resid_cap) { set_error("flush: min_r=%d outside [0, resid_cap]", min_r); return MILLION_ERR_ARG; }
    const FlushLayers ly = {n_layers, (long long)rows_layer_stride, (long long)ids_layer_stride, (long long)lengths_layer_stride, advance ? 1 : 0};
    return launch_flush(k, v, dev_lengths, resid_cap, min_r, ly, (hipStream_t)stream);
}

int million_pq_decode(const void *codes, const void *cents, void *out, int64_t n_rows, int d, int M, int C,
                      million_stream_t stream) {
    if (n_rows < 0 || M <= 0 || d <= 0 || d % M) { set_error("decode: n_rows=%lld d=%d M=%d", (long long)n_rows, d, M); return MILLION_ERR_SHAPE; }
    if (C < 1 || C > 65536) { set_error("decode: C=%d (uint8 codes up to 256, uint16 codes up to 65536)", C); return MILLION_ERR_SHAPE; }
    if (n_rows == 0) return MILLION_OK;
    if (!codes || !cents || !out) { set_error("decode: null pointer"); return MILLION_ERR_ARG; }
    return launch_decode(codes, (const f16 *)cents, (f16 *)out, n_rows, M, C, d / M, (hipStream_t)stream);
}

int million_transpose_v_codes(const void *v_codes, void *v_pages, int bs, int nh_k, int n_tokens, int M,
                              int64_t v_stride_b, int64_t v_stride_h, million_stream_t stream) {
    if (!v_codes || !v_pages) { set_error("transpose_v_codes: null pointer"); return MILLION_ERR_ARG; }
    if (bs <= 0 || nh_k <= 0 || n_tokens <= 0 || M <= 0 || M > 64 || M % 16) { set_error("transpose_v_codes: bs=%d nh_k=%d T=%d M=%d", bs, nh_k, n_tokens, M); return MILLION_ERR_SHAPE; }
    if ((((uintptr_t)v_codes | (uintptr_t)v_pages) & 15) || ((v_stride_b | v_stride_h) & 15)) { set_error("transpose_v_codes: 16-byte alignment"); return MILLION_ERR_ALIGN; }
    const int n_pages = (n_tokens + 63) / 64;
    hipLaunchKernelGGL(codes_transpose_kernel, dim3(n_pages, bs * nh_k), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)v_codes, (uint8_t *)v_pages, nh_k, n_tokens, M, (long long)v_stride_b, (long long)v_stride_h, n_pages);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("codes_transpose launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

size_t million_attn_workspace_bytes(const million_attn_desc *desc) {
    if (!desc || desc->nh_k <= 0 || desc->nh % desc->nh_k) return 0;
    const int Gt = desc->nh / desc->nh_k, G = Gt < kMaxGMfma ? Gt : kMaxGMfma;
    size_t bytes = attn_partial_bytes(desc->bs, desc->nh_k, G, desc->d);
    // row-major V on the MFMA shapes: room for the transposed copy of the V codes (64-token pages)
    if (desc->v_layout == MILLION_KV_ROWMAJOR && desc->k_layout == MILLION_KV_ROWMAJOR && desc->n_tokens > 0 && desc->M > 0)
        bytes += (size_t)desc->bs * desc->nh_k * ((desc->n_tokens + 63) / 64) * desc->M * 64;
    return bytes;
}

int million_workspace_init(void *workspace, size_t bytes, million_stream_t stream) {
    if (!workspace) { set_error("workspace_init: null"); return MILLION_ERR_ARG; }
    const hipError_t e = hipMemsetAsync(workspace, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("workspace_init: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

// p restricted to n query heads per kv head from g0 on (one launch of a bigger group)
static AttnParams with_heads(const AttnParams &p, int g0, int n) {
    AttnParams q = p;
    q.g0 = g0;
    q.G = n;
    q.slot_floats = (int)slot_floats_for(n, p.d);
    return q;
}
// the launches after the first of a call: a fused append has happened (the row is in the window; device-resident
// lengths were advanced by the first launch)
static void after_first_launch(AttnParams &p) {
    if (p.k_new) {
        p.k_new = p.v_new = nullptr;
        if (!p.dev_lengths) p.r += 1;
    }
}

int million_attn_kernel_kind(const million_attn_desc *desc) {
    AttnParams p;
    if (fill_attn_params(desc, p) != MILLION_OK) return -1;
    if (g_force_generic) return 0;
    if (attn_mfma_supported(p) && attn_mfma_handles(p)) return attn_mfma_streams(p) ? 1 : 5;
    if (attn_mfma_shape_ok(p) && !p.v_paged && !p.k_paged && p.T > 0) {               // transpose + MFMA kernel
        AttnParams pt = p;
        pt.v_paged = 1; pt.v_identity = 1; pt.page_size = 64; pt.ps_shift = 6;
        if (attn_mfma_handles(pt)) return attn_mfma_streams(pt) ? 2 : 5;
    }
    const AttnParams p8 = with_heads(p, 0, p.Gt < kMaxGMfma ? p.Gt : kMaxGMfma);      // tile kernel: up to kMaxGMfma heads per launch
    if (attn_tile_supported(p8)) return 3;                                            // tile kernel
    if (attn_tile_shape_ok(p8) && !p.v_paged && !p.k_paged) return 4;                 // transpose + tile kernel
    return 0;
}

// one query-head group (p.G heads per kv head from p.g0 on): 1. the streaming / grouped MFMA kernels (d = 128,
// M in {64, 32} with up to kMaxGMfma heads, M = 16 with up to 4); 2. the tile kernel (every other shape of the binding surface); 3. the scalar
// kernel (both up to kMaxG heads: a bigger group the MFMA kernels hand back is split here)
static int launch_group(const AttnParams &p, hipStream_t stream) {
    if (!g_force_generic) {
        if (attn_mfma_supported(p)) {
            const int rc_fast = launch_attn_mfma(p, stream);
            if (rc_fast != kAttnNotHandled) return rc_fast;
        }
    }
    if (!g_force_generic && attn_tile_supported(p)) return launch_attn_tile(p, stream);      // up to kMaxGMfma heads per launch
    if (p.G > kMaxG) {
        for (int g1 = 0; g1 < p.G; g1 += kMaxG) {
            AttnParams h = with_heads(p, p.g0 + g1, p.G - g1 < kMaxG ? p.G - g1 : kMaxG);
            if (g1 > 0) after_first_launch(h);
            const int rc_h = launch_group(h, stream);
            if (rc_h != MILLION_OK) return rc_h;
        }
        return MILLION_OK;
    }
    if (!g_force_generic) {
        if (attn_tile_supported(p)) return launch_attn_tile(p, stream);
    }
    if (!g_force_generic) {
        // not a silent cliff: the scalar kernel is ~60x slower than the MFMA kernels.  Said once per process
        // on stderr (MILLION_QUIET=1 silences it); million_attn_kernel_kind() answers the same question without launching.
        static std::once_flag once;
        std::call_once(once, [&] {
            const char *q = getenv("MILLION_QUIET");
            if (!q || !*q || *q == '0')
                fprintf(stderr, "libmillion_hip: decode attention fell back to the scalar kernel (d=%d M=%d C=%d, k %s / v %s%s): "
                                "~60x slower than the MFMA kernels; they take d=128 with M in {64,32} or d in {64,128} with the "
                                "reference's other (M, C) and V in transposed pages (row-major K and V are transposed for you).\n",
                        p.d, p.M, p.C, p.k_paged ? "paged" : "row-major", p.v_paged ? "paged" : "row-major",
                        p.T == 0 ? ", no quantised tokens" : "");
        });
    }
    AttnParams g = p;
    choose_splits(g, 256);
    g.nslots = g.nsplit + 1;
    return launch_attn_generic(g, stream);
}

static int attn_impl(const million_attn_desc *desc, const void *q, const void *k_new, const void *v_new,
                     const void *k_codes, const void *v_codes,
                     const void *k_page_ids, const void *v_page_ids, const void *k_cents_prepared, const void *v_cents_prepared,
                     const void *k_resid, const void *v_resid, void *out, void *workspace,
                     size_t workspace_bytes, million_stream_t stream) {
    AttnParams p;
    const int rc = fill_attn_params(desc, p);
    if (rc != MILLION_OK) return rc;
    if ((k_new == nullptr) != (v_new == nullptr)) { set_error("attn: k_new and v_new must both be given"); return MILLION_ERR_ARG; }
    if (k_new) {
        if (!p.dev_lengths && p.r >= p.rcap) { set_error("attn: fused append into a full window (r=%d, cap=%d)", p.r, p.rcap); return MILLION_ERR_ARG; }
        if (((uintptr_t)k_new | (uintptr_t)v_new) & 15) { set_error("attn: k_new / v_new must be 16-byte aligned"); return MILLION_ERR_ALIGN; }
    }
    if (!q || !k_cents_prepared || !v_cents_prepared || !k_resid || !v_resid || !out || !workspace) { set_error("attn: null pointer"); return MILLION_ERR_ARG; }
    if (p.T > 0 && (!k_codes || !v_codes)) { set_error("attn: null code pointer with n_tokens=%d", p.T); return MILLION_ERR_ARG; }
    if (p.T > 0 && ((p.k_paged && !k_page_ids) || (p.v_paged && !p.v_identity && !v_page_ids))) { set_error("attn: paged layout without page ids"); return MILLION_ERR_ARG; }
    if (workspace_bytes < million_attn_workspace_bytes(desc)) { set_error("attn: workspace %zu < %zu bytes", workspace_bytes, million_attn_workspace_bytes(desc)); return MILLION_ERR_WORKSPACE; }
    if (((uintptr_t)q | (uintptr_t)k_codes | (uintptr_t)v_codes | (uintptr_t)k_resid | (uintptr_t)v_resid |
         (uintptr_t)out | (uintptr_t)workspace | (uintptr_t)k_cents_prepared | (uintptr_t)v_cents_prepared) & 15) {
        set_error("attn: every pointer must be 16-byte aligned");
        return MILLION_ERR_ALIGN;
    }
    if ((!p.k_paged && ((p.k_sb | p.k_sh) & 15)) || (!p.v_paged && ((p.v_sb | p.v_sh) & 15))) { set_error("attn: code strides must be multiples of 16 bytes"); return MILLION_ERR_ALIGN; }
    if ((p.res_sb | p.res_sh) & 7) { set_error("attn: residual strides must be multiples of 8 elements"); return MILLION_ERR_ALIGN; }
    const size_t tab = (size_t)p.M * p.C * p.dm;
    p.q = (const f16 *)q; p.k_codes = (const uint8_t *)k_codes; p.v_codes = (const uint8_t *)v_codes;
    p.k_ids32 = (const int *)k_page_ids; p.k_ids64 = (const long long *)k_page_ids;
    p.v_ids32 = (const int *)v_page_ids; p.v_ids64 = (const long long *)v_page_ids;
    p.k_tab = (const f16 *)k_cents_prepared; p.k_tab_col = p.k_tab + tab;
    p.v_tab = (const f16 *)v_cents_prepared; p.v_tab_col = p.v_tab + tab;
    p.k_res = (const f16 *)k_resid; p.v_res = (const f16 *)v_resid; p.out = (f16 *)out;
    p.k_new = (const f16 *)k_new; p.v_new = (const f16 *)v_new;
    p.k_res_w = (f16 *)k_resid; p.v_res_w = (f16 *)v_resid;
    p.dev_lengths_w = (int *)desc->dev_lengths;
    p.ws_cnt = (unsigned long long *)workspace;
    p.ws_cnt2 = (int *)((unsigned *)workspace + head_pairs(p.bs, p.nh_k) * kRecWords);
    p.ws_flags = (unsigned *)((char *)workspace + attn_cnt_bytes(p.bs, p.nh_k));
    p.dbg = g_dbg;
    p.k_pool_pages = desc->k_pool_pages; p.v_pool_pages = desc->v_pool_pages;
#ifdef MILLION_DEBUG_CHECK_IDS
    p.bad_ids = bad_ids_counter();
#endif
    p.ws_part = (float *)((char *)workspace + attn_cnt_bytes(p.bs, p.nh_k) + attn_flag_bytes(p.bs, p.nh_k));
    // The fast kernels want V in transposed pages: the reference's 10-arg row-major layout is transposed into scratch
    // pages first (once per call, whatever the number of query-head groups below).
    AttnParams pl = p;
    const AttnParams p8 = with_heads(p, 0, p.Gt < kMaxGMfma ? p.Gt : kMaxGMfma);
    if (!g_force_generic && !p.v_paged && !p.k_paged && (attn_mfma_shape_ok(p) || attn_tile_shape_ok(p8))) {
        pl.v_paged = 1; pl.v_identity = 1; pl.page_size = 64; pl.ps_shift = 6;
        pl.n_pages_cap = p.T > 0 ? (p.T + 63) / 64 : 1;
        if (p.T > 0) {
            uint8_t *scratch = (uint8_t *)workspace + attn_partial_bytes(p.bs, p.nh_k, p.G, p.d);
            hipLaunchKernelGGL(codes_transpose_kernel, dim3(pl.n_pages_cap, p.bs * p.nh_k), dim3(256), 0, (hipStream_t)stream,
                               p.v_codes, scratch, p.nh_k, p.T, p.M, p.v_sb, p.v_sh, pl.n_pages_cap);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) { set_error("codes_transpose launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
            pl.v_codes = scratch;
        }
    }
    // One launch of the MFMA kernels serves up to 16 query heads per kv head (the 16 columns of the score tile: Llama-3.1-405B's
    // nh / nh_k = 16 reads its codes once), of the other kernels up to 8; bigger groups run as several launches on the same
    // stream and workspace, each re-reading the codes.  A fused append happens in the first one: the later ones find the
    // row in the window.
    const AttnParams p16 = with_heads(pl, 0, p.Gt < kMaxGMfma ? p.Gt : kMaxGMfma);
    const int step = (!g_force_generic && (attn_mfma_supported(pl) || attn_tile_supported(p16))) ? kMaxGMfma : kMaxG;
    for (int g0 = 0; g0 < p.Gt; g0 += step) {
        AttnParams pg = with_heads(pl, g0, p.Gt - g0 < step ? p.Gt - g0 : step);
        if (g0 > 0) after_first_launch(pg);
        const int rc_g = launch_group(pg, (hipStream_t)stream);
        if (rc_g != MILLION_OK) return rc_g;
    }
    return MILLION_OK;
}

int million_pq_decode_attn(const million_attn_desc *desc, const void *q, const void *k_codes, const void *v_codes,
                           const void *k_page_ids, const void *v_page_ids, const void *k_cents_prepared, const void *v_cents_prepared,
                           const void *k_resid, const void *v_resid, void *out, void *workspace,
                           size_t workspace_bytes, million_stream_t stream) {
    return attn_impl(desc, q, nullptr, nullptr, k_codes, v_codes, k_page_ids, v_page_ids, k_cents_prepared,
                     v_cents_prepared, k_resid, v_resid, out, workspace, workspace_bytes, stream);
}

int million_pq_decode_attn_append(const million_attn_desc *desc, const void *q, const void *k_new, const void *v_new,
                                  const void *k_codes, const void *v_codes, const void *k_page_ids,
                                  const void *v_page_ids, const void *k_cents_prepared, const void *v_cents_prepared,
                                  void *k_resid, void *v_resid, void *out, void *workspace, size_t workspace_bytes,
                                  million_stream_t stream) {
    if (!k_new || !v_new) { set_error("attn_append: k_new / v_new null"); return MILLION_ERR_ARG; }
    return attn_impl(desc, q, k_new, v_new, k_codes, v_codes, k_page_ids, v_page_ids, k_cents_prepared,
                     v_cents_prepared, k_resid, v_resid, out, workspace, workspace_bytes, stream);
}

int million_residual_append(const void *k_new, const void *v_new, void *k_resid, void *v_resid, int bs, int nh_k,
                            int d, int resid_cap, int64_t resid_stride_b, int64_t resid_stride_h, int r,
                            int resid_start, int32_t *dev_lengths, million_stream_t stream) {
    if (!k_new || !v_new || !k_resid || !v_resid) { set_error("residual_append: null pointer"); return MILLION_ERR_ARG; }
    if (bs <= 0 || nh_k <= 0 || d <= 0 || resid_cap <= 0) { set_error("residual_append: bad shape"); return MILLION_ERR_SHAPE; }
    if (!dev_lengths && (r < 0 || r >= resid_cap || resid_start < 0 || resid_start >= resid_cap)) {
        set_error("residual_append: r=%d start=%d cap=%d (window full?)", r, resid_start, resid_cap);
        return MILLION_ERR_ARG;
    }
    hipLaunchKernelGGL(residual_append_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const f16 *)k_new,
                       (const f16 *)v_new, (f16 *)k_resid, (f16 *)v_resid, bs, nh_k, d, resid_cap,
                       (long long)resid_stride_b, (long long)resid_stride_h, r, resid_start, dev_lengths);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("residual_append launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

int million_lengths_advance(int32_t *dev_lengths, int bs, int n_flushed, int resid_cap, million_stream_t stream) {
    if (!dev_lengths || bs <= 0 || resid_cap <= 0) { set_error("lengths_advance: bad argument"); return MILLION_ERR_ARG; }
    hipLaunchKernelGGL(lengths_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_lengths, bs, n_flushed, resid_cap);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("lengths_advance launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // extern "C"
