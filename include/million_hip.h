/*
 * million_hip.h — C ABI of libmillion_hip.so: MI355X-native (gfx950) PQ-KV attention hot path.
 *
 * This is the drop-in boundary below the reference's Python module `bindings`
 * (reference: scripts/modeldb/bindings/bindings.template.cpp:11-63 declares, and
 * scripts/modeldb/bindings/Interface.template.cu:16-147 defines, one torch-typed C++ symbol per
 * (Ns, Lt, d, M, C) tuple).  Here ONE set of plain-C entry points sits behind all those names:
 * plain pointers and sizes, no torch types.  The Python shim `bindings/` (built by `make bindings`)
 * re-exports the reference's function names on top of these entry points.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name starts with `host_`;
 *   - fp16 tensors are IEEE binary16 (`_Float16`), codes are uint8 (reference setup.py:10-11);
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is what the
 *     reference launches on, Interface.template.cu:65,88,109);
 *   - functions return MILLION_OK (0) or a negative error code and NEVER exit the process
 *     (the reference's gpuErrchk calls exit(), Interface.template.cu:3-11); million_last_error()
 *     returns a thread-local description of the last failure;
 *   - no entry point allocates, frees or synchronises: all are legal inside hipGraph capture.
 */
#ifndef MILLION_HIP_H
#define MILLION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MILLION_HIP_VERSION 1

enum {
    MILLION_OK = 0,
    MILLION_ERR_SHAPE = -1,       /* unsupported or inconsistent shape */
    MILLION_ERR_ALIGN = -2,       /* pointer / stride alignment */
    MILLION_ERR_ARG = -3,         /* null pointer, bad enum, r out of range ... */
    MILLION_ERR_WORKSPACE = -4,   /* workspace too small */
    MILLION_ERR_LAUNCH = -5       /* hipGetLastError() after launch */
};

typedef void *million_stream_t;

int million_version(void);
const char *million_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Codebook preparation.
 * Replaces: the per-call `.contiguous()` / transposes the reference applies to the centroid table
 * (scripts/utils/pq_utils.py:149-159 set_cent; Interface.template.cu:49-50 key_cents.transpose).
 * Input : cents (M, C, d_m) fp16 contiguous — the reference's codebook tensor (main_pq.py:252-260).
 * Output: `prepared`, million_prepared_cents_bytes() bytes, two LDS-ready images back to back:
 *           [0, M*C*d_m*2)            "row image"  [m][c][d_m]   (bank = code: used for K lookups)
 *           [M*C*d_m*2, 2*M*C*d_m*2)  "col image"  [c][m][d_m]   (bank = subspace: used for V lookups)
 * Call once per codebook (set_cent time); the decode entry points take prepared tables. */
size_t million_prepared_cents_bytes(int M, int C, int d_m);
int million_prepare_cents(const void *cents, int M, int C, int d_m, void *prepared,
                          million_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * PQ encode.
 * Replaces: sa_encode_4d_keops (scripts/utils/pq_utils.py:451-499; call sites :189-190,:235-236,
 * :292-293 and paged_pq_utils.py:161,167,241-242) plus the permute+cat that stores the codes
 * (pq_utils.py:140-147, paged_pq_utils.py:162,173-175).
 * codes[b,hk,t,m] = argmin_c sum_k (x[b,hk,t,m*d_m+k] - cents[m,c,k])^2, fp32 direct form,
 * sequential k, lowest c wins ties; bit-exact with oracle/pq_oracle.c:pq_encode_direct. */
enum {
    MILLION_CODES_ROWMAJOR = 0,   /* dst (bs, nh_k, T_cap, M): reference layout, pq_utils.py:497-499 */
    MILLION_CODES_KPAGES = 1,     /* dst pool (n_pool, page_size, M), via page ids                     */
    MILLION_CODES_VPAGES = 2      /* dst pool (n_pool, M, page_size): transposed pages,                */
                                  /*     paged_pq_utils.py:173-175, MILLION_技术分析文档.md:1330-1340   */
};

typedef struct {
    uint32_t struct_size;         /* = sizeof(million_encode_desc) */
    int32_t bs, nh_k, n;          /* X is (bs, nh_k, n, d) */
    int32_t d, M, C;
    int64_t x_stride_b, x_stride_h, x_stride_n;   /* in fp16 elements; innermost dim contiguous */
    int32_t x_row_start, x_row_mod;   /* row t of X is read at ((x_row_start + t) % x_row_mod) if x_row_mod > 0
                                         (residual ring buffer); else at t */
    int32_t dst_layout;           /* MILLION_CODES_* */
    int32_t dst_token_start;      /* first destination token index t0: tokens [t0, t0+n) are written */
    int64_t dst_stride_b, dst_stride_h;   /* bytes; ROWMAJOR only (row stride is M bytes) */
    int32_t page_size;            /* KPAGES / VPAGES */
    int32_t n_pages_cap;          /* page_ids is (bs, nh_k, n_pages_cap) int32 */
    const int32_t *dev_lengths;   /* optional device array (bs, 4) = {n_tokens, r, resid_start, 0}: when set,
                                     dst_token_start := n_tokens and x_row_start := resid_start are read on the
                                     device (flush of the residual window inside a replayed hipGraph) */
    const void *cents_prepared;   /* optional: the same codebook through million_prepare_cents (its fp32 image
                                     feeds the scalar operands of the distance loop: ~1.6x faster); codes are
                                     identical with or without it */
} million_encode_desc;

int million_pq_encode(const million_encode_desc *desc, const void *x, const void *cents /* (M,C,d_m) fp16 */,
                      void *dst, const int32_t *page_ids, million_stream_t stream);
/* Code width: C <= 256 -> dst holds uint8 codes; 256 < C <= 65536 (nbits 9..16) -> uint16 codes in the same three
 * layouts (reference nbits2dtype, scripts/utils/pq_utils.py:542-552; sa_encode_4d*(target_dtype=...), :449/:499).
 * dst_stride_b / dst_stride_h stay in BYTES.  The decode-attention kernels are uint8-only, as the reference's are
 * (setup.py:10-11); wide codes serve the dequantise-then-attend path (DynamicPQCache.update, pq_utils.py:166-220). */

/* One launch per window flush.
 * Replaces: PagedPQCache.flush_to_pages (scripts/utils/paged_pq_utils.py:130-210): encode of the oldest `desc->n` K
 * rows and V rows of the residual window (two sa_encode_4d_keops calls, :161,:167), the permute + torch.cat that store
 * them (:162,:173-175) and the window shift (:188-204; here: the ring start advances).
 * desc describes the K side: X = k_rows (the K window buffer; the V window must have the same shape and strides),
 * dst_layout = MILLION_CODES_KPAGES into k_pool; the V side is written as MILLION_CODES_VPAGES into v_pool through the
 * same page ids.  dev_lengths (must equal desc->dev_lengths; may be NULL): the destination token and the ring start are
 * read on the device and, once every workgroup has read them, advanced there (n_tokens += n, r -= n, resid_start =
 * (resid_start + n) % resid_cap; the 4th word of each row is the workgroups' ticket and is left at 0).  uint8 codes.
 * min_r (device lengths only; 0 = every batch item): batch items whose window holds fewer than min_r rows are left alone -
 * requests of different lengths share the launch and only those whose window is full flush (pass resid_cap). */
int million_pq_flush(const million_encode_desc *desc, const void *k_rows, const void *v_rows,
                     const void *k_cents, const void *v_cents, void *k_pool, void *v_pool,
                     const int32_t *page_ids, int32_t *dev_lengths, int resid_cap, int min_r, million_stream_t stream);

/* The same for n_layers layers of a cache in ONE launch, and optionally WITHOUT moving the window (encode-ahead).
 * The layers' K / V window buffers, page tables and length rows lie rows_layer_stride fp16 elements, ids_layer_stride
 * int32 and lengths_layer_stride int32 apart (layer 0 at the pointers given); pools and codebooks are shared, as in the
 * reference (one PagedPQCache, one codebook pair for all layers: paged_pq_utils.py:70-80, pq_utils.py:149-159).
 * advance = 0: the rows are encoded into the pages of tokens [n_tokens, n_tokens + n) and nothing else changes - the oldest
 * page of window rows is complete long before the window is full (the reference flushes at r >= extended_residual_size,
 * paged_pq_utils.py:359-361; the rows exist from r >= page_size on), so a cache can encode them during any earlier step,
 * beside that step's attention launches (the kernel runs 4 waves of <= 32 registers per workgroup: it fits on a CU next to
 * an attention workgroup), and commit the flush later with million_lengths_advance alone.  Codes and final state are
 * identical to million_pq_flush at the flush step.  min_r (device lengths) then means "rows present": pass page_size. */
int million_pq_flush_layers(const million_encode_desc *desc, const void *k_rows, const void *v_rows,
                            const void *k_cents, const void *v_cents, void *k_pool, void *v_pool,
                            const int32_t *page_ids, int32_t *dev_lengths, int resid_cap, int min_r,
                            int n_layers, int64_t rows_layer_stride, int64_t ids_layer_stride, int64_t lengths_layer_stride,
                            int advance, million_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * PQ decode (reconstruction).
 * Replaces: sa_decode_4d (scripts/utils/pq_utils.py:501-540): out[row, m*d_m + k] = cents[m, codes[row, m], k].
 * codes: (n_rows, M) u8 contiguous (any leading dims flattened); cents: the RAW (M, C, d_m) fp16 codebook;
 * out: (n_rows, d) fp16 contiguous.  Exact (a gather).  Used by the reference only in fallbacks and the
 * perplexity path (pq_utils.py:198-204), so this is a plain bandwidth kernel with the codebook staged in LDS. */
int million_pq_decode(const void *codes /* uint8 for C <= 256, uint16 above */, const void *cents, void *out,
                      int64_t n_rows, int d, int M, int C, million_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused decode-step attention.
 * Replaces: flash_decoding_allocated_buffer<> (Interface.template.cu:26-120), i.e. the LUT matmul
 * (:49-50), flash_decoding_split_kernel (Kernel.cuh:11-166), flash_decoding_residual_kernel
 * (Kernel.cuh:1038-1209), torch::zeros (:104) and flash_decoding_reduce_kernel (Kernel.cuh:1211-1270)
 * — one launch; and the intended 13-argument flash_decoding_paged_v (call site
 * scripts/utils/paged_pq_utils.py:621-635; spec MILLION_技术分析文档.md:1292-1345) when the codes
 * live in page pools.
 *
 * out[b,h,:] = softmax_j( q[b,h,:] . Kfull[b,hk,j,:] / sqrt(d) ) Vfull[b,hk,j,:],  hk = h / (nh/nh_k),
 * Kfull = [ dequant(k codes, T tokens) ; k_resid valid rows (r rows) ], likewise V.
 * T = 0 and r = 0 (nothing to attend to): out = 0 (the reference divides 0 by 0).
 */
enum {
    MILLION_KV_ROWMAJOR = 0,      /* codes (bs, nh_k, T_cap, M) u8: reference layout (Interface.template.cu:29-30) */
    MILLION_KV_PAGED = 1          /* K: pool (n_pool, page_size, M); V: pool (n_pool, M, page_size) — transposed  */
                                  /* pages (paged_pq_utils.py:173-175); page ids (bs, nh_k, n_pages_cap).          */
                                  /* The reference's 13-arg paged call mixes row-major K with paged V              */
                                  /* (paged_pq_utils.py:621-635): k_layout and v_layout are independent.           */
};

typedef struct {
    uint32_t struct_size;         /* = sizeof(million_attn_desc) */
    int32_t bs, nh, nh_k;         /* any nh / nh_k >= 1: up to 16 query heads per kv head are one launch on the d = 128, M in {64, 32}
                                     shapes (8 on the others); bigger groups run as several launches inside the call */
    int32_t d, M, C;
    int32_t n_tokens;             /* T: quantised tokens per (b, hk); upper bound if dev_lengths != NULL */
    int32_t r;                    /* valid residual rows, 0 <= r <= resid_cap */
    int32_t resid_start;          /* first valid residual row (ring buffer); reference: always 0 */
    int32_t resid_cap;            /* Lt: rows of the residual buffers (reference: Lt == d) */
    int64_t resid_stride_b, resid_stride_h;   /* fp16 elements; row stride is d */
    int32_t k_layout, v_layout;   /* MILLION_KV_* */
    int32_t page_size;            /* PAGED: tokens per page (32, 64 or 128) */
    int32_t n_pages_cap;          /* PAGED: row length of the page-id arrays */
    int32_t page_ids_i64;         /* PAGED: 0 = page ids are int32, 1 = int64 (reference passes int64, paged_pq_utils.py:440) */
    int32_t v_pages_dense;        /* v_layout PAGED only: 1 = the V pool passed to the call is the dense run of transposed
                                     64-token pages million_transpose_v_codes writes (page p of (b, hk) is pool page
                                     (b*nh_k + hk) * n_pages_cap + p): no V page-id array is read (v_page_ids may be NULL) */
    int64_t k_stride_b, k_stride_h;   /* ROWMAJOR: bytes between batches / kv heads of k_codes */
    int64_t v_stride_b, v_stride_h;   /* ROWMAJOR: same for v_codes */
    const int32_t *dev_lengths;   /* optional device array (bs, 4) = {n_tokens, r, resid_start, 0}: when set,
                                     lengths are read on the device (graph replay with changing lengths); every
                                     batch item has its own row, so requests of different lengths can share a
                                     launch - n_tokens above is then only the bound the grid is sized for */
    int32_t k_pool_pages;         /* PAGED: pages in the K / V pools handed to the call; 0 = not given.  Page ids are trusted, */
    int32_t v_pool_pages;         /* as in the reference (paged_pq_utils.py:440-441): only a library built with
                                     -DMILLION_DEBUG_CHECK_IDS (make debug-ids -> libmillion_hip_dbgids.so) reads these,
                                     maps ids outside [0, pool_pages) to page 0 and counts them (million_debug_bad_page_ids) */
} million_attn_desc;

size_t million_attn_workspace_bytes(const million_attn_desc *desc);
/* The workspace must be zeroed once after allocation (million_workspace_init or any memset); every call
 * leaves it ready for the next one.  Calls of DIFFERENT shapes may share a workspace (sized for the largest) as long as
 * bs * nh_k <= 2048; a shape with more (b, kv head) pairs needs a workspace of its own. */
int million_workspace_init(void *workspace, size_t bytes, million_stream_t stream);

int million_pq_decode_attn(const million_attn_desc *desc,
                           const void *q,              /* (bs, nh, 1, d) fp16 contiguous */
                           const void *k_codes,        /* ROWMAJOR tensor or K page pool */
                           const void *v_codes,        /* ROWMAJOR tensor or V page pool */
                           const void *k_page_ids,     /* k_layout PAGED only: int32 or int64, see page_ids_i64 */
                           const void *v_page_ids,     /* v_layout PAGED only (may alias k_page_ids) */
                           const void *k_cents_prepared, const void *v_cents_prepared,
                           const void *k_resid, const void *v_resid,   /* (bs, nh_k, resid_cap, d) fp16 */
                           void *out,                  /* (bs, nh, 1, d) fp16 */
                           void *workspace, size_t workspace_bytes,
                           million_stream_t stream);

/* Same, fused with the residual-window append of the new token (replaces the two slice-assign copies of
 * DynamicPQCache.decoding, scripts/utils/pq_utils.py:304-312, AND the attention launch that follows them,
 * :314-326): k_new / v_new (bs, nh_k, 1, d) fp16 are attended to as one more window row and stored into
 * row (resid_start + r) % resid_cap of k_resid / v_resid, where r = desc->r is the number of valid rows
 * BEFORE the call (r < resid_cap).  With dev_lengths the row count is read on the device and incremented
 * there once every workgroup has read it. */
int million_pq_decode_attn_append(const million_attn_desc *desc, const void *q, const void *k_new, const void *v_new,
                                  const void *k_codes, const void *v_codes, const void *k_page_ids,
                                  const void *v_page_ids, const void *k_cents_prepared,
                                  const void *v_cents_prepared, void *k_resid, void *v_resid, void *out,
                                  void *workspace, size_t workspace_bytes, million_stream_t stream);

/* Row-major V codes (bs, nh_k, T, M) u8 (the reference's 10-argument layout, Interface.template.cu:30) -> the dense run
 * of transposed 64-token pages ((bs*nh_k) * ceil(T/64), M, 64) that the fast kernels read with v_pages_dense = 1,
 * page_size = 64, n_pages_cap = ceil(T/64).  Replaces the per-call pad + view + transpose + contiguous of
 * PagedPQCache._call_paged_kernel (scripts/utils/paged_pq_utils.py:464-500).  A caller that passes the same V code
 * tensor on many decode steps (the reference does, between two flushes) transposes once and reuses the pages;
 * million_pq_decode_attn with v_layout = ROWMAJOR does the same transpose into its workspace on EVERY call. */
int million_transpose_v_codes(const void *v_codes, void *v_pages, int bs, int nh_k, int n_tokens, int M,
                              int64_t v_stride_b, int64_t v_stride_h, million_stream_t stream);

/* Which kernel million_pq_decode_attn would pick for a descriptor: 1 = streaming MFMA kernels (d = 128 with M in {64, 32}, and
 * d = 128 / M = 16 with up to 16 query heads per kv head; any batch and any context up to 1M tokens per (b, kv head):
 * calls with more than 64 rounds per wave get more splits; with 256 or 128 centroids, up to 4 query heads per kv head and pages of 64 / 128
 * tokens the "lean" form of it runs - csrc/attn_lean.h - which also takes d = 64 with M in {64, 32, 16} at up to 16
 * query heads per kv head: 5 and more run as ceil(G / 4) virtual kv heads of 3 / 4 query heads, while bs * nh_k * parts <= 2048),
 * 2 = the same after transposing row-major V codes into workspace scratch (one extra launch), 3 = tile MFMA kernel
 * (d = 64 on 32-token pages; d = 64 or d = 128 / M = 16 with more than 2048 virtual (b, kv head) pairs),
 * 4 = the same after the transpose, 5 = the grouped MFMA kernel (the streaming kernel's fallback on its M = 64 / 32 shapes: no
 * quantised token yet, or more than 1M tokens), 0 = scalar fallback (anything else the descriptor allows: C not 128 / 256, paged K
 * with row-major V, ...), -1 = bad descriptor. */
int million_attn_kernel_kind(const million_attn_desc *desc);
/* Kernel choice for A/B measurements and tests: 0 = auto (default), 1 = generic kernel only, 2 = MFMA grouped kernel
 * only (never the streaming one), 4 = auto, but the helper workgroups of the split merge give up at once (exercises the
 * last arriver's take-over path of the MFMA kernels' tail: every give-up bit is set before the launch's first ticket),
 * 8 = auto, but the helpers have no patience: each gives up through the real path (its atomic on the ticket word) unless
 * every workgroup has already taken its ticket, 16 = auto, but the shapes of the lean kernel (csrc/attn_lean.h) stay on the
 * streaming / tile kernels and no call runs as virtual kv heads (A/B and the tests of those kernels' forms), 64 = auto, but million_prefill_attn runs its plain
 * tile loop at d = 128 instead of the pipelined one (csrc/prefill.hip; A/B and the tests of both forms). */
void million_set_force_generic(int on);

/* ------------------------------------------------------------------------------------------------
 * Prompt (prefill) attention on fp16 K/V.
 * Replaces: scaled_dot_product_attention(q, repeat_kv(k), repeat_kv(v), is_causal=True) of the reference's prompt pass
 * (scripts/utils/pq_utils.py:249-260 DynamicPQCache.prefill, scripts/utils/paged_pq_utils.py:216-320
 * PagedPQCache.prefill; baseline: scripts/modeldb/models/modeling_llama.py:403-443) - GQA without materialising
 * repeat_kv: the nh / nh_k query heads of a kv head share the K/V tiles of one workgroup.
 *   out[b,h,i,:] = softmax_{j <= q_pos0 + i (causal), j < n_kv}( q[b,h,i,:] . k[b,hk,j,:] / sqrt(d) ) v[b,hk,j,:],  hk = h / (nh/nh_k)
 * fp16 in / out, fp32 scores, online softmax and accumulation; d = 128 or 64.  causal = 0: every key (j < n_kv).
 * torch's is_causal=True with q_len == kv_len is q_pos0 = 0; a prompt chunk behind n_past cached fp16 rows is q_pos0 = n_past. */
typedef struct {
    uint32_t struct_size;         /* = sizeof(million_prefill_desc) */
    int32_t bs, nh, nh_k, d;
    int32_t n_q, n_kv;            /* query rows, key/value rows */
    int32_t q_pos0;               /* position of query row 0 among the keys */
    int32_t causal;
    int64_t q_stride_b, q_stride_h, q_stride_n;   /* fp16 elements; the d elements of a row are contiguous; multiples of 8 */
    int64_t k_stride_b, k_stride_h, k_stride_n;
    int64_t v_stride_b, v_stride_h, v_stride_n;
    int64_t o_stride_b, o_stride_h, o_stride_n;
} million_prefill_desc;

int million_prefill_attn(const million_prefill_desc *desc, const void *q /* (bs, nh, n_q, d) */,
                         const void *k /* (bs, nh_k, n_kv, d) */, const void *v, void *out /* (bs, nh, n_q, d) */,
                         million_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Residual-window append.
 * Replaces: the two slice-assign copies of DynamicPQCache.decoding (pq_utils.py:304-312) /
 * PagedPQCache.decoding_with_pages (paged_pq_utils.py:377-380).
 * Writes k_new/v_new (bs, nh_k, 1, d) into row (resid_start + r) % resid_cap of the residual buffers;
 * when dev_lengths != NULL the row comes from the device array and r is incremented there. */
int million_residual_append(const void *k_new, const void *v_new, void *k_resid, void *v_resid,
                            int bs, int nh_k, int d, int resid_cap,
                            int64_t resid_stride_b, int64_t resid_stride_h,
                            int r, int resid_start, int32_t *dev_lengths, million_stream_t stream);

/* After a flush of `n_flushed` residual rows into the code store (reference: flush_to_pages,
 * paged_pq_utils.py:181-208 / DynamicPQCache.decoding, pq_utils.py:297-301) advance the device-resident
 * lengths: n_tokens += n_flushed, r -= n_flushed, resid_start = (resid_start + n_flushed) % resid_cap. */
int million_lengths_advance(int32_t *dev_lengths, int bs, int n_flushed, int resid_cap, million_stream_t stream);

/* Diagnostics only.  -1 in the product build.  In a library built with -DMILLION_DEBUG_CHECK_IDS: waits for the device and
 * returns (and clears) the number of page ids outside [0, k_pool_pages) / [0, v_pool_pages) that the three decode-attention
 * kernels have met since the last call; such ids were read as page 0 instead of as out-of-bounds addresses. */
int million_debug_bad_page_ids(void);
/* Diagnostics: waits for the device and returns (and clears) the number of (b, kv head) merges of the MFMA decode-attention
 * kernels that gave up waiting for a split's partial (a workgroup of the launch died, or the workspace was not zeroed).  The
 * heads concerned were written as NaN.  0 in every healthy run; after a non-zero answer zero the workspace again
 * (million_workspace_init).  -1: the runtime refused the read. */
int million_debug_tail_faults(void);
/* Diagnostics only: when `buf` is non-NULL the decode-attention kernels store up to 16 x uint64 realtime-counter
 * stamps (100 MHz) per wave at their phase boundaries into buf (grid_size * 8 waves * 32 slots entries).  NULL = off. */
void million_debug_set_stamp_buffer(void *buf);
/* Diagnostics only: runs the kernel's row-swap reductions on one wave: out_max[l] / out_sum[l] = max / sum of
 * in[l % 16 + 16*k], k = 0..3 (device pointers to 64 floats each). */
int million_debug_rows_reduce(const float *in64, float *out_max64, float *out_sum64, million_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MILLION_HIP_H */
