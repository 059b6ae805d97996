// encode.hip — PQ encode (argmin over centroids per sub-vector), gfx950.
//
// Replaces sa_encode_4d_keops (reference scripts/utils/pq_utils.py:451-499: fp32 upcast :483-484,
// ((x-c)**2).sum(-1).argmin :491-494) and the permute/cat that stores the codes (:497-499,
// paged_pq_utils.py:162,173-175): codes are written straight into their final place — row-major K
// store, K page pool, or transposed V page pool.
//
// Arithmetic contract (bit-exact with oracle/pq_oracle.c:pq_encode_direct): e = x - c, sq = e * e,
// acc = sq_0 + sq_1 + ... sequentially, every operation one IEEE fp32 round-to-nearest, no FMA
// contraction; strict '<' scan over increasing c, so the lowest index wins exact ties.
//
// Mapping: a wave owns 4 consecutive subspaces of 64 tokens (lane = token): the lane's 4*d_m input halfs are
// one vector load, the 4 code bytes of a token leave as one 32-bit store (row-major / K pages) or as four
// lane-contiguous byte rows (transposed V pages).  The centroid row of a subspace is wave-uniform and comes
// through the scalar cache as SGPR operands: from the fp32 image of a prepared codebook (million_prepare_cents)
// the inner loop is, for d_m = 2, v_pk_add_f32 (x - c), v_pk_mul_f32, v_add_f32, v_cmp_lt_f32, v_cndmask (index),
// v_min_f32 (distance): 6 vector instructions per centroid test; from the raw fp16 codebook two more
// (v_cvt_f32_f16 of the centroid pair).  Packed fp32 operations round each half exactly like the scalar ones.
//
// Roofline: pure vector ALU.  32K tokens x 8 kv heads x 64 subspaces x 256 centroids = 4.3e9 tests per layer and
// side; at 6 instructions per test and one 64-lane instruction per 4 cycles per SIMD (1024 SIMDs, ~2.4 GHz:
// 39e12 lane-instructions/s) the floor is ~0.66 ms; HBM traffic (67 MB in, 17 MB out) is two orders below that.
#include "common.h"

#pragma clang fp contract(off)

namespace million {

constexpr int kEncBlock = 256;
constexpr int kEncSub = 4;      // subspaces per wave

typedef float v2f __attribute__((ext_vector_type(2)));

// CodeT = uint8_t (C <= 256) or uint16_t (nbits 9..16: reference nbits2dtype, pq_utils.py:542-552).
template <int DM, bool F32TAB, typename CodeT>
__global__ __launch_bounds__(kEncBlock) void pq_encode_kernel(EncParams p) {
    constexpr int kCB = sizeof(CodeT) * 8;      // bits per stored code
    const int lane = threadIdx.x & 63;
    const int mg = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * (kEncBlock / 64) + (threadIdx.x >> 6)));
    const int m0 = mg * kEncSub;
    const int bh = blockIdx.z;
    const int b = bh / p.nh_k, hk = bh % p.nh_k;
    const int t = blockIdx.x * 64 + lane;
    if (m0 >= p.M) return;
    const bool valid = t < p.n;
    const int tc = valid ? t : p.n - 1;
    int tok0 = p.tok0, xrow_start = p.xrow_start;
    if (p.dev_lengths) { tok0 = p.dev_lengths[b * 4 + 0]; xrow_start = p.dev_lengths[b * 4 + 2]; }
    // device-resident values are not trusted: a ring start outside [0, xrow_mod) becomes 0, a destination token outside
    // the page table (or negative) drops the store (below)
    if (p.xrow_mod > 0 && (unsigned)xrow_start >= (unsigned)p.xrow_mod) xrow_start = 0;
    const int xrow = p.xrow_mod > 0 ? (xrow_start + tc) % p.xrow_mod : tc;
    const f16 *xp = p.x + b * p.xsb + hk * p.xsh + (long long)xrow * p.xsn + m0 * DM;
    const int nsub = p.M - m0 < kEncSub ? p.M - m0 : kEncSub;      // wave-uniform
    float x[kEncSub][DM];
#pragma unroll
    for (int j = 0; j < kEncSub; ++j)
#pragma unroll
        for (int k = 0; k < DM; ++k) x[j][k] = j < nsub ? (float)xp[j * DM + k] : 0.f;

    unsigned long long codes = 0;
#pragma unroll
    for (int j = 0; j < kEncSub; ++j) {
        if (j >= nsub) break;
        const int m = m0 + j;
        const float *__restrict__ c32 = p.cents32 + (long long)m * p.C * DM;   // wave-uniform rows
        const f16 *__restrict__ c16 = p.cents + (long long)m * p.C * DM;
        float best = INFINITY;
        int best_c = 0;
#pragma unroll 16
        for (int c = 0; c < p.C; ++c) {
            float cv[DM];
#pragma unroll
            for (int k = 0; k < DM; ++k) cv[k] = F32TAB ? c32[c * DM + k] : (float)c16[c * DM + k];
            float acc = 0.f;
            if (DM % 2 == 0) {
                // pairs of dims as packed fp32: every lane-half is one IEEE round-to-nearest operation
#pragma unroll
                for (int k = 0; k < DM; k += 2) {
                    const v2f xv = {x[j][k], x[j][k + 1]}, cc = {cv[k], cv[k + 1]};
                    const v2f e = xv - cc;
                    const v2f sq = e * e;
                    acc = (k == 0) ? sq[0] : acc + sq[0];
                    acc = acc + sq[1];
                }
            } else {
#pragma unroll
                for (int k = 0; k < DM; ++k) {
                    const float e = x[j][k] - cv[k];
                    const float sq = e * e;
                    acc = (k == 0) ? sq : acc + sq;
                }
            }
            // strict '<', increasing c: the lowest index wins exact ties.  Written as "keep unless smaller" so that the
            // select is (condition ? register : constant): v_cndmask takes the constant c as a literal, no v_mov
            best_c = !(acc < best) ? best_c : c;
            best = fminf(best, acc);
        }
        codes |= (unsigned long long)(unsigned)best_c << (kCB * j);
    }
    if (!valid) return;
    const int tok = tok0 + t;
    if (tok0 < 0 || (p.layout != MILLION_CODES_ROWMAJOR && tok / p.page_size >= p.n_pages_cap)) return;
    CodeT *dst = (CodeT *)p.dst;      // strides dsb / dsh are in bytes
    if (p.layout == MILLION_CODES_VPAGES) {
        const long long pid = p.page_ids[(long long)bh * p.n_pages_cap + tok / p.page_size];
        const int off = tok % p.page_size;
#pragma unroll
        for (int j = 0; j < kEncSub; ++j)
            if (j < nsub) dst[(pid * p.M + m0 + j) * p.page_size + off] = (CodeT)(codes >> (kCB * j));
        return;
    }
    CodeT *row;
    if (p.layout == MILLION_CODES_ROWMAJOR) {
        row = (CodeT *)(p.dst + b * p.dsb + hk * p.dsh) + (long long)tok * p.M + m0;
    } else {
        const long long pid = p.page_ids[(long long)bh * p.n_pages_cap + tok / p.page_size];
        row = dst + (pid * p.page_size + tok % p.page_size) * p.M + m0;
    }
    if (nsub == kEncSub && ((size_t)row & (4 * sizeof(CodeT) - 1)) == 0) {
        if (sizeof(CodeT) == 1) *(unsigned *)row = (unsigned)codes;
        else *(unsigned long long *)row = codes;
    } else {
#pragma unroll
        for (int j = 0; j < kEncSub; ++j)
            if (j < nsub) row[j] = (CodeT)(codes >> (kCB * j));
    }
}

// Small calls (a flush of 64 window rows: 512 (row, subspace) pairs per kv head) cannot hide the scalar-cache
// round trips of the kernel above behind other waves: 32 dependent s_load batches of ~0.6 us each.  This variant
// puts the centroid row into LDS with ONE vector round trip (converted to fp32 on the way) and reads it back with
// wave-uniform (broadcast) ds_reads that pipeline; one subspace per workgroup, a quarter of the centroids per wave.
template <int DM>
__device__ __forceinline__ void encode_small_body(const EncParams &p, int m, int bh, int tblock) {
    // one subspace and 64 tokens per WORKGROUP: its four waves scan a quarter of the centroids each (a 4x shorter
    // dependent chain), wave 0 picks among the four candidates in centroid order (strict '<': lowest index on ties)
    constexpr int kW = kEncBlock / 64;
    __shared__ float rows[256 * DM];
    __shared__ float cand_d[kW][64];
    __shared__ int cand_c[kW][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = bh / p.nh_k, hk = bh % p.nh_k;
    const int t = tblock * 64 + lane;
    const int cq = (p.C + kW - 1) / kW;                   // centroids per wave
    const int c0 = w * cq, c1 = min(c0 + cq, p.C);
    const f16 *cm = p.cents + (long long)m * p.C * DM;
    for (int e = c0 * DM + lane; e < c1 * DM; e += 64) rows[e] = (float)cm[e];      // own quarter, own wave: LDS order suffices
    const bool valid = t < p.n;
    const int tc = valid ? t : p.n - 1;
    int tok0 = p.tok0, xrow_start = p.xrow_start;
    if (p.dev_lengths) { tok0 = p.dev_lengths[b * 4 + 0]; xrow_start = p.dev_lengths[b * 4 + 2]; }
    if (p.xrow_mod > 0 && (unsigned)xrow_start >= (unsigned)p.xrow_mod) xrow_start = 0;      // not trusted (see pq_encode_kernel)
    const int xrow = p.xrow_mod > 0 ? (xrow_start + tc) % p.xrow_mod : tc;
    const f16 *xp = p.x + b * p.xsb + hk * p.xsh + (long long)xrow * p.xsn + m * DM;
    float x[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) x[k] = (float)xp[k];
    float best = INFINITY;
    int best_c = c0;
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            const float e = x[k] - rows[c * DM + k];
            const float sq = e * e;
            acc = (k == 0) ? sq : acc + sq;
        }
        best_c = acc < best ? c : best_c;
        best = fminf(best, acc);
    }
    cand_d[w][lane] = best;
    cand_c[w][lane] = best_c;
    __syncthreads();
    if (w != 0 || !valid) return;
    // an empty quarter (C < 4 quarters) leaves +inf: never smaller than a real candidate; all-inf keeps quarter 0's index
#pragma unroll
    for (int j = 1; j < kW; ++j) {
        const float dj = cand_d[j][lane];
        if (dj < best) { best = dj; best_c = cand_c[j][lane]; }
    }
    const int tok = tok0 + t;
    if (tok0 < 0) return;
    if (p.layout == MILLION_CODES_ROWMAJOR) {
        p.dst[b * p.dsb + hk * p.dsh + (long long)tok * p.M + m] = (uint8_t)best_c;
    } else {
        if (tok / p.page_size >= p.n_pages_cap) return;
        const long long pid = p.page_ids[(long long)bh * p.n_pages_cap + tok / p.page_size];
        const int off = tok % p.page_size;
        if (p.layout == MILLION_CODES_KPAGES) p.dst[(pid * p.page_size + off) * p.M + m] = (uint8_t)best_c;
        else p.dst[(pid * p.M + m) * p.page_size + off] = (uint8_t)best_c;
    }
}

// Small calls (a flush of 64 window rows: 512 (row, subspace) pairs per kv head) cannot hide the scalar-cache
// round trips of the kernel above behind other waves: 32 dependent s_load batches of ~0.6 us each.  This variant
// puts the centroid row into LDS with ONE vector round trip (converted to fp32 on the way) and reads it back with
// wave-uniform (broadcast) ds_reads that pipeline; one subspace per workgroup, a quarter of the centroids per wave.
template <int DM>
__global__ __launch_bounds__(kEncBlock) void pq_encode_small_kernel(EncParams p) {
    encode_small_body<DM>(p, blockIdx.y, blockIdx.z, blockIdx.x);
}

// ---- one launch per flush (reference flush_to_pages, paged_pq_utils.py:130-210: encode the oldest page of K rows, of
// V rows, then move the window).
//
// History.  Round 2: one 256-thread workgroup per (head, side, subspace), 1024 workgroups, a 1024-arrival lengths ticket
// behind five dependent cold round trips: 18.8 us per launch inside a decode step.  Round 3, first form: 16-wave
// workgroups (wave = head x centroid part), everything independent of the lengths requested first, early ticket: 9.5 us
// in situ - but a 16-wave workgroup needs a CU to itself, and a decode step has none to spare: every attention launch puts
// one 138-KiB workgroup on each CU, so each flush launch on the side stream (PagedPQCache.flush_ahead) held some
// attention workgroups back for its own duration (flush step = 1.21 x a plain step).
// Now the flush is built to run BESIDE the attention workgroups instead of between them: 4 waves (one per SIMD),
// <= 32 VGPRs, 5 KiB of LDS.  An attention workgroup holds 2 x 240 of a SIMD's 512 registers and 138 of the CU's 160 KiB,
// so one flush workgroup fits on every CU while attention runs, and a new attention workgroup fits while a flush
// workgroup is resident: neither waits for the other; the flush's ~2300 vector instructions per wave fill issue slots
// the latency-bound front and tail of the attention kernel leave empty.
//   * a workgroup = (64-token block, side, subspace, group of hp <= 4 kv heads): wave = (head, part of the centroid
//     range; parts = 4 / hp, 1 at nh_k >= 4); the subspace's codebook row is fetched once per workgroup;
//   * no load sits in a conditional (clamped indices instead: hipcc answers a conditional load with vmcnt(0)): the
//     codebook row goes out first, the lengths come through the scalar cache meanwhile, the window row and the page id
//     follow the lengths in ONE round trip;
//   * the ticket (4th word of the batch item's device lengths) is taken as soon as every wave of the workgroup HAS READ
//     the lengths - all it protects - so its fan-in (<= 256 arrivals) overlaps the centroid scan; the workgroup whose
//     ticket was last advances the lengths at its end.
// Codes are bit-identical to pq_encode_kernel / the oracle: same IEEE operations in the same order per (row, centroid),
// strict '<' over increasing c within a part, parts combined in centroid order.
constexpr int kFlushWaves = 4;
struct FlushParams {
    EncParams k, v;
    int *dev_lengths_w;      // writable alias of k.dev_lengths (null: host lengths, nothing to advance)
    int n_flush, rcap;
    int min_r;               // device lengths only: batch items whose window holds fewer rows are skipped (0: flush every item)
    int hp;                  // kv heads per workgroup (1, 2 or 4)
    int hgroups;             // ceil(nh_k / hp)
    // several layers of a cache in one launch (grid z = layer * bs + b): element strides between the layers' window
    // buffers, page tables and length rows; one layer: 0
    long long x_ls, ids_ls, len_ls;
    int n_layers;
    int advance;             // 0: encode only (encode-ahead: the lengths move later, million_lengths_advance)
};

template <int DM>
__global__ __launch_bounds__(kFlushWaves * 64) void pq_flush_kernel(FlushParams f) {
    typedef f16 hvec __attribute__((ext_vector_type(DM)));
    __shared__ __attribute__((aligned(16))) float rows[256 * DM];
    __shared__ float cand_d[kFlushWaves][64];
    __shared__ int cand_c[kFlushWaves][64];
    const int M = f.k.M;
    const int side_m = (int)blockIdx.y / f.hgroups, hgrp = (int)blockIdx.y % f.hgroups;
    const bool vside = side_m >= M;                              // workgroup-uniform
    const EncParams &p = vside ? f.v : f.k;
    const int m = vside ? side_m - M : side_m;
    // grid z covers the (layer, request) pairs - all of them at once (a flush the attention waits for), or a slice that the
    // workgroups walk (encode-ahead: about one workgroup per CU in flight, see launch_flush)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hp = f.hp;
    const int parts = kFlushWaves / hp;                          // 1, 2 or 4
    const int hl = w % hp, part = w / hp;
    const int hk = hgrp * hp + hl;
    const bool active = hk < p.nh_k;                             // wave-uniform
    const int hkc = active ? hk : p.nh_k - 1;
    const int C = p.C;
    // (1) independent of the lengths, and the same for every layer: this subspace's codebook row, one centroid per
    //     thread (clamped, not predicated) -> LDS as fp32
    {
        const hvec cv = *(const hvec *)(p.cents + ((long long)m * C + (tid < C ? tid : C - 1)) * DM);
        if (tid < C) {
#pragma unroll
            for (int k = 0; k < DM; ++k) rows[tid * DM + k] = (float)cv[k];
        }
    }
    const int cq = (C + parts - 1) / parts;
    const int c0 = part * cq, c1 = min(c0 + cq, C);
    for (int zi = blockIdx.z; zi < f.n_layers * p.bs; zi += gridDim.z) {
        const int layer = zi / p.bs, b = zi % p.bs;
        const int *const dev_lengths = p.dev_lengths ? p.dev_lengths + layer * f.len_ls : nullptr;
        // (2) lengths (scalar cache) -> window row of this lane and the page id of the destination token, one round trip
        int tok0 = p.tok0, xrow_start = p.xrow_start;
        if (dev_lengths) {
            tok0 = dev_lengths[b * 4 + 0]; xrow_start = dev_lengths[b * 4 + 2];
            // ragged batches: only the requests whose window is full are flushed (every workgroup of item b decides
            // alike, before any barrier and before the ticket)
            if (f.min_r > 0 && dev_lengths[b * 4 + 1] < f.min_r) continue;
        }
        // device-resident values are not trusted (cf. clamp_lengths): a start outside the ring becomes 0, a destination
        // outside the page table drops the store
        if (p.xrow_mod > 0 && (unsigned)xrow_start >= (unsigned)p.xrow_mod) xrow_start = 0;
        float x[DM];
        {
            const int t = blockIdx.x * 64 + lane;
            int xrow = xrow_start + (t < p.n ? t : p.n - 1);
            if (p.xrow_mod > 0) xrow %= p.xrow_mod;
            const hvec xv = *(const hvec *)(p.x + layer * f.x_ls + b * p.xsb + hkc * p.xsh + (long long)xrow * p.xsn + m * DM);
#pragma unroll
            for (int k = 0; k < DM; ++k) x[k] = (float)xv[k];
        }
        int pid;
        {
            const int page = (tok0 + (int)blockIdx.x * 64 + lane) / p.page_size;
            const bool ok = tok0 >= 0 && page < p.n_pages_cap;
            pid = p.page_ids[layer * f.ids_ls + (b * p.nh_k + hkc) * p.n_pages_cap + (ok ? page : 0)];      // < 2^31 entries per layer (host check)
        }
        // every wave has its lengths in registers (and, first layer, the codebook row is in LDS) before the barrier lets
        // thread 0 take the ticket
        asm volatile("" :: "s"(tok0), "s"(xrow_start) : "memory");
        __syncthreads();
        // returning atomic of thread 0 as an out-of-range-predicated buffer operation of every thread (a plain atomicrmw in
        // a one-lane branch makes hipcc's atomic optimizer wait vmcnt(0) on the spot: wave 0 would start its scan a memory
        // round trip late); null lengths: an empty descriptor, every lane out of range
        int ticket;
        {
            int *const lw = f.dev_lengths_w ? f.dev_lengths_w + layer * f.len_ls + b * 4 : nullptr;
            __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)lw, 0, lw ? 16 : 0, 0x00020000);
            ticket = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rl, tid == 0 ? 12 : (1 << 20), 0, 0);
        }
        // (3) scan this wave's part of the centroids
        float best = INFINITY;
        int best_c = c0 < C ? c0 : 0;
        if (active) {
#pragma unroll DM <= 2 ? 4 : 2       // <= 32 VGPRs up to d_m = 4 (see above)
            for (int c = c0; c < c1; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < DM; ++k) {
                    const float e = x[k] - rows[c * DM + k];
                    const float sq = e * e;
                    acc = (k == 0) ? sq : acc + sq;
                }
                best_c = acc < best ? c : best_c;
                best = fminf(best, acc);
            }
        }
        if (parts > 1) {      // workgroup-uniform
            cand_d[w][lane] = best;
            cand_c[w][lane] = best_c;
            __syncthreads();
            if (part == 0) {
                // parts in centroid order, strict '<': the lowest index wins exact ties; an empty part left +inf
                for (int j = 1; j < parts; ++j) {
                    const float dj = cand_d[j * hp + hl][lane];
                    if (dj < best) { best = dj; best_c = cand_c[j * hp + hl][lane]; }
                }
            }
        }
        {
            const int t = blockIdx.x * 64 + lane;
            const int tok = tok0 + t;
            const bool tok_ok = t < p.n && tok0 >= 0 && tok / p.page_size < p.n_pages_cap;
            if (active && part == 0 && tok_ok) {
                const int off = tok % p.page_size;
                if (p.layout == MILLION_CODES_KPAGES) p.dst[((long long)pid * p.page_size + off) * p.M + m] = (uint8_t)best_c;
                else p.dst[((long long)pid * p.M + m) * p.page_size + off] = (uint8_t)best_c;
            }
        }
        // (4) the workgroup whose ticket was the last one moves the window: every workgroup had read the lengths by then
        if (f.dev_lengths_w && tid == 0) {
            const int total = (int)(gridDim.x * gridDim.y);              // workgroups that read batch item b's lengths
            if (ticket == total - 1) {
                int *dl = f.dev_lengths_w + layer * f.len_ls + b * 4;
                const int cap_tok = p.n_pages_cap * p.page_size;
                int T = dl[0], r = dl[1], st = dl[2];
                T = T < 0 ? 0 : T;
                T = T + f.n_flush > cap_tok ? cap_tok : T + f.n_flush;
                r = r - f.n_flush < 0 ? 0 : r - f.n_flush;
                st = (unsigned)st < (unsigned)f.rcap ? st : 0;
                dl[0] = T;
                dl[1] = r;
                dl[2] = (st + f.n_flush) % f.rcap;
                __hip_atomic_store(dl + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int launch_flush(const EncParams &k, const EncParams &v, int *dev_lengths_w, int rcap, int min_r, const FlushLayers &ly,
                 hipStream_t s) {
    if (k.n <= 0 || k.bs * k.nh_k <= 0 || ly.n_layers <= 0) return MILLION_OK;
    if (k.C > 256) { set_error("flush: uint8 codes only (C=%d)", k.C); return MILLION_ERR_SHAPE; }
    if ((long long)ly.n_layers * k.bs > 65535) { set_error("flush: %d layers x %d requests exceed the grid", ly.n_layers, k.bs); return MILLION_ERR_SHAPE; }
    FlushParams f;
    f.k = k; f.v = v; f.n_flush = k.n; f.rcap = rcap; f.min_r = k.dev_lengths ? min_r : 0;
    f.dev_lengths_w = ly.advance ? dev_lengths_w : nullptr;      // encode-ahead: nothing to advance, no ticket
    f.x_ls = ly.x_ls; f.ids_ls = ly.ids_ls; f.len_ls = ly.len_ls; f.advance = ly.advance; f.n_layers = ly.n_layers;
    // (layer, request) pairs in flight.  A flush the attention waits for (advance): all of them.  Encode-ahead: as many as
    // keep about ONE workgroup per CU in flight - the most that fits beside the attention workgroups (a second flush
    // workgroup on a CU would hold the registers the next attention workgroup needs, and the attention launch would wait
    // for it); the workgroups walk the remaining pairs.
    const int pairs = ly.n_layers * k.bs;
    // kv heads per workgroup: 4 (one per wave), 2 or 1 - halved while the grid would leave CUs idle (the waves then split
    // the centroid range instead)
    const int tblocks = (k.n + 63) / 64;
    int hp = k.nh_k >= 4 ? 4 : k.nh_k >= 2 ? 2 : 1;
    const int cus = device_cus();
    while (hp > 1 && (long long)tblocks * 2 * k.M * ((k.nh_k + hp - 1) / hp) * pairs < cus) hp /= 2;
    f.hp = hp;
    f.hgroups = (k.nh_k + hp - 1) / hp;
    int gz = pairs;
    if (!ly.advance) {
        const long long per_pair = (long long)tblocks * 2 * k.M * f.hgroups;
        gz = (int)(cus / per_pair);
        gz = gz < 1 ? 1 : gz > pairs ? pairs : gz;
    }
    const dim3 grid(tblocks, 2 * k.M * f.hgroups, gz);
    switch (k.dm) {
        case 1: hipLaunchKernelGGL((pq_flush_kernel<1>), grid, dim3(kFlushWaves * 64), 0, s, f); break;
        case 2: hipLaunchKernelGGL((pq_flush_kernel<2>), grid, dim3(kFlushWaves * 64), 0, s, f); break;
        case 4: hipLaunchKernelGGL((pq_flush_kernel<4>), grid, dim3(kFlushWaves * 64), 0, s, f); break;
        case 8: hipLaunchKernelGGL((pq_flush_kernel<8>), grid, dim3(kFlushWaves * 64), 0, s, f); break;
        default: set_error("flush: d/M=%d unsupported (1,2,4,8)", k.dm); return MILLION_ERR_SHAPE;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("flush launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

template <int DM>
static void launch_dm(const EncParams &p, dim3 grid, hipStream_t s) {
    if (p.C > 256) {      // uint16 codes: the raw fp16 codebook through the scalar cache (any C)
        hipLaunchKernelGGL((pq_encode_kernel<DM, false, uint16_t>), grid, dim3(kEncBlock), 0, s, p);
        return;
    }
    // fewer than ~one wave per SIMD with 4 subspaces per wave: latency-bound, take the LDS variant
    const long long waves4 = (long long)grid.x * grid.y * (kEncBlock / 64) * grid.z;
    if (waves4 < 1024) {
        dim3 g1(grid.x, p.M, grid.z);
        hipLaunchKernelGGL((pq_encode_small_kernel<DM>), g1, dim3(kEncBlock), 0, s, p);
        return;
    }
    if (p.cents32) hipLaunchKernelGGL((pq_encode_kernel<DM, true, uint8_t>), grid, dim3(kEncBlock), 0, s, p);
    else hipLaunchKernelGGL((pq_encode_kernel<DM, false, uint8_t>), grid, dim3(kEncBlock), 0, s, p);
}

int launch_encode(const EncParams &p, hipStream_t s) {
    if (p.n <= 0 || p.bs * p.nh_k <= 0) return MILLION_OK;
    const int groups = (p.M + kEncSub - 1) / kEncSub;
    dim3 grid((p.n + 63) / 64, (groups + kEncBlock / 64 - 1) / (kEncBlock / 64), p.bs * p.nh_k);
    switch (p.dm) {
        case 1: launch_dm<1>(p, grid, s); break;
        case 2: launch_dm<2>(p, grid, s); break;
        case 4: launch_dm<4>(p, grid, s); break;
        case 8: launch_dm<8>(p, grid, s); break;
        default: set_error("encode: d/M=%d unsupported (1,2,4,8)", p.dm); return MILLION_ERR_SHAPE;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("encode launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

// ---- PQ decode: out[row, m*dm + k] = cents[m, codes[row, m], k] (sa_decode_4d, pq_utils.py:501-540) -------------
// The codebook (<= 64 KiB) is staged in LDS once per workgroup; thread = (row, subspace): consecutive threads read
// consecutive code bytes and write consecutive dm-half groups.  Grid-stride over row tiles.
template <int DM>
__global__ __launch_bounds__(256) void pq_decode_kernel(const uint8_t *__restrict__ codes, const f16 *__restrict__ cents,
                                                        f16 *__restrict__ out, long long n_rows, int M, int C) {
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    typedef struct { f16 v[DM]; } __attribute__((aligned(2 * DM))) Entry;
    Entry *tab = (Entry *)dec_smem;
    const int n_ent = M * C;
    for (int i = threadIdx.x; i < n_ent; i += 256) tab[i] = ((const Entry *)cents)[i];
    __syncthreads();
    const long long total = n_rows * M;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int m = (int)(idx % M);
        const unsigned c = codes[idx];
        ((Entry *)out)[idx] = tab[m * C + (c < (unsigned)C ? c : 0u)];
    }
}

// Codebooks that do not fit the 64 KiB LDS stage, and uint16 codes (C > 256): the same gather straight from the global
// codebook (L2 / L1 resident).
template <int DM, typename CodeT>
__global__ __launch_bounds__(256) void pq_decode_global_kernel(const CodeT *__restrict__ codes, const f16 *__restrict__ cents,
                                                               f16 *__restrict__ out, long long n_rows, int M, int C) {
    typedef struct { f16 v[DM]; } __attribute__((aligned(2 * DM))) Entry;
    const Entry *tab = (const Entry *)cents;
    const long long total = n_rows * M;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int m = (int)(idx % M);
        const unsigned c = codes[idx];
        ((Entry *)out)[idx] = tab[(long long)m * C + (c < (unsigned)C ? c : 0u)];
    }
}

template <int DM>
static void launch_decode_dm(const void *codes, const f16 *cents, f16 *out, long long n_rows, int M, int C, unsigned blocks,
                             hipStream_t s) {
    const size_t lds = (size_t)M * C * DM * sizeof(f16);
    if (C > 256)
        hipLaunchKernelGGL((pq_decode_global_kernel<DM, uint16_t>), dim3(blocks), dim3(256), 0, s, (const uint16_t *)codes, cents, out, n_rows, M, C);
    else if (lds > 64 * 1024)
        hipLaunchKernelGGL((pq_decode_global_kernel<DM, uint8_t>), dim3(blocks), dim3(256), 0, s, (const uint8_t *)codes, cents, out, n_rows, M, C);
    else
        hipLaunchKernelGGL(pq_decode_kernel<DM>, dim3(blocks), dim3(256), lds, s, (const uint8_t *)codes, cents, out, n_rows, M, C);
}

// codes: uint8 for C <= 256, uint16 above (reference nbits2dtype, pq_utils.py:542-552)
int launch_decode(const void *codes, const f16 *cents, f16 *out, long long n_rows, int M, int C, int dm, hipStream_t s) {
    if (n_rows <= 0) return MILLION_OK;
    long long blocks = (n_rows * M + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    switch (dm) {
        case 1: launch_decode_dm<1>(codes, cents, out, n_rows, M, C, (unsigned)blocks, s); break;
        case 2: launch_decode_dm<2>(codes, cents, out, n_rows, M, C, (unsigned)blocks, s); break;
        case 4: launch_decode_dm<4>(codes, cents, out, n_rows, M, C, (unsigned)blocks, s); break;
        case 8: launch_decode_dm<8>(codes, cents, out, n_rows, M, C, (unsigned)blocks, s); break;
        default: set_error("decode: d/M=%d unsupported (1,2,4,8)", dm); return MILLION_ERR_SHAPE;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("decode launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million
