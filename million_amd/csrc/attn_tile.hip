// attn_tile.hip — fused decode attention over PQ codes for the shapes the streaming kernel does not take:
// d = 64 with M in {16, 32, 64}, d = 128 with M = 16, and whatever launch_attn_mfma hands back (attn_tile_supported),
// gfx950 / CDNA4.
//
// Same job and same single launch as attn_mfma.hip (reference: LUT matmul + flash_decoding_split_kernel +
// flash_decoding_residual_kernel + flash_decoding_reduce_kernel, Interface.template.cu:26-120, Kernel.cuh:11-166,
// 1038-1270), written once for every sub-vector width d_m in {1, 2, 4, 8}:
//   * both codebooks (row image [m][c][d_m], fp16) sit in LDS for the whole kernel;
//   * a wave walks 16-token tiles on its own (no workgroup barrier inside the loop).  Per tile it
//       1. dequantises the K codes into a fp16 tile K^[16 tokens][d] in LDS (lane = (token, quarter of the code row):
//          one codebook entry per code byte, written back as 16-byte pieces),
//       2. scores: v_mfma_f32_16x16x32_f16, A = K^ rows (one ds_read_b128 per 32 dims), B = the query heads,
//       3. online softmax in the exp2 domain, fp32 (head = lane column; the probabilities are already laid out as the
//          B operand of step 5),
//       4. dequantises the V codes into the transposed tile V^T[d][16 tokens] over the same LDS bytes (lane = (pair of
//          dims, 8 tokens): eight 4-byte entry reads, eight v_perm to put the tokens of one dim side by side),
//       5. values: v_mfma_f32_16x16x16_f16, A = V^T rows (dims), B = P -> O^T[dim][head]: the running rescale is
//          lane-local;
//   * the residual window goes through the same five steps with the tile filled from the fp16 rows instead of the
//     codebook (one extra workgroup per (b, kv head), as in attn_generic.hip), the fused append included;
//   * the waves' partials are merged through LDS, the workgroups' through the workspace by the last arriver
//     (common.h: publish_and_merge), exactly as in the other two kernels.
// Code bytes are read once per kv head (all G = nh / nh_k query heads share a workgroup) straight into registers,
// kPF tiles ahead; page ids 2 kPF tiles ahead.  V must be in transposed pages (the reference's 10-argument row-major call
// is transposed first, million_api.hip).
#include "common.h"

namespace million {

typedef _Float16 t8f16 __attribute__((ext_vector_type(8)));
typedef _Float16 t4f16 __attribute__((ext_vector_type(4)));
typedef float t4f32 __attribute__((ext_vector_type(4)));
typedef unsigned t4u __attribute__((ext_vector_type(4)));

// Waves per workgroup (template parameter TW): the two codebooks (d*C*2 bytes each) decide.  d = 128: 128 KiB of tables
// leave room for four 6-KiB wave tiles (one workgroup of 4 waves per CU).  d = 64: 64 KiB of tables: either two
// workgroups of 4 waves per CU (best throughput: 57 vs 62 us at 4 requests x 32K, M = 32) or one workgroup of 16 waves
// (fewer tiles per wave, shorter launch when there is little work per CU: 23.9 vs 26.5 us at 1 request).
constexpr int kTT = 16;    // tokens per wave tile

template <int D>
struct TileGeom {
    static constexpr int kKRow = D * 2 + 16;                    // K^ row stride (bytes): 16 tokens x (d halves + pad)
    static constexpr int kVRow = kTT * 2 + 16;                  // V^T row stride (bytes): d rows x (16 halves + pad)
    static constexpr int kTile = (kTT * kKRow > D * kVRow) ? kTT * kKRow : D * kVRow;
};

static size_t tile_lds_bytes(int d, int C, int slot_floats, int waves) {
    const size_t tile = d == 128 ? TileGeom<128>::kTile : TileGeom<64>::kTile;
    return 2 * (size_t)d * C * 2 + waves * tile + (size_t)slot_floats * 4 + 16;
}

__device__ __forceinline__ float fast_exp2_tile(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32: exp2(-inf) = 0

__device__ __forceinline__ void lds_phase() {      // LDS writes of this wave before, LDS reads of this wave after
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

template <int D, int DM, int TW>
__global__ __launch_bounds__(TW * 64, TW == 4 ? 2 : 1) void attn_tile_kernel(AttnParams p) {
    constexpr int kTW = TW;
    constexpr int M = D / DM;
    constexpr int NS = D / 32;                 // score stages (32 dims each)
    constexpr int NC = D / 16;                 // output tiles of 16 dims
    constexpr int KD = M / 16;                 // K code dwords per lane and tile: lane (token, quarter) owns M/4 bytes
    constexpr int NTASK = D / 64;              // V tasks per lane and tile: (d/2 dim pairs) x (2 token octets) / 64
    constexpr int VR = DM == 1 ? 2 : 1;        // V page rows a task reads (a pair of dims spans two subspaces at d_m = 1)
    constexpr int kKRow = TileGeom<D>::kKRow, kVRow = TileGeom<D>::kVRow, kTile = TileGeom<D>::kTile;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, c16 = lane & 15;
    const int slot = blockIdx.x, bh = blockIdx.y;
    const int b = bh / p.nh_k, hk = bh % p.nh_k;
    const int G = p.G, C = p.C;
    const unsigned cmask = (unsigned)C - 1u;   // C is 128 or 256: a stray byte past T still indexes inside its table row
    int T, r, rstart;
    load_lengths(p, b, T, r, rstart);
    const int r_old = r;
    if (p.k_new) r += 1;                       // fused append: the new token is window row r_old

    const int tab_bytes = D * C * 2;
    char *tk = smem, *tv = smem + tab_bytes;
    char *tile = smem + 2 * tab_bytes + wave * kTile;
    float *part = (float *)(smem + 2 * tab_bytes + kTW * kTile);
    int *flag = (int *)(part + p.slot_floats);

    const bool is_resid = slot == p.nsplit;
    const int t_begin = is_resid ? 0 : min(slot * p.split_len, T);
    const int t_end = is_resid ? r : min(t_begin + p.split_len, T);
    const int n_tiles = (t_end - t_begin + kTT - 1) / kTT;

    if (p.k_new && is_resid && tid < D) {      // fused append: park the new row in the window (read back from k_new below)
        const long long o = b * p.res_sb + hk * p.res_sh + (long long)((rstart + r_old) % p.rcap) * D + tid;
        p.k_res_w[o] = p.k_new[(long long)bh * D + tid];
        p.v_res_w[o] = p.v_new[(long long)bh * D + tid];
    }
    if (!is_resid && n_tiles > 0) {
        for (int i = tid * 16; i < tab_bytes; i += kTW * 64 * 16) {
            *(t4u *)(tk + i) = *(const t4u *)((const char *)p.k_tab + i);
            *(t4u *)(tv + i) = *(const t4u *)((const char *)p.v_tab + i);
        }
    }
    t8f16 qb[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        t4u z = {0u, 0u, 0u, 0u};
        if (c16 < G) z = *(const t4u *)(p.q + ((long long)b * p.nh + head0(p, hk) + c16) * D + 32 * s + 8 * q4);
        qb[s] = __builtin_bit_cast(t8f16, z);
    }
    __syncthreads();

    float m_run = -INFINITY, l_run = 0.f;
    t4f32 O[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) O[n] = t4f32{0.f, 0.f, 0.f, 0.f};

    // ---- steps 2-3: scores of the K^ tile, online softmax; returns the probabilities (B operand of step 5) ----
    auto scores = [&](int n_valid) -> t4f16 {
        t4f32 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const t4u a = *(const t4u *)(tile + c16 * kKRow + (32 * s + 8 * q4) * 2);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t8f16, a), qb[s], acc, 0, 0, 0);
        }
        float sc[4], mx = -INFINITY;
#pragma unroll
        for (int v = 0; v < 4; ++v) {          // acc[v] = S[token 4*q4 + v][head c16]
            sc[v] = (4 * q4 + v < n_valid) ? acc[v] * p.scale_log2e : -INFINITY;
            mx = fmaxf(mx, sc[v]);
        }
        // m_run is the softmax reference of head c16, moved (with the cross-lane maximum and the rescale of O, l) only
        // when a score exceeds it by more than 2^8: the common tile needs no reduction (attn_mfma.hip: softmax_online_raw)
        if (__builtin_amdgcn_ballot_w64(mx > m_run + 8.0f)) {      // also the first tile: m_run = -inf, mx finite
            const float m_new = fmaxf(m_run, rows_max(mx));        // over the four lane rows: all 16 tokens of head c16
            const float alpha = fast_exp2_tile(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int n = 0; n < NC; ++n) O[n] *= alpha;        // O^T[dim][head c16]: this lane's own head
            }
        }
        float pe[4], ls = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) { pe[v] = fast_exp2_tile(sc[v] - m_run); ls += pe[v]; }
        l_run += ls;                           // per-lane partial sum; the four lane rows are added once, at the end
        return t4f16{(_Float16)pe[0], (_Float16)pe[1], (_Float16)pe[2], (_Float16)pe[3]};
    };
    // ---- step 5: O^T += V^T P ----
    auto values = [&](t4f16 P) {
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const v2u a = *(const v2u *)(tile + (16 * n + c16) * kVRow + (4 * q4) * 2);
            O[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(t4f16, a), P, O[n], 0, 0, 0);
        }
    };
    // eight (d0, d1) pairs of consecutive tokens -> the two V^T rows of that dim pair
    auto store_vt = [&](int dp, int oct, const unsigned (&e)[8]) {
        t4u lo, hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = __builtin_amdgcn_perm(e[2 * i + 1], e[2 * i], 0x05040100u);
            hi[i] = __builtin_amdgcn_perm(e[2 * i + 1], e[2 * i], 0x07060302u);
        }
        *(t4u *)(tile + (2 * dp) * kVRow + oct * 16) = lo;
        *(t4u *)(tile + (2 * dp + 1) * kVRow + oct * 16) = hi;
    };

#ifdef MILLION_TILE_PROF
    unsigned long long prof_t[5] = {0, 0, 0, 0, 0}, prof_acc[5] = {0, 0, 0, 0, 0};
    const unsigned long long prof_start = __builtin_amdgcn_s_memtime();
#define TILE_PROF_T(i) do { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); prof_t[i] = t_; } while (0)
#define TILE_PROF_ACC() do { for (int i_ = 0; i_ < 4; ++i_) prof_acc[i_] += prof_t[i_ + 1] - prof_t[i_]; prof_acc[4] += 1; } while (0)
#else
#define TILE_PROF_T(i) do { } while (0)
#define TILE_PROF_ACC() do { } while (0)
#endif
    if (!is_resid) {
        // ================= code tiles =================
        const int tl = c16, mq = q4;           // step 1 lane roles: token, quarter of the code row
        const int ps_mask = p.page_size - 1;
        struct Codes { unsigned k[KD]; unsigned v[NTASK][VR][2]; };
        struct Pids { long long k, v; };
        auto tile_t0 = [&](int j) {            // first token of this wave's j-th tile (clamped: re-request, never past the split)
            int ti = wave + kTW * j;
            if (ti > n_tiles - 1) ti = n_tiles - 1;
            return t_begin + kTT * ti;
        };
        // Page ids without a branch around the load (hipcc answers a load in a conditional with s_waitcnt vmcnt(0),
        // which would drain the code prefetch every tile): one 4-byte load of the low dword of the id (ids are < 2^31)
        // from a table chosen before the loop; a side without a table reads a dummy word and ignores it.
        const int *kid_base = p.k_paged ? (p.ids64 ? (const int *)p.k_ids64 : p.k_ids32) : (const int *)p.q;
        const int *vid_base = !p.v_identity ? (p.ids64 ? (const int *)p.v_ids64 : p.v_ids32) : (const int *)p.q;
        const int kid_stride = p.k_paged ? (p.ids64 ? 2 : 1) : 0;
        const int vid_stride = !p.v_identity ? (p.ids64 ? 2 : 1) : 0;
        auto load_pids = [&](int t0) -> Pids {
            const int idx = bh * p.n_pages_cap + (t0 >> p.ps_shift);      // < 2^31 (the host checks the page table size)
            Pids o;
            o.k = kid_base[idx * kid_stride];                               // 32-bit index arithmetic: a 64-bit multiply
            const int v = vid_base[idx * vid_stride];                       // here made hipcc tie a wait to an in-flight load
            o.v = p.v_identity ? idx : v;
#ifdef MILLION_DEBUG_CHECK_IDS
            if (p.k_paged) o.k = MILLION_CHECK_KID(p, o.k);
            if (!p.v_identity) o.v = MILLION_CHECK_VID(p, o.v);
#endif
            return o;
        };
        auto load_codes = [&](int t0, Pids id) -> Codes {
            Codes c;
            const int tk_ = min(t0 + tl, t_end - 1);
            const long long koff = p.k_paged ? ((id.k << p.ps_shift) + (tk_ & ps_mask)) * M
                                             : b * p.k_sb + hk * p.k_sh + (long long)tk_ * M;
            const uint8_t *krow = p.k_codes + koff + mq * (M / 4);
            if constexpr (KD == 1) c.k[0] = *(const unsigned *)krow;
            else if constexpr (KD == 2) { const v2u w = *(const v2u *)krow; c.k[0] = w[0]; c.k[1] = w[1]; }
            else { const t4u w = *(const t4u *)krow; c.k[0] = w[0]; c.k[1] = w[1]; c.k[2] = w[2]; c.k[3] = w[3]; }
#pragma unroll
            for (int j = 0; j < NTASK; ++j) {
                const int task = j * 64 + lane, dp = task >> 1, oct = task & 1;
                const int m0 = (2 * dp) / DM;
#pragma unroll
                for (int rr = 0; rr < VR; ++rr) {
                    const v2u w = *(const v2u *)(p.v_codes + ((id.v * M + m0 + rr) << p.ps_shift) + (t0 & ps_mask) + 8 * oct);
                    c.v[j][rr][0] = w[0]; c.v[j][rr][1] = w[1];
                }
            }
            return c;
        };
        // Lookup addresses: one v_bfe_u32 + one v_lshl_add_u32 per code byte.  The per-subspace table bases are
        // lane constants kept in registers, and the "& (C - 1)" that keeps a stray byte inside its table row is applied
        // to four bytes at once.
        constexpr unsigned kEsh = DM == 8 ? 4 : DM == 4 ? 3 : DM == 2 ? 2 : 1;      // log2(bytes per codebook entry)
        const unsigned cm4 = cmask * 0x01010101u;
        const char *kb[M / 4];
#pragma unroll
        for (int j = 0; j < M / 4; ++j) kb[j] = tk + (mq * (M / 4) + j) * C * (DM * 2);
        const char *vb[NTASK][VR];
#pragma unroll
        for (int j = 0; j < NTASK; ++j) {
            const int dp = (j * 64 + lane) >> 1;
#pragma unroll
            for (int rr = 0; rr < VR; ++rr)
                vb[j][rr] = DM == 1 ? tv + (2 * dp + rr) * C * 2 : tv + (((2 * dp) / DM) * C * DM + (2 * dp) % DM) * 2;
        }
        auto fill_k = [&](const Codes &c) {
            unsigned buf[D / 8];               // this lane's d/4 dims of its token
#pragma unroll
            for (int j = 0; j < M / 4; ++j) {
                const unsigned code = ((c.k[j >> 2] & cm4) >> (8 * (j & 3))) & 0xffu;
                const char *e = kb[j] + (code << kEsh);
                if constexpr (DM == 8) { const t4u x = *(const t4u *)e; buf[4 * j] = x[0]; buf[4 * j + 1] = x[1]; buf[4 * j + 2] = x[2]; buf[4 * j + 3] = x[3]; }
                else if constexpr (DM == 4) { const v2u x = *(const v2u *)e; buf[2 * j] = x[0]; buf[2 * j + 1] = x[1]; }
                else if constexpr (DM == 2) buf[j] = *(const unsigned *)e;
                else {
                    const unsigned x = *(const unsigned short *)e;
                    if (j & 1) buf[j >> 1] |= x << 16; else buf[j >> 1] = x;
                }
            }
            char *dst = tile + tl * kKRow + mq * (D / 2);
#pragma unroll
            for (int i = 0; i < D / 32; ++i) *(t4u *)(dst + 16 * i) = t4u{buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]};
        };
        auto fill_v = [&](const Codes &c) {
#pragma unroll
            for (int j = 0; j < NTASK; ++j) {
                const int task = j * 64 + lane, dp = task >> 1, oct = task & 1;
                unsigned e[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned c0 = ((c.v[j][0][t >> 2] & cm4) >> (8 * (t & 3))) & 0xffu;
                    if constexpr (DM == 1) {
                        const unsigned c1 = ((c.v[j][1][t >> 2] & cm4) >> (8 * (t & 3))) & 0xffu;
                        const unsigned x0 = *(const unsigned short *)(vb[j][0] + (c0 << 1));
                        const unsigned x1 = *(const unsigned short *)(vb[j][1] + (c1 << 1));
                        e[t] = x0 | (x1 << 16);
                    } else {
                        e[t] = *(const unsigned *)(vb[j][0] + (c0 << kEsh));
                    }
                }
                store_vt(dp, oct, e);
            }
        };

        if (n_tiles > 0) {
            const int my_tiles = wave < n_tiles ? (n_tiles - wave + kTW - 1) / kTW : 0;
            // Code bytes kPF tiles ahead, page ids 2 kPF tiles ahead, in a register ring with compile-time slots (the loop
            // is unrolled by the ring size; a tile is 5-9 registers).  One tile ahead is not enough: an iteration is
            // ~0.6 us of work and a load under traffic takes 1-2 us, so the loop ran at the load latency.
            constexpr int kPF = TW == 4 ? 4 : 2;       // (sixteen waves per CU hide latency by themselves; <= 128 VGPRs)
            Codes ring[kPF];
            Pids idr[kPF];
#pragma unroll
            for (int k = 0; k < kPF; ++k) idr[k] = load_pids(tile_t0(k));
#pragma unroll
            for (int k = 0; k < kPF; ++k) {
                ring[k] = load_codes(tile_t0(k), idr[k]);
                idr[k] = load_pids(tile_t0(k + kPF));
            }
            for (int j0 = 0; j0 < my_tiles; j0 += kPF) {
#pragma unroll
                for (int k = 0; k < kPF; ++k) {
                    const int j = j0 + k;
                    const Codes cur = ring[k];
                    ring[k] = load_codes(tile_t0(j + kPF), idr[k]);
                    idr[k] = load_pids(tile_t0(j + 2 * kPF));
                    if (j < my_tiles) {                                   // wave-uniform; no global load inside
                        const int t0 = tile_t0(j);
                        TILE_PROF_T(0);
                        fill_k(cur);
                        lds_phase();
                        TILE_PROF_T(1);
                        const t4f16 P = scores(min(kTT, t_end - t0));
                        lds_phase();
                        TILE_PROF_T(2);
                        fill_v(cur);
                        lds_phase();
                        TILE_PROF_T(3);
                        values(P);
                        lds_phase();
                        TILE_PROF_T(4);
                        TILE_PROF_ACC();
                    }
                }
            }
        }
    } else {
        // ================= residual window: the same steps, tiles filled from the fp16 rows =================
        const int tl = c16, mq = q4;
        const f16 *kres = p.k_res + b * p.res_sb + hk * p.res_sh;
        const f16 *vres = p.v_res + b * p.res_sb + hk * p.res_sh;
        auto row_of = [&](int i, const f16 *base, const f16 *fresh) -> const f16 * {      // window row i (clamped to a valid one)
            if (i > r - 1) i = r - 1;
            if (fresh && i == r_old) return fresh + (long long)bh * D;
            return base + (long long)((rstart + i) % p.rcap) * D;
        };
        for (int ti = wave; ti < n_tiles; ti += kTW) {
            const int t0 = kTT * ti;
            {
                const f16 *src = row_of(t0 + tl, kres, p.k_new) + mq * (D / 4);
                char *dst = tile + tl * kKRow + mq * (D / 2);
#pragma unroll
                for (int i = 0; i < D / 32; ++i) *(t4u *)(dst + 16 * i) = *(const t4u *)((const char *)src + 16 * i);
            }
            lds_phase();
            const t4f16 P = scores(min(kTT, t_end - t0));
            lds_phase();
#pragma unroll
            for (int j = 0; j < NTASK; ++j) {
                const int task = j * 64 + lane, dp = task >> 1, oct = task & 1;
                unsigned e[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = *(const unsigned *)(row_of(t0 + 8 * oct + t, vres, p.v_new) + 2 * dp);
                store_vt(dp, oct, e);
            }
            lds_phase();
            values(P);
            lds_phase();
        }
    }

#ifdef MILLION_TILE_PROF
    if (p.dbg && lane == 0) {
        unsigned long long *o = p.dbg + (((long long)blockIdx.y * gridDim.x + blockIdx.x) * kTW + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = prof_acc[i];
        o[5] = __builtin_amdgcn_s_memtime() - prof_start;
    }
#endif
    // ---- merge the waves' partials through LDS, then hand the workgroup's partial over ----
    __syncthreads();
    {
        const int wf = G * D + 2 * G;                         // floats per wave
        const float l_sum = rows_sum(l_run);
        float *ws = (float *)(smem + 2 * tab_bytes) + wave * wf;
        if (c16 < G) {
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int v = 0; v < 4; ++v) ws[c16 * D + 16 * n + 4 * q4 + v] = O[n][v];
            if (q4 == 0) { ws[G * D + c16] = m_run; ws[G * D + G + c16] = l_sum; }
        }
        __syncthreads();
        const float *w0 = (const float *)(smem + 2 * tab_bytes);
        for (int e = tid; e < wf; e += kTW * 64) {            // e: O[g][dim], then m[g], then l[g]
            const int g = e < G * D ? e / D : (e - G * D) % G;
            float mm = -INFINITY;
#pragma unroll
            for (int w = 0; w < kTW; ++w) mm = fmaxf(mm, w0[w * wf + G * D + g]);
            const float ms = mm > -INFINITY ? mm : 0.f;
            float o = mm;
            if (e < G * D || e >= G * D + G) {
                o = 0.f;
#pragma unroll
                for (int w = 0; w < kTW; ++w) o += w0[w * wf + e] * exp2f(w0[w * wf + G * D + g] - ms);
            }
            part[e] = o;
        }
        __syncthreads();
    }
    publish_and_merge(p, b, hk, slot, part, (float *)smem, flag);      // the codebooks are dead: merge scratch
}

bool attn_tile_supported(const AttnParams &p) {
    const bool shape = (p.d == 128 || p.d == 64) && (p.M == 16 || p.M == 32 || p.M == 64) && (p.C == 128 || p.C == 256) &&
                       p.G <= kMaxG;
    if (!shape || !p.v_paged) return false;
    return tile_lds_bytes(p.d, p.C, p.slot_floats, 4) <= 160 * 1024;
}
bool attn_tile_shape_ok(const AttnParams &p) {
    AttnParams q = p;
    q.v_paged = 1;
    return attn_tile_supported(q);
}

int launch_attn_tile(const AttnParams &p_in, hipStream_t s) {
    AttnParams p = p_in;
    // d = 64: sixteen waves in one workgroup per CU while a CU has at most 128 tiles to walk, else two 4-wave workgroups
    const int bh = p.bs * p.nh_k;
    const int cus = device_cus();
    const bool wide = p.d == 64 && (long long)p.T * bh <= 128ll * kTT * cus;
    const int waves = wide ? 16 : 4;
    // split policy: one workgroup per CU (two of the 4-wave ones at d = 64); a split is a multiple of 64 tokens and at
    // least 256 tokens long
    const int target = cus * (p.d == 64 && !wide ? 2 : 1);
    int ns = (target + bh - 1) / bh;
    if (ns > kMaxSplits) ns = kMaxSplits;
    int by_len = (p.T + 255) / 256;
    if (by_len < 1) by_len = 1;
    if (ns > by_len) ns = by_len;
    int len = (p.T + ns - 1) / ns;
    len = (len + 63) / 64 * 64;
    if (len < 64) len = 64;
    ns = p.T > 0 ? (p.T + len - 1) / len : 1;
    p.nsplit = ns;
    p.split_len = len;
    p.nslots = ns + 1;
    const size_t lds = tile_lds_bytes(p.d, p.C, p.slot_floats, waves);
    if (device_once(2)) {
#define TILE_ATTR(D_, DM_, TW_) (void)hipFuncSetAttribute((const void *)attn_tile_kernel<D_, DM_, TW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        TILE_ATTR(128, 8, 4); TILE_ATTR(128, 4, 4); TILE_ATTR(128, 2, 4);
        TILE_ATTR(64, 4, 4); TILE_ATTR(64, 2, 4); TILE_ATTR(64, 1, 4);
        TILE_ATTR(64, 4, 16); TILE_ATTR(64, 2, 16); TILE_ATTR(64, 1, 16);
#undef TILE_ATTR
    }
    const dim3 grid(p.nslots, bh), block(waves * 64);
#define TILE_LAUNCH(D_, DM_, TW_) hipLaunchKernelGGL((attn_tile_kernel<D_, DM_, TW_>), grid, block, lds, s, p)
    const int key = p.d * 16 + p.dm;
    switch (key) {
    case 128 * 16 + 8: TILE_LAUNCH(128, 8, 4); break;
    case 128 * 16 + 4: TILE_LAUNCH(128, 4, 4); break;
    case 128 * 16 + 2: TILE_LAUNCH(128, 2, 4); break;
    case 64 * 16 + 4: if (wide) TILE_LAUNCH(64, 4, 16); else TILE_LAUNCH(64, 4, 4); break;
    case 64 * 16 + 2: if (wide) TILE_LAUNCH(64, 2, 16); else TILE_LAUNCH(64, 2, 4); break;
    case 64 * 16 + 1: if (wide) TILE_LAUNCH(64, 1, 16); else TILE_LAUNCH(64, 1, 4); break;
    default: set_error("attn_tile: d=%d d_m=%d", p.d, p.dm); return MILLION_ERR_SHAPE;
    }
#undef TILE_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("attn_tile launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million
