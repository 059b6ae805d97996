// attn_tile.hip — fused decode attention over PQ codes for the shapes the streaming kernel does not take:
// d = 64 with M in {16, 32, 64}, d = 128 with M = 16, and whatever launch_attn_mfma hands back (attn_tile_supported),
// gfx950 / CDNA4.
//
// Same job and same single launch as attn_mfma.hip (reference: LUT matmul + flash_decoding_split_kernel +
// flash_decoding_residual_kernel + flash_decoding_reduce_kernel, Interface.template.cu:26-120, Kernel.cuh:11-166,
// 1038-1270), written once for every sub-vector width d_m in {1, 2, 4, 8}:
//   * both codebooks (row image [m][c][d_m], fp16) sit in LDS for the whole kernel;
//   * a wave walks 16-token tiles on its own (no workgroup barrier inside the loop).  Round 4: a code tile no longer goes
//     through LDS (rounds 1-3 dequantised K^ and V^T into an LDS tile and read them back: four dependent LDS round trips per
//     tile, 3-15 % of the HBM peak) - the centroids go from the LDS-resident codebooks STRAIGHT into MFMA operands, as in
//     attn_mfma.hip:
//       1. scores: lane (quarter q4 of the dims, token): the lane's M/4 code bytes are its quarter of the code row; stage s
//          (8 dims of the quarter) gathers 8/d_m entries of the K row image - one ds_read_b128 / two b64 / four b32 / eight
//          u16 - which ARE the A operand of v_mfma_f32_16x16x32_f16 (B = the query heads, same dim order);
//       2. online softmax in the exp2 domain, fp32 (head = lane column; the probabilities are already laid out as the B
//          operand of step 4);
//       3. values: lane (4 tokens q4, subspace m = 16 cg + c16): 4 gathers of the V col image per column group cg (bank =
//          subspace); for every dim e of the sub-vector the four tokens' halves are packed (2 v_perm) into the A operand of
//          v_mfma_f32_16x16x16_f16 (rows = 16 subspaces, k = 16 tokens), B = P -> tile (cg, e) of O^T[(16 cg + row) d_m + e][head]:
//          d/16 accumulator tiles as before, the running rescale is lane-local;
//     M/4 + M/4 gathers, d/8 packs and d/32 + d/16 MFMAs per tile and wave; up to 16 query heads per launch (the 16 columns);
//   * the residual window keeps the LDS tile (one extra workgroup per (b, kv head), as in attn_generic.hip: K rows -> K^ tile ->
//     scores, V rows -> transposed V^T tile -> values), the fused append included;
//   * the waves' partials are merged through LDS, the workgroups' through the workspace by the last arriver
//     (common.h: publish_and_merge), exactly as in the other two kernels.
// Code bytes are read once per kv head (all G = nh / nh_k query heads share a workgroup) straight into registers,
// kPF tiles ahead; page ids 2 kPF tiles ahead.  V must be in transposed pages (the reference's 10-argument row-major call
// is transposed first, million_api.hip).
#include "common.h"
#include "dev_switches.h"      // MILLION_TILE_PROF (per-tile stamps) is a development switch: never in the product build

namespace million {

typedef _Float16 t8f16 __attribute__((ext_vector_type(8)));
typedef _Float16 t4f16 __attribute__((ext_vector_type(4)));
typedef float t4f32 __attribute__((ext_vector_type(4)));
typedef unsigned t4u __attribute__((ext_vector_type(4)));

// Waves per workgroup (template parameter TW): see launch_attn_tile.  Only the residual-window workgroup uses LDS wave tiles.
constexpr int kTT = 16;    // tokens per wave tile

template <int D>
struct TileGeom {
    static constexpr int kKRow = D * 2 + 16;                    // K^ row stride (bytes): 16 tokens x (d halves + pad)
    static constexpr int kVRow = kTT * 2 + 16;                  // V^T row stride (bytes): d rows x (16 halves + pad)
    static constexpr int kTile = (kTT * kKRow > D * kVRow) ? kTT * kKRow : D * kVRow;
};

// LDS: [region][partial of the workgroup][flag].  region = the two codebooks (code workgroups), reused after the loop as the
// waves' merge scratch; the residual workgroup (no codebooks) keeps its K^ / V^T wave tiles there.
static size_t tile_region_bytes(int d, int C, int G, int waves) {
    const size_t tile = d == 128 ? TileGeom<128>::kTile : TileGeom<64>::kTile;
    size_t r = 2 * (size_t)d * C * 2;
    if (waves * tile > r) r = waves * tile;
    const size_t wf = (size_t)waves * (G * d + 2 * G) * 4;
    if (wf > r) r = wf;
    return (r + 15) / 16 * 16;
}
static size_t tile_lds_bytes(int d, int C, int slot_floats, int waves, int G) {
    return tile_region_bytes(d, C, G, waves) + (size_t)slot_floats * 4 + 16;
}

__device__ __forceinline__ float fast_exp2_tile(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32: exp2(-inf) = 0

__device__ __forceinline__ void lds_phase() {      // LDS writes of this wave before, LDS reads of this wave after
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

template <int D, int DM, int TW>
__global__ __launch_bounds__(TW * 64, TW / 4) void attn_tile_kernel(AttnParams p) {
    constexpr int kTW = TW;
    constexpr int M = D / DM;
    constexpr int NS = D / 32;                 // score stages (32 dims each)
    constexpr int NC = D / 16;                 // output tiles of 16 dims
    constexpr int KD = M / 16;                 // K code dwords per lane and tile: lane (token, quarter) owns M/4 bytes
    constexpr int CG = M / 16;                 // column groups of 16 subspaces: V code dwords per lane and tile (4 tokens each)
    constexpr int EW = DM >= 2 ? DM / 2 : 1;   // dwords per gathered codebook entry
    constexpr int NTASK = D / 64;              // residual window: V tasks per lane and tile: (d/2 dim pairs) x (2 token octets) / 64
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
    char *tile = smem + wave * kTile;          // residual workgroup only (it loads no codebook)
    int region = 2 * tab_bytes;
    if (kTW * kTile > region) region = kTW * kTile;
    if (kTW * (G * D + 2 * G) * 4 > region) region = kTW * (G * D + 2 * G) * 4;
    region = (region + 15) / 16 * 16;
    float *part = (float *)(smem + region);
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
            *(t4u *)(tv + i) = *(const t4u *)((const char *)p.v_tab_col + i);      // col image [c][m][d_m]: bank = subspace
        }
    }
    t8f16 qb[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        t4u z = {0u, 0u, 0u, 0u};
        // stage s of lane quarter q4 = dims q4 * D/4 + 8 s .. + 8: the dims of the lane's own code bytes (direct gathers)
        if (c16 < G) z = *(const t4u *)(p.q + ((long long)b * p.nh + head0(p, hk) + c16) * D + q4 * (D / 4) + 8 * s);
        qb[s] = __builtin_bit_cast(t8f16, z);
    }
    __syncthreads();

    float m_run = -INFINITY, l_run = 0.f;
    t4f32 O[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) O[n] = t4f32{0.f, 0.f, 0.f, 0.f};

    // ---- residual window: scores of the K^ tile in LDS ----
    auto scores_lds = [&]() -> t4f32 {
        t4f32 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const t4u a = *(const t4u *)(tile + c16 * kKRow + (q4 * (D / 4) + 8 * s) * 2);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t8f16, a), qb[s], acc, 0, 0, 0);
        }
        return acc;
    };
    auto softmax_tile = [&](t4f32 acc, int n_valid) -> t4f16 {
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
        struct Codes { unsigned k[KD]; unsigned v[CG]; };      // K: the lane's quarter of its token's code row; V: 4 tokens of subspace 16 cg + c16
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
            // V: a tile never straddles a page (16 | page_size); bytes of tokens >= T inside the page are whatever the page
            // holds: "& (C - 1)" keeps them inside the table and their probabilities are exact zeros
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
                c.v[cg] = *(const unsigned *)(p.v_codes + ((id.v * M + 16 * cg + c16) << p.ps_shift) + (t0 & ps_mask) + 4 * q4);
            return c;
        };
        // Lookup addresses: one v_bfe_u32 + one v_lshl_add_u32 per code byte.  The per-subspace table bases are lane
        // constants kept in registers, and the "& (C - 1)" that keeps a stray byte inside its table row is applied to four
        // bytes at once.
        constexpr unsigned kEsh = DM == 8 ? 4 : DM == 4 ? 3 : DM == 2 ? 2 : 1;      // log2(bytes per codebook entry)
        constexpr unsigned kVsh = D == 128 ? 8 : 7;                                  // log2(bytes per code of the V col image = 2 d)
        const unsigned cm4 = cmask * 0x01010101u;
        const char *kb[M / 4];
#pragma unroll
        for (int j = 0; j < M / 4; ++j) kb[j] = tk + (mq * (M / 4) + j) * C * (DM * 2);
        const char *vb[CG];
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) vb[cg] = tv + (16 * cg + c16) * (DM * 2);
        auto code_of = [&](unsigned w, int i) -> unsigned { return ((w & cm4) >> (8 * (i & 3))) & 0xffu; };
        // ---- step 1: scores straight from the K row image ----
        auto scores_direct = [&](const Codes &c) -> t4f32 {
            t4f32 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                t4u a;
                if constexpr (DM == 8) {
                    a = *(const t4u *)(kb[st] + (code_of(c.k[st >> 2], st) << kEsh));
                } else if constexpr (DM == 4) {
                    const v2u x = *(const v2u *)(kb[2 * st] + (code_of(c.k[(2 * st) >> 2], 2 * st) << kEsh));
                    const v2u y = *(const v2u *)(kb[2 * st + 1] + (code_of(c.k[(2 * st + 1) >> 2], 2 * st + 1) << kEsh));
                    a = t4u{x[0], x[1], y[0], y[1]};
                } else if constexpr (DM == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = *(const unsigned *)(kb[4 * st + i] + (code_of(c.k[(4 * st + i) >> 2], 4 * st + i) << kEsh));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned lo = *(const unsigned short *)(kb[8 * st + 2 * i] + (code_of(c.k[(8 * st + 2 * i) >> 2], 8 * st + 2 * i) << kEsh));
                        const unsigned hi = *(const unsigned short *)(kb[8 * st + 2 * i + 1] + (code_of(c.k[(8 * st + 2 * i + 1) >> 2], 8 * st + 2 * i + 1) << kEsh));
                        a[i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                    }
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t8f16, a), qb[st], acc, 0, 0, 0);
            }
            return acc;
        };
        // ---- step 3: the V entries of this lane's 4 tokens x CG subspaces (independent of the softmax: issued with the K gathers) ----
        struct VEnt { unsigned e[CG][4][EW]; };
        auto gather_v = [&](const Codes &c) -> VEnt {
            VEnt v;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const char *ep = vb[cg] + (code_of(c.v[cg], t) << kVsh);
                    if constexpr (DM == 8) { const t4u x = *(const t4u *)ep; v.e[cg][t][0] = x[0]; v.e[cg][t][1] = x[1]; v.e[cg][t][2] = x[2]; v.e[cg][t][3] = x[3]; }
                    else if constexpr (DM == 4) { const v2u x = *(const v2u *)ep; v.e[cg][t][0] = x[0]; v.e[cg][t][1] = x[1]; }
                    else if constexpr (DM == 2) v.e[cg][t][0] = *(const unsigned *)ep;
                    else v.e[cg][t][0] = *(const unsigned short *)ep;
                }
            return v;
        };
        // ---- step 4: O^T tile (cg, e) += V^T_e (rows = subspaces 16 cg + .., k = 16 tokens) P ----
        auto values_direct = [&](const VEnt &v, t4f16 P) {
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int e = 0; e < DM; ++e) {
                    const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    const v2u a = {__builtin_amdgcn_perm(v.e[cg][1][e >> 1], v.e[cg][0][e >> 1], sel),
                                   __builtin_amdgcn_perm(v.e[cg][3][e >> 1], v.e[cg][2][e >> 1], sel)};
                    O[cg * DM + e] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(t4f16, a), P, O[cg * DM + e], 0, 0, 0);
                }
        };

        if (n_tiles > 0) {
            const int my_tiles = wave < n_tiles ? (n_tiles - wave + kTW - 1) / kTW : 0;
            // Code bytes kPF tiles ahead, page ids 2 kPF tiles ahead, in a register ring with compile-time slots (the loop
            // is unrolled by the ring size; a tile is 5-9 registers).  One tile ahead is not enough: an iteration is
            // ~0.6 us of work and a load under traffic takes 1-2 us, so the loop ran at the load latency.
            constexpr int kPF = TW <= 8 ? 4 : 2;       // (sixteen waves per CU hide latency by themselves; <= 128 VGPRs)
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
                        const t4f32 acc = scores_direct(cur);
                        const VEnt ve = gather_v(cur);
                        TILE_PROF_T(1);
                        const t4f16 P = softmax_tile(acc, min(kTT, t_end - t0));
                        TILE_PROF_T(2);
                        values_direct(ve, P);
                        TILE_PROF_T(3);
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
            const t4f16 P = softmax_tile(scores_lds(), min(kTT, t_end - t0));
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
        float *ws = (float *)smem + wave * wf;                 // the codebooks / wave tiles are dead (barrier above)
        if (c16 < G) {
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    // code workgroups: tile n = (column group n / d_m, dim e = n % d_m), row = subspace 16 cg + 4 q4 + v;
                    // residual workgroup: tile n = dims 16 n .. 16 n + 15
                    const int dim = is_resid ? 16 * n + 4 * q4 + v : (16 * (n / DM) + 4 * q4 + v) * DM + n % DM;
                    ws[c16 * D + dim] = O[n][v];
                }
            if (q4 == 0) { ws[G * D + c16] = m_run; ws[G * D + G + c16] = l_sum; }
        }
        __syncthreads();
        const float *w0 = (const float *)smem;
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
    // merge scratch of the last arriver: the waves' partials above are dead once `part` is complete (barrier above)
    publish_and_merge(p, b, hk, slot, part, (float *)smem, flag);
}

bool attn_tile_supported(const AttnParams &p) {
    const bool shape = (p.d == 128 || p.d == 64) && (p.M == 16 || p.M == 32 || p.M == 64) && (p.C == 128 || p.C == 256) &&
                       p.G <= kMaxGMfma;      // the 16 columns of the score tile: up to 16 query heads per kv head and launch (round 4)
    if (!shape || !p.v_paged) return false;
    return tile_lds_bytes(p.d, p.C, p.slot_floats, 16, p.G) <= 160 * 1024;
}
bool attn_tile_shape_ok(const AttnParams &p) {
    AttnParams q = p;
    q.v_paged = 1;
    return attn_tile_supported(q);
}

int launch_attn_tile(const AttnParams &p_in, hipStream_t s) {
    AttnParams p = p_in;
    // Waves per workgroup (round 4: a code tile needs no LDS of its own any more, so the codebooks alone decide how many
    // workgroups fit - one per CU at d = 128 - and the registers how many waves: 16 where the kernel stays within 128
    // registers (d = 64; d = 128 with 16-byte entries), 8 otherwise).  More waves = fewer 16-token tiles per wave and more
    // LDS round trips in flight.
    const int bh = p.bs * p.nh_k;
    const int cus = device_cus();
    const int waves = (p.d == 64 || p.dm == 8) ? 16 : 8;
    // split policy: one workgroup per CU - the residual-window workgroup of every (b, kv head) included (rounds 2-3 sized the
    // splits for all CUs and launched bh more workgroups than the chip holds at d = 128: the last ones started when the first
    // finished); a split is a multiple of 64 tokens and at least 256 tokens long
    int ns = cus > 2 * bh ? (cus - bh) / bh : (cus + bh - 1) / bh;
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
    const size_t lds = tile_lds_bytes(p.d, p.C, p.slot_floats, waves, p.G);
    if (device_once(2)) {
#define TILE_ATTR(D_, DM_, TW_) (void)hipFuncSetAttribute((const void *)attn_tile_kernel<D_, DM_, TW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        TILE_ATTR(128, 8, 16); TILE_ATTR(128, 4, 8); TILE_ATTR(128, 2, 8);
        TILE_ATTR(64, 4, 16); TILE_ATTR(64, 2, 16); TILE_ATTR(64, 1, 16);
#undef TILE_ATTR
    }
    const dim3 grid(p.nslots, bh), block(waves * 64);
#define TILE_LAUNCH(D_, DM_, TW_) hipLaunchKernelGGL((attn_tile_kernel<D_, DM_, TW_>), grid, block, lds, s, p)
    const int key = p.d * 16 + p.dm;
    switch (key) {
    case 128 * 16 + 8: TILE_LAUNCH(128, 8, 16); break;
    case 128 * 16 + 4: TILE_LAUNCH(128, 4, 8); break;
    case 128 * 16 + 2: TILE_LAUNCH(128, 2, 8); break;
    case 64 * 16 + 4: TILE_LAUNCH(64, 4, 16); break;
    case 64 * 16 + 2: TILE_LAUNCH(64, 2, 16); break;
    case 64 * 16 + 1: TILE_LAUNCH(64, 1, 16); break;
    default: set_error("attn_tile: d=%d d_m=%d", p.d, p.dm); return MILLION_ERR_SHAPE;
    }
#undef TILE_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("attn_tile launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million
