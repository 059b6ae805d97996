// attn_lean.h - the lean decode kernel (round 5).  Included by attn_mfma.hip (inside namespace million, behind the streaming
// kernel): it shares that file's launch skeleton, residual-window helpers and L2 tail.
//
// The streaming kernel spends its vector issue on PQ address arithmetic (two instructions per K lookup: a lane's table base depends
// on the lane) and multiplies idle rows / columns (scores: 4 of the 16 head columns of a 16 x 16 x 32 tile, values: 8 of the 32 rows
// of a 32 x 32 x 16 tile).  This kernel keeps the skeleton - page-strided units, ONE page-id vector load, both codebooks in LDS
// behind one barrier, the residual tile, the fused append, the L2 tail - and replaces the core (tools/micro/lean_core.hip measures
// it alone: 1.15-1.18 x the units per us and SIMD of the parity-V core on cache-resident codes at +0.26 GHz of in-kernel clock,
// 1.05-1.14 x streamed from HBM; profiles/r05_core_micro.txt):
//   * a unit is 64 tokens and LANE = TOKEN on the K side: a lane owns its token's whole code row (M bytes: M / 16 16-byte loads);
//   * scores by v_mfma_f32_4x4x4_16B_f16 - sixteen independent blocks of 4 tokens x 4 heads x 4 dims: every MAC useful at G = 4.
//     A = the gathered centroid words of the k-step's 4 dims as gathered (d_m = 2: two 4-byte entries, subspaces 2 s and 2 s + 1;
//     d_m = 4: one 8-byte entry).  The subspace is wave-uniform, so the table base rides in the ds_read offset field and a K lookup
//     address is ONE instruction (an SDWA shift of the code byte).  B = the query heads: d / 16 register pairs; k-step 4 u + lambda
//     sits in lane group lambda of pair u and the MFMA's blgp field broadcasts that group to every block.  The scores land in lane
//     (token quad b, head j), register i = token 4 b + i: all 64 lanes carry useful scores, the softmax is 4 exponentials per 64
//     tokens and lane (16 x 16 tiles: 16);
//   * values, d_m = 2, by v_mfma_f32_16x16x32_f16 in "z-rows": rows = (z, parity of the dim, head < 4) - all 16 rows in use; the
//     reduction index is (token, parity) and the gathered V word is the B operand as it stands (as in parity-V).  Row (z, p, g) is
//     fed by the token groups kg (16 lanes = 16 tokens each) with (kg & 1) == z ^ phi in the MFMA of phase phi, and those lanes
//     gather subspace 32 pi + 16 z + n: the two token groups of a 32-lane LDS half read two different sets of 16 subspaces - 32
//     distinct banks, conflict-free (with rows = (parity, head < 8) both groups read the SAME subspace for different tokens: every V
//     gather a 2-way conflict, and the core was LDS-bound: micro VAR 0) - phase 1 swaps the roles, both phases accumulate into the
//     same 4 registers: 4 accumulator registers per 32 subspaces (parity-V: 32 for 64).  The A operand needs "4 tokens of ONE head"
//     per lane - exactly what a score lane holds: (P, P) pairs are broadcast from lane bank s to the whole 16-lane row by ONE
//     ds_swizzle per register and token step and ANDed with a lane-constant mask per phase (half of the dword by parity; zero where
//     the lane's token group does not feed the row);
//   * values, d_m = 4: the d_m = 4 form of the streaming kernel (rows = (dim position dq, head): all 16 rows in use already;
//     reduction = (token of 2, dim position of 4); column tiles of 16 subspaces; lanes (kg, n) and (kg + 1, n) gather the same
//     subspace for different tokens - its 2-way conflict stays) fed by the same swizzle-broadcast pairs: A = the pairs of the step's
//     two tokens ANDed with the lane's dim-position masks.
// Per 64 tokens and wave at d = 128 / M = 64: 128 LDS gathers (as before), 48 MFMAs (32 of them 4 x 4 x 4: 512 matrix-pipe cycles;
// parity-V: 768), ~195 vector instructions (parity-V: ~300), 16 swizzles.
// Shapes: C = 256 / 128, up to 4 query heads per kv head (d = 64: also 6 .. 16 as parts), pages of 64 or 128 tokens (or row-major K);
// d = 128 with M = 64 / 32 and d = 64 with M = 64 / 32 / 16 (d_m = 1 as zero-padded d_m = 2, 2, 4).  Everything else stays on the streaming / tile kernels (million_set_force_generic(16)
// keeps the lean shapes there too: A/B and tests).
// =====================================================================================================

// 4 * (8 *) byte B of w - the LDS offset of a 4- (8-) byte table entry - in ONE instruction (hipcc finds the SDWA form for bytes 1-3
// by itself and emits shift + mask for byte 0)
template <int B, int SH>
__device__ __forceinline__ unsigned byte_shl(unsigned w) {
    unsigned r;
    if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"((unsigned)SH), "v"(w));
    else if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"((unsigned)SH), "v"(w));
    else if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"((unsigned)SH), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"((unsigned)SH), "v"(w));
    return r;
}

// One 64-token unit of a wave: NQ = M / 16 16-byte pieces per token row (K) and per lane (V).
//   K: lane t: bytes [16 q, 16 q + 16) of token t's code row
//   V, d_m = 2: x = 2 pi + t: lane (kg, n): subspace 32 pi + 16 (t ^ (kg & 1)) + n, tokens 16 kg .. 16 kg + 15
//   V, d_m = 4: tile x: lane (kg, n): subspace n + 16 x, tokens 16 kg .. 16 kg + 15
template <int NQ>
struct LeanUnit { v4u k[NQ], v[NQ]; };

// Residual tile: scores on 16 x 16 x 32 tiles as in the streaming kernel (A = the fp16 K rows, lane (q4, c16): row c16, dims
// (DD / 4) q4 + 8 s .. in product s < DD / 32).  Values:
//   d_m = 2 (z-rows): v[2 pi + z][i] = dims (2 m, 2 m + 1), m = 32 pi + 16 z + n, of tile row 4 kg + i: the B operand of product (pi, z)
//   d_m = 4: v[2 j + s_][2 rr .. + 1] = dims 4 (n + 16 j) .. + 3 of tile row 4 kg + 2 s_ + rr: the B operand of product (tile j, k-step s_)
//   d_m = 1 (DR = 64 rows computed as d_m = 2 with every odd dim zero, DD = 128): v[x][i] = (dim 16 x + n, 0) of tile row 4 kg + i
template <int DD, int DR = DD>
struct LeanResTile {
    v4u k[DR / 32];
    unsigned v[DD / 32][4];
};
template <int DD, bool DM2, int DR = DD>
__device__ __forceinline__ void lean_load_res_tile(const AttnParams &p, int bh, const f16 *kr, const f16 *vr, int wave, int rcnt,
                                                   int split, int rstart, int r_old, int lane, LeanResTile<DD, DR> &t) {
    const int q4 = lane >> 4, c16 = lane & 15;
    {
        bool is_new;
        const long long off = (res_row_off(p, kResRows * wave + c16, wave, rcnt, split, rstart, r_old, is_new) >> 7) * DR;
        const f16 *kp = (is_new ? p.k_new + (long long)bh * DR : kr + off) + (DR / 4) * q4;
#pragma unroll
        for (int s = 0; s < DR / 32; ++s) t.k[s] = *(const v4u *)(kp + 8 * s);
    }
    if constexpr (DR != DD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool is_new;
            const long long off = (res_row_off(p, kResRows * wave + 4 * q4 + i, wave, rcnt, split, rstart, r_old, is_new) >> 7) * DR;
            const unsigned short *vp = (const unsigned short *)(is_new ? p.v_new + (long long)bh * DR : vr + off) + c16;
#pragma unroll
            for (int x = 0; x < DD / 32; ++x) t.v[x][i] = vp[16 * x];
        }
    } else if constexpr (DM2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bool is_new;
            const long long off = (res_row_off(p, kResRows * wave + 4 * q4 + i, wave, rcnt, split, rstart, r_old, is_new) >> 7) * DD;
            const f16 *vp = (is_new ? p.v_new + (long long)bh * DD : vr + off) + 2 * c16;
#pragma unroll
            for (int x = 0; x < DD / 32; ++x) t.v[x][i] = *(const unsigned *)(vp + 32 * x);      // dims 2 (16 x + n): x = 2 pi + z
        }
    } else {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                bool is_new;
                const long long off = (res_row_off(p, kResRows * wave + 4 * q4 + 2 * s_ + rr, wave, rcnt, split, rstart, r_old, is_new) >> 7) * DD;
                const f16 *vp = (is_new ? p.v_new + (long long)bh * DD : vr + off) + 4 * c16;
#pragma unroll
                for (int j = 0; j < DD / 64; ++j) {
                    const v2u w = *(const v2u *)(vp + 64 * j);
                    t.v[2 * j + s_][2 * rr + 0] = w[0];
                    t.v[2 * j + s_][2 * rr + 1] = w[1];
                }
            }
    }
}

// MS = M (subspaces), DD = d.  d_m = DD / MS is 2 (z-row values) or 4 (dim-position values).
// DR = 64 with DD = 128, MS = 64 is the d_m = 1 shape (d = 64, M = 64) run as d_m = 2 with every odd dim ZERO: the 2-byte codebook
// entries are widened to (value, 0) words while they are copied to LDS (the same 64 + 64 KiB as d = 128 / M = 64), the query words
// are (q, 0), the residual rows are read at their real length, and the accumulators - whose odd-parity rows stay zero - are
// packed into the d = 64 layout by 8 ds_bpermute in front of the tail.  Same gathers and products per token as d = 128 / M = 64
// (half of the score MACs multiply zeros: the matrix pipe is not what bounds this kernel).
// FL: which run-time extras the instance carries (the BASELINE shapes' instances carry none: this code runs from a cold instruction
// cache and every branch of its prologue is on the launch's critical path - 0.25 us per launch with both compiled in everywhere):
//   1 = may be given 128 centroids (p.C): the LDS copy spreads the K rows;  2 = may run as query-head parts (AttnParams::nhk_real)
template <int MODE, int MS = 64, int DD = 128, int DR = DD, int FL = 0>
__global__ __launch_bounds__(kNW * 64, 2) void attn_lean_kernel(AttnParams p) {
    static_assert((DD == 128 && (MS == 64 || MS == 32)) || (DD == 64 && (MS == 32 || MS == 16)), "lean kernel: d_m = 2 or 4 at d = 128 / 64");
    static_assert(DR == DD || (DR == 64 && DD == 128 && MS == 64), "lean kernel: the padded form is d = 64 / M = 64 only");
    constexpr bool PAD = DR != DD;
    constexpr bool DM2 = DD / MS == 2;
    constexpr int kLog2M = MS == 64 ? 6 : MS == 32 ? 5 : 4;
    constexpr int NK = DD / 4;                       // score k-steps (4 dims each) per unit
    constexpr int NQ = MS / 16;                      // 16-byte requests per unit and side
    constexpr int NPI = DD / 64;                     // value accumulators: d_m = 2: pairs pi of 32 subspaces; d_m = 4: column tiles of 16 subspaces
    constexpr int NV = 8 * NPI;                      // value products per unit: (token step s) x (phase phi | token pair h) x (pi | tile)
    constexpr int NT = DR / 16;                      // 16-byte pieces of a codebook image per thread (64 KiB at d = 128, 32 at d = 64)
    constexpr int MSTAG = DM2 ? 640 : 320;           // accumulator layout for the tail (merge_and_publish)
    constexpr int RING = 2;      // ring slots of one 64-token unit (8 NQ registers each).  Three slots (the K bytes two blocks ahead
                                 // instead of one) were measured and are slower at every shape: 24.98 vs 22.65 us at two requests,
                                 // 60.5 vs 59.8 at eight, 67.2 vs 65.1 at 8 x 36864 (profiles/r05_ab_lean.txt)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int split = blockIdx.x, bh = blockIdx.y;      // all splits of a (b, kv head) on one XCD: see attn_stream_kernel
    if ((gridDim.y & 7) == 0) {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        bh = id % (int)gridDim.y;
        split = id / (int)gridDim.y;
    }
    const int b = bh / p.nh_k, hk = bh % p.nh_k;      // hk, bh: VIRTUAL when the launch splits the query heads of a kv head into parts
    // real kv head / pair (what codes, page ids, window rows and the new rows are indexed by) and this workgroup's part
    const int part = (FL & 2) ? head_part(p, hk) : 0, hkr = hk - part * p.nhk_mul;
    const int bhr = (FL & 2) ? bh - (b * p.hparts_m1 + part) * p.nhk_mul : bh;
    const int G = (FL & 2) && p.nhk_mul ? min(p.G, p.G_all - part * p.G) : p.G;      // (the last part of an odd head group holds fewer)
    const bool k_paged = MODE == 0 ? true : MODE == 1 ? false : (p.k_paged != 0);
    const bool v_ident = MODE == 0 ? false : MODE == 1 ? true : (p.v_identity != 0);
    const bool ids64 = MODE == 2 ? (p.ids64 != 0) : false;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i dl = {p.T, p.r, p.rstart, 0};
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    const bool dbg_on = p.dbg != nullptr;
#define STAMP(i) stamp_lds(dbg_on, lane, wave, i)
    stamp_lds_clear(dbg_on, lane, wave);
    STAMP(0);
    const int kg = lane >> 4, n16 = lane & 15, hj = lane & 3;

    // ---- where this wave reads: page pg0 + j * pg_step in round j, tokens [tin, tin + 64) of it ----
    const int ups = p.ps_shift - 6;                       // log2(units per page): pages of 64 or 128 tokens
    const int wp = wave >> ups, uw = wave & ((1 << ups) - 1);
    const int pg0 = wp * p.nsplit + split;
    const int pg_step = p.nsplit << (3 - ups);
    const int tin = uw << 6;
    int vpk = 0, vpv = 0;      // page ids of rounds 0..63 (lane = round): the oldest loads of the wave
    {
        int pgl = pg0 + lane * pg_step;
        pgl = pgl < p.n_pages_cap ? pgl : p.n_pages_cap - 1;
        const long long idx = (long long)bhr * p.n_pages_cap + pgl;
        if (k_paged) vpk = ids64 ? (int)p.k_ids64[idx] : p.k_ids32[idx];
        if (v_ident) vpv = (int)idx;
        else vpv = ids64 ? (int)p.v_ids64[idx] : p.v_ids32[idx];
#ifdef MILLION_DEBUG_CHECK_IDS
        {
            const bool live = pg0 + lane * pg_step < p.n_pages_cap && ((long long)(pg0 + lane * pg_step) << p.ps_shift) < p.T;
            if (k_paged) vpk = MILLION_CHECK_KID(p, ids64 ? (long long)p.k_ids64[idx] : (long long)vpk, live);
            if (!v_ident) vpv = MILLION_CHECK_VID(p, ids64 ? (long long)p.v_ids64[idx] : (long long)vpv, live);
        }
#endif
    }
    // query operand of the 4 x 4 x 4 products: register pair u, lane group kg holds the 4 dims of k-step (u, kg) of head (lane & 3).
    // A lane's pairs 2 v, 2 v + 1 are ONE 16-byte load (dims 32 v + 8 kg .. + 7), so k-step (u, lambda) covers the 4-dim group
    // sigma = 8 (u >> 1) + 2 lambda + (u & 1) (LEAN_SIGMA below; d_m = 2: subspaces 2 sigma, 2 sigma + 1; d_m = 4: subspace sigma):
    // any order of the groups will do
    const f16 *qrow = p.q + ((long long)b * p.nh + head0(p, hk)) * DR;
    v2u Q[NK / 4];
#pragma unroll
    for (int v = 0; v < DD / 32; ++v) {
        v4u t;
        if constexpr (PAD) {      // padded dims 32 v + 8 kg .. + 7 = real dims 16 v + 4 kg .. + 3, each widened to (q, 0)
            const v2u t2 = *(const v2u *)(qrow + (hj < G ? hj : 0) * DR + 16 * v + 4 * kg);
            t = v4u{t2[0] & 0xffffu, t2[0] >> 16, t2[1] & 0xffffu, t2[1] >> 16};
        } else {
            t = *(const v4u *)(qrow + (hj < G ? hj : 0) * DD + 32 * v + 8 * kg);
        }
        if (hj >= G) t = v4u{0, 0, 0, 0};
        Q[2 * v] = v2u{t[0], t[1]};
        Q[2 * v + 1] = v2u{t[2], t[3]};
    }
    const bool append_wave = p.k_new && split == 0 && wave == kNW - 1 && part == 0;      // wave-uniform
    h2 new_k = {}, new_v = {};
    if (append_wave && 2 * lane < DR) {
        new_k = *(const h2 *)(p.k_new + (long long)bhr * DR + 2 * lane);
        new_v = *(const h2 *)(p.v_new + (long long)bhr * DR + 2 * lane);
    }
    // (128 centroids: the same loads - the second half of them reads the image that follows in the prepared buffer and is dropped
    // by the copy into LDS below, which spreads the K rows to the 256-entry row stride the gathers use)
    v4u tabk[NT], tabv[NT];
    const int rot = (blockIdx.x + 5 * blockIdx.y) & (NT - 1);
    {
        const v4u *ks = (const v4u *)p.k_tab;
#pragma unroll
        for (int i = 0; i < NT; ++i) tabk[i] = ks[((i + rot) & (NT - 1)) * (kNW * 64) + tid];
        const v4u *vs = (const v4u *)p.v_tab_col;
#pragma unroll
        for (int i = 0; i < NT; ++i) tabv[i] = vs[((i + rot) & (NT - 1)) * (kNW * 64) + tid];
    }
    if (p.dev_lengths)
        asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dl) : "s"(p.dev_lengths), "s"((unsigned)b * 16u) : "memory");
    int T = dl[0], r_old = dl[1], rstart = dl[2];
    clamp_lengths(p, T, r_old, rstart);
    const int r = r_old + (p.k_new ? 1 : 0);
    const int t0 = (pg0 << p.ps_shift) + tin;             // first token of round 0
    const int t_step = pg_step << p.ps_shift;             // tokens between rounds
    const int n_mine = T > t0 ? (T - t0 + t_step - 1) / t_step : 0;      // rounds (= units) of this wave; host: <= 64
    const int j_last = n_mine > 0 ? n_mine - 1 : 0;
    const int T_ld = T > 0 ? T : 1;

    // ---- residual window rows of this split (see load_res_tile) ----
    const int rcnt = split < r ? (r - split + p.nsplit - 1) / p.nsplit : 0;
    const bool has_res = kResRows * wave < rcnt;
    const f16 *kr = p.k_res + b * p.res_sb + hkr * p.res_sh;
    const f16 *vr = p.v_res + b * p.res_sb + hkr * p.res_sh;
    LeanResTile<DD, DR> rt;
    v8f16 qb[DR / 32];      // the residual tile's query operand (16 x 16 x 32 layout: lane (q4, c16): head c16, dims (DD / 4) q4 + 8 s ..;
                            // d_m = 4: head c16 & 3 - the four column groups carry copies of the heads, as in the streaming kernel's form)
    if (has_res) {
        lean_load_res_tile<DD, DM2, DR>(p, bhr, kr, vr, wave, rcnt, split, rstart, r_old, lane, rt);
        const int hq = DM2 ? n16 : hj;
        const f16 *qv = qrow + (hq < G ? hq : 0) * DR + (DR / 4) * kg;
#pragma unroll
        for (int s = 0; s < DR / 32; ++s) {
            v4u t = *(const v4u *)(qv + 8 * s);
            if (hq >= G) t = v4u{0, 0, 0, 0};
            qb[s] = __builtin_bit_cast(v8f16, t);
        }
    }

    // ---- one unit's 16-byte requests into ring slot SL (rounds past the wave's last unit re-request it: no load in a conditional) ----
    LeanUnit<NQ> ring[RING];
    const unsigned k_lane_off = (unsigned)lane << kLog2M;
    // V, d_m = 2: x even reads subspace row 16 (kg & 1) + n, x odd the other 16 of the pair's 32; + 32 subspace rows per pi
    //    d_m = 4: tile x reads subspace row n + 16 x
    const unsigned v_lane_off1 = ((unsigned)(n16 + (DM2 ? 16 * (kg & 1) : 0)) << p.ps_shift) + 16u * kg;
    const unsigned v_lane_off2 = ((unsigned)(n16 + (DM2 ? 16 * (1 - (kg & 1)) : 16)) << p.ps_shift) + 16u * kg;
#define UNIT_REQ_K(SL, J)                                                                                          \
    {                                                                                                              \
        const int jc_ = (J) < n_mine ? (J) : j_last;                                                               \
        gptr_u8 kb_;                                                                                               \
        if (k_paged) {                                                                                             \
            const long long pk_ = (long long)__builtin_amdgcn_readlane(vpk, jc_);                                  \
            kb_ = uniform_ptr(p.k_codes + (((pk_ << p.ps_shift) + tin) << kLog2M));                                \
            _Pragma("unroll") for (int q_ = 0; q_ < NQ; ++q_) ring[SL].k[q_] = *(gptr_v4u)(kb_ + k_lane_off + 16u * q_); \
        } else {      /* row-major K: absolute row per lane, rows past T - 1 re-read it (masked later) */          \
            const int tu_ = t0 + jc_ * t_step;                                                                     \
            kb_ = uniform_ptr(p.k_codes + b * p.k_sb + hkr * p.k_sh);                                               \
            const unsigned ro_ = (unsigned)min(tu_ + lane, T_ld - 1) << kLog2M;                                    \
            _Pragma("unroll") for (int q_ = 0; q_ < NQ; ++q_) ring[SL].k[q_] = *(gptr_v4u)(kb_ + ro_ + 16u * q_);  \
        }                                                                                                          \
    }
#define UNIT_REQ_V(SL, J)                                                                                          \
    {                                                                                                              \
        const int jc_ = (J) < n_mine ? (J) : j_last;                                                               \
        const long long pv_ = (long long)__builtin_amdgcn_readlane(vpv, jc_);                                      \
        const gptr_u8 vb_ = uniform_ptr(p.v_codes + (pv_ << (kLog2M + p.ps_shift)) + tin);                         \
        _Pragma("unroll") for (int x_ = 0; x_ < NQ; ++x_)                                                          \
            ring[SL].v[x_] = *(gptr_v4u)(vb_ + ((x_ & 1) ? v_lane_off2 : v_lane_off1) + ((32u * (x_ >> 1)) << p.ps_shift)); \
    }
#define UNIT_REQ(SL, J) { UNIT_REQ_K(SL, J) UNIT_REQ_V(SL, J) }
    // only unit 0 (8 KiB per wave at M = 64 - what the streaming kernel asks for up front) goes out before the codebooks are in LDS:
    // the CU's request queue is in order, and what is asked for in front of the barrier delays it (both units up front: +1.3 us at
    // one request).  (Its K bytes in the coalesced shape of the streaming kernel + a 4 x 4 in-register transpose by row swaps:
    // no difference, profiles/r05_ab_unit0_coalesced.txt.)
    UNIT_REQ(0, 0)
    STAMP(7);
    {
        v4u *ld = (v4u *)smem;
        v4u *ldv = (v4u *)(smem + kVBase);
        if ((FL & 1) && p.C == 128) {      // (wave-uniform)
            // K row image [m][128][d_m]: row m goes to the 256-entry row stride of the C = 256 image (codes are < 128: the second half of
            // a row is never read); V col image [c][m][d_m]: the first 128 rows, placed as for C = 256.  Pieces of the images' size only
            constexpr unsigned RBS = 128u * (DR / MS) * 2u;               // source bytes of a K row
            constexpr unsigned LROW = 256u * (DD / MS) * 2u;              // its LDS stride
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (((i + rot) & (NT - 1)) >= NT / 2) continue;
                const unsigned pi_ = ((i + rot) & (NT - 1)) * (kNW * 64) + tid;
                const unsigned off = pi_ * 16u, kd = (off / RBS) * LROW + (off % RBS) * (PAD ? 2u : 1u);
                if constexpr (PAD) {
                    *(v4u *)(smem + kd) = v4u{tabk[i][0] & 0xffffu, tabk[i][0] >> 16, tabk[i][1] & 0xffffu, tabk[i][1] >> 16};
                    *(v4u *)(smem + kd + 16) = v4u{tabk[i][2] & 0xffffu, tabk[i][2] >> 16, tabk[i][3] & 0xffffu, tabk[i][3] >> 16};
                    ldv[2 * pi_] = v4u{tabv[i][0] & 0xffffu, tabv[i][0] >> 16, tabv[i][1] & 0xffffu, tabv[i][1] >> 16};
                    ldv[2 * pi_ + 1] = v4u{tabv[i][2] & 0xffffu, tabv[i][2] >> 16, tabv[i][3] & 0xffffu, tabv[i][3] >> 16};
                } else {
                    *(v4u *)(smem + kd) = tabk[i];
                    if constexpr (DD == 128) ldv[pi_] = tabv[i];
                    else *(v4u *)(smem + kVBase + ((off >> 7) << 8) + (off & 127u)) = tabv[i];
                }
            }
        } else
        if constexpr (PAD) {      // 2-byte entries -> (value, 0) words: piece pi of an image becomes pieces 2 pi, 2 pi + 1 of its LDS copy
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const unsigned pi_ = ((i + rot) & (NT - 1)) * (kNW * 64) + tid;
                ld[2 * pi_] = v4u{tabk[i][0] & 0xffffu, tabk[i][0] >> 16, tabk[i][1] & 0xffffu, tabk[i][1] >> 16};
                ld[2 * pi_ + 1] = v4u{tabk[i][2] & 0xffffu, tabk[i][2] >> 16, tabk[i][3] & 0xffffu, tabk[i][3] >> 16};
                ldv[2 * pi_] = v4u{tabv[i][0] & 0xffffu, tabv[i][0] >> 16, tabv[i][1] & 0xffffu, tabv[i][1] >> 16};
                ldv[2 * pi_ + 1] = v4u{tabv[i][2] & 0xffffu, tabv[i][2] >> 16, tabv[i][3] & 0xffffu, tabv[i][3] >> 16};
            }
        } else {
#pragma unroll
        for (int i = 0; i < NT; ++i) ld[((i + rot) & (NT - 1)) * (kNW * 64) + tid] = tabk[i];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const unsigned pi_ = ((i + rot) & (NT - 1)) * (kNW * 64) + tid;      // 16-byte piece of the dense col image [c][m][d_m]
            if constexpr (DD == 128) ldv[pi_] = tabv[i];
            else      // d = 64: a code's row is 128 bytes; in LDS it sits at c * 256, so that a code byte is byte 1 of its row's address
                *(v4u *)(smem + kVBase + (((pi_ * 16u) >> 7) << 8) + ((pi_ * 16u) & 127u)) = tabv[i];
        }
        }
    }
    STAMP(8);
    __syncthreads();
    STAMP(1);

    // O.t[j][g]: head g; d_m = 2: lane (rg, n): dim 64 j + 32 (rg >> 1) + 2 n + (rg & 1); d_m = 4: lane (dq, n): dim 4 (n + 16 j) + dq
    Acc8 O;
    O.t[0] = v4f32{0.f, 0.f, 0.f, 0.f};
    O.t[1] = v4f32{0.f, 0.f, 0.f, 0.f};
    if (append_wave && 2 * lane < DR) {
        int row_n = rstart + r_old;
        row_n = row_n >= p.rcap ? row_n - p.rcap : row_n;
        const long long o = b * p.res_sb + hkr * p.res_sh + (long long)row_n * DR + 2 * lane;
        *(h2 *)(p.k_res_w + o) = new_k;
        *(h2 *)(p.v_res_w + o) = new_v;
    }
    // lane constants of the value side.  d_m = 2: the half of a (P, P) pair this lane's row takes (parity p = bit 2 of the lane), the
    // phase masks (row z = bit 3 is fed by this lane's token group kg in phase phi iff (kg & 1) == z ^ phi), the gather constants
    // (V col image base | 4 x the lane's subspace within the pair's 32); d_m = 4: the dim-position masks, base | 8 n
    const unsigned zr_half = ((lane >> 2) & 1) ? 0xffff0000u : 0x0000ffffu;
    const bool z_own = (kg & 1) == ((lane >> 3) & 1);
    const unsigned zmask0 = z_own ? zr_half : 0u, zmask1 = z_own ? 0u : zr_half;
    const unsigned vcz0 = DM2 ? ((unsigned)kVBase | ((unsigned)(n16 + 16 * (kg & 1)) << 2)) : ((unsigned)kVBase | ((unsigned)n16 << 3));
    const unsigned vcz1 = (unsigned)kVBase | ((unsigned)(n16 + 16 * (1 - (kg & 1))) << 2);
    unsigned d4mx, d4my;
    d8_masks(lane, d4mx, d4my);
    const float c_ = p.scale_log2e, inv_c = 1.0f / p.scale_log2e;
    const float idle = hj < G ? 0.f : -INFINITY;      // lanes of heads that do not exist: probabilities come out as exact zeros
    // softmax state of head (lane & 3): the reference m (see SoftRef), this lane's part of the row sums
    float s_m = -INFINITY, s_l = 0.f;
    if (has_res) {      // residual tile of this wave first: it needs neither codebook
        float scr[4], m_run = -INFINITY, l_run = 0.f;
        {
            v4f32 Dr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DR / 32; ++s) Dr = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8f16, rt.k[s]), qb[s], Dr, 0, 0, 0);
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) scr[rho] = (kResRows * wave + 4 * kg + rho) < rcnt ? Dr[rho] * p.scale_log2e : -INFINITY;
        }
        softmax_online<4, false>(scr, m_run, l_run, O, G, lane);           // lane (q4, head c16): rows 4 q4 + rho; O is zero: nothing is rescaled
        if constexpr (!DM2) {      // every lane holds its own head's probabilities (replicated column groups) and reference
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const h2 a0 = {(f16)scr[2 * s_], (f16)scr[2 * s_]}, a1 = {(f16)scr[2 * s_ + 1], (f16)scr[2 * s_ + 1]};
                const unsigned w0 = __builtin_bit_cast(unsigned, a0), w1 = __builtin_bit_cast(unsigned, a1);
                const v8f16 Ar = as_v8f16(w0 & d4mx, w0 & d4my, w1 & d4mx, w1 & d4my);
#pragma unroll
                for (int j = 0; j < NPI; ++j)
                    O.t[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ar, as_v8f16(rt.v[2 * j + s_][0], rt.v[2 * j + s_][1], rt.v[2 * j + s_][2], rt.v[2 * j + s_][3]),
                                                                    O.t[j], 0, 0, 0);
            }
            s_m = m_run;
            s_l = (lane & 12) == 0 ? l_run : 0.f;
        } else {
            // (P, P) of lanes 0-3 of every 16-lane row (heads 0-3, rows 4 kg + i) -> the whole row; rows of z: one product per (pi, z)
            unsigned sw_[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const h2 pp = {(f16)scr[i], (f16)scr[i]};
                sw_[i] = (unsigned)__builtin_amdgcn_ds_swizzle((int)__builtin_bit_cast(unsigned, pp), 0x13);
            }
            const unsigned rz0 = ((lane >> 3) & 1) ? 0u : zr_half, rz1 = ((lane >> 3) & 1) ? zr_half : 0u;
#pragma unroll
            for (int x = 0; x < DD / 32; ++x) {
                const unsigned mk = (x & 1) ? rz1 : rz0;
                O.t[x >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_v8f16(sw_[0] & mk, sw_[1] & mk, sw_[2] & mk, sw_[3] & mk),
                                                                     as_v8f16(rt.v[x][0], rt.v[x][1], rt.v[x][2], rt.v[x][3]), O.t[x >> 1], 0, 0, 0);
            }
            // state into the lean layout: head j's reference sits in lane j of every row (bank 0); a row's sums are counted once
            s_m = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, m_run), 0x13));
            s_l = (lane & 12) == 0 ? l_run : 0.f;
        }
    }
    STAMP(2);
    float s_neg = (s_m > -INFINITY ? -s_m : 0.f) + idle, s_thr = (s_m + 8.0f) * inv_c;

    unsigned L[4];                  // (P, P) pairs of this lane's (token quad, head)
    unsigned sw[2][4];              // the pairs of token step s, broadcast over the 16-lane row (sw[s & 1])
    unsigned Az[2][4];              // A operands (d_m = 2: of the two phases)
    v2u a[8];                       // K gathers in flight (k-step sg in a[sg & 7]: 3 steps ahead inside a block, 6 when the scores run alone)
    unsigned e[4][4];               // V gathers in flight (value step i in e[i & 3]: 2 steps ahead inside a block, 3 alone)
    v4f32 D[2];

    // ---- K side: k-step sg = the 4-dim group sigma (wave-uniform: the table base is the read's offset field) ----
#define KBYTE4(SL, B) byte_shl<(B) & 3, 2>(ring[SL].k[((B) >> 4) & (NQ - 1)][((B) >> 2) & 3])      /* 4 * code byte B of the lane's row */
#define KBYTE8(SL, B) byte_shl<(B) & 3, 3>(ring[SL].k[((B) >> 4) & (NQ - 1)][((B) >> 2) & 3])      /* 8 * code byte B */
#define LEAN_SIGMA(SG) (8 * ((SG) >> 3) + 2 * ((SG) & 3) + (((SG) >> 2) & 1))
#define KG(SL, SG)                                                                                                 \
    {                                                                                                              \
        if constexpr (DM2) {                                                                                       \
            a[(SG) & 7][0] = lds32(KBYTE4(SL, 2 * LEAN_SIGMA(SG)) + (2 * LEAN_SIGMA(SG)) * 1024u);                 \
            a[(SG) & 7][1] = lds32(KBYTE4(SL, 2 * LEAN_SIGMA(SG) + 1) + (2 * LEAN_SIGMA(SG) + 1) * 1024u);         \
        } else {      /* d_m = 4: k-step = subspace sigma: one 8-byte entry = the lane's 4 dims */                   \
            a[(SG) & 7] = lds64(KBYTE8(SL, LEAN_SIGMA(SG)) + LEAN_SIGMA(SG) * 2048u);                              \
        }                                                                                                          \
    }
    // (the empty asm behind a product pins it: an MFMA is register-only, so hipcc moves it across sched_barrier() at will - it
    // bunched the score products in runs of 4-12 behind runs of 4-5 value products; with its accumulator made opaque at this
    // point the product stays in the slot it is written in, cdna_hip_programming.md 5.7 item 3)
#define KM(SG)                                                                                                     \
    {                                                                                                              \
        D[(SG) & 1] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4f16_t, a[(SG) & 7]), __builtin_bit_cast(v4f16_t, Q[(SG) >> 2]), \
                                                         (SG) < 2 ? v4f32{0.f, 0.f, 0.f, 0.f} : D[(SG) & 1], 0, 0, 4 + ((SG) & 3)); \
        asm volatile("" : "+v"(D[(SG) & 1]));                                                                      \
    }
    // ---- V side: value product i = (2 s + mid) NPI + lo, lo = accumulator (pi | tile), s = token step (tokens 16 kg + 4 s + 0..3 =
    //      dword s of the lane's 16 bytes):
    //      d_m = 2: mid = phase phi: gathers from ring.v[2 lo + mid], all four bytes of dword s
    //      d_m = 4: mid = token pair h: gathers bytes 2 h, 2 h + 1 of dword s of ring.v[lo] ----
#define V_LO(I) ((I) % NPI)
#define V_MID(I) (((I) / NPI) & 1)
#define V_S(I) ((I) / (2 * NPI))
#define VG(SL, I)                                                                                                  \
    {                                                                                                              \
        if constexpr (DM2) {                                                                                       \
            const unsigned w_ = ring[SL].v[(2 * V_LO(I) + V_MID(I)) & (NQ - 1)][V_S(I)];                           \
            const unsigned vc_ = V_MID(I) ? vcz1 : vcz0;                                                           \
            e[(I) & 3][0] = lds32(__builtin_amdgcn_perm(w_, vc_, 0x03020400u) + 128u * V_LO(I));                   \
            e[(I) & 3][1] = lds32(__builtin_amdgcn_perm(w_, vc_, 0x03020500u) + 128u * V_LO(I));                   \
            e[(I) & 3][2] = lds32(__builtin_amdgcn_perm(w_, vc_, 0x03020600u) + 128u * V_LO(I));                   \
            e[(I) & 3][3] = lds32(__builtin_amdgcn_perm(w_, vc_, 0x03020700u) + 128u * V_LO(I));                   \
        } else {                                                                                                   \
            const unsigned w_ = ring[SL].v[V_LO(I) & (NQ - 1)][V_S(I)];                                            \
            const v2u x0_ = lds64(__builtin_amdgcn_perm(w_, vcz0, V_MID(I) ? 0x03020600u : 0x03020400u) + 128u * V_LO(I)); \
            const v2u x1_ = lds64(__builtin_amdgcn_perm(w_, vcz0, V_MID(I) ? 0x03020700u : 0x03020500u) + 128u * V_LO(I)); \
            e[(I) & 3][0] = x0_[0]; e[(I) & 3][1] = x0_[1]; e[(I) & 3][2] = x1_[0]; e[(I) & 3][3] = x1_[1];        \
        }                                                                                                          \
    }
    // (P, P) of lane bank s -> every lane of the 16-lane row (bit mode: lane' = (lane & 0x13) | (s << 2))
#define ZSWZ(S) { _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) sw[(S) & 1][ii] = (unsigned)__builtin_amdgcn_ds_swizzle((int)L[ii], 0x13 | (((S) << 2) << 5)); }
#define VS(I)                                                                                                      \
    {                                                                                                              \
        if constexpr (DM2) {                                                                                       \
            if (V_LO(I) == 0 && V_MID(I) == 0) {      /* a new token step: both phases' A operands, the next step's swizzles */ \
                _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) { Az[0][ii] = sw[V_S(I) & 1][ii] & zmask0; Az[1][ii] = sw[V_S(I) & 1][ii] & zmask1; } \
                if (V_S(I) < 3) ZSWZ(V_S(I) + 1)                                                                   \
            }                                                                                                      \
        } else if (V_LO(I) == 0) {      /* d_m = 4: the A operand of the step's two tokens (pairs 2 h, 2 h + 1 of block s), for every tile */ \
            Az[0][0] = sw[V_S(I) & 1][2 * V_MID(I)] & d4mx; Az[0][1] = sw[V_S(I) & 1][2 * V_MID(I)] & d4my;        \
            Az[0][2] = sw[V_S(I) & 1][2 * V_MID(I) + 1] & d4mx; Az[0][3] = sw[V_S(I) & 1][2 * V_MID(I) + 1] & d4my; \
            if (V_MID(I) == 1 && V_S(I) < 3) ZSWZ(V_S(I) + 1)                                                      \
        }                                                                                                          \
        constexpr int az_ = DM2 ? V_MID(I) : 0;                                                                    \
        O.t[V_LO(I)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_v8f16(Az[az_][0], Az[az_][1], Az[az_][2], Az[az_][3]), \
                                                              as_v8f16(e[(I) & 3][0], e[(I) & 3][1], e[(I) & 3][2], e[(I) & 3][3]), \
                                                              O.t[V_LO(I)], 0, 0, 0);                              \
        asm volatile("" : "+v"(O.t[V_LO(I)]));                                                                     \
    }
    // ---- online softmax of the unit of round J whose scores are in D: lane (quad b, head j) holds tokens t_u + 4 b + i.  The
    //      reference moves only when a raw score exceeds thr (SoftRef); the new one is the maximum over the head's 16 lanes ----
#define SOFTMAX(J)                                                                                                 \
    {                                                                                                              \
        float x_[4];                                                                                               \
        const int t_u = t0 + (J) * t_step;                                                                         \
        if (t_u + 64 <= T) {                                                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) x_[i] = D[0][i] + D[1][i];                               \
        } else {      /* the unit that holds token T - 1 (wave-uniform); a unit past it gives -inf everywhere */   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) x_[i] = t_u + (lane & ~3) + i < T ? D[0][i] + D[1][i] : -INFINITY; \
        }                                                                                                          \
        const float mx_ = max3_raw(max3_raw(x_[0], x_[1], x_[2]), x_[3], x_[3]);                                   \
        if (__any(mx_ > s_thr)) {                                                                                  \
            float mq_ = mx_;                                                                                       \
            mq_ = fmaxf(mq_, MILLION_DPP(mq_, 0x124));      /* row_ror:4 */                                        \
            mq_ = fmaxf(mq_, MILLION_DPP(mq_, 0x128));      /* row_ror:8 */                                        \
            const float m_new = fmaxf(s_m, rows_max(mq_) * c_);                                                    \
            const float m_safe = m_new > -INFINITY ? m_new : 0.f;                                                  \
            const float alpha = fast_exp2(s_m - m_safe);                                                           \
            if (__any(m_new > s_m && s_m > -INFINITY)) rescale_acc(O, alpha, G, lane);                             \
            s_l *= alpha;                                                                                          \
            s_m = m_new;                                                                                           \
            s_neg = (m_new > -INFINITY ? -m_new : 0.f) + idle;                                                     \
            s_thr = (m_new + 8.0f) * inv_c;                                                                        \
        }                                                                                                          \
        float ps_ = 0.f;                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            const float pr_ = fast_exp2(fmaf(x_[i], c_, s_neg));                                                   \
            ps_ += pr_;                                                                                            \
            const h2 pp_ = {(f16)pr_, (f16)pr_};                                                                   \
            L[i] = __builtin_bit_cast(unsigned, pp_);                                                              \
        }                                                                                                          \
        s_l += ps_;                                                                                                \
        ZSWZ(0)                                                                                                    \
    }
#define FOR8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define FOR16(X) FOR8(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define FOR32(X) FOR16(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
    // (every step index is a literal - the MFMA's blgp field and the swizzle patterns are immediates - and the lists are as long as
    // the larger shape needs: a step past NK / NV is an empty statement)
#define FOR_NK(X) FOR32(X)
#define FOR_NV(X) FOR16(X)
    // the scores of the unit in slot 0 alone (the prologue: nothing to interleave with, so the gathers run 6 k-steps ahead); the
    // requests of round 1 go out in between - behind the barrier, one group every few k-steps (all eight right behind the barrier:
    // the waves sat ~1 us in their load instructions while the CU's request queue was full)
#define SA_STEP(SG)                                                                                                \
    if constexpr ((SG) < NK) {                                                                                     \
        KM(SG)                                                                                                     \
        if constexpr ((SG) + 6 < NK) KG(0, ((SG) + 6) & (NK - 1))                                                  \
        if ((SG) == 4) UNIT_REQ_K(1, 1)                                                                            \
        if ((SG) == 12) UNIT_REQ_V(1, 1)                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
    // BLOCK: the NV value products of the unit in slot BL_SL (round bl_j) interleaved with the NK score products of the unit in the
    // other slot (round bl_j + 1), whose first three gathers are in flight; the K bytes of round bl_j + 2 are requested at the start
    // into slot BL_SL (its K bytes were used up by the previous block), its V bytes once the last value gather of round bl_j is out;
    // the first three K gathers of round bl_j + 2 close the block
#define BL_STEP(I)                                                                                                 \
    if constexpr ((I) < NV) {                                                                                      \
        VS(I)                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        VG(((I) + 2 < NV ? BL_SL : BL_SLN), ((I) + 2) & (NV - 1))                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        KM(2 * (I))                                                                                                \
        KG((2 * (I) + 3 < NK ? BL_SLN : BL_SL), (2 * (I) + 3) & (NK - 1))                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        KM(2 * (I) + 1)                                                                                            \
        KG((2 * (I) + 4 < NK ? BL_SLN : BL_SL), (2 * (I) + 4) & (NK - 1))                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if ((I) == NV - 3) UNIT_REQ_V(BL_SL, bl_j + RING)                                                          \
    }
#define BLOCK() { UNIT_REQ_K(BL_SL, bl_j + RING) FOR_NV(BL_STEP) SOFTMAX(bl_j + 1) ++bl_j; }
    // the value products of the unit in slot VA_SL alone (gathers 0 .. 2 are in flight; 3 steps ahead)
#define VA_STEP(I) if constexpr ((I) < NV) { if constexpr ((I) + 3 < NV) VG(VA_SL, (I) + 3) VS(I) __builtin_amdgcn_sched_barrier(0); }
    // A wave with n units runs: scores of unit 0 | n - 1 blocks (values of unit j beside the scores of unit j + 1) | values of unit
    // n - 1.  Blocks alternate between the two ring slots (unit j lives in slot j & 1): one block, pairs in a loop, one more when n is odd.
    const int nb = n_mine > 1 ? n_mine - 1 : 0;
    TailReq treq;
    treq.idx = 0; treq.gen = 0; treq.cen = 0; treq.base = 0; treq.done = false;
    tail_mark_xcd(p, bh, split, wave, lane);      // this split's slot of the XCD census
    int bl_j = 0;
    {
        KG(0, 0) KG(0, 1) KG(0, 2) KG(0, 3) KG(0, 4) KG(0, 5)
        FOR_NK(SA_STEP)
        SOFTMAX(0)      // (a wave without units: every score is masked to -inf, every probability 0)
        STAMP(16);
    }
    if (n_mine > 0) {
        VG(0, 0) VG(0, 1)
        if (nb > 0) {
            KG(1, 0) KG(1, 1) KG(1, 2)
#define BL_SL 0
#define BL_SLN 1
            BLOCK()
#undef BL_SL
#undef BL_SLN
            for (int w = 0; w < (nb - 1) >> 1; ++w) {
#define BL_SL 1
#define BL_SLN 0
                BLOCK()
#undef BL_SL
#undef BL_SLN
#define BL_SL 0
#define BL_SLN 1
                BLOCK()
#undef BL_SL
#undef BL_SLN
                if (w == 0) STAMP(17);
            }
        }
        tail_request(p, bh, p.nslots, wave, lane, treq);      // ~3 us ahead of the point where the tail needs the answers
        STAMP(19);
        if (n_mine & 1) {      // an odd number of units: the last one sits in slot 0 - behind one more block unless it is the only one
            if (nb > 0) {
#define BL_SL 1
#define BL_SLN 0
                BLOCK()
#undef BL_SL
#undef BL_SLN
            }
#define VA_SL 0
            VG(0, 2)
            FOR_NV(VA_STEP)
#undef VA_SL
        } else {
#define VA_SL 1
            VG(1, 2)
            FOR_NV(VA_STEP)
#undef VA_SL
        }
    }
#undef BLOCK
#undef BL_STEP
#undef VA_STEP
#undef SA_STEP
#undef FOR_NK
#undef FOR_NV
#undef FOR32
#undef FOR16
#undef FOR8
#undef SOFTMAX
#undef VS
#undef ZSWZ
#undef VG
#undef V_LO
#undef V_MID
#undef V_S
#undef KM
#undef KG
#undef LEAN_SIGMA
#undef KBYTE4
#undef KBYTE8
#undef UNIT_REQ
#undef UNIT_REQ_K
#undef UNIT_REQ_V
    STAMP(3);
    // the tail wants head g's reference in lane g and row sums that rows_sum() completes: add up the head's four lanes of a row
    float l_row = s_l;
    l_row += MILLION_DPP(l_row, 0x124);
    l_row += MILLION_DPP(l_row, 0x128);
    if constexpr (PAD) {
        // padded accumulators: O.t[j'][g], lane (rg', n'), rg' even: real dim 32 j' + 16 (rg' >> 1) + n' (odd rg': the zero dims).
        // The d = 64 layout wants lane (rg, n) of t[0] to hold dim D = 32 (rg >> 1) + 2 n + (rg & 1): j' = rg >> 1,
        // lane' = 32 (n >> 3) + 2 (n & 7) + (rg & 1)
        const int src = (32 * (n16 >> 3) + 2 * (n16 & 7) + (kg & 1)) << 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float f0 = O.t[0][g], f1 = O.t[1][g];      // (copies: __builtin_bit_cast of a vector ELEMENT reads element 0 with this hipcc)
            const int v0 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, f0));
            const int v1 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, f1));
            O.t[0][g] = __builtin_bit_cast(float, (kg >> 1) ? v1 : v0);
            O.t[1][g] = 0.f;
        }
    }
    merge_and_publish<MSTAG, false, DR>(p, smem, b, hk, split, G, tid, lane, wave, dbg_on, O, s_m, l_row, treq);
#undef STAMP
}

