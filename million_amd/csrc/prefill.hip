// prefill.hip — causal prompt attention on fp16 K/V (flash-style, MFMA), gfx950.
//
// Replaces: the scaled_dot_product_attention(q, repeat_kv(k), repeat_kv(v), is_causal=True) of the reference's prompt
// pass (scripts/utils/pq_utils.py:249-260 DynamicPQCache.prefill; scripts/utils/paged_pq_utils.py:216-320
// PagedPQCache.prefill) - without materialising repeat_kv: the G = nh / nh_k query heads of a kv head are served by the
// SAME workgroup, so a K/V tile is fetched once per kv head and q block, not once per query head.
//
//   out[b, h, i, :] = softmax_j<=q_pos0+i( q[b,h,i,:] . k[b,hk,j,:] / sqrt(d) ) v[b,hk,j,:],   hk = h / G
//
// Design
//   * workgroup = 8 waves = HPW query heads of one kv head x (256 / HPW) query rows; a wave owns 32 query rows of one head.
//     HPW = the largest of {8, 4, 2, 1} dividing G.  Linear block id -> kv head fastest (8 kv heads = 8 XCDs: every
//     workgroup that reads a kv head's K/V runs on one XCD, next to its L2), heaviest (last) query blocks first.
//   * K/V tiles of 64 keys go global -> registers -> LDS, issued one tile ahead (registers filled while the current tile
//     is computed, written to the other LDS buffer afterwards: one barrier per tile); ONE LDS image serves the row reads
//     (ds_read_b128, K as the A operand) and the transposed reads (ds_read_b64_tr_b16, V^T as the A operand), both
//     conflict-free (pf_off: cdna_hip_programming.md T10 image (b) for d = 128, a searched swizzle for d = 64).
//   * everything is computed TRANSPOSED so that a query row lives on a lane: S^T = K Q^T (A = K rows from LDS, B = Q^T
//     from registers, v_mfma_f32_32x32x16_f16): lane (q, h) holds 16 of a 32-key tile's scores of query q; softmax is
//     in-lane plus ONE half-wave exchange; P^T, converted pairwise to fp16, IS the B operand of O^T += V^T P^T (the
//     accumulator's row index is the next product's reduction index: no lane movement, no LDS), and the running rescale
//     of O^T is lane-local.  fp32 online softmax in the exp2 domain, fp32 accumulation, fp16 output.
//   * causal: a workgroup walks the key tiles up to its last query row's diagonal; a wave skips tiles wholly above its own
//     rows and masks only the tiles its diagonal crosses.
//
// Roofline: MFMA (2.5 PFLOP/s dense fp16).  FLOPs = 4 d nh (number of unmasked (i, j) pairs).  Per 64-key tile a wave
// issues 32 MFMAs (32 cycles each) and reads 32 KiB of LDS (K and V^T fragments are re-read by each of the 8 waves: 256
// B/clk/CU at the MFMA rate, i.e. the LDS ceiling equals the MFMA ceiling in this 8 x 32-row decomposition).
#include <cstdlib>
#include <type_traits>

#include "common.h"

#include <utility>
#include "dev_switches.h"  // MILLION_EXP: development ablation switches (tools/ab_build.py, tools/pf_ab.sh); 0 in the product build:
                           // 1 no exponentials, 2 no barrier, 4 no PV MFMAs, 8 no QK MFMAs, 16 no global -> LDS staging

namespace million {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned pv4u __attribute__((ext_vector_type(4)));
typedef short pv4s __attribute__((__vector_size__(4 * sizeof(short))));

struct PrefillParams {
    const f16 *q, *k, *v;
    f16 *out;
    int bs, nh, nh_k, G, d;
    int n_q, n_kv, q_pos0, causal;
    long long q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, o_sb, o_sh, o_sn;      // strides in elements; d contiguous
    int hpw;            // query heads per workgroup (1, 2, 4, 8)
    int n_qb;           // query blocks per head (of 256 / hpw rows)
    float scale_log2e;
};

constexpr int kPWDefault = 8;     // waves per workgroup (template parameter kPW of the kernel)
constexpr int kKV = 64;           // keys per tile

// LDS image of a [64 keys][D] fp16 tile; off(row, ch) = byte offset of 16-byte chunk ch of a row.  D = 128 (256-byte rows):
// image (b) of cdna_hip_programming.md T10.  D = 64 (128-byte rows, two to a bank row): slot = ((row & 1) << 3 | ch) ^
// (((rp & 1) << 2) | ((rp >> 2) & 3)) with rp = row >> 1 - found by exhaustive search over the linear maps rp -> 4 bits for
// the one that leaves BOTH the ds_read_b128 row reads of the 32x32x16 A operand and the ds_read_b64_tr_b16 reads
// conflict-free (checked lane group by lane group against the bank rules of MI355X_MICROARCH.md, LDS).
template <int D>
__device__ __forceinline__ unsigned pf_off(int row, int ch) {
    if (D == 128) return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    const int rp = row >> 1;
    return 256u * rp + 16u * ((((row & 1) << 3) | ch) ^ (((rp & 1) << 2) | ((rp >> 2) & 3)));
}
typedef __attribute__((address_space(3))) pv4u *lds_v4u_p;
typedef __attribute__((address_space(3))) pv4s *lds_v4s_p;

// D = 128 (the Llama head size of every BASELINE config) and D = 64 (the other head size the reference builds, setup.py:12).
template <int D, int kPW>
__global__ __launch_bounds__(kPW * 64, 2) void prefill_attn_kernel(PrefillParams p) {
    constexpr int DS = D / 16;                 // k-steps of the score product
    constexpr int NB = D / 32;                 // 32-row blocks of O^T
    constexpr int kTileBytes = kKV * 2 * D;    // one [64][D] fp16 tile
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];      // [2 buffers][K tile | V tile]
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)pf_smem != 0u) __builtin_trap();      // absolute LDS addressing below
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    // ---- which rows ----
    int id = blockIdx.x;
    const int hk = id % p.nh_k;
    id /= p.nh_k;
    const int n_hg = p.G / p.hpw;
    const int hg = id % n_hg;
    id /= n_hg;
    const int qb = p.n_qb - 1 - id % p.n_qb;      // heaviest query blocks first
    const int b = id / p.n_qb;
    const int wph = kPW / p.hpw;                  // waves per head
    const int QB = wph * 32;                      // query rows per head in this workgroup
    const int g = hg * p.hpw + wave / wph;        // query head within the kv head's group
    const int head = hk * p.G + g;
    const int q_lo = qb * QB + (wave % wph) * 32; // first query row of this wave
    const int q_row = q_lo + r32;                 // this lane's query row
    const int q_pos = p.q_pos0 + q_row;           // its position among the keys (causal: keys <= q_pos)

    // ---- Q^T fragments: B operand, lane (q, h): Q[q][16 s + 8 h .. + 8] ----
    v8h qf[DS];
    {
        const int qr = q_row < p.n_q ? q_row : p.n_q - 1;
        const f16 *qp = p.q + b * p.q_sb + head * p.q_sh + (long long)qr * p.q_sn + 8 * hh;
#pragma unroll
        for (int s = 0; s < DS; ++s) qf[s] = *(const v8h *)(qp + 16 * s);
    }
    // ---- key tiles of this workgroup / of this wave ----
    const int wg_q_hi = qb * QB + QB - 1 < p.n_q - 1 ? qb * QB + QB - 1 : p.n_q - 1;      // last query row of the workgroup
    int kv_end_wg = p.causal ? p.q_pos0 + wg_q_hi + 1 : p.n_kv;
    kv_end_wg = kv_end_wg < p.n_kv ? kv_end_wg : p.n_kv;
    const int nt = kv_end_wg > 0 ? (kv_end_wg + kKV - 1) / kKV : 0;
    const int w_pos_lo = p.q_pos0 + q_lo, w_pos_hi = p.q_pos0 + q_lo + 31;                // positions of this wave's rows
    const bool wave_live = q_lo < p.n_q;                                                   // wave-uniform

    // ---- staging: global -> LDS directly (global_load_lds_dwordx4: no registers, no ds_write).  A wave instruction moves
    //      64 x 16 bytes into 1 KiB of CONSECUTIVE LDS, so the swizzle of the image is applied on the global side: lane l of
    //      piece j fills 16-byte slot 64 j + l of a tile and fetches the (row, chunk) that pf_off puts there.  The ablation
    //      of the register-staged form priced staging at 1.3 of 9.2 ms (address arithmetic, 4 global loads + 4 ds_write_b128
    //      per thread and tile, the wait in front of the stores).  The instruction is issued from inline asm: hipcc answers the
    //      builtin with s_waitcnt vmcnt(0) in front of every later LDS read; the waits are explicit (dma_wait) instead. ----
    const f16 *kbase = p.k + b * p.k_sb + hk * p.k_sh;
    const f16 *vbase = p.v + b * p.v_sb + hk * p.v_sh;
    constexpr int kPieces = kTileBytes / 1024;          // 1-KiB pieces per tile side (16 / 8)
    constexpr int NPW = 2 * kPieces / kPW;              // pieces per wave and tile: K and V (4 / 2)
    int prow[NPW], pch[NPW];                            // the (row, 16-byte chunk) this lane fetches for its piece i
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int pc = (wave + kPW * i) % kPieces;      // piece within its side
        const int pos = 64 * pc + lane;                 // 16-byte slot of the tile image
        if (D == 128) {
            prow[i] = pos >> 4;
            pch[i] = (pos & 15) ^ (((prow[i] & 3) << 2) | ((prow[i] >> 2) & 3));
        } else {
            const int rp = pos >> 4, x = (pos & 15) ^ (((rp & 1) << 2) | ((rp >> 2) & 3));
            prow[i] = 2 * rp + (x >> 3);
            pch[i] = x & 7;
        }
    }
    auto dma_tile = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pcg = wave + kPW * i;             // wave-uniform: pieces [0, kPieces) are K, the rest V
            const bool is_v = pcg >= kPieces;
            int kvr = t * kKV + prow[i];
            kvr = kvr < p.n_kv ? kvr : p.n_kv - 1;      // clamped: rows past the end are masked below
            const f16 *src = (is_v ? vbase + (long long)kvr * p.v_sn : kbase + (long long)kvr * p.k_sn) + 8 * pch[i];
            const unsigned dst = 2u * kTileBytes * buf + (is_v ? kTileBytes : 0) + 1024u * (pcg % kPieces);
            asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    v16f O[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[i][j] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;      // m in the scaled exp2 domain
    const float c = p.scale_log2e;

    if (nt > 0) dma_tile(0, 0);
    dma_wait();
    __syncthreads();
    // One key tile.  The buffer index is a compile-time constant (the tile loop below is unrolled by two) and the reads are
    // written as lane pointer + constant element index: the buffer's base then rides in the immediate offset of every
    // ds_read instead of one v_or per read (48 per tile; written as integer arithmetic hipcc hoisted a second set of 48
    // address registers instead and spilled).
    auto tile = [&](auto bufc, const int t) {
        constexpr int buf = decltype(bufc)::value;
#if !(MILLION_EXP & 16)
        // the other buffer was last read in iteration t - 1 and every wave has passed that iteration's barrier: the next
        // tile's bytes fly into it during this tile's products
        if (t + 1 < nt) dma_tile(t + 1, buf ^ 1);
#endif
        const int kv0 = t * kKV;
        const bool tile_live = wave_live && (!p.causal || kv0 <= w_pos_hi);      // wave-uniform
        if (tile_live) {
            constexpr unsigned kb = 2u * kTileBytes * buf, vb = kb + kTileBytes;
            // ---- S^T = K Q^T: two 32-key x 32-query tiles ----
            v16f S0, S1;
#pragma unroll
            for (int j = 0; j < 16; ++j) { S0[j] = 0.f; S1[j] = 0.f; }
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const v8h a0 = __builtin_bit_cast(v8h, ((lds_v4u_p)(size_t)pf_off<D>(r32, 2 * s + hh))[kb / 16]);
                const v8h a1 = __builtin_bit_cast(v8h, ((lds_v4u_p)(size_t)pf_off<D>(32 + r32, 2 * s + hh))[kb / 16]);
#if MILLION_EXP & 8
                S0[s] += (float)a0[0] * (float)qf[s][0];
                S1[s] += (float)a1[0] * (float)qf[s][0];
#else
                S0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[s], S0, 0, 0, 0);
                S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[s], S1, 0, 0, 0);
#endif
            }
            // ---- mask (only where the diagonal or the end of the keys crosses this tile), scale, online softmax ----
            float sc[32];
#pragma unroll
            for (int j = 0; j < 16; ++j) { sc[j] = S0[j]; sc[16 + j] = S1[j]; }
            const bool need_mask = (p.causal && kv0 + kKV - 1 > w_pos_lo) || kv0 + kKV > p.n_kv;      // wave-uniform
            if (need_mask) {
                const int lim = p.causal ? (q_pos < p.n_kv - 1 ? q_pos : p.n_kv - 1) : p.n_kv - 1;     // last key this row attends to
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int kv = kv0 + 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * hh;
                    sc[j] = kv <= lim ? sc[j] : -INFINITY;
                }
            }
            float mx = sc[0];
#pragma unroll
            for (int j = 1; j < 32; ++j) mx = fmaxf(mx, sc[j]);
            {
                const v2u ex = swap32_self(__float_as_uint(mx));      // both halves of the wave: the same query rows
                const unsigned e0 = ex[0], e1 = ex[1];
                mx = fmaxf(__uint_as_float(e0), __uint_as_float(e1));
            }
            const float m_new = fmaxf(m_run, mx * c);
            const float m_safe = m_new > -INFINITY ? m_new : 0.f;
            if (__any(m_new > m_run && m_run > -INFINITY)) {      // some row's maximum moved: rescale (lane-local: a row is a lane)
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) O[i][j] *= alpha;
                l_run *= alpha;
            }
            m_run = m_new;
            float ls = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
#if MILLION_EXP & 1
                sc[j] = fmaf(sc[j], c, -m_safe);
#else
                sc[j] = __builtin_amdgcn_exp2f(fmaf(sc[j], c, -m_safe));
#endif
                ls += sc[j];
            }
            l_run += ls;
            // ---- O^T += V^T P^T: P^T registers 8 ks .. 8 ks + 7 of a score tile are k-step ks of the B operand ----
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                    unsigned pw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const h2v t2 = {(f16)sc[16 * jt + 8 * ks + 2 * e], (f16)sc[16 * jt + 8 * ks + 2 * e + 1]};
                        pw[e] = __builtin_bit_cast(unsigned, t2);
                    }
                    const pv4u pwv = {pw[0], pw[1], pw[2], pw[3]};
                    const v8h pb = __builtin_bit_cast(v8h, pwv);
                    // A = V^T: element e of lane half h is key 32 jt + 16 ks + 8 (e >> 2) + 4 h + (e & 3), row = value dim
                    const int kvr0 = 32 * jt + 16 * ks + 4 * hh;
                    const int qd = (lane >> 2) & 3, pp = lane & 3, g16 = (lane >> 4) & 1;      // lane 4 qd + pp of its 16-lane group
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk) {
                        const int chn = 4 * blk + 2 * g16 + (pp >> 1);
                        const pv4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(size_t)(pf_off<D>(kvr0 + qd, chn) + 8 * (pp & 1)) + vb / 8);
                        const pv4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(size_t)(pf_off<D>(kvr0 + 8 + qd, chn) + 8 * (pp & 1)) + vb / 8);
                        typedef short v8s __attribute__((ext_vector_type(8)));
                        const v8s av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#if MILLION_EXP & 4
                        O[blk][0] += (float)av[0] * (float)pb[0];
#else
                        O[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, av), pb, O[blk], 0, 0, 0);
#endif
                    }
                }
        }
        dma_wait();
#if !(MILLION_EXP & 2)
        __syncthreads();
#endif
    };
    for (int t = 0; t < nt; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nt) tile(std::integral_constant<int, 1>{}, t + 1);
    }
    // ---- normalise and store: lane (q, h) holds dims 32 blk + 8 i + 4 h + (0..3) of its row ----
    {
        const v2u ex = swap32_self(__float_as_uint(l_run));
        const unsigned e0 = ex[0], e1 = ex[1];
        l_run = __uint_as_float(e0) + __uint_as_float(e1);
    }
    if (wave_live && q_row < p.n_q) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        f16 *op = p.out + b * p.o_sb + head * p.o_sh + (long long)q_row * p.o_sn + 4 * hh;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef f16 h4 __attribute__((ext_vector_type(4)));
                const h4 o = {(f16)(O[blk][4 * i] * inv), (f16)(O[blk][4 * i + 1] * inv), (f16)(O[blk][4 * i + 2] * inv),
                              (f16)(O[blk][4 * i + 3] * inv)};
                *(h4 *)(op + 32 * blk + 8 * i) = o;
            }
    }
}

// =====================================================================================================
// Pipelined form (round 5; d = 128, 8 waves): the same tiles, layouts and numerics as prefill_attn_kernel above, but a wave's
// instruction stream is software-pipelined across 32-key HALF tiles and every matrix instruction is PINNED where it is written:
//   half-step h:   phase 1   the 8 score products of half h + 1 (K operands requested three steps ahead), each followed by the
//                            exponentials / row sums / fp16 pack of TWO scores of half h; its last three gaps request phase 2's
//                            first three V operands;
//                  phase 2   the 8 value products of half h (V operands three steps ahead), each followed by one v_max3 of the running maximum over half h + 1's
//                            raw scores; then the reference decision for half h + 1 (lazy reference as in the decode kernels: it
//                            moves - rescale of O and l, a wave-uniform branch - only when a score exceeds it by 2^8: P <= 2^8 is an
//                            exact fp16 operand; the rescale sits behind ALL value products of half h, cdna_hip_programming.md T13).
// Whole-tile pipelining (16 + 16 products per phase) holds both score halves of two tiles, 16 probability pairs and their operands:
// 256 registers and 41-127 spilled dwords - qf and the fragment addresses reloaded in front of every product with vmcnt(0) waits
// that also wait for the tile DMA: 646 TFLOP/s against 979 for the plain kernel.  Half tiles keep 16 + 16 score registers.
// The K stream runs half a tile ahead of V: iteration t reads K halves 2 t + 1, 2 t + 2 and V tile t, and its DMA brings K halves
// 2 t + 3, 2 t + 4 and V tile t + 1 (K half h sits in quarter h & 3 of the K region = tile (h >> 1) & 1's buffer: nothing moves).
// Round 4 built a cross-tile pipeline at source level and measured no gain: hipcc kept 'ds_read, s_waitcnt, v_mfma' triples and runs
// of 30-100 vector instructions - an MFMA is register-only and has no chain to sched_barrier() in the selection DAG, so the compiler
// moves it at will (seen again in the lean decode kernel this round).  Here an empty asm volatile("" : "+v"(accumulator)) behind
// every product and behind every softmax slice fixes the order: M, 7 vector, M, 7 vector ... as written.
// =====================================================================================================
#define PF_PIN(x) asm volatile("" : "+v"(x))
__global__ __launch_bounds__(8 * 64, 2) void prefill_attn_pipe_kernel(PrefillParams p) {
    constexpr int D = 128, kPW = 8, DS = D / 16, NB = D / 32, kTileBytes = kKV * 2 * D;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)pf_smem != 0u) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    int id = blockIdx.x;
    const int hk = id % p.nh_k;
    id /= p.nh_k;
    const int n_hg = p.G / p.hpw;
    const int hg = id % n_hg;
    id /= n_hg;
    const int qb = p.n_qb - 1 - id % p.n_qb;
    const int b = id / p.n_qb;
    const int wph = kPW / p.hpw;
    const int QB = wph * 32;
    const int g = hg * p.hpw + wave / wph;
    const int head = hk * p.G + g;
    const int q_lo = qb * QB + (wave % wph) * 32;
    const int q_row = q_lo + r32;
    const int q_pos = p.q_pos0 + q_row;
    v8h qf[DS];
    {
        const int qr = q_row < p.n_q ? q_row : p.n_q - 1;
        const f16 *qp = p.q + b * p.q_sb + head * p.q_sh + (long long)qr * p.q_sn + 8 * hh;
#pragma unroll
        for (int s = 0; s < DS; ++s) qf[s] = *(const v8h *)(qp + 16 * s);
        // Q carries the softmax scale and log2 e (one more fp16 rounding of Q: rel-L2 against fp32 1.6e-4 -> 1.8e-4, bar 1e-3), and
        // a half's score products start from -reference instead of 0: the accumulator IS the exponent, p = exp2(S') with no
        // fused multiply-add per score (16 of ~164 vector instructions per half and wave; +2 % measured, profiles/r05_prefill.txt)
#pragma unroll
        for (int s = 0; s < DS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = (f16)((float)qf[s][e] * p.scale_log2e);
    }
    const int wg_q_hi = qb * QB + QB - 1 < p.n_q - 1 ? qb * QB + QB - 1 : p.n_q - 1;
    int kv_end_wg = p.causal ? p.q_pos0 + wg_q_hi + 1 : p.n_kv;
    kv_end_wg = kv_end_wg < p.n_kv ? kv_end_wg : p.n_kv;
    const int nt = kv_end_wg > 0 ? (kv_end_wg + kKV - 1) / kKV : 0;
    const int nh2 = 2 * nt;                                              // 32-key halves of this workgroup
    const int w_pos_lo = p.q_pos0 + q_lo, w_pos_hi = p.q_pos0 + q_lo + 31;
#if MILLION_EXP & 16384
    const bool wave_live = q_lo < p.n_q && wave < 4;      // diagnostic: one computing wave per SIMD (its partner only issues DMA and joins barriers)
#else
    const bool wave_live = q_lo < p.n_q;
#endif
    const f16 *kbase = p.k + b * p.k_sb + hk * p.k_sh;
    const f16 *vbase = p.v + b * p.v_sb + hk * p.v_sh;
    // one-KiB pieces (4 rows of a tile): a K half = pieces 8 jt .. 8 jt + 7 of its tile: one per wave; a V tile = 16: two per wave
    int prow[2], pch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = 64 * (wave + kPW * i) + lane;
        prow[i] = pos >> 4;
        pch[i] = (pos & 15) ^ (((prow[i] & 3) << 2) | ((prow[i] >> 2) & 3));
    }
    auto dma_piece = [&](int t, int i, bool is_v) {      // piece wave + 8 i of tile t's K or V side -> buffer t & 1
        int kvr = t * kKV + prow[i];
        kvr = kvr < p.n_kv ? kvr : p.n_kv - 1;
        const f16 *src = (is_v ? vbase + (long long)kvr * p.v_sn : kbase + (long long)kvr * p.k_sn) + 8 * pch[i];
        const unsigned dst = 2u * kTileBytes * (t & 1) + (is_v ? kTileBytes : 0) + 1024u * (wave + kPW * i);
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
    };
    auto dma_k_half = [&](int h) { if (h < nh2) dma_piece(h >> 1, h & 1, false); };
    auto dma_v_tile = [&](int t) { if (t < nt) { dma_piece(t, 0, true); dma_piece(t, 1, true); } };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    v16f O[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[i][j] = 0.f;
    float m_ref = -INFINITY, neg_ref = 0.f, thr_rel = -INFINITY, l_run = 0.f;      // thr_rel: 8 once a reference exists (scores are relative to it)
    const int qd = (lane >> 2) & 3, pp = lane & 3, g16 = (lane >> 4) & 1;

    // wave-uniform predicates of half h (keys 32 h .. 32 h + 31) for this wave
    auto live = [&](int h) { return h < nh2 && wave_live && (!p.causal || 32 * h <= w_pos_hi); };
    auto masked = [&](int h) { return (p.causal && 32 * h + 31 > w_pos_lo) || 32 * h + 32 > p.n_kv; };
    // raw scores of half h -> -inf where the row does not attend; compares against immediates: key = 32 h + 4 hh + (j & 3) + 8 (j >> 2) <= lim
    auto mask_half = [&](int h, v16f &S) {
        const int lim = p.causal ? (q_pos < p.n_kv - 1 ? q_pos : p.n_kv - 1) : p.n_kv - 1;
        const int rel = lim - 32 * h - 4 * hh;
#pragma unroll
        for (int j = 0; j < 16; ++j) S[j] = (j & 3) + 8 * (j >> 2) <= rel ? S[j] : -INFINITY;
    };
    v16f NEG;      // -reference of the lane's query row in every register: the initial accumulator of a half's score products
#pragma unroll
    for (int j = 0; j < 16; ++j) NEG[j] = 0.f;
    // the reference decision for a half whose per-lane maximum (relative to the current reference) is mx (both half-waves hold the
    // same query rows); Nx = that half's scores, already computed against the old reference: they move with it
    auto decide = [&](float mx, v16f &Nx) {
        {
            const v2u ex = swap32_self(__float_as_uint(mx));
            const unsigned e0 = ex[0], e1 = ex[1];
            mx = fmaxf(__uint_as_float(e0), __uint_as_float(e1));
        }
        if (__any(mx > thr_rel)) {
            const float m_new = fmaxf(m_ref, mx - neg_ref);      // mx is relative to the old reference (neg_ref = -m_safe)
            const float m_safe = m_new > -INFINITY ? m_new : 0.f;
            const float alpha = __builtin_amdgcn_exp2f(m_ref - m_safe);
#pragma unroll
            for (int i = 0; i < NB; ++i)      // alpha = 0 on the first move (O = 0 then), 1 when only another row's maximum moved
#pragma unroll
                for (int j = 0; j < 16; ++j) O[i][j] *= alpha;
            l_run *= alpha;
            m_ref = m_new;
            const float shift = -m_safe - neg_ref;      // the scores already computed move to the new reference
#pragma unroll
            for (int j = 0; j < 16; ++j) { Nx[j] += shift; NEG[j] = -m_safe; }
            neg_ref = -m_safe;
            thr_rel = 8.0f;
        }
    };
    // Fragment addresses as 8 + 8 lane constants with everything else in the ds_read immediates (the plain kernel keeps 16 + 32
    // address registers live across its loop; the pipeline has no room for them).  pf_off(row, ch) = 256 row + 16 (ch ^ X(row)) with
    // X = ((row & 3) << 2) | ((row >> 2) & 3):
    //   K rows 32 jt + r32, chunk 2 s + hh: X depends on r32 only, so  addr = ka[s] + 256 * 32 jt,  ka[s] = 256 r32 + 16 ((2 s + hh) ^ X(r32))
    //   V rows 32 jt + 16 ks + 8 hi + 4 hh + qd, chunk 4 blk + 2 g16 + (pp >> 1): row & 3 = qd, (row >> 2) & 3 = 2 hi + hh, so
    //   addr = va[hi][blk] + 256 (32 jt + 16 ks),  va[hi][blk] = 256 (8 hi + 4 hh + qd) + 16 (((blk ^ qd) << 2) | ((2 g16 + (pp >> 1)) ^ (2 hi + hh))) + 8 (pp & 1)
    unsigned ka[DS], va[2][NB];
    {
        const unsigned X = ((r32 & 3) << 2) | ((r32 >> 2) & 3);
#pragma unroll
        for (int s_ = 0; s_ < DS; ++s_) ka[s_] = 256u * r32 + 16u * ((2u * s_ + hh) ^ X);
#pragma unroll
        for (int hi_ = 0; hi_ < 2; ++hi_)
#pragma unroll
            for (int blk_ = 0; blk_ < NB; ++blk_)
                va[hi_][blk_] = 256u * (8 * hi_ + 4 * hh + qd) + 16u * ((((unsigned)blk_ ^ qd) << 2) | ((2u * g16 + (pp >> 1)) ^ (2u * hi_ + hh))) + 8u * (pp & 1);
    }
    // K half HQ (0..3: buffer HQ >> 1, rows 32 (HQ & 1) ..), k-step S;  V of buffer BUFV, half JT, key step KS, operand half HI, dim block BLK
#define PF_KFRAG(HQ, S) __builtin_bit_cast(v8h, ((lds_v4u_p)(size_t)ka[S])[(2u * kTileBytes * ((HQ) >> 1) + 256u * 32u * ((HQ) & 1)) / 16])
#define PF_VFRAG(BUFV, JT, KS, HI, BLK) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(size_t)va[HI][BLK] + (2u * kTileBytes * (BUFV) + kTileBytes + 256u * (32u * (JT) + 16u * (KS))) / 8)

#if MILLION_EXP & 2048
    // phase clock of this wave (tools/prefill_prof.py): cycles since the previous stamp are added to slot I.  s_memtime needs
    // lgkmcnt(0), so a stamp also drains the wave's LDS reads - the profile runs a few percent slow and over-states the slot behind
    // a stamp that cuts a read-ahead.  Slots: 0 DMA issue, 1 phase 1, 2 mask, 3 phase 2, 4 reference decision, 5 DMA wait + barrier
    unsigned long long pt_last = __builtin_readcyclecounter();
    unsigned pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PT(I) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_acc[I] += (unsigned)(n_ - pt_last); pt_last = n_; }
#else
#define PT(I)
#endif
    // prologue: K halves 0, 1, 2 and V tile 0
    dma_k_half(0); dma_k_half(1); dma_k_half(2); dma_v_tile(0);
#pragma unroll
    for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));      // hipcc's vmcnt model: Q has landed here, not 'somewhere in the loop'
    dma_wait();
    __syncthreads();
    v16f SA, SB;      // raw scores: of the current half / accumulating the next one (roles swap every half-step)
    bool cur_live = live(0);
    if (cur_live) {      // scores of half 0 (plain), mask, reference
#pragma unroll
        for (int s = 0; s < DS; ++s)
            SA = s == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(PF_KFRAG(0, s), qf[s], v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
                        : __builtin_amdgcn_mfma_f32_32x32x16_f16(PF_KFRAG(0, s), qf[s], SA, 0, 0, 0);
        if (masked(0)) mask_half(0, SA);
        float mx = SA[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, SA[j]);
        decide(mx, SA);
    }
    // One half-step: half h (scores in C, reference decided; its V rows: buffer HQ >> 1, half HQ & 1) and half h + 1 (scores into N;
    // its K rows: quarter (HQ + 1) & 3).  HQ = h & 3 is a compile-time constant: the tile loop is unrolled by two.
    // iteration t: half-steps 2 t and 2 t + 1; its DMA: K halves 2 t + 3, 2 t + 4 (quarters last read in iteration t - 1) and V tile t + 1
#if MILLION_EXP & 512
#define PF_DMA(T) (void)0
#else
#define PF_DMA(T) dma_k_half(2 * (T) + 3); dma_k_half(2 * (T) + 4); dma_v_tile((T) + 1)
#endif
#if MILLION_EXP & 1024
#define PF_SYNC() (void)0
#else
#define PF_SYNC() dma_wait(); __syncthreads()
#endif
    // (Tried, no change: an iteration's first half-step requesting its first three K operands BEFORE it issues the iteration's DMA
    // pieces, and the second half-step's in the first one's last gaps - the LDS round trip at a phase's start is not what the
    // loop waits for: 1060 vs 1062 TFLOP/s, profiles/r05_prefill.txt.)
    // (Tried, slower: a wave's four DMA pieces issued inside the MFMA gaps of the iteration's first half-step, two per phase, at
    // wave-dependent gaps - the issue stall moves into the phases and grows: phase 1 732 -> 982 cycles, phase 2 330 -> 488 per half-step
    // for 250 saved: 1024 vs 1055 TFLOP/s, profiles/r05_prefill.txt.  A burst outside the phases is the cheapest place.)
    auto half_step = [&](auto hqc, v16f &C, v16f &N, const int h) {
        constexpr int HQ = decltype(hqc)::value, HN = (HQ + 1) & 3, BUFV = HQ >> 1, JT = HQ & 1;
        const bool nxt_live = live(h + 1);

        if (cur_live) {      // a wave's last live half runs the same code: the scores of the half behind it are masked whole
            float ls = 0.f, mx = -INFINITY;
            unsigned pw8[8];
            // phase 1: score products of half h + 1 (operands three k-steps ahead) | exponentials of half h; its last three gaps
            // also request the first three value operands of phase 2
            v8h af[3] = {PF_KFRAG(HN, 0), PF_KFRAG(HN, 1), PF_KFRAG(HN, 2)};
            pv4s lo[3], hi[3];
#pragma unroll
            for (int s = 0; s < DS; ++s) {
#if MILLION_EXP & 8
                if (s == 0) N = NEG;      // ablation: no score products (the operand reads stay: kept alive below)
                N[s] += (float)af[s % 3][0];
#else
                if (s == 0) N = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s % 3], qf[s], NEG, 0, 0, 0);
                else N = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s % 3], qf[s], N, 0, 0, 0);
#endif
                PF_PIN(N);
                if (s + 3 < DS) af[s % 3] = PF_KFRAG(HN, (s + 3) & 7);
                else { lo[s - 5] = PF_VFRAG(BUFV, JT, 0, 0, (s - 5) & 3); hi[s - 5] = PF_VFRAG(BUFV, JT, 0, 1, (s - 5) & 3); }
#if MILLION_EXP & 256
                const float p0 = C[2 * s] + 1.0f, p1 = C[2 * s + 1] + 1.0f;
#else
                const float p0 = __builtin_amdgcn_exp2f(C[2 * s]), p1 = __builtin_amdgcn_exp2f(C[2 * s + 1]);
#endif
                ls += p0;
                ls += p1;
                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                const h2v t2 = {(f16)p0, (f16)p1};
                pw8[s] = __builtin_bit_cast(unsigned, t2);
                asm volatile("" : "+v"(pw8[s]), "+v"(ls));
            }
            PT(1)
            if (!nxt_live || masked(h + 1)) mask_half(h + 1, N);
            PT(2)
            // phase 2: value products of half h (operands three steps ahead) | running maximum of half h + 1's raw scores
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ks = i >> 2, blk = i & 3;
                typedef short v8s __attribute__((ext_vector_type(8)));
                const v8s av = {lo[i % 3][0], lo[i % 3][1], lo[i % 3][2], lo[i % 3][3], hi[i % 3][0], hi[i % 3][1], hi[i % 3][2], hi[i % 3][3]};
                const pv4u pwv = {pw8[4 * ks], pw8[4 * ks + 1], pw8[4 * ks + 2], pw8[4 * ks + 3]};
#if MILLION_EXP & 4
                O[blk][i] += __builtin_bit_cast(float, ((unsigned)(unsigned short)av[0] | pwv[0]) & 1u);      // ablation: no value products
#else
                O[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, av), __builtin_bit_cast(v8h, pwv), O[blk], 0, 0, 0);
#endif
                PF_PIN(O[blk]);
                if (i + 3 < 8) { lo[i % 3] = PF_VFRAG(BUFV, JT, ((i + 3) >> 2) & 1, 0, (i + 3) & 3); hi[i % 3] = PF_VFRAG(BUFV, JT, ((i + 3) >> 2) & 1, 1, (i + 3) & 3); }
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(N[2 * i]), "v"(N[2 * i + 1]));
                PF_PIN(mx);
            }
            PT(3)
            l_run += ls;
            decide(mx, N);
            PT(4)
        }
        cur_live = nxt_live;
    };
    PT(7)
    for (int t = 0; t < nt; t += 2) {
        PF_DMA(t);
        PT(0)
        half_step(std::integral_constant<int, 0>{}, SA, SB, 2 * t);
        half_step(std::integral_constant<int, 1>{}, SB, SA, 2 * t + 1);
        PF_SYNC();
        PT(5)
        if (t + 1 < nt) {
            PF_DMA(t + 1);
            PT(0)
            half_step(std::integral_constant<int, 2>{}, SA, SB, 2 * t + 2);
            half_step(std::integral_constant<int, 3>{}, SB, SA, 2 * t + 3);
            PF_SYNC();
            PT(5)
        }
    }
#if MILLION_EXP & 2048
    if (blockIdx.x == 0 && lane < 8) {      // into this wave's own first query row (read into registers long ago; nobody else reads it)
        unsigned *dst = (unsigned *)(p.q + b * p.q_sb + head * p.q_sh + (long long)q_lo * p.q_sn);
        unsigned v_ = pt_acc[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) v_ = lane == i ? pt_acc[i] : v_;
        dst[lane] = v_;
    }
#endif
#undef PT
#undef PF_DMA
#undef PF_SYNC
#undef PF_KFRAG
#undef PF_VFRAG
    {
        const v2u ex = swap32_self(__float_as_uint(l_run));
        const unsigned e0 = ex[0], e1 = ex[1];
        l_run = __uint_as_float(e0) + __uint_as_float(e1);
    }
    if (wave_live && q_row < p.n_q) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        f16 *op = p.out + b * p.o_sb + head * p.o_sh + (long long)q_row * p.o_sn + 4 * hh;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                typedef f16 h4 __attribute__((ext_vector_type(4)));
                const h4 o = {(f16)(O[blk][4 * i] * inv), (f16)(O[blk][4 * i + 1] * inv), (f16)(O[blk][4 * i + 2] * inv),
                              (f16)(O[blk][4 * i + 3] * inv)};
                *(h4 *)(op + 32 * blk + 8 * i) = o;
            }
    }
}
#undef PF_PIN

#ifdef MILLION_DEV_BUILD
// =====================================================================================================
// Experiment (development builds only; million_set_force_generic(128)): the pipelined kernel with FOUR waves of 64 query rows - one wave
// per SIMD, two 32-row blocks per wave sharing every K / V operand read (half the LDS reads per product), two independent chains for
// the in-order wave to interleave.  Same tiles, LDS image, DMA stream, numerics.  profiles/r05_prefill.txt section 10.
// =====================================================================================================
template <class F, int... I> __device__ __forceinline__ void w64_sfor_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void w64_sfor(F &&f) { w64_sfor_impl(f, std::make_integer_sequence<int, N>{}); }      // f(integral_constant<int, 0>) .. f(<N - 1>)
// Accumulator-file registers OWNED by the asm of prefill_attn_w64_kernel (named literally, listed as clobbers where written; the compiler
// never allocates AGPRs of its own there - audited in the .s: no v_accvgpr_* outside ;;#ASMSTART / ;;#ASMEND):
//   a[16 K .. 16 K + 15], K = 4 j + blk < 8: the output tile O[j][blk];   a[128 + 4 Q .. + 3], Q = 8 j + s < 16: the query operand qf[j][s]
__device__ __forceinline__ void w64_o_zero() { asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"); }
template <int Q> __device__ __forceinline__ void w64_q_store(pv4u w) {
    if constexpr (Q == 0) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a128, %0\n\tv_accvgpr_write_b32 a129, %1\n\tv_accvgpr_write_b32 a130, %2\n\tv_accvgpr_write_b32 a131, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a128", "a129", "a130", "a131");
    else if constexpr (Q == 1) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a132, %0\n\tv_accvgpr_write_b32 a133, %1\n\tv_accvgpr_write_b32 a134, %2\n\tv_accvgpr_write_b32 a135, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a132", "a133", "a134", "a135");
    else if constexpr (Q == 2) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %1\n\tv_accvgpr_write_b32 a138, %2\n\tv_accvgpr_write_b32 a139, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a136", "a137", "a138", "a139");
    else if constexpr (Q == 3) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a140, %0\n\tv_accvgpr_write_b32 a141, %1\n\tv_accvgpr_write_b32 a142, %2\n\tv_accvgpr_write_b32 a143, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a140", "a141", "a142", "a143");
    else if constexpr (Q == 4) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a144, %0\n\tv_accvgpr_write_b32 a145, %1\n\tv_accvgpr_write_b32 a146, %2\n\tv_accvgpr_write_b32 a147, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a144", "a145", "a146", "a147");
    else if constexpr (Q == 5) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a148, %0\n\tv_accvgpr_write_b32 a149, %1\n\tv_accvgpr_write_b32 a150, %2\n\tv_accvgpr_write_b32 a151, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a148", "a149", "a150", "a151");
    else if constexpr (Q == 6) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a152, %0\n\tv_accvgpr_write_b32 a153, %1\n\tv_accvgpr_write_b32 a154, %2\n\tv_accvgpr_write_b32 a155, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a152", "a153", "a154", "a155");
    else if constexpr (Q == 7) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a156, %0\n\tv_accvgpr_write_b32 a157, %1\n\tv_accvgpr_write_b32 a158, %2\n\tv_accvgpr_write_b32 a159, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a156", "a157", "a158", "a159");
    else if constexpr (Q == 8) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a160, %0\n\tv_accvgpr_write_b32 a161, %1\n\tv_accvgpr_write_b32 a162, %2\n\tv_accvgpr_write_b32 a163, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a160", "a161", "a162", "a163");
    else if constexpr (Q == 9) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a164, %0\n\tv_accvgpr_write_b32 a165, %1\n\tv_accvgpr_write_b32 a166, %2\n\tv_accvgpr_write_b32 a167, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a164", "a165", "a166", "a167");
    else if constexpr (Q == 10) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a168, %0\n\tv_accvgpr_write_b32 a169, %1\n\tv_accvgpr_write_b32 a170, %2\n\tv_accvgpr_write_b32 a171, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a168", "a169", "a170", "a171");
    else if constexpr (Q == 11) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a172, %0\n\tv_accvgpr_write_b32 a173, %1\n\tv_accvgpr_write_b32 a174, %2\n\tv_accvgpr_write_b32 a175, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a172", "a173", "a174", "a175");
    else if constexpr (Q == 12) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a176, %0\n\tv_accvgpr_write_b32 a177, %1\n\tv_accvgpr_write_b32 a178, %2\n\tv_accvgpr_write_b32 a179, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a176", "a177", "a178", "a179");
    else if constexpr (Q == 13) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a180, %0\n\tv_accvgpr_write_b32 a181, %1\n\tv_accvgpr_write_b32 a182, %2\n\tv_accvgpr_write_b32 a183, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a180", "a181", "a182", "a183");
    else if constexpr (Q == 14) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a184, %0\n\tv_accvgpr_write_b32 a185, %1\n\tv_accvgpr_write_b32 a186, %2\n\tv_accvgpr_write_b32 a187, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a184", "a185", "a186", "a187");
    else if constexpr (Q == 15) asm volatile("s_nop 0\n\tv_accvgpr_write_b32 a188, %0\n\tv_accvgpr_write_b32 a189, %1\n\tv_accvgpr_write_b32 a190, %2\n\tv_accvgpr_write_b32 a191, %3" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "a188", "a189", "a190", "a191");
}
template <int Q, bool FIRST> __device__ __forceinline__ void w64_qk(v16f &N, v8h af, const v16f &NEG) {
    if constexpr (Q == 0) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[128:131], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[128:131], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 1) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[132:135], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[132:135], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 2) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[136:139], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[136:139], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 3) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[140:143], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[140:143], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 4) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[144:147], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[144:147], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 5) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[148:151], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[148:151], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 6) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[152:155], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[152:155], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 7) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[156:159], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[156:159], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 8) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[160:163], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[160:163], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 9) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[164:167], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[164:167], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 10) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[168:171], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[168:171], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 11) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[172:175], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[172:175], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 12) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[176:179], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[176:179], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 13) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[180:183], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[180:183], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 14) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[184:187], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[184:187], %0" : "+v"(N) : "v"(af)); }
    else if constexpr (Q == 15) { if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[188:191], %2" : "=&v"(N) : "v"(af), "v"(NEG)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[188:191], %0" : "+v"(N) : "v"(af)); }
}
template <int K> __device__ __forceinline__ void w64_pv(v8h av, v8h pw) {
    if constexpr (K == 0) asm volatile("v_mfma_f32_32x32x16_f16 a[0:15], %0, %1, a[0:15]" :: "v"(av), "v"(pw) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    else if constexpr (K == 1) asm volatile("v_mfma_f32_32x32x16_f16 a[16:31], %0, %1, a[16:31]" :: "v"(av), "v"(pw) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    else if constexpr (K == 2) asm volatile("v_mfma_f32_32x32x16_f16 a[32:47], %0, %1, a[32:47]" :: "v"(av), "v"(pw) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    else if constexpr (K == 3) asm volatile("v_mfma_f32_32x32x16_f16 a[48:63], %0, %1, a[48:63]" :: "v"(av), "v"(pw) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    else if constexpr (K == 4) asm volatile("v_mfma_f32_32x32x16_f16 a[64:79], %0, %1, a[64:79]" :: "v"(av), "v"(pw) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    else if constexpr (K == 5) asm volatile("v_mfma_f32_32x32x16_f16 a[80:95], %0, %1, a[80:95]" :: "v"(av), "v"(pw) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    else if constexpr (K == 6) asm volatile("v_mfma_f32_32x32x16_f16 a[96:111], %0, %1, a[96:111]" :: "v"(av), "v"(pw) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    else if constexpr (K == 7) asm volatile("v_mfma_f32_32x32x16_f16 a[112:127], %0, %1, a[112:127]" :: "v"(av), "v"(pw) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
}
template <int K> __device__ __forceinline__ void w64_o_scale(float alpha) {      // O[K] *= alpha (the caller has waited for the products that wrote it)
    float t_;
    if constexpr (K == 0) asm volatile("v_accvgpr_read_b32 %0, a0\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a1\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_read_b32 %0, a2\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_read_b32 %0, a3\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a3, %0\n\tv_accvgpr_read_b32 %0, a4\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a4, %0\n\tv_accvgpr_read_b32 %0, a5\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_read_b32 %0, a6\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_read_b32 %0, a7\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a7, %0\n\tv_accvgpr_read_b32 %0, a8\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_read_b32 %0, a9\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a9, %0\n\tv_accvgpr_read_b32 %0, a10\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_read_b32 %0, a11\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a11, %0\n\tv_accvgpr_read_b32 %0, a12\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a12, %0\n\tv_accvgpr_read_b32 %0, a13\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a13, %0\n\tv_accvgpr_read_b32 %0, a14\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_read_b32 %0, a15\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a15, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    else if constexpr (K == 1) asm volatile("v_accvgpr_read_b32 %0, a16\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a16, %0\n\tv_accvgpr_read_b32 %0, a17\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a17, %0\n\tv_accvgpr_read_b32 %0, a18\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_read_b32 %0, a19\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a19, %0\n\tv_accvgpr_read_b32 %0, a20\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a20, %0\n\tv_accvgpr_read_b32 %0, a21\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a21, %0\n\tv_accvgpr_read_b32 %0, a22\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_read_b32 %0, a23\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a23, %0\n\tv_accvgpr_read_b32 %0, a24\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a24, %0\n\tv_accvgpr_read_b32 %0, a25\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a25, %0\n\tv_accvgpr_read_b32 %0, a26\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a26, %0\n\tv_accvgpr_read_b32 %0, a27\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a27, %0\n\tv_accvgpr_read_b32 %0, a28\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a28, %0\n\tv_accvgpr_read_b32 %0, a29\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a29, %0\n\tv_accvgpr_read_b32 %0, a30\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a30, %0\n\tv_accvgpr_read_b32 %0, a31\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a31, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    else if constexpr (K == 2) asm volatile("v_accvgpr_read_b32 %0, a32\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a32, %0\n\tv_accvgpr_read_b32 %0, a33\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_read_b32 %0, a34\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_read_b32 %0, a35\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a35, %0\n\tv_accvgpr_read_b32 %0, a36\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a36, %0\n\tv_accvgpr_read_b32 %0, a37\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a37, %0\n\tv_accvgpr_read_b32 %0, a38\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_read_b32 %0, a39\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a39, %0\n\tv_accvgpr_read_b32 %0, a40\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a40, %0\n\tv_accvgpr_read_b32 %0, a41\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a41, %0\n\tv_accvgpr_read_b32 %0, a42\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a42, %0\n\tv_accvgpr_read_b32 %0, a43\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a43, %0\n\tv_accvgpr_read_b32 %0, a44\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a44, %0\n\tv_accvgpr_read_b32 %0, a45\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a45, %0\n\tv_accvgpr_read_b32 %0, a46\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a46, %0\n\tv_accvgpr_read_b32 %0, a47\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a47, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    else if constexpr (K == 3) asm volatile("v_accvgpr_read_b32 %0, a48\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a48, %0\n\tv_accvgpr_read_b32 %0, a49\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a49, %0\n\tv_accvgpr_read_b32 %0, a50\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a50, %0\n\tv_accvgpr_read_b32 %0, a51\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a51, %0\n\tv_accvgpr_read_b32 %0, a52\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a52, %0\n\tv_accvgpr_read_b32 %0, a53\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a53, %0\n\tv_accvgpr_read_b32 %0, a54\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a54, %0\n\tv_accvgpr_read_b32 %0, a55\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a55, %0\n\tv_accvgpr_read_b32 %0, a56\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a56, %0\n\tv_accvgpr_read_b32 %0, a57\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a57, %0\n\tv_accvgpr_read_b32 %0, a58\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a58, %0\n\tv_accvgpr_read_b32 %0, a59\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a59, %0\n\tv_accvgpr_read_b32 %0, a60\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a60, %0\n\tv_accvgpr_read_b32 %0, a61\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a61, %0\n\tv_accvgpr_read_b32 %0, a62\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a62, %0\n\tv_accvgpr_read_b32 %0, a63\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a63, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    else if constexpr (K == 4) asm volatile("v_accvgpr_read_b32 %0, a64\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a64, %0\n\tv_accvgpr_read_b32 %0, a65\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a65, %0\n\tv_accvgpr_read_b32 %0, a66\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_read_b32 %0, a67\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a67, %0\n\tv_accvgpr_read_b32 %0, a68\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a68, %0\n\tv_accvgpr_read_b32 %0, a69\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a69, %0\n\tv_accvgpr_read_b32 %0, a70\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a70, %0\n\tv_accvgpr_read_b32 %0, a71\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a71, %0\n\tv_accvgpr_read_b32 %0, a72\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a72, %0\n\tv_accvgpr_read_b32 %0, a73\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a73, %0\n\tv_accvgpr_read_b32 %0, a74\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a74, %0\n\tv_accvgpr_read_b32 %0, a75\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a75, %0\n\tv_accvgpr_read_b32 %0, a76\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a76, %0\n\tv_accvgpr_read_b32 %0, a77\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a77, %0\n\tv_accvgpr_read_b32 %0, a78\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a78, %0\n\tv_accvgpr_read_b32 %0, a79\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a79, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    else if constexpr (K == 5) asm volatile("v_accvgpr_read_b32 %0, a80\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a80, %0\n\tv_accvgpr_read_b32 %0, a81\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a81, %0\n\tv_accvgpr_read_b32 %0, a82\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a82, %0\n\tv_accvgpr_read_b32 %0, a83\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a83, %0\n\tv_accvgpr_read_b32 %0, a84\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a84, %0\n\tv_accvgpr_read_b32 %0, a85\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a85, %0\n\tv_accvgpr_read_b32 %0, a86\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a86, %0\n\tv_accvgpr_read_b32 %0, a87\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a87, %0\n\tv_accvgpr_read_b32 %0, a88\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a88, %0\n\tv_accvgpr_read_b32 %0, a89\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a89, %0\n\tv_accvgpr_read_b32 %0, a90\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a90, %0\n\tv_accvgpr_read_b32 %0, a91\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a91, %0\n\tv_accvgpr_read_b32 %0, a92\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a92, %0\n\tv_accvgpr_read_b32 %0, a93\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a93, %0\n\tv_accvgpr_read_b32 %0, a94\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a94, %0\n\tv_accvgpr_read_b32 %0, a95\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a95, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    else if constexpr (K == 6) asm volatile("v_accvgpr_read_b32 %0, a96\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a96, %0\n\tv_accvgpr_read_b32 %0, a97\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a97, %0\n\tv_accvgpr_read_b32 %0, a98\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a98, %0\n\tv_accvgpr_read_b32 %0, a99\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a99, %0\n\tv_accvgpr_read_b32 %0, a100\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a100, %0\n\tv_accvgpr_read_b32 %0, a101\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a101, %0\n\tv_accvgpr_read_b32 %0, a102\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a102, %0\n\tv_accvgpr_read_b32 %0, a103\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a103, %0\n\tv_accvgpr_read_b32 %0, a104\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a104, %0\n\tv_accvgpr_read_b32 %0, a105\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a105, %0\n\tv_accvgpr_read_b32 %0, a106\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a106, %0\n\tv_accvgpr_read_b32 %0, a107\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a107, %0\n\tv_accvgpr_read_b32 %0, a108\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a108, %0\n\tv_accvgpr_read_b32 %0, a109\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a109, %0\n\tv_accvgpr_read_b32 %0, a110\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a110, %0\n\tv_accvgpr_read_b32 %0, a111\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a111, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    else if constexpr (K == 7) asm volatile("v_accvgpr_read_b32 %0, a112\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a112, %0\n\tv_accvgpr_read_b32 %0, a113\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a113, %0\n\tv_accvgpr_read_b32 %0, a114\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a114, %0\n\tv_accvgpr_read_b32 %0, a115\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a115, %0\n\tv_accvgpr_read_b32 %0, a116\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a116, %0\n\tv_accvgpr_read_b32 %0, a117\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a117, %0\n\tv_accvgpr_read_b32 %0, a118\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a118, %0\n\tv_accvgpr_read_b32 %0, a119\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a119, %0\n\tv_accvgpr_read_b32 %0, a120\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a120, %0\n\tv_accvgpr_read_b32 %0, a121\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a121, %0\n\tv_accvgpr_read_b32 %0, a122\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a122, %0\n\tv_accvgpr_read_b32 %0, a123\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a123, %0\n\tv_accvgpr_read_b32 %0, a124\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a124, %0\n\tv_accvgpr_read_b32 %0, a125\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a125, %0\n\tv_accvgpr_read_b32 %0, a126\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a126, %0\n\tv_accvgpr_read_b32 %0, a127\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a127, %0\n\ts_nop 1" : "=&v"(t_) : "v"(alpha) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
}
template <int K> __device__ __forceinline__ v16f w64_o_read() {
    float e_[16];
    if constexpr (K == 0) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\tv_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\tv_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    else if constexpr (K == 1) asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\tv_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\tv_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\tv_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    else if constexpr (K == 2) asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\tv_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\tv_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\tv_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\tv_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\tv_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\tv_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    else if constexpr (K == 3) asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\tv_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\tv_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\tv_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\tv_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\tv_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    else if constexpr (K == 4) asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\tv_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\tv_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\tv_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\tv_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\tv_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    else if constexpr (K == 5) asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83\n\tv_accvgpr_read_b32 %4, a84\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a86\n\tv_accvgpr_read_b32 %7, a87\n\tv_accvgpr_read_b32 %8, a88\n\tv_accvgpr_read_b32 %9, a89\n\tv_accvgpr_read_b32 %10, a90\n\tv_accvgpr_read_b32 %11, a91\n\tv_accvgpr_read_b32 %12, a92\n\tv_accvgpr_read_b32 %13, a93\n\tv_accvgpr_read_b32 %14, a94\n\tv_accvgpr_read_b32 %15, a95\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    else if constexpr (K == 6) asm volatile("v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\tv_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99\n\tv_accvgpr_read_b32 %4, a100\n\tv_accvgpr_read_b32 %5, a101\n\tv_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a103\n\tv_accvgpr_read_b32 %8, a104\n\tv_accvgpr_read_b32 %9, a105\n\tv_accvgpr_read_b32 %10, a106\n\tv_accvgpr_read_b32 %11, a107\n\tv_accvgpr_read_b32 %12, a108\n\tv_accvgpr_read_b32 %13, a109\n\tv_accvgpr_read_b32 %14, a110\n\tv_accvgpr_read_b32 %15, a111\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    else if constexpr (K == 7) asm volatile("v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\tv_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115\n\tv_accvgpr_read_b32 %4, a116\n\tv_accvgpr_read_b32 %5, a117\n\tv_accvgpr_read_b32 %6, a118\n\tv_accvgpr_read_b32 %7, a119\n\tv_accvgpr_read_b32 %8, a120\n\tv_accvgpr_read_b32 %9, a121\n\tv_accvgpr_read_b32 %10, a122\n\tv_accvgpr_read_b32 %11, a123\n\tv_accvgpr_read_b32 %12, a124\n\tv_accvgpr_read_b32 %13, a125\n\tv_accvgpr_read_b32 %14, a126\n\tv_accvgpr_read_b32 %15, a127\n\ts_nop 0" : "=v"(e_[0]), "=v"(e_[1]), "=v"(e_[2]), "=v"(e_[3]), "=v"(e_[4]), "=v"(e_[5]), "=v"(e_[6]), "=v"(e_[7]), "=v"(e_[8]), "=v"(e_[9]), "=v"(e_[10]), "=v"(e_[11]), "=v"(e_[12]), "=v"(e_[13]), "=v"(e_[14]), "=v"(e_[15]));
    v16f r_;
#pragma unroll
    for (int e = 0; e < 16; ++e) r_[e] = e_[e];
    return r_;
}
__global__ __launch_bounds__(4 * 64, 1) void prefill_attn_w64_kernel(PrefillParams p) {
    constexpr int D = 128, kPW = 4, NQ = 2, DS = D / 16, NB = D / 32, kTileBytes = kKV * 2 * D;
    extern __shared__ __attribute__((aligned(16))) char pf_smem[];
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)pf_smem != 0u) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    int id = blockIdx.x;
    const int hk = id % p.nh_k;
    id /= p.nh_k;
    const int n_hg = p.G / p.hpw;
    const int hg = id % n_hg;
    id /= n_hg;
    const int qb = p.n_qb - 1 - id % p.n_qb;
    const int b = id / p.n_qb;
    const int wph = kPW / p.hpw;                  // waves per head (hpw <= 4 here)
    const int QB = wph * 64;
    const int g = hg * p.hpw + wave / wph;
    const int head = hk * p.G + g;
    const int q_lo = qb * QB + (wave % wph) * 64;
    auto q_all = [&](auto jc, auto sc, const f16 *qp) {
        constexpr int J = decltype(jc)::value, S = decltype(sc)::value;
        v8h t = *(const v8h *)(qp + 16 * S);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (f16)((float)t[e] * p.scale_log2e);
        w64_q_store<8 * J + S>(__builtin_bit_cast(pv4u, t));
    };
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q_row = q_lo + 32 * j + r32;
        const int qr = q_row < p.n_q ? q_row : p.n_q - 1;
        const f16 *qp = p.q + b * p.q_sb + head * p.q_sh + (long long)qr * p.q_sn + 8 * hh;
        if (j == 0) { q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, qp); q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, qp); q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, qp); q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, qp);
                      q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, qp); q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, qp); q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}, qp); q_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 7>{}, qp); }
        else        { q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, qp); q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, qp); q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, qp); q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, qp);
                      q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, qp); q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 5>{}, qp); q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 6>{}, qp); q_all(std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{}, qp); }
    }
    const int wg_q_hi = qb * QB + QB - 1 < p.n_q - 1 ? qb * QB + QB - 1 : p.n_q - 1;
    int kv_end_wg = p.causal ? p.q_pos0 + wg_q_hi + 1 : p.n_kv;
    kv_end_wg = kv_end_wg < p.n_kv ? kv_end_wg : p.n_kv;
    const int nt = kv_end_wg > 0 ? (kv_end_wg + kKV - 1) / kKV : 0;
    const int nh2 = 2 * nt;
    const int w_pos_hi = p.q_pos0 + q_lo + 63;
    const bool wave_live = q_lo < p.n_q;
    const f16 *kbase = p.k + b * p.k_sb + hk * p.k_sh;
    const f16 *vbase = p.v + b * p.v_sb + hk * p.v_sh;
    // one-KiB pieces: piece wave + 4 i, i < 4: a K half = pieces 8 jt .. 8 jt + 7: two per wave (i = 2 jt, 2 jt + 1); a V tile = 16: four per wave
    int prow[4], pch[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = 64 * (wave + kPW * i) + lane;
        prow[i] = pos >> 4;
        pch[i] = (pos & 15) ^ (((prow[i] & 3) << 2) | ((prow[i] >> 2) & 3));
    }
    auto dma_piece = [&](int t, int i, bool is_v) {
        int kvr = t * kKV + prow[i];
        kvr = kvr < p.n_kv ? kvr : p.n_kv - 1;
        const f16 *src = (is_v ? vbase + (long long)kvr * p.v_sn : kbase + (long long)kvr * p.k_sn) + 8 * pch[i];
        const unsigned dst = 2u * kTileBytes * (t & 1) + (is_v ? kTileBytes : 0) + 1024u * (wave + kPW * i);
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
    };
    auto dma_k_half = [&](int h) { if (h < nh2) { dma_piece(h >> 1, 2 * (h & 1), false); dma_piece(h >> 1, 2 * (h & 1) + 1, false); } };
    auto dma_v_tile = [&](int t) { if (t < nt) { dma_piece(t, 0, true); dma_piece(t, 1, true); dma_piece(t, 2, true); dma_piece(t, 3, true); } };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // The products are inline asm on asm-owned accumulator registers (helpers above): hipcc left to itself moved ~570 registers per 16
    // products between the two halves of the register file; with "+a" / "a" constraints still 128.  Wait states are ours
    // (cdna_hip_programming.md 5.7 item 2): s_nop 1 opens every product; a score tile is read by vector code only behind W64_D_WAIT.
#define W64_D_WAIT() asm volatile("s_nop 7\n\ts_nop 7" ::: "memory")
    w64_o_zero();
    float m_ref[NQ], neg_ref[NQ], thr_rel[NQ], l_run[NQ];
    v16f NEG[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        m_ref[j] = -INFINITY; neg_ref[j] = 0.f; thr_rel[j] = -INFINITY; l_run[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) NEG[j][e] = 0.f;
    }
    const int qd = (lane >> 2) & 3, pp = lane & 3, g16 = (lane >> 4) & 1;
    auto live = [&](int h) { return h < nh2 && wave_live && (!p.causal || 32 * h <= w_pos_hi); };
    auto masked = [&](int h) { return (p.causal && 32 * h + 31 > p.q_pos0 + q_lo) || 32 * h + 32 > p.n_kv; };      // some row of the wave does not see the whole half
    auto mask_half = [&](int h, int j, v16f &S) {
        const int q_pos = p.q_pos0 + q_lo + 32 * j + r32;
        const int lim = p.causal ? (q_pos < p.n_kv - 1 ? q_pos : p.n_kv - 1) : p.n_kv - 1;
        const int rel = lim - 32 * h - 4 * hh;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = (e & 3) + 8 * (e >> 2) <= rel ? S[e] : -INFINITY;
    };
    auto decide = [&](auto jc, float mx, v16f &Nx) {
        constexpr int j = decltype(jc)::value;
        {
            const v2u ex = swap32_self(__float_as_uint(mx));
            const unsigned e0 = ex[0], e1 = ex[1];
            mx = fmaxf(__uint_as_float(e0), __uint_as_float(e1));
        }
        if (__any(mx > thr_rel[j])) {
            const float m_new = fmaxf(m_ref[j], mx - neg_ref[j]);
            const float m_safe = m_new > -INFINITY ? m_new : 0.f;
            const float alpha = __builtin_amdgcn_exp2f(m_ref[j] - m_safe);
            W64_D_WAIT();      // (the value products that wrote O may be in flight)
            w64_sfor<NB>([&](auto ic) { w64_o_scale<4 * j + decltype(ic)::value>(alpha); });
            l_run[j] *= alpha;
            m_ref[j] = m_new;
            const float shift = -m_safe - neg_ref[j];
#pragma unroll
            for (int e = 0; e < 16; ++e) { Nx[e] += shift; NEG[j][e] = -m_safe; }
            neg_ref[j] = -m_safe;
            thr_rel[j] = 8.0f;
        }
    };
    unsigned ka[DS], va[2][NB];
    {
        const unsigned X = ((r32 & 3) << 2) | ((r32 >> 2) & 3);
#pragma unroll
        for (int s_ = 0; s_ < DS; ++s_) ka[s_] = 256u * r32 + 16u * ((2u * s_ + hh) ^ X);
#pragma unroll
        for (int hi_ = 0; hi_ < 2; ++hi_)
#pragma unroll
            for (int blk_ = 0; blk_ < NB; ++blk_)
                va[hi_][blk_] = 256u * (8 * hi_ + 4 * hh + qd) + 16u * ((((unsigned)blk_ ^ qd) << 2) | ((2u * g16 + (pp >> 1)) ^ (2u * hi_ + hh))) + 8u * (pp & 1);
    }
#define PF_PIN(x) asm volatile("" : "+v"(x))
#define PF_KFRAG(HQ, S) __builtin_bit_cast(v8h, ((lds_v4u_p)(size_t)ka[S])[(2u * kTileBytes * ((HQ) >> 1) + 256u * 32u * ((HQ) & 1)) / 16])
#define PF_VFRAG(BUFV, JT, KS, HI, BLK) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(size_t)va[HI][BLK] + (2u * kTileBytes * (BUFV) + kTileBytes + 256u * (32u * (JT) + 16u * (KS))) / 8)
#if MILLION_EXP & 2048
    unsigned long long pt_last = __builtin_readcyclecounter();
    unsigned pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PT(I) { const unsigned long long n_ = __builtin_readcyclecounter(); pt_acc[I] += (unsigned)(n_ - pt_last); pt_last = n_; }
#else
#define PT(I)
#endif
    dma_k_half(0); dma_k_half(1); dma_k_half(2); dma_v_tile(0);
    dma_wait();
    __syncthreads();
    v16f SA[NQ], SB[NQ];
    bool cur_live = live(0);
    if (cur_live) {
        w64_sfor<NQ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            w64_sfor<DS>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                const v8h kf = PF_KFRAG(0, s_);
                w64_qk<8 * j + s_, s_ == 0>(SA[j], kf, NEG[j]);
            });
            W64_D_WAIT();
            if (masked(0)) mask_half(0, j, SA[j]);
            float mx = SA[j][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, SA[j][e]);
            decide(jc, mx, SA[j]);
        });
    }
    auto dma_k_half_piece = [&](int h, int jj) { if (h < nh2) dma_piece(h >> 1, 2 * (h & 1) + jj, false); };
    v8h af[3];      // K operands in flight across a phase (and across the two half-steps of an iteration)
    auto half_step = [&](auto hqc, v16f (&C)[NQ], v16f (&N)[NQ], const int h, const int t_dma) {
        constexpr int HQ = decltype(hqc)::value, HN = (HQ + 1) & 3, BUFV = HQ >> 1, JT = HQ & 1;
        const bool nxt_live = live(h + 1);
        if (cur_live) {
            float ls[NQ] = {0.f, 0.f}, ls2[NQ] = {0.f, 0.f}, ex[NQ][2] = {{0.f, 0.f}, {0.f, 0.f}}, mx[NQ] = {-INFINITY, -INFINITY};
            pv4u pwq[NQ][2];      // the packed probabilities as the value products' operand tuples (written in place: no moves in front of a product)
            if constexpr ((HQ & 1) == 0) { af[0] = PF_KFRAG(HN, 0); af[1] = PF_KFRAG(HN, 1); af[2] = PF_KFRAG(HN, 2); }
            pv4s lo[3], hi[3];
            w64_sfor<DS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                w64_sfor<NQ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    w64_qk<8 * j + s, s == 0>(N[j], af[s % 3], NEG[j]);
                    if constexpr (j == NQ - 1) {
                        if constexpr (s + 3 < DS) af[s % 3] = PF_KFRAG(HN, (s + 3) & 7);
                        else { lo[s - 5] = PF_VFRAG(BUFV, JT, 0, 0, (s - 5) & 3); hi[s - 5] = PF_VFRAG(BUFV, JT, 0, 1, (s - 5) & 3); }
                    }
                    // one wave per SIMD: nothing else fills an exponential's latency, so a gap consumes the pair issued in the PREVIOUS gap
                    // (the next pair is issued first) and the two row sums of a block run as separate chains
                    if constexpr (s == 0) { ex[j][0] = __builtin_amdgcn_exp2f(C[j][0]); ex[j][1] = __builtin_amdgcn_exp2f(C[j][1]); }
                    const float p0 = ex[j][0], p1 = ex[j][1];
                    if constexpr (s + 1 < DS) { ex[j][0] = __builtin_amdgcn_exp2f(C[j][2 * s + 2]); ex[j][1] = __builtin_amdgcn_exp2f(C[j][2 * s + 3]); }
                    ls[j] += p0;
                    ls2[j] += p1;
                    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                    const h2v t2 = {(f16)p0, (f16)p1};
                    {
                        unsigned pw_ = __builtin_bit_cast(unsigned, t2);
                        float ls_ = ls[j], ls2_ = ls2[j], e0_ = ex[j][0], e1_ = ex[j][1];
                        asm volatile("" : "+v"(pw_), "+v"(ls_), "+v"(ls2_), "+v"(e0_), "+v"(e1_));
                        pwq[j][s >> 2][s & 3] = pw_;
                        ls[j] = ls_; ls2[j] = ls2_; ex[j][0] = e0_; ex[j][1] = e1_;
                    }
                });
            });
            PT(1)
            if (!nxt_live || masked(h + 1)) {
                W64_D_WAIT();
#pragma unroll
                for (int j = 0; j < NQ; ++j) mask_half(h + 1, j, N[j]);
            }
            PT(2)
            w64_sfor<8>([&](auto ic) {
                constexpr int i = decltype(ic)::value, ks = i >> 2, blk = i & 3;
                typedef short v8s __attribute__((ext_vector_type(8)));
                const v8s av = {lo[i % 3][0], lo[i % 3][1], lo[i % 3][2], lo[i % 3][3], hi[i % 3][0], hi[i % 3][1], hi[i % 3][2], hi[i % 3][3]};
                w64_sfor<NQ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    w64_pv<4 * j + blk>(__builtin_bit_cast(v8h, av), __builtin_bit_cast(v8h, pwq[j][ks]));
                });
                if constexpr (i + 3 < 8) { lo[i % 3] = PF_VFRAG(BUFV, JT, ((i + 3) >> 2) & 1, 0, (i + 3) & 3); hi[i % 3] = PF_VFRAG(BUFV, JT, ((i + 3) >> 2) & 1, 1, (i + 3) & 3); }
                else if constexpr ((HQ & 1) == 0) af[i - 5] = PF_KFRAG((HN + 1) & 3, i - 5);      // the next half-step's first K operands (same iteration)
                // this iteration's DMA pieces ride in the value phase's gaps (one wave per SIMD: a burst of eight at the barrier idles the matrix pipe)
                if (t_dma >= 0) {
                    if constexpr ((HQ & 1) == 0) { if (i == 1) dma_k_half_piece(2 * t_dma + 3, 0); if (i == 3) dma_k_half_piece(2 * t_dma + 3, 1); if (i == 5) dma_k_half_piece(2 * t_dma + 4, 0); if (i == 7) dma_k_half_piece(2 * t_dma + 4, 1); }
                    else if (t_dma + 1 < nt) { if (i == 0) dma_piece(t_dma + 1, 0, true); if (i == 1) dma_piece(t_dma + 1, 1, true); if (i == 2) dma_piece(t_dma + 1, 2, true); if (i == 3) dma_piece(t_dma + 1, 3, true); }
                }
                if constexpr (i == 0) asm volatile("s_nop 3");      // (the last score products are two value products back: padded to 12 states and more)
                w64_sfor<NQ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    {      // (locals: clang refuses asm operands that name a variable of an enclosing lambda)
                        float m_ = mx[j];
                        const float n0_ = N[j][2 * i], n1_ = N[j][2 * i + 1];
                        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m_) : "v"(m_), "v"(n0_), "v"(n1_));
                        mx[j] = m_;
                    }
                });
            });
            PT(3)
            w64_sfor<NQ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                l_run[j] += ls[j] + ls2[j];
                decide(jc, mx[j], N[j]);
            });
            PT(4)
        } else if (t_dma >= 0) {      // a wave with nothing to compute still owns its share of the tile
            if constexpr ((HQ & 1) == 0) { dma_k_half(2 * t_dma + 3); dma_k_half(2 * t_dma + 4); }
            else dma_v_tile(t_dma + 1);
        }
        cur_live = nxt_live;
    };
    PT(7)
    for (int t = 0; t < nt; t += 2) {
        half_step(std::integral_constant<int, 0>{}, SA, SB, 2 * t, t);
        half_step(std::integral_constant<int, 1>{}, SB, SA, 2 * t + 1, t);
        PT(0)
        dma_wait();
        __syncthreads();
        PT(5)
        if (t + 1 < nt) {
            half_step(std::integral_constant<int, 2>{}, SA, SB, 2 * t + 2, t + 1);
            half_step(std::integral_constant<int, 3>{}, SB, SA, 2 * t + 3, t + 1);
            PT(0)
            dma_wait();
            __syncthreads();
            PT(5)
        }
    }
#if MILLION_EXP & 2048
    if (blockIdx.x == 0 && lane < 8) {
        unsigned *dst = (unsigned *)(p.q + b * p.q_sb + head * p.q_sh + (long long)q_lo * p.q_sn);
        unsigned v_ = pt_acc[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) v_ = lane == i ? pt_acc[i] : v_;
        dst[lane] = v_;
    }
#endif
#undef PT
#undef PF_KFRAG
#undef PF_VFRAG
#undef PF_PIN
    W64_D_WAIT();
    w64_sfor<NQ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        float l = l_run[j];
        {
            const v2u ex = swap32_self(__float_as_uint(l));
            const unsigned e0 = ex[0], e1 = ex[1];
            l = __uint_as_float(e0) + __uint_as_float(e1);
        }
        const int q_row = q_lo + 32 * j + r32;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        f16 *op = p.out + b * p.o_sb + head * p.o_sh + (long long)q_row * p.o_sn + 4 * hh;
        w64_sfor<NB>([&](auto bc) {
            constexpr int blk = decltype(bc)::value;
            const v16f o16 = w64_o_read<4 * j + blk>();
            if (wave_live && q_row < p.n_q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    typedef f16 h4 __attribute__((ext_vector_type(4)));
                    const h4 o = {(f16)(o16[4 * i] * inv), (f16)(o16[4 * i + 1] * inv), (f16)(o16[4 * i + 2] * inv), (f16)(o16[4 * i + 3] * inv)};
                    *(h4 *)(op + 32 * blk + 8 * i) = o;
                }
            }
        });
    });
}
#undef W64_D_WAIT
#endif      // MILLION_DEV_BUILD

template <int D, int PW>
static void launch_prefill_t(const PrefillParams &p, long long blocks, int lds, hipStream_t s) {
    hipLaunchKernelGGL((prefill_attn_kernel<D, PW>), dim3((unsigned)blocks), dim3(PW * 64), lds, s, p);
}
// dynamic-LDS attribute of the four instances: once per device, under the library's per-device mutex (common.h: device_once)
static int g_prefill_plain = 0;      // A/B and tests (million_set_force_generic(64)): the plain (round-3) form at d = 128 too
void set_prefill_policy(int plain) { g_prefill_plain = plain; }
static void prefill_attrs_once() {
    if (!device_once(3)) return;
    (void)hipFuncSetAttribute((const void *)prefill_attn_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKV * 2 * 128);
    (void)hipFuncSetAttribute((const void *)prefill_attn_kernel<128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKV * 2 * 128);
    (void)hipFuncSetAttribute((const void *)prefill_attn_kernel<128, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKV * 2 * 128);
    (void)hipFuncSetAttribute((const void *)prefill_attn_kernel<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKV * 2 * 64);
    (void)hipFuncSetAttribute((const void *)prefill_attn_kernel<64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKV * 2 * 64);
}

int launch_prefill(const PrefillParams &p_in, hipStream_t s) {
    PrefillParams p = p_in;
    static const int pw = [] { const char *e = getenv("MILLION_PREFILL_WAVES"); return e && e[0] == '4' ? 4 : kPWDefault; }();      // development A/B
    int hpw = 1;
    for (int c = 8; c >= 1; c >>= 1)
        if (p.G % c == 0 && c <= pw) { hpw = c; break; }
    p.hpw = hpw;
    const int QB = (pw / hpw) * 32;
    p.n_qb = (p.n_q + QB - 1) / QB;
    const long long blocks = (long long)p.bs * p.nh_k * (p.G / hpw) * p.n_qb;
    if (blocks <= 0) return MILLION_OK;
    if (blocks > 0x7fffffffLL) { set_error("prefill: %lld workgroups", blocks); return MILLION_ERR_SHAPE; }
    const int lds = 4 * kKV * 2 * p.d;      // two buffers of (K tile, V tile)
    prefill_attrs_once();
#ifdef MILLION_DEV_BUILD
    if (p.d == 128 && pw == 8 && g_prefill_plain == 2 && hpw <= 4) {
        static bool once = [] { (void)hipFuncSetAttribute((const void *)prefill_attn_w64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKV * 2 * 128); return true; }();
        (void)once;
        hipLaunchKernelGGL(prefill_attn_w64_kernel, dim3((unsigned)blocks), dim3(4 * 64), lds, s, p);
    } else
#endif
    if (p.d == 128 && pw == 8 && g_prefill_plain != 1) hipLaunchKernelGGL(prefill_attn_pipe_kernel, dim3((unsigned)blocks), dim3(8 * 64), lds, s, p);
    else if (p.d == 128) { if (pw == 4) launch_prefill_t<128, 4>(p, blocks, lds, s); else launch_prefill_t<128, 8>(p, blocks, lds, s); }
    else { if (pw == 4) launch_prefill_t<64, 4>(p, blocks, lds, s); else launch_prefill_t<64, 8>(p, blocks, lds, s); }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("prefill launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million

using namespace million;

extern "C" int million_prefill_attn(const million_prefill_desc *desc, const void *q, const void *k, const void *v, void *out,
                                    million_stream_t stream) {
    if (!desc || desc->struct_size != sizeof(million_prefill_desc)) { set_error("prefill: bad desc / struct_size"); return MILLION_ERR_ARG; }
    if (!q || !k || !v || !out) { set_error("prefill: null pointer"); return MILLION_ERR_ARG; }
    PrefillParams p;
    p.q = (const f16 *)q; p.k = (const f16 *)k; p.v = (const f16 *)v; p.out = (f16 *)out;
    p.bs = desc->bs; p.nh = desc->nh; p.nh_k = desc->nh_k; p.d = desc->d;
    p.n_q = desc->n_q; p.n_kv = desc->n_kv; p.q_pos0 = desc->q_pos0; p.causal = desc->causal != 0;
    if (p.bs <= 0 || p.nh <= 0 || p.nh_k <= 0 || p.nh % p.nh_k) { set_error("prefill: bs=%d nh=%d nh_k=%d", p.bs, p.nh, p.nh_k); return MILLION_ERR_SHAPE; }
    if (p.d != 128 && p.d != 64) { set_error("prefill: d=%d (64 or 128)", p.d); return MILLION_ERR_SHAPE; }
    if (p.n_q < 0 || p.n_kv < 0 || p.q_pos0 < 0) { set_error("prefill: n_q=%d n_kv=%d q_pos0=%d", p.n_q, p.n_kv, p.q_pos0); return MILLION_ERR_ARG; }
    if (p.n_q == 0) return MILLION_OK;
    if (p.n_kv == 0) { set_error("prefill: no keys (n_kv = 0) for %d query rows", p.n_q); return MILLION_ERR_ARG; }
    p.G = p.nh / p.nh_k;
    p.q_sb = desc->q_stride_b; p.q_sh = desc->q_stride_h; p.q_sn = desc->q_stride_n;
    p.k_sb = desc->k_stride_b; p.k_sh = desc->k_stride_h; p.k_sn = desc->k_stride_n;
    p.v_sb = desc->v_stride_b; p.v_sh = desc->v_stride_h; p.v_sn = desc->v_stride_n;
    p.o_sb = desc->o_stride_b; p.o_sh = desc->o_stride_h; p.o_sn = desc->o_stride_n;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) { set_error("prefill: every pointer must be 16-byte aligned"); return MILLION_ERR_ALIGN; }
    if ((p.q_sb | p.q_sh | p.q_sn | p.k_sb | p.k_sh | p.k_sn | p.v_sb | p.v_sh | p.v_sn | p.o_sb | p.o_sh | p.o_sn) & 7) {
        set_error("prefill: strides must be multiples of 8 elements (16-byte rows)");
        return MILLION_ERR_ALIGN;
    }
    p.scale_log2e = 1.4426950408889634f / sqrtf((float)p.d);
    return launch_prefill(p, (hipStream_t)stream);
}
