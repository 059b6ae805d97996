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

template <int D, int PW>
static void launch_prefill_t(const PrefillParams &p, long long blocks, int lds, hipStream_t s) {
    hipLaunchKernelGGL((prefill_attn_kernel<D, PW>), dim3((unsigned)blocks), dim3(PW * 64), lds, s, p);
}
// dynamic-LDS attribute of the four instances: once per device, under the library's per-device mutex (common.h: device_once)
static void prefill_attrs_once() {
    if (!device_once(3)) return;
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
    if (p.d == 128) { if (pw == 4) launch_prefill_t<128, 4>(p, blocks, lds, s); else launch_prefill_t<128, 8>(p, blocks, lds, s); }
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
