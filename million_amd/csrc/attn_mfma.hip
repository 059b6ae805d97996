// attn_mfma.hip — fused decode attention over PQ code pages for the headline shapes
// (d = 128, M = 64, C = 256, transposed V pages), hand-written for gfx950 / CDNA4.
//
// Replaces (one launch): the LUT matmul + flash_decoding_split_kernel + flash_decoding_residual_kernel
// + torch::zeros + flash_decoding_reduce_kernel of the reference (Interface.template.cu:26-120,
// Kernel.cuh:11-166, 1038-1270), and the intended paged-V kernel (MILLION_技术分析文档.md:1292-1345).
//
// Design (DESIGN.md "decode attention kernel"):
//   * one 512-thread workgroup per CU; both codebooks live in LDS for the whole kernel:
//       K table, row image [m][c] (4-byte entries): bank = code  -> random gather
//       V table, col image [c][m] (4-byte entries): bank = m     -> conflict-free gather (lane = m)
//   * all G = nh/nh_k query heads of a kv head are served by the same workgroup, so every code byte is
//     read from HBM once per kv head (the reference re-reads it G times);
//   * a wave walks 32-token units.  K side: lane = (token, 16-byte quarter of the code row); each code
//     byte fetches its 2-dim centroid from LDS straight into the A operand of
//     v_mfma_f32_16x16x32_f16 (rows = 16 tokens, K = 32 dims), B = the query heads -> fp32 scores with
//     exact fp16 centroids (no fp16 LUT rounding).  V side: lane = subspace m; 16 consecutive token
//     bytes of a transposed page row are one 16-byte load; looked-up centroids are packed into the B
//     operand of v_mfma_f32_32x32x16_f16 (K = 16 tokens, cols = 32 subspaces), A = the probabilities of
//     the G heads, moved from the score layout with v_permlane32_swap / v_permlane16_swap;
//   * online softmax per wave in the exp2 domain, fp32; wave partials are merged through LDS, split
//     partials through the workspace by the last-arriving workgroup (common.h:publish_and_merge);
//   * the residual window (r <= 128 fp16 rows) is dealt round-robin to the splits and, inside a split, to its
//     waves: each wave's rows are ONE 16-row MFMA tile (scores: A = the fp16 K rows; values: B = the fp16 V rows)
//     that rides along with the wave's code units - no separate partial, no scalar FMA loop.
//   * round 4: M = 16 (d_m = 8) and M = 32 at G <= 4 run the streaming kernel in the "d_m = 8 / 4 forms" (see there): the query
//     heads are replicated over the four column groups of the score tile, a gathered codebook entry is a lane's reduction slots of
//     a 16x16x32 operand as it stands (K: A operand, V: B operand), 8 accumulator registers, no pack and no lane movement;
//   * two kernels share everything above and the merge tail: attn_stream_kernel (the one that runs: value steps of
//     unit u interleaved with the score stages of unit u + 1, online softmax per unit, any split length) and
//     attn_mfma_kernel (groups of 4 units: score pass, softmax, value pass; the fallback for T = 0 and for splits of
//     more than 64 units per wave, and the A/B reference).
#include <type_traits>

#include "common.h"
#include "dev_switches.h"      // MILLION_EXP (0 in the product build) and the other development A/B switches

namespace million {

typedef _Float16 v8f16 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 v4f16_t __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int kNW = 8;                       // waves per workgroup
constexpr int kRing = 4;                     // 32-token units in flight per wave (16 VGPRs each)
constexpr int kTabBytes = 64 * 1024;         // one codebook image (M*C*dm*2)
constexpr int kVBase = kTabBytes;            // V col image behind the K row image
constexpr int kPartOff = 2 * kTabBytes;      // [128K,136K): final partial, flag
constexpr int kStampOff = kPartOff + 8192;   // [136K,138K): diagnostic stamps (only touched when a stamp buffer is set)
constexpr int kLdsBytes = kStampOff + kStampWaves * kStampSlots * 8;  // 138 KiB
constexpr int kResRows = 16;                 // residual-window rows per wave (one MFMA tile): 8 waves x 16 = 128 rows per split

struct UnitCodes {
    v4u k[2];   // K code bytes: score tile g2, lane (q, c): tile row c, bytes [16q, 16q+16) of that token's code row
                // (grouped kernel: row c = token 16 g2 + c; streaming kernel: stream_token_of_row(g2, c))
    v4u v[2];   // V code bytes: half n (32 subspaces), lane (h, c): m = 32n + c, tokens [16h, 16h+16)
};

// M = 32 (d_m = 4): a token's code row is 32 bytes, a codebook entry 8 bytes (one ds_read_b64)
struct UnitCodes32 {
    v2u k[2];   // K code bytes: score tile g2, lane (q, c): tile row c (see UnitCodes), bytes [8q, 8q+8)
    v4u v[1];   // V code bytes: lane (h, c): subspace m = c, tokens [16h, 16h+16)
};

// M = 16 (d_m = 8, streaming kernel only, round 4): a token's code row is 16 bytes, a codebook entry 16 bytes (one ds_read_b128)
// - the whole A operand of a 16x16x32 k-step on the K side and the whole B operand of one on the V side (see "d_m = 8 form")
struct UnitCodes16 {
    unsigned k[2];   // K code bytes: score tile g2, lane (q, c): tile row c, bytes [4q, 4q+4) = subspaces 4q .. 4q+3 of that token
    unsigned v[2];   // V code bytes: tile g2, lane (t, n): subspace n, tile rows 4t .. 4t+3 (tokens 8t + 4 g2 + 0..3 of the unit)
};
// M = 32 in the d_m = 4 form (streaming kernel, G <= 4, round 4): K bytes as UnitCodes32; V bytes as in UnitCodes16, for the two
// column tiles (subspaces n and n + 16)
struct UnitCodes32D {
    v2u k[2];             // as UnitCodes32
    unsigned v[2][2];     // V code bytes: tile g2, column tile j, lane (t, n): subspace n + 16 j, tile rows 4t .. 4t+3
};
// accumulators of the d_m = 8 form: row tile h (dim position 4h + dq), lane (dq = lane >> 4, n = lane & 15): register i = head i,
// dims 8n + 4h + dq
// (d_m = 4 form: column tile j, lane (dq, n): register i = head i, dims 4 (n + 16 j) + dq)
struct Acc8 { float __attribute__((ext_vector_type(4))) t[2]; };

// LDS by absolute byte address: the dynamic LDS segment of this kernel starts at 0 (no static LDS; the
// kernel traps otherwise), so a lookup address needs no base add.
__device__ __forceinline__ unsigned lds32(unsigned addr) {
    return *(const __attribute__((address_space(3))) unsigned *)(size_t)addr;
}
__device__ __forceinline__ v2u lds64(unsigned addr) {
    return *(const __attribute__((address_space(3))) v2u *)(size_t)addr;
}
__device__ __forceinline__ v4u lds128(unsigned addr) {
    return *(const __attribute__((address_space(3))) v4u *)(size_t)addr;
}
// Diagnostic stamps go to LDS (lane 0 of each wave) and are copied out at the very end of the kernel: a global
// store per stamp would put a vmcnt(0) into the phases being timed (and a generic-pointer store a FLAT op,
// which makes hipcc wait vmcnt(0) on the non-diagnostic path too).
__device__ __forceinline__ void stamp_lds(bool on, int lane, int wave, int i) {
    if (on && lane == 0)
        *(volatile __attribute__((address_space(3))) unsigned long long *)(size_t)(kStampOff + (wave * kStampSlots + i) * 8) =
            __builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void stamp_lds_clear(bool on, int lane, int wave) {
    if (on && lane < kStampSlots)
        *(volatile __attribute__((address_space(3))) unsigned long long *)(size_t)(kStampOff + (wave * kStampSlots + lane) * 8) = 0ull;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32

__device__ __forceinline__ float lane_bcast(float x, int lane_const) {       // v_readlane -> SGPR operand
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane_const));
}
// Rescale of the value accumulators when a head's softmax reference moves.  alpha of head g sits in lane g; O rows (32x32
// tile, register 4 j + rho): lanes < 32 hold head 8 j + rho, lanes >= 32 head 8 j + 4 + rho; j = 1 only exists for groups
// of more than 8 query heads (wave-uniform branch).  Rows of heads >= G are scaled by whatever their idle column holds:
// they are never read.  One ds_bpermute per register row (no SGPRs: 16 v_readlane results spilled scalar registers in
// the streaming loop).
// PV = parity-V accumulators (streaming kernel, M = 64; see "parity-V" below): only the tiles O[n][0] exist, tile rows are
// (parity of the dim, head): register 4 j + rho of lane (h, col) = row 8 j + 4 h + rho = parity j >> 1, head 8 (j & 1) + 4 h + rho.
template <bool PV = false>
__device__ __forceinline__ void rescale_heads(v16f32 (&O)[2][PV ? 1 : 2], float alpha, int G, int lane) {
    const int sel = lane < 32 ? 0 : 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 1 && G <= 8) break;
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
            const float f = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel + 4 * (8 * j + rho), __builtin_bit_cast(int, alpha)));
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (PV) {
                    O[n][0][4 * j + rho] *= f;
                    O[n][0][4 * (j + 2) + rho] *= f;
                } else {
#pragma unroll
                    for (int kk = 0; kk < (PV ? 1 : 2); ++kk) O[n][kk][4 * j + rho] *= f;
                }
            }
        }
    }
}
__device__ __forceinline__ v8f16 as_v8f16(unsigned a, unsigned b, unsigned c, unsigned d) {
    v4u t = {a, b, c, d};
    return __builtin_bit_cast(v8f16, t);
}

// ---- how the vector-memory queue is kept deep without fighting hipcc's waitcnt insertion ------------
// Every code / codebook / q load is a plain load the compiler can count, and NONE of them sits in a
// conditional: slots past a wave's last unit re-request the unit holding token T-1 (L2 hits).  The
// pending-load pattern at the loop header is then identical on entry and on the back edge, and hipcc
// emits counted waits (vmcnt(12) before a unit: the three younger units stay in flight).  Versions with
// conditional refills, or with LDS-DMA for the tables, made hipcc wait vmcnt(0) and drained the ring.
//
// Page ids come through the scalar cache in ONE asm statement (issue + wait): hipcc does not pick s_load
// for them by itself, and a vector load of an id would sit in the vmcnt queue in front of the codes.
struct PidPair { long long k, v; int k32, v32; };
__device__ __forceinline__ void load_pids4(const AttnParams &p, int bh, const int (&page)[4], PidPair (&o)[4]);

__device__ __forceinline__ PidPair load_pids(const AttnParams &p, int bh, int page) {
    int pg[kRing] = {page, page, page, page};
    PidPair o[kRing];
    load_pids4(p, bh, pg, o);
    return o[0];
}

// The page ids of the ring's kRing first units: ONE scalar round trip.  Issue and wait live in the SAME asm
// statement on purpose: an earlier version split them to overlap the latency and hipcc, on an unrelated
// edit, placed SGPR copies between the two statements — copies of values still in flight — which sent
// wild addresses to the code loads.  The statement sits after the independent vector loads have been
// issued, so the scalar latency still overlaps with them.
__device__ __forceinline__ void load_pids4(const AttnParams &p, int bh, const int (&page)[kRing], PidPair (&o)[kRing]) {
    unsigned off[kRing];
#pragma unroll
    for (int k = 0; k < kRing; ++k) { o[k].k = 0; o[k].v = 0; o[k].k32 = 0; o[k].v32 = 0; }
    if (p.v_identity && !p.k_paged) {           // dense scratch pages + row-major K: no id table at all
#pragma unroll
        for (int k = 0; k < kRing; ++k) o[k].v = bh * p.n_pages_cap + page[k];
        return;
    }
#pragma unroll
    for (int k = 0; k < kRing; ++k) off[k] = (unsigned)(bh * p.n_pages_cap + page[k]) * (p.ids64 ? 8u : 4u);
    if (p.ids64) {
        long long v0, v1, v2, v3;
        asm volatile("s_load_dwordx2 %0, %4, %5\n\ts_load_dwordx2 %1, %4, %6\n\ts_load_dwordx2 %2, %4, %7\n\t"
                     "s_load_dwordx2 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
                     : "s"(p.v_ids64), "s"(off[0]), "s"(off[1]), "s"(off[2]), "s"(off[3]) : "memory");
        o[0].v = v0; o[1].v = v1; o[2].v = v2; o[3].v = v3;
        if (p.k_paged) {
            asm volatile("s_load_dwordx2 %0, %4, %5\n\ts_load_dwordx2 %1, %4, %6\n\ts_load_dwordx2 %2, %4, %7\n\t"
                         "s_load_dwordx2 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
                         : "s"(p.k_ids64), "s"(off[0]), "s"(off[1]), "s"(off[2]), "s"(off[3]) : "memory");
            o[0].k = v0; o[1].k = v1; o[2].k = v2; o[3].k = v3;
        }
    } else {
        int v0, v1, v2, v3;
        if (p.v_identity) {
            v0 = bh * p.n_pages_cap + page[0]; v1 = bh * p.n_pages_cap + page[1];
            v2 = bh * p.n_pages_cap + page[2]; v3 = bh * p.n_pages_cap + page[3];
        } else {
            asm volatile("s_load_dword %0, %4, %5\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %4, %7\n\t"
                         "s_load_dword %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
                         : "s"(p.v_ids32), "s"(off[0]), "s"(off[1]), "s"(off[2]), "s"(off[3]) : "memory");
        }
        o[0].v = v0; o[1].v = v1; o[2].v = v2; o[3].v = v3;
        if (p.k_paged) {
            asm volatile("s_load_dword %0, %4, %5\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %4, %7\n\t"
                         "s_load_dword %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
                         : "s"(p.k_ids32), "s"(off[0]), "s"(off[1]), "s"(off[2]), "s"(off[3]) : "memory");
            o[0].k = v0; o[1].k = v1; o[2].k = v2; o[3].k = v3;
        }
    }
#ifdef MILLION_DEBUG_CHECK_IDS
#pragma unroll
    for (int k = 0; k < kRing; ++k) {
        if (p.k_paged) o[k].k = MILLION_CHECK_KID(p, o[k].k);
        if (!p.v_identity) o[k].v = MILLION_CHECK_VID(p, o[k].v);
    }
#endif
}

// Request the 16-byte loads of one 32-token unit (see UnitCodes): two for the K bytes, two for the V bytes.
// t_unit: multiple of 32, < T, wave-uniform.  Addresses are a wave-uniform 64-bit base (scalar ALU, forced
// into SGPRs) plus a 32-bit per-lane offset, so that the loads take the saddr + voffset form: the per-lane
// 64-bit pointer arithmetic of the obvious formulation was ~10 vector instructions per request.
typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;      // global address space: an integer -> pointer
                                                                       // cast would otherwise make FLAT loads
__device__ __forceinline__ gptr_u8 uniform_ptr(const uint8_t *q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gptr_u8)(((unsigned long long)hi << 32) | lo);
}
typedef const __attribute__((address_space(1))) v4u *gptr_v4u;
__device__ __forceinline__ void load_unit_k(const AttnParams &p, int b, int hk, const PidPair &pid, int t_unit, int T,
                                            int lane, UnitCodes &u) {
    const int q4 = lane >> 4, c16 = lane & 15;
    const int inpage = t_unit - ((t_unit >> p.ps_shift) << p.ps_shift);          // a unit never straddles a page
    const gptr_u8 base = uniform_ptr(p.k_paged ? p.k_codes + (((pid.k << p.ps_shift) + inpage) << 6)
                                                : p.k_codes + b * p.k_sb + hk * p.k_sh + ((long long)t_unit << 6));
    const int lim = T - 1 - t_unit;                  // rows past token T-1 re-read it (stay inside the store; masked later)
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        const int row = min(16 * g2 + c16, lim);
        const unsigned off = ((unsigned)row << 6) + 16u * q4;
        u.k[g2] = *(gptr_v4u)(base + off);
    }
}
__device__ __forceinline__ void load_unit_v(const AttnParams &p, const PidPair &pid, int t_unit, int lane, UnitCodes &u) {
    const int h2i = lane >> 5, c32 = lane & 31;
    const int inpage = t_unit - ((t_unit >> p.ps_shift) << p.ps_shift);
    const gptr_u8 base = uniform_ptr(p.v_codes + (pid.v << (6 + p.ps_shift)) + inpage);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const unsigned off = ((unsigned)(32 * n + c32) << p.ps_shift) + 16u * h2i;      // loop-invariant per lane
        u.v[n] = *(gptr_v4u)(base + off);
    }
}
typedef const __attribute__((address_space(1))) v2u *gptr_v2u;
__device__ __forceinline__ void load_unit_k(const AttnParams &p, int b, int hk, const PidPair &pid, int t_unit, int T,
                                            int lane, UnitCodes32 &u) {
    const int q4 = lane >> 4, c16 = lane & 15;
    const int inpage = t_unit - ((t_unit >> p.ps_shift) << p.ps_shift);
    const gptr_u8 base = uniform_ptr(p.k_paged ? p.k_codes + (((pid.k << p.ps_shift) + inpage) << 5)
                                               : p.k_codes + b * p.k_sb + hk * p.k_sh + ((long long)t_unit << 5));
    const int lim = T - 1 - t_unit;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        const int row = min(16 * g2 + c16, lim);
        const unsigned off = ((unsigned)row << 5) + 8u * q4;
        u.k[g2] = *(gptr_v2u)(base + off);
    }
}
__device__ __forceinline__ void load_unit_v(const AttnParams &p, const PidPair &pid, int t_unit, int lane, UnitCodes32 &u) {
    const int h2i = lane >> 5, c32 = lane & 31;
    const int inpage = t_unit - ((t_unit >> p.ps_shift) << p.ps_shift);
    const gptr_u8 base = uniform_ptr(p.v_codes + (pid.v << (5 + p.ps_shift)) + inpage);
    const unsigned off = ((unsigned)c32 << p.ps_shift) + 16u * h2i;
    u.v[0] = *(gptr_v4u)(base + off);
}
template <class Unit>
__device__ __forceinline__ void load_unit_pid(const AttnParams &p, int b, int hk, const PidPair &pid, int t_unit, int T,
                                              int lane, Unit &u) {
    load_unit_k(p, b, hk, pid, t_unit, T, lane, u);
    load_unit_v(p, pid, t_unit, lane, u);
}
template <class Unit>
__device__ __forceinline__ void load_unit(const AttnParams &p, int b, int hk, int bh, int t_unit, int T,
                                          int lane, Unit &u) {
    load_unit_pid(p, b, hk, load_pids(p, bh, t_unit >> p.ps_shift), t_unit, T, lane, u);
}

// Four centroid gathers of one score step: code bytes w -> A operand words (subspaces 4s..4s+3 of this lane's quarter).
__device__ __forceinline__ void k_gather(unsigned w, unsigned base, unsigned (&a)[4]) {
    a[0] = lds32(base + 0 * 1024 + ((w & 0xffu) << 2));
    a[1] = lds32(base + 1 * 1024 + (((w >> 8) & 0xffu) << 2));
    a[2] = lds32(base + 2 * 1024 + (((w >> 16) & 0xffu) << 2));
    a[3] = lds32(base + 3 * 1024 + ((w >> 24) << 2));
}

// Scores of one 32-token unit (MFMA 16x16x32): sc[g2*4 + rho] = scaled score (exp2 domain) of token
// 16*g2 + 4*q' + rho for the head of this lane's column (lane & 15); -inf beyond the split's last token.
// 8 steps (g2, s) of 4 gathers + 1 MFMA; the gathers run kDepth steps ahead of the MFMAs so that the LDS
// queue of this wave never drains (two waves per SIMD do not hide an LDS round trip per step).
template <bool MASK>
__device__ __forceinline__ void score_unit(const v4u (&kc)[2], const v8f16 (&qb)[4], int t_unit, int t_end,
                                           float scale_log2e, int lane, unsigned kbase, float (&sc)[8]) {
    const int q4 = lane >> 4;
    constexpr int kDepth = 3;
    unsigned a[8][4];
#pragma unroll
    for (int st = 0; st < kDepth; ++st) k_gather(kc[st >> 2][st & 3], kbase + (st & 3) * 4096, a[st]);
    v4f32 D[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        if (st + kDepth < 8) k_gather(kc[(st + kDepth) >> 2][(st + kDepth) & 3], kbase + ((st + kDepth) & 3) * 4096, a[st + kDepth]);
        D[st >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_v8f16(a[st][0], a[st][1], a[st][2], a[st][3]), qb[st & 3],
                                                            D[st >> 2], 0, 0, 0);
    }
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
            const float v = D[g2][rho] * scale_log2e;
            if (MASK) {
                const int tok = t_unit + 16 * g2 + 4 * q4 + rho;
                sc[g2 * 4 + rho] = tok < t_end ? v : -INFINITY;
            } else {
                sc[g2 * 4 + rho] = v;
            }
        }
}

// Values of one 32-token unit: O[n][kk] (rows = heads, cols = subspaces 32n..32n+31) += P (heads x tokens) * Vhat.
// pr[g2*4 + rho] = probability of token 16*g2 + 4*q' + rho for the head of this lane's column.
__device__ __forceinline__ void value_unit(const v4u (&vc)[2], const float (&pr)[8], unsigned vconst0, unsigned vconst1,
                                           v16f32 (&O)[2][2]) {
    // probabilities -> A operand of the value MFMA (rows = heads, K = 16 tokens per step)
    // pk[g2][i]: tokens 16*g2 + 4*q' + {2i, 2i+1} as packed fp16
    unsigned pk[2][2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            h2 t = {(f16)pr[g2 * 4 + 2 * i], (f16)pr[g2 * 4 + 2 * i + 1]};
            pk[g2][i] = __builtin_bit_cast(unsigned, t);
        }
    // step s uses tokens 16h + 8s + j: rows q' = 2s (j<4) and 2s+1 (j>=4) of group h
    unsigned P[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const v2u x = __builtin_amdgcn_permlane32_swap(pk[0][i], pk[1][i], false, false);
        // x[0] = {grp0 rows 0,1 | grp1 rows 0,1}  (step 0)   x[1] = {grp0 rows 2,3 | grp1 rows 2,3}  (step 1)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const v2u y = swap16_self(x[s]);
            // y[0] rows {0,0,2,2} of x[s] ; y[1] rows {1,1,3,3} of x[s]
            P[s][i] = y[0];
            P[s][2 + i] = y[1];
        }
    }
    // 4 steps (n, s) of 8 gathers + 2 MFMAs; the gathers of the next step are issued before this step's MFMAs
    unsigned e[4][8];
#define V_GATHER(ST)                                                                                               \
    {                                                                                                              \
        const unsigned vconst = ((ST) >> 1) ? vconst1 : vconst0;                                                   \
        const unsigned w0 = vc[(ST) >> 1][2 * ((ST) & 1)], w1 = vc[(ST) >> 1][2 * ((ST) & 1) + 1];                 \
        e[ST][0] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020400u));                                          \
        e[ST][1] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020500u));                                          \
        e[ST][2] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020600u));                                          \
        e[ST][3] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020700u));                                          \
        e[ST][4] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020400u));                                          \
        e[ST][5] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020500u));                                          \
        e[ST][6] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020600u));                                          \
        e[ST][7] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020700u));                                          \
    }
    V_GATHER(0)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st == 0) V_GATHER(1)
        if (st == 1) V_GATHER(2)
        if (st == 2) V_GATHER(3)
        const int n = st >> 1, sidx = st & 1;
        const v8f16 B0 = as_v8f16(__builtin_amdgcn_perm(e[st][1], e[st][0], 0x05040100u), __builtin_amdgcn_perm(e[st][3], e[st][2], 0x05040100u),
                                  __builtin_amdgcn_perm(e[st][5], e[st][4], 0x05040100u), __builtin_amdgcn_perm(e[st][7], e[st][6], 0x05040100u));
        const v8f16 B1 = as_v8f16(__builtin_amdgcn_perm(e[st][1], e[st][0], 0x07060302u), __builtin_amdgcn_perm(e[st][3], e[st][2], 0x07060302u),
                                  __builtin_amdgcn_perm(e[st][5], e[st][4], 0x07060302u), __builtin_amdgcn_perm(e[st][7], e[st][6], 0x07060302u));
        const v8f16 A = as_v8f16(P[sidx][0], P[sidx][1], P[sidx][2], P[sidx][3]);
        O[n][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B0, O[n][0], 0, 0, 0);
        O[n][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B1, O[n][1], 0, 0, 0);
    }
#undef V_GATHER
}

// ---- M = 32 forms of the two unit functions ------------------------------------------------------------
// Scores: lane (q4, token) holds 8 code bytes = subspaces 8*q4 .. 8*q4+7; step s uses subspaces 8*q4 + 2s, +1 =
// dims 32*q4 + 8s .. +8 (the SAME dim <-> k mapping as M = 64, so the query operand qb is shared); two 8-byte
// gathers per step.
__device__ __forceinline__ void k_gather32(unsigned w, int s, unsigned base, unsigned (&a)[4]) {
    const unsigned sh = 16 * (s & 1);
    const v2u lo = lds64(base + 0 * 2048 + (((w >> sh) & 0xffu) << 3));
    const v2u hi = lds64(base + 1 * 2048 + (((w >> (sh + 8)) & 0xffu) << 3));
    a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
}
template <bool MASK>
__device__ __forceinline__ void score_unit(const v2u (&kc)[2], const v8f16 (&qb)[4], int t_unit, int t_end,
                                           float scale_log2e, int lane, unsigned kbase, float (&sc)[8]) {
    const int q4 = lane >> 4;
    constexpr int kDepth = 3;
    unsigned a[8][4];
#pragma unroll
    for (int st = 0; st < kDepth; ++st) k_gather32(kc[st >> 2][(st & 3) >> 1], st & 3, kbase + (st & 3) * 4096, a[st]);
    v4f32 D[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        if (st + kDepth < 8) {
            const int t = st + kDepth;
            k_gather32(kc[t >> 2][(t & 3) >> 1], t & 3, kbase + (t & 3) * 4096, a[t]);
        }
        D[st >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_v8f16(a[st][0], a[st][1], a[st][2], a[st][3]), qb[st & 3],
                                                            D[st >> 2], 0, 0, 0);
    }
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
            const float v = D[g2][rho] * scale_log2e;
            if (MASK) {
                const int tok = t_unit + 16 * g2 + 4 * q4 + rho;
                sc[g2 * 4 + rho] = tok < t_end ? v : -INFINITY;
            } else {
                sc[g2 * 4 + rho] = v;
            }
        }
}
// Values: lane (h, m): 16 token bytes of subspace m; an 8-byte gather brings the 4 dims of one (token, m); tile
// O[i][j] holds dim 4m + 2i + j of the 32 subspaces (cols).  2 steps of 8 gathers + 16 packs + 4 MFMAs.
__device__ __forceinline__ void value_unit(const v4u (&vc)[1], const float (&pr)[8], unsigned vconst0, unsigned /*vconst1*/,
                                           v16f32 (&O)[2][2]) {
    unsigned pk[2][2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            h2 t = {(f16)pr[g2 * 4 + 2 * i], (f16)pr[g2 * 4 + 2 * i + 1]};
            pk[g2][i] = __builtin_bit_cast(unsigned, t);
        }
    unsigned P[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const v2u x = __builtin_amdgcn_permlane32_swap(pk[0][i], pk[1][i], false, false);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const v2u y = swap16_self(x[s]);
            P[s][i] = y[0];
            P[s][2 + i] = y[1];
        }
    }
    unsigned e[2][2][8];      // [step][dword of the entry][token]
#define V_GATHER32(ST)                                                                                             \
    {                                                                                                              \
        const unsigned w0 = vc[0][2 * (ST)], w1 = vc[0][2 * (ST) + 1];                                             \
        const unsigned sel[4] = {0x03020400u, 0x03020500u, 0x03020600u, 0x03020700u};                              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                            \
            const v2u t = lds64(__builtin_amdgcn_perm(j < 4 ? w0 : w1, vconst0, sel[j & 3]));                      \
            e[ST][0][j] = t[0];                                                                                    \
            e[ST][1][j] = t[1];                                                                                    \
        }                                                                                                          \
    }
    V_GATHER32(0)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        if (st == 0) V_GATHER32(1)
        const v8f16 A = as_v8f16(P[st][0], P[st][1], P[st][2], P[st][3]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned(&x)[8] = e[st][i];
            const v8f16 B0 = as_v8f16(__builtin_amdgcn_perm(x[1], x[0], 0x05040100u), __builtin_amdgcn_perm(x[3], x[2], 0x05040100u),
                                      __builtin_amdgcn_perm(x[5], x[4], 0x05040100u), __builtin_amdgcn_perm(x[7], x[6], 0x05040100u));
            const v8f16 B1 = as_v8f16(__builtin_amdgcn_perm(x[1], x[0], 0x07060302u), __builtin_amdgcn_perm(x[3], x[2], 0x07060302u),
                                      __builtin_amdgcn_perm(x[5], x[4], 0x07060302u), __builtin_amdgcn_perm(x[7], x[6], 0x07060302u));
            O[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B0, O[i][0], 0, 0, 0);
            O[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B1, O[i][1], 0, 0, 0);
        }
    }
#undef V_GATHER32
}

// ---- residual window -------------------------------------------------------------------------------
// The window rows j = split, split + nsplit, ... < r of a split are dealt to its waves in runs of kResRows: wave w
// owns list entries idx = 16 w + i, i < kResRows (round 3; rounds 1-2 dealt them round-robin, idx = w + 8 i: with 4-7
// rows per split that made 4-7 waves load a whole 16-row tile - 12 requests each in the kernel's front - for ONE row;
// now the older, faster wave 0 takes them all), as ONE 16-row MFMA tile that rides along with the code
// units of the first group: scores with A = the fp16 K rows themselves, values with B = the fp16 V rows
// (k = 16 rows, cols = 32 subspaces; even / odd dims by v_perm like the looked-up centroids).  Rows past the
// list re-read the wave's first row and are masked to -inf.
struct ResTile {
    v4u k[4];          // lane (q4, c16): row c16 of the tile, dims 32*q4 + 8*s .. + 8
    unsigned v[2][8];  // lane (h, c32): rows 8*h + j, dims (2m, 2m+1) of subspace m = 32*n + c32
};

// Row pointer of list entry idx (clamped to the wave's first entry, which exists when the tile is used).
__device__ __forceinline__ long long res_row_off(const AttnParams &p, int idx, int wave, int rcnt, int split, int rstart,
                                                 int r_old, bool &is_new) {
    const int idc = idx < rcnt ? idx : kResRows * wave;
    const int j = split + idc * p.nsplit;
    int row = rstart + j;
    row = row >= p.rcap ? row - p.rcap : row;          // rstart, j < rcap: one wrap at most
    is_new = p.k_new && j == r_old;                    // fused append: the new token is window row r_old
    return (long long)row * 128;
}

template <int MS = 64>
__device__ __forceinline__ void load_res_tile(const AttnParams &p, int bh, const f16 *kr, const f16 *vr, int wave, int rcnt,
                                              int split, int rstart, int r_old, int lane, ResTile &t) {
    const int q4 = lane >> 4, c16 = lane & 15, h = lane >> 5, c32 = lane & 31;
    {
        bool is_new;
        const long long off = res_row_off(p, kResRows * wave + c16, wave, rcnt, split, rstart, r_old, is_new);
        const f16 *kp = (is_new ? p.k_new + (long long)bh * 128 : kr + off) + 32 * q4;
#pragma unroll
        for (int s = 0; s < 4; ++s) t.k[s] = *(const v4u *)(kp + 8 * s);
    }
    if constexpr (MS == 320) {     // d_m = 4 form: lane (t = q4, n = c16): k-step s_: rows 4 t + 2 s_ (+ 1), dims 4 (n + 16 j) .. + 3
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                bool is_new;
                const long long off = res_row_off(p, kResRows * wave + 4 * q4 + 2 * s_ + rr, wave, rcnt, split, rstart, r_old, is_new);
                const f16 *vp = (is_new ? p.v_new + (long long)bh * 128 : vr + off) + 4 * c16;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const v2u w = *(const v2u *)(vp + 64 * j);
                    t.v[s_][4 * j + 2 * rr + 0] = w[0];
                    t.v[s_][4 * j + 2 * rr + 1] = w[1];
                }
            }
        return;
    }
    if constexpr (MS == 16) {      // d_m = 8 form: lane (t = q4, n = c16): k-step s_: rows 4 t + 2 s_ (+ 1), dims 8 n + 4 h .. + 3
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                bool is_new;
                const long long off = res_row_off(p, kResRows * wave + 4 * q4 + 2 * s_ + rr, wave, rcnt, split, rstart, r_old, is_new);
                const v4u w = *(const v4u *)((is_new ? p.v_new + (long long)bh * 128 : vr + off) + 8 * c16);
                t.v[s_][2 * rr + 0] = w[0]; t.v[s_][2 * rr + 1] = w[1];              // half h = 0
                t.v[s_][4 + 2 * rr + 0] = w[2]; t.v[s_][4 + 2 * rr + 1] = w[3];      // half h = 1
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bool is_new;
        const long long off = res_row_off(p, kResRows * wave + 8 * h + j, wave, rcnt, split, rstart, r_old, is_new);
        if (MS == 64) {      // tile (n, kk): dim 2*(32n + c32) + kk
            const f16 *vp = (is_new ? p.v_new + (long long)bh * 128 : vr + off) + 2 * c32;
            t.v[0][j] = *(const unsigned *)vp;
            t.v[1][j] = *(const unsigned *)(vp + 64);
        } else {             // M = 32, tile (i, jj): dim 4*c32 + 2i + jj
            const f16 *vp = (is_new ? p.v_new + (long long)bh * 128 : vr + off) + 4 * c32;
            const v2u w = *(const v2u *)vp;
            t.v[0][j] = w[0];
            t.v[1][j] = w[1];
        }
    }
}

// scores of the tile: sc[rho] = row 4*q' + rho for the head of this lane's column
template <class RT>      // ResTile / ResTileLean (lean kernel): the K rows are laid out alike
__device__ __forceinline__ void score_res_tile(const RT &t, const v8f16 (&qb)[4], float scale_log2e, int wave, int rcnt,
                                               int lane, float (&sc)[4]) {
    const int q4 = lane >> 4;
    v4f32 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
        D = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8f16, t.k[s]), qb[s], D, 0, 0, 0);
#pragma unroll
    for (int rho = 0; rho < 4; ++rho)
        sc[rho] = (kResRows * wave + 4 * q4 + rho) < rcnt ? D[rho] * scale_log2e : -INFINITY;
}

// O += P (heads x 16 rows) * V rows.  pr[rho] = probability of row 4*q' + rho for the head of this lane's column.
__device__ __forceinline__ void value_res_tile(const ResTile &t, const float (&pr)[4], v16f32 (&O)[2][2]) {
    h2 t0 = {(f16)pr[0], (f16)pr[1]}, t1 = {(f16)pr[2], (f16)pr[3]};
    const v2u y0 = swap16_self(__builtin_bit_cast(unsigned, t0));     // [0]: rows 2h of the score layout, [1]: rows 2h + 1
    const v2u y1 = swap16_self(__builtin_bit_cast(unsigned, t1));
    const unsigned y00 = y0[0], y01 = y0[1], y10 = y1[0], y11 = y1[1];
    const v8f16 A = as_v8f16(y00, y10, y01, y11);                     // rows 8h + (0,1), (2,3), (4,5), (6,7)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const unsigned(&e)[8] = t.v[n];
        const v8f16 B0 = as_v8f16(__builtin_amdgcn_perm(e[1], e[0], 0x05040100u), __builtin_amdgcn_perm(e[3], e[2], 0x05040100u),
                                  __builtin_amdgcn_perm(e[5], e[4], 0x05040100u), __builtin_amdgcn_perm(e[7], e[6], 0x05040100u));
        const v8f16 B1 = as_v8f16(__builtin_amdgcn_perm(e[1], e[0], 0x07060302u), __builtin_amdgcn_perm(e[3], e[2], 0x07060302u),
                                  __builtin_amdgcn_perm(e[5], e[4], 0x07060302u), __builtin_amdgcn_perm(e[7], e[6], 0x07060302u));
        O[n][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B0, O[n][0], 0, 0, 0);
        O[n][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B1, O[n][1], 0, 0, 0);
    }
}

// ---- parity-V (round 4): the value product without the pack -------------------------------------------------------
// A gathered V word is (dim 2m, dim 2m + 1) of ONE token; the value MFMA's operand register wants two reduction indices of one
// column.  Rounds 1-3 re-packed: 8 gathers -> 8 v_perm -> B0 (even dims), B1 (odd dims), two MFMAs, 64 accumulator registers.
// Here the reduction index IS (token, parity of the dim): the gathered word is the B operand as it stands (4 gathers = one
// lane's 8 reduction slots = 4 tokens), and the zero pattern moves to the cheap side - tile row (parity p, head g) holds
// P[g][token] in half p of the register and 0 in the other half, so that
//   D[(p, g)][m] = sum over (token, e) of P[g][token] [e == p] * Vhat[token][2m + e] = O[g][2m + p].
// Per 32-token unit: 32 gathers, 32 address v_perm, 16 placement v_perm, 8 MFMA (one per 8 tokens x 32 subspaces), 32
// accumulator registers - against 32 + 32 + 32 pack + 8 and 64 (tools/micro/core_micro.hip: +12 % units per SIMD and us).
// Tile rows: r = 16 p + g (g < 16 heads); lane (h, r) of the A operand, step s (tokens 16h + 4s + t, t = 0..3): register t.
// The score tiles leave, in lane (q4 = 2h + p', g), the probabilities of tokens 16h + 8p' + x, x = 0..7: W[k] = cvt_pk(x = 2k,
// 2k + 1); swap16_self hands every lane pair (p' = 0, 1) both rows' W (E: tokens 16h + 0..7, F: 16h + 8..15); sel_lo / sel_hi
// (lane constants, by the lane's OWN row parity) move one half of a W into the lane's half of the register.
struct ParA { unsigned E[4], F[4]; };
__device__ __forceinline__ void value_prep_par(const float (&pr)[8], ParA &pa) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h2 t = {(f16)pr[2 * k], (f16)pr[2 * k + 1]};
        const v2u y = swap16_self(__builtin_bit_cast(unsigned, t));
        const unsigned y0 = y[0], y1 = y[1];
        pa.E[k] = y0;
        pa.F[k] = y1;
    }
}
// A operand of token step s (0..3)
__device__ __forceinline__ v8f16 value_A_par(const ParA &pa, int s, unsigned sel_lo, unsigned sel_hi) {
    const unsigned w0 = s < 2 ? pa.E[2 * (s & 1)] : pa.F[2 * (s & 1)], w1 = s < 2 ? pa.E[2 * (s & 1) + 1] : pa.F[2 * (s & 1) + 1];
    return as_v8f16(__builtin_amdgcn_perm(0u, w0, sel_lo), __builtin_amdgcn_perm(0u, w0, sel_hi),
                    __builtin_amdgcn_perm(0u, w1, sel_lo), __builtin_amdgcn_perm(0u, w1, sel_hi));
}
__device__ __forceinline__ void par_selectors(int lane, unsigned &sel_lo, unsigned &sel_hi) {
    const bool odd = (lane >> 4) & 1;      // v_perm selectors: bytes 0-3 = the W register, 0x0c = zero
    sel_lo = odd ? 0x01000c0cu : 0x0c0c0100u;
    sel_hi = odd ? 0x03020c0cu : 0x0c0c0302u;
}
// the 4 gathers of value step (token step s, subspace half n): tokens 16h + 4s + t of subspace 32n + c32
__device__ __forceinline__ void v_gather_par(const v4u (&vc)[2], int s, int n, unsigned vconst0, unsigned vconst1, unsigned (&e)[4]) {
    const unsigned vconst = n ? vconst1 : vconst0;
    const unsigned w = vc[n][s];
    e[0] = lds32(__builtin_amdgcn_perm(w, vconst, 0x03020400u));
    e[1] = lds32(__builtin_amdgcn_perm(w, vconst, 0x03020500u));
    e[2] = lds32(__builtin_amdgcn_perm(w, vconst, 0x03020600u));
    e[3] = lds32(__builtin_amdgcn_perm(w, vconst, 0x03020700u));
}
// residual tile in the parity form: t.v[n][j] (rows 8h + j, dims (2m, 2m + 1)) IS the B operand of step s = j >> 2; pr[rho] =
// probability of row 4 q4 + rho = 8h + 4p' + rho: step 0's rows sit in the even lane rows, step 1's in the odd ones
__device__ __forceinline__ void value_res_tile_par(const ResTile &t, const float (&pr)[4], unsigned sel_lo, unsigned sel_hi, v16f32 (&O)[2][1]) {
    h2 t0 = {(f16)pr[0], (f16)pr[1]}, t1 = {(f16)pr[2], (f16)pr[3]};
    const v2u y0 = swap16_self(__builtin_bit_cast(unsigned, t0));
    const v2u y1 = swap16_self(__builtin_bit_cast(unsigned, t1));
    const unsigned e0 = y0[0], f0 = y0[1], e1 = y1[0], f1 = y1[1];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned w0 = s ? f0 : e0, w1 = s ? f1 : e1;
        const v8f16 A = as_v8f16(__builtin_amdgcn_perm(0u, w0, sel_lo), __builtin_amdgcn_perm(0u, w0, sel_hi),
                                 __builtin_amdgcn_perm(0u, w1, sel_lo), __builtin_amdgcn_perm(0u, w1, sel_hi));
#pragma unroll
        for (int n = 0; n < 2; ++n)
            O[n][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, as_v8f16(t.v[n][4 * s], t.v[n][4 * s + 1], t.v[n][4 * s + 2], t.v[n][4 * s + 3]),
                                                             O[n][0], 0, 0, 0);
    }
}

// =====================================================================================================
// Tail shared by the MFMA kernels (round 3): wave partials -> LDS -> the split's partial -> workspace -> merge.
//
// Round 2 handed the split partials over through MEMORY: write-through (sc1) stores, drain, barrier, one returning
// ticket atomic, barrier, and in the last arriver 64 KiB of sc1 loads by one workgroup - three dependent fabric round
// trips, 3.4 us of an 18 us launch (profiles/r02_stamps.txt).  Now (protocol measured in isolation by
// tools/micro/l2_handoff.hip -> profiles/r03_l2_handoff.txt):
//   * the kernel deals all splits of a (b, kv head) to ONE XCD (workgroup i runs on XCD i % 8).  A plain store is in
//     that XCD's L2 when its vmcnt retires, and an sc1 load issued on the same XCD is served from there (it bypasses
//     only the L1): a same-XCD hand-off never leaves the chiplet.  Whether the placement really holds is checked, not
//     assumed: in its prologue every workgroup MARKS its slot of the (b, kv head)'s census line with its XCC id (one
//     write-through 4-byte store), and at the start of its tail it reads the line: only if all nsplit slots carry its own
//     XCC id does it store its partial plain; otherwise (another XCD, or a workgroup that has not started yet)
//     write-through (sc1), which any XCD can read.  Loads and polls are sc1 in both cases;
//   * nobody waits for a ticket: the arrival index is requested ~3 us ahead of the tail (wave 7, which stores nothing);
//     the storing waves drain their stores, and behind the workgroup barrier they then join the split's FLAG (= generation
//     + 1) is raised; the workgroup whose index is ns - 1 is the merger (round 4: the only one): its waves poll the flags
//     (lane = split, bounded) and merge the query heads, four waves per head and two heads per pass - 16 KiB of loads per
//     head, wave reductions by DPP / row swaps, no LDS, no barrier.  Every workgroup it waits for has taken its index, so
//     it is resident, past its loop and waits for nothing: the polls end under any dispatch order and any residency.  A
//     poll that runs out of its bound is COUNTED (g_tail_faults, million_debug_tail_faults) and the heads are written as
//     NaN, never as the sum of stale partials;
//   * the workgroup with the highest index clears the census line and the counter and advances the generation once its
//     own poll has seen every flag (all census reads and stores of the launch are behind those flags);
//   * nsplit = 1: the only workgroup normalises and writes the output itself.
// =====================================================================================================
__device__ __forceinline__ unsigned *tail_rec(const AttnParams &p, int bh) { return (unsigned *)p.ws_cnt + (long long)bh * kRecWords; }
__device__ __forceinline__ unsigned *tail_flags(const AttnParams &p, int bh) { return p.ws_flags + (long long)bh * (2 * kFlagWords); }
__device__ __forceinline__ unsigned tail_xcc() { return __builtin_amdgcn_s_getreg(6164) & 7u; }      // hwreg(HW_REG_XCC_ID, 0, 4)

// Census mark: thread 0, write-through (every XCD must be able to read it, and no copy may linger dirty in an L2 when the
// last workgroup clears the line), in a wave-uniform branch of wave 0.  hipcc sizes wave 0's later vmcnt waits as if the
// store had not been issued, so wave 0's next wait for an OLDER load also waits for this store's acknowledgement: it is
// placed in the prologue behind the first gathers, where that next wait is ~1 us away, in the wave that reaches the
// wave-merge barrier 1.6 us early anyway.  The same wave drains it (vmcnt(0)) before this workgroup's flag goes up.
// (First form of this tail: a returning start-counter atomic, a census atomic and a generation load up here, by all eight
// waves: 512 same-line memory-side operations per (b, kv head) queued at one channel and launches took 20-25 us; by one
// lane: the body still ran 1.4 us longer.)
__device__ __forceinline__ void tail_mark_xcd(const AttnParams &p, int bh, int split, int wave, int lane) {
    if (wave == 0) {
        if (lane == 0) {
            __hip_atomic_store(tail_flags(p, bh) + kFlagWords + split, tail_xcc() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // diagnostics (million_set_force_generic(4)): every helper "has given up" before anybody's ticket
            if (p.tail_test == 1) __hip_atomic_fetch_or(tail_rec(p, bh) + 2, 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The tail's three requests - census line (wave 0, lane = split), arrival index (returning atomic) and generation (thread
// (kNW-1)*64) - issued ~3 us ahead of the tail, between the last blocks of the streaming loop: under the full code stream a
// memory round trip takes 2-3 us, more than the wave merge hides (the census read at the start of the tail held barrier B
// for 0.7 us).  Every lane of every wave issues the three instructions - no branch between two blocks of the pipeline, and
// no conditional vector-memory operation for hipcc's wait counting - but only the lanes named above address inside the
// descriptors; the hardware drops out-of-range lanes (loads return 0).
struct TailReq {
    int idx;        // RAW ticket word as the atomic returned it: give-up bits [7:0], arrival count [31:8] (see merge_and_publish)
    unsigned gen, cen, base;
    int nm, tt;     // mergers per (b, kv head) and the tail's test mode: kernel arguments, read here - not on the tail's critical path
    bool done;      // wave-uniform: false = this wave never passed the early request point (it had no whole round)
};
__device__ __forceinline__ void tail_request(const AttnParams &p, int bh, int ns, int wave, int lane, TailReq &t) {
    constexpr int kOut = 1 << 20;
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)tail_rec(p, bh), 0, kRecWords * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(tail_flags(p, bh) + kFlagWords), 0, kFlagWords * 4, 0x00020000);
    const bool one = wave == kNW - 1 && lane == 0;
    t.cen = __builtin_amdgcn_raw_buffer_load_b32(rc, wave == 0 ? (lane < ns ? lane : 0) * 4 : kOut, 0, 16);
    t.idx = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(256, rr, one ? 2 * 4 : kOut, 0, 0);
    t.gen = __builtin_amdgcn_raw_buffer_load_b32(rr, one ? 3 * 4 : kOut, 0, 16);
    t.base = __builtin_amdgcn_raw_buffer_load_b32(rr, one ? 4 * 4 : kOut, 0, 16);
    t.nm = ns > 1 ? (p.nmerge < ns ? (p.nmerge > 0 ? p.nmerge : 1) : ns) : 1;
    t.tt = p.tail_test;
    t.done = true;
}

// 16-lane row reductions by DPP (quad swaps, half-row mirror, row mirror), then the four rows by the row swaps
#define MILLION_DPP(x, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (CTRL), 0xf, 0xf, false))
__device__ __forceinline__ float wave_max_valu(float x) {
    x = fmaxf(x, MILLION_DPP(x, 0xB1));      // quad_perm [1,0,3,2]
    x = fmaxf(x, MILLION_DPP(x, 0x4E));      // quad_perm [2,3,0,1]
    x = fmaxf(x, MILLION_DPP(x, 0x141));     // row_half_mirror
    x = fmaxf(x, MILLION_DPP(x, 0x140));     // row_mirror
    return rows_max(x);
}
__device__ __forceinline__ float wave_sum_valu(float x) {
    x += MILLION_DPP(x, 0xB1);
    x += MILLION_DPP(x, 0x4E);
    x += MILLION_DPP(x, 0x141);
    x += MILLION_DPP(x, 0x140);
    return rows_sum(x);
}

// ---- d_m = 8 form (M = 16, streaming kernel, G <= 4; round 4) ----------------------------------------------------------
// A 16-byte codebook entry is 8 dims of ONE token and ONE subspace: exactly one lane's 8 reduction slots of a 16x16x32 operand.
//   scores: A[row = token][k = (quarter q4, dim 8)] = the gathered K entry of subspace 4 q4 + s in k-step s; B = the query heads,
//           REPLICATED over the four column groups (column c = 4 dq + g holds head g): the score tile then has head g's
//           probabilities in every lane row the value operand wants them in - no lane movement at all;
//   values: a 16-byte V entry is handled as its two 8-byte halves (dims 4 h .. 4 h + 3, h = 0 / 1) in the d_m = 4 form below: the
//           reduction index is (token of 2, dim position of 4), B = the halves h of the two tokens' entries (two ds_read_b64), A
//           carries the two tokens' probabilities at dim position dq (lane-constant masks), one product per half:
//           D_h[(dq, g)][n] = out[g][8 n + 4 h + dq].  (First version: whole entries by ds_read_b128, reduction index (token of 4,
//           dim position of 8), A = {x, y, 0, 0} / {0, 0, x, y} for the two row tiles: twice the value MFMAs, and hipcc rebuilt
//           the zero-padded operands with 8 v_mov per step.)
// Per 32-token unit: 8 gathers (ds_read_b128) + 16 (ds_read_b64), 8 score + 8 value MFMAs (16x16x32), 8 accumulator registers, no
// pack and no cross-lane instruction.
__device__ __forceinline__ void d8_masks(int lane, unsigned &mx, unsigned &my) {
    const int dq = (lane >> 2) & 3;      // column group of this lane = dim position (mod 4) of its rows
    mx = dq == 0 ? 0x0000ffffu : dq == 1 ? 0xffff0000u : 0u;
    my = dq == 2 ? 0x0000ffffu : dq == 3 ? 0xffff0000u : 0u;
}
// ---- d_m = 4 form (M = 32, G <= 4): the same idea with 8-byte entries.  A lane's 8 reduction slots are TWO tokens x 4 dim
// positions: k-step s of a 16-token tile takes tile rows 4 t + 2 s and 4 t + 2 s + 1 (t = lane >> 4) - registers 2 s, 2 s + 1 of the
// lane's own scores; the B operand is the two gathered entries of those rows, for subspace n (column tile 0) and n + 16 (tile 1);
// rows = (dim position dq, head g): ONE row tile.  Per unit: 16 + 16 gathers (ds_read_b64), 8 + 8 MFMAs (16x16x32), 8
// accumulator registers (the packed form: 8 + 8 MFMAs of which the value ones are 32x32x16, 48 pack v_perm, 64 accumulators).
__device__ __forceinline__ void d4_vstep(float p0, float p1, const unsigned (&e)[8], unsigned mx, unsigned my, Acc8 &O) {
    const h2 a0 = {(f16)p0, (f16)p0}, a1 = {(f16)p1, (f16)p1};
    const unsigned w0 = __builtin_bit_cast(unsigned, a0), w1 = __builtin_bit_cast(unsigned, a1);
    const v8f16 A = as_v8f16(w0 & mx, w0 & my, w1 & mx, w1 & my);
    O.t[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, as_v8f16(e[0], e[1], e[2], e[3]), O.t[0], 0, 0, 0);
    O.t[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, as_v8f16(e[4], e[5], e[6], e[7]), O.t[1], 0, 0, 0);
}
// d_m = 8 form: the same value step on the two 8-byte HALVES of a 16-byte entry (row tile h = dims 4 h .. 4 h + 3 of the entry):
// vconst = V col image base | 16 n, the halves 8 bytes apart
__device__ __forceinline__ void d8_vgather(const unsigned (&vc)[2], int i, unsigned vconst, unsigned (&e)[8]) {
    const unsigned sel[4] = {0x03020400u, 0x03020500u, 0x03020600u, 0x03020700u};
    const int g2 = i >> 1, s = i & 1;
    const unsigned w = vc[g2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const v2u x0 = lds64(__builtin_amdgcn_perm(w, vconst + 8u * h, sel[2 * s]));
        const v2u x1 = lds64(__builtin_amdgcn_perm(w, vconst + 8u * h, sel[2 * s + 1]));
        e[4 * h + 0] = x0[0]; e[4 * h + 1] = x0[1]; e[4 * h + 2] = x1[0]; e[4 * h + 3] = x1[1];
    }
}
// the 4 gathers of value step i = (tile g2 = i >> 1, k-step s = i & 1): bytes 2 s, 2 s + 1 of the lane's code words of the tile;
// vconst_j = V col image base | 8 (n + 16 j) (entries of 8 bytes, 256 bytes per code)
__device__ __forceinline__ void d4_vgather(const unsigned (&vc)[2][2], int i, unsigned vconst0, unsigned vconst1, unsigned (&e)[8]) {
    const unsigned sel[4] = {0x03020400u, 0x03020500u, 0x03020600u, 0x03020700u};
    const int g2 = i >> 1, s = i & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned w = vc[g2][j], vconst = j ? vconst1 : vconst0;
        const v2u x0 = lds64(__builtin_amdgcn_perm(w, vconst, sel[2 * s]));
        const v2u x1 = lds64(__builtin_amdgcn_perm(w, vconst, sel[2 * s + 1]));
        e[4 * j + 0] = x0[0]; e[4 * j + 1] = x0[1]; e[4 * j + 2] = x1[0]; e[4 * j + 3] = x1[1];
    }
}
// residual tile: t.v[s][4 j + ..] = (row 4 t + 2 s, row 4 t + 2 s + 1) x dims 4 (n + 16 j) .. + 3
__device__ __forceinline__ void value_res_tile_d4(const ResTile &t, const float (&pr)[4], unsigned mx, unsigned my, Acc8 &O) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) d4_vstep(pr[2 * s_], pr[2 * s_ + 1], t.v[s_], mx, my, O);
}
// rescale of the accumulators: register i belongs to head i, whose alpha sits in lane i of this lane's quad
__device__ __forceinline__ void rescale_acc(Acc8 &O, float alpha, int, int) {
    const float f0 = MILLION_DPP(alpha, 0x00), f1 = MILLION_DPP(alpha, 0x55), f2 = MILLION_DPP(alpha, 0xAA), f3 = MILLION_DPP(alpha, 0xFF);
#pragma unroll
    for (int h = 0; h < 2; ++h) { O.t[h][0] *= f0; O.t[h][1] *= f1; O.t[h][2] *= f2; O.t[h][3] *= f3; }
}
template <int KK>
__device__ __forceinline__ void rescale_acc(v16f32 (&O)[2][KK], float alpha, int G, int lane) { rescale_heads<KK == 1>(O, alpha, G, lane); }

// Count of merges that gave up waiting for a split's flag (million_debug_tail_faults): never non-zero unless a workgroup of the
// launch died or the workspace was not zeroed; the heads concerned are written as NaN, never as a stale partial's sum.
__device__ unsigned g_tail_faults = 0;

// One query head is merged by FOUR waves: wave part (0..3) owns outputs [32 part, 32 part + 32) of the head; its lane
// (h, q8) owns float4 q8 of those for the splits s = h (mod 8): ns / 8 16-byte loads per lane, the eight split subsets are
// summed with DPP / row swaps.  (A two-heads-per-pass variant - both heads' loads in flight before the first reduction - paid
// when ONE workgroup merged every head; with the helpers back each merger has one head per wave group and the second
// instantiation only made the cold tail longer.)
template <int DD = 128>      // DD = d: rows of 64 dims (lean kernel, d = 64) keep two of the four waves of a head busy
__device__ __forceinline__ void tail_merge_head(const AttnParams &p, int b, int hk, int g, int part, int ns, const float *src, int lane,
                                                bool fault) {
    if (32 * part >= DD) return;      // wave-uniform
    const int q8 = lane & 7, h = lane >> 3;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
    // softmax weights of the splits (lane = split)
    const bool on = lane < ns;
    const int sl = on ? lane : 0;
    const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (sl * p.slot_floats + p.G * DD + g) * 4, 0, 16));
    const float l1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (sl * p.slot_floats + p.G * DD + p.G + g) * 4, 0, 16));
    v4u v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int slot = 8 * k + h;
        const int sc = slot < ns ? slot : ns - 1;                      // clamped: never a conditional load (weight 0)
        if (k < 4 || ns > 32)                                          // wave-uniform: the second half only for more than 32 splits
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (sc * p.slot_floats + g * DD + 32 * part + 4 * q8) * 4, 0, 16);
        else
            v[k] = v4u{0, 0, 0, 0};
    }
    const float m0 = on ? m1 : -INFINITY;
    const float l0 = on ? l1 : 0.f;
    const float mx = wave_max_valu(m0);
    const float ms_ = mx > -INFINITY ? mx : 0.f;
    const float w0 = fast_exp2(m0 - ms_);                               // -inf -> 0 (lanes >= ns: 0)
    // unnormalised sum first, 1 / (sum of w l) at the end: the denominator's reduction runs beside the accumulation
    v4f32 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (8 * k + h), __builtin_bit_cast(int, w0)));
        acc += w * __builtin_bit_cast(v4f32, v[k]);
    }
    const float den = wave_sum_valu(w0 * l0);
    // nothing to attend to: 0; a merge that gave up on a flag: NaN, never a stale partial's sum
    const float inv = fault ? __builtin_nanf("") : den > 0.f ? __builtin_amdgcn_rcpf(den) : 0.f;
    // sum over the eight split subsets: lanes l, l ^ 8 (same 16-lane row), then the four rows
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x = acc[c];
        x += MILLION_DPP(x, 0x128);      // row_ror:8
        acc[c] = rows_sum(x) * inv;
    }
    if (lane < 8) {
        typedef f16 h4 __attribute__((ext_vector_type(4)));
        const h4 o = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
        *(h4 *)(p.out + ((long long)b * p.nh + head0(p, hk) + g) * DD + 32 * part + 4 * q8) = o;
    }
}

template <int MS = 64, bool PV = false, int DD = 128, class ACC>
__device__ __forceinline__ void merge_and_publish(const AttnParams &p, char *smem, int b, int hk, int split, int G, int tid,
                                                  int lane, int wave, bool dbg_on, ACC &O, float m_run, float l_run, TailReq &treq) {
#define STAMP(i) stamp_lds(dbg_on, lane, wave, i)
    const int ns = p.nslots;
    const int bh = b * p.nh_k + hk;
    // LDS words by absolute address (a generic pointer made these FLAT accesses): [1] arrival index, [2] generation,
    // [3] 1 = every split of this (b, kv head) runs on this XCD, [5] give-up bits as this workgroup's ticket returned them,
    // [6] count base of this launch (common.h: record words [2] and [4])
    typedef volatile __attribute__((address_space(3))) int *lds_int_p;
    const lds_int_p tl = (lds_int_p)(size_t)kPartOff;
    // ---- census line / arrival index / generation: requested ~3 us ago by the streaming loop (tail_request); a wave that
    //      had no whole round asks now ----
    if (!treq.done) tail_request(p, bh, ns, wave, lane, treq);
    const unsigned raw_v = (unsigned)treq.idx;
    const int idx_v = (int)(((raw_v >> 8) - treq.base) & 0xffffffu);      // arrival index of this workgroup within this launch
    const unsigned gen_v = treq.gen, cen_v = treq.cen;
    // ---- merge the waves of this workgroup through LDS (tables are dead after the barrier) ----
    l_run = rows_sum(l_run);
    __syncthreads();
    STAMP(4);
    if (wave == 0) {
        const bool all_here = __all(cen_v == tail_xcc() + 1u);
        if (lane == 0) tl[3] = all_here ? 1 : 0;
    }
    const int wstride = G * DD + 2 * kMaxGMfma;           // floats per wave (G = 16: 65 KiB for the 8 waves, the dead tables' space)
    float *scr_l = (float *)smem;
    float *mine = scr_l + wave * wstride;
    {
        const bool hi = lane >= 32;
        const int c32 = lane & 31;
        if constexpr (MS == 640) {     // lean kernel (z-rows): accumulator pi, lane (rg = lane >> 4, n = lane & 15), register i = head i:
                                       // row 4 rg + i = (z = rg >> 1, parity rg & 1, head i), column n = subspace 32 pi + 16 z + n
#pragma unroll
            for (int j = 0; j < DD / 64; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < G) mine[i * DD + 64 * j + 32 * (lane >> 5) + 2 * (lane & 15) + ((lane >> 4) & 1)] = O.t[j][i];
        } else
        if constexpr (MS == 320) {     // d_m = 4 form: column tile j, lane (dq = lane >> 4, n = lane & 15), register i = head i
#pragma unroll
            for (int j = 0; j < DD / 64; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < G) mine[i * DD + 4 * ((lane & 15) + 16 * j) + (lane >> 4)] = O.t[j][i];
        } else
        if constexpr (MS == 16) {      // d_m = 8 form: row tile h, lane (dq = lane >> 4, n = lane & 15), register i = head i
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < G) mine[i * 128 + 8 * (lane & 15) + 4 * h + (lane >> 4)] = O.t[h][i];
        } else
        if constexpr (PV) {      // parity-V tiles O[n][0]: register 4 j + rho = row 8 j + 4 hi + rho = (parity j >> 1, head 8 (j & 1) + 4 hi + rho)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rho = 0; rho < 4; ++rho) {
                    const int g = 8 * (j & 1) + (hi ? 4 + rho : rho);
                    if (g < G) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) mine[g * 128 + 2 * (32 * n + c32) + (j >> 1)] = O[n][0][4 * j + rho];
                    }
                }
        } else if constexpr (MS != 16 && MS != 320 && MS != 640) {
#pragma unroll
        for (int j = 0; j < 2; ++j)                      // tile rows 8 j + 4 hi + rho = register 4 j + rho; j = 1: groups above 8 heads
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) {
                const int g = 8 * j + (hi ? 4 + rho : rho);
                if (g < G) {
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int kk = 0; kk < (PV ? 1 : 2); ++kk)
                            mine[g * 128 + (MS == 64 ? 2 * (32 * n + c32) + kk : 4 * c32 + 2 * n + kk)] = O[n][kk][4 * j + rho];
                }
            }
        }
        if (lane < G) {                                  // lane g: row q' = 0, col g
            mine[G * DD + lane] = m_run;
            mine[G * DD + kMaxGMfma + lane] = l_run;
        }
    }
    __syncthreads();
    // a thread combines the 8 wave partials of 4 consecutive output elements (16-byte LDS reads) and publishes them
    // straight from registers with one 16-byte store into this split's workspace slot: plain (stays in this XCD's L2)
    // when the census says every workgroup of this (b, kv head) runs on this XCD, write-through (sc1) otherwise
    const bool same_xcd = tl[3] != 0;
    const int nsw = (G * (DD / 4) + 63) >> 6;            // waves that store
    float *dst = slot_ptr(p, b, hk, split);
    {
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, 0x7fffffff, 0x00020000);
        for (int q = tid; q < G * (DD / 4); q += kNW * 64) {
            const int g = q / (DD / 4);
            float mw[kNW], lw[kNW];
            v4f32 vw[kNW];
#pragma unroll
            for (int w = 0; w < kNW; ++w) {
                mw[w] = scr_l[w * wstride + G * DD + g];
                vw[w] = *(const v4f32 *)(scr_l + w * wstride + 4 * q);
                lw[w] = scr_l[w * wstride + G * DD + kMaxGMfma + g];
            }
            float Mx = mw[0];
#pragma unroll
            for (int w = 1; w < kNW; ++w) Mx = fmaxf(Mx, mw[w]);
            const float Ms = Mx > -INFINITY ? Mx : 0.f;
            v4f32 acc = {0.f, 0.f, 0.f, 0.f};
            float lsum = 0.f;
#pragma unroll
            for (int w = 0; w < kNW; ++w) {
                const float f = fast_exp2(mw[w] - Ms);      // -inf -> 0
                acc += f * vw[w];
                lsum = fmaf(f, lw[w], lsum);
            }
            if (ns == 1) {      // the only split of this (b, kv head): normalise and write the output (nothing to attend to: 0)
                const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
                typedef f16 h4 __attribute__((ext_vector_type(4)));
                const h4 o = {(f16)(acc[0] * inv), (f16)(acc[1] * inv), (f16)(acc[2] * inv), (f16)(acc[3] * inv)};
                *(h4 *)(p.out + ((long long)b * p.nh + head0(p, hk)) * DD + 4 * q) = o;
            } else if (same_xcd) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, acc), rsrc, q * 16, 0, 0);
                if (q % (DD / 4) == 0) {      // (a slot is laid out for p.G heads: the last part of an odd head group holds fewer, G < p.G)
                    dst[p.G * DD + g] = Mx;
                    dst[p.G * DD + p.G + g] = lsum;
                }
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, acc), rsrc, q * 16, 0, 16 /* sc1 */);
                if (q % (DD / 4) == 0) {
                    st_agent(dst + p.G * DD + g, Mx);
                    st_agent(dst + p.G * DD + p.G + g, lsum);
                }
            }
        }
    }
    STAMP(5);
    if (wave == kNW - 1 && lane == 0) { tl[1] = idx_v; tl[2] = (int)gen_v; tl[5] = (int)(raw_v & 0xffu); tl[6] = (int)treq.base; }      // the index, the generation, the give-up bits and the base have arrived
    // a storing wave's partial is out of the CU (in L2, or in memory) when its vmcnt retires; the flag is raised behind the
    // barrier every storing wave then joins (cdna_hip_programming.md Guideline 16, R1)
    if (ns > 1 && wave < nsw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // wave scratch is dead; index and generation are visible to every wave
    MILLION_STAMP(p, 10);
    const int idx = tl[1];
    const unsigned want = (unsigned)tl[2] + 1u;
    if (ns > 1 && tid == 0) {
        __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)tail_flags(p, bh), 0, kFlagWords * 4, 0x00020000);
        if (same_xcd) __builtin_amdgcn_raw_buffer_store_b32(want, rf, split * 4, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(want, rf, split * 4, 0, 16 /* sc1 */);
    }
    if (p.dbg && tid == 0)      // diagnostics: slot 12 = 1 + "stored plain (every split on this XCD)", slot 13 = 1 + arrival index
        { unsigned long long *d_ = p.dbg + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * kStampWaves * kStampSlots; d_[12] = 1 + (same_xcd ? 1 : 0); d_[13] = 1 + idx; }
    // ---- the merge (round 4).  The workgroup whose arrival index is ns - 1 - the PRIMARY - is responsible for every head:
    //      every workgroup it waits for has taken its index, i.e. is resident, on its way to its own flag store, and waits
    //      for nothing itself, so its polls end under ANY dispatch order and residency.  The nm - 1 workgroups that arrived
    //      just before it are HELPERS (merger k = idx - (ns - nm) takes heads k, k + nm, ...).  A helper needs the flags of
    //      workgroups that arrived AFTER it and may not even be dispatched (more workgroups than resident slots; two launches
    //      sharing the chip), so its patience is BOUNDED (48 polls of the flags, ~30 us).  What happens then is decided on ONE
    //      word, the ticket word of the (b, kv head): arrival count in bits 31:8 (a ticket is an atomic add of 256), give-up
    //      bits 7:0.  A helper wave out of patience ORs bit k in and looks at the count the atomic returns: incomplete - the
    //      last ticket comes later and RETURNS the bit to the primary (fetched ~3 us before its tail): leave, the primary merges
    //      head k too; complete - every workgroup is resident, the flags will come: poll on and merge.  Atomics on one word are
    //      serialised, so there is no window between the two cases, and nobody polls anybody's status.  The count is never
    //      reset (a straggling wave must never read a count that looks incomplete): a launch's indices are counted from
    //      `base`, which the primary moves on by ns at the end, together with the generation; it also clears the bits (one set
    //      behind the clear costs the next launch's primary a redundant merge of the same values, nothing else).
    //      (Round 3 let all nm mergers wait for flags without bound: when every resident workgroup is such a merger the launch
    //      stalls for the spin bound and merges stale partials.  The primary alone pulls all 64 KiB of a (b, kv head)'s
    //      partials through ONE CU: +1.2 us per launch at one request; helpers that report through status words the primary
    //      polls: +0.8 us, profiles/r04_ab_merge.txt.)  The host sets nmerge = 1 when the grid does not fit the chip.
    //      Every merging wave polls the flags itself (lane = split) and merges behind its own match; four waves per head:
    //      waves 0-3 heads k, k + 2 nm, ..., waves 4-7 heads k + nm, k + 3 nm, ... ----
    const int nm = treq.nm;
    const int km = idx - (ns - nm);                          // merger number; nm - 1 = the primary
    if (ns > 1 && km >= 0) {
        const bool primary = idx == ns - 1;
        const float *src = p.ws_part + (long long)bh * ns * p.slot_floats;
        __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)tail_flags(p, bh), 0, kFlagWords * 4, 0x00020000);
        const int fo = (lane < ns ? lane : 0) * 4;
        const int tt = treq.tt;
        // The common path is short and straight (this code runs once per workgroup from a cold instruction cache: round 4
        // measured +0.3 us from the barrier to "flags seen" and +0.35 us over the merge for a tail with loops over head masks
        // and kernel arguments read here): poll, merge; what happens when a helper's patience runs out, and the primary's
        // extra heads, sit behind unlikely branches.
        int state = 0;      // 0 = not polled, 1 = every flag seen, 2 = fault, 3 = gave up
        for (int g = km + (wave >> 2) * nm; g < G; g += 2 * nm) {
            if (state == 0) {
                // a helper's patience: ~30 us of polls (test modes: none); the primary's: the fault bound
                const int bound = primary ? (1 << 20) : (tt ? 0 : 48);
                state = 2;
                for (int spin = 0; spin < bound; ++spin) {
                    const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(rf, fo, 0, 16);
                    if (__all(f == want)) { state = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (__builtin_expect(state != 1 && !primary, 0)) {
                    // out of patience (this WAVE: the decision needs no barrier).  An atomic OR of bit km into the ticket word:
                    // if it returns an incomplete count the primary's ticket comes later and returns the bit - leave; if the
                    // count is complete every workgroup is resident and the flags will come - poll on to the fault bound and
                    // merge (the primary may have seen a bit another wave of this workgroup set: it then merges the head as
                    // well, same values).  A bit set after the primary's clear survives into the next launch and costs its
                    // primary one merge more, nothing else.  Test mode 1: the bits were all set in the prologue.
                    unsigned old = 0;
                    if (tt != 1 && lane == 0) old = __hip_atomic_fetch_or(tail_rec(p, bh) + 2, 1u << km, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
                    if (tt == 1 || (((old >> 8) - (unsigned)tl[6]) & 0xffffffu) < (unsigned)ns) state = 3;
                    else
                        for (int spin = 0; spin < (1 << 20); ++spin) {
                            const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(rf, fo, 0, 16);
                            if (__all(f == want)) { state = 1; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                }
                MILLION_STAMP(p, 11);
                if (state == 3) break;
                if (__builtin_expect(state == 2, 0) && lane == 0) atomicAdd(&g_tail_faults, 1u);      // this wave's outputs are written as NaN
            }
            tail_merge_head<DD>(p, b, hk, g, wave & 3, ns, src, lane, state == 2);
        }
        // the primary also merges the heads of the helpers that gave up before it took its index (bits of its own ticket):
        // heads h, h + nm, ... of helper h, the same four-waves-per-head split
        const unsigned gave = primary ? (unsigned)tl[5] & ((1u << (nm - 1)) - 1u) : 0u;
        if (__builtin_expect(gave != 0, 0)) {
            if (state == 0) {      // waves 4-7 of a primary with one head of its own have not polled yet
                state = 2;
                for (int spin = 0; spin < (1 << 20); ++spin) {
                    const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(rf, fo, 0, 16);
                    if (__all(f == want)) { state = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (state == 2 && lane == 0) atomicAdd(&g_tail_faults, 1u);
            }
            int pos = 0;
            for (int h = 0; h < nm - 1; ++h)
                if (gave >> h & 1u)
                    for (int g = h; g < G; g += nm) {
                        if ((pos & 1) == (wave >> 2)) tail_merge_head<DD>(p, b, hk, g, wave & 3, ns, src, lane, state == 2);
                        ++pos;
                    }
        }
    }
    if (idx == ns - 1 && tid == 0) {
        // the workgroup that arrived last: its wave 0 has seen every flag of this launch (or ns == 1), so every workgroup of
        // this (b, kv head) has read the census line and the generation and stored its partial
        unsigned *rec = tail_rec(p, bh);
        __hip_atomic_fetch_and(rec + 2, ~0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // give-up bits off; the count stays
        __hip_atomic_store(rec + 4, ((unsigned)tl[6] + (unsigned)ns) & 0xffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 3, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // fused append with device-resident lengths: every workgroup of batch b has read its lengths once all nh_k
        // heads have got this far; the last of them advances r
        if (p.k_new && p.dev_lengths_w) {
            const int t2 = __hip_atomic_fetch_add(p.ws_cnt2 + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 == p.nh_k - 1) {
                __hip_atomic_store(p.ws_cnt2 + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p.dev_lengths_w[b * 4 + 1] += 1;
            }
        }
    }
    if (idx == ns - 1 && wave == 0) {      // census line back to zero, behind this wave's own poll (write-through: the next launch may run anywhere)
        __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(tail_flags(p, bh) + kFlagWords), 0, kFlagWords * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(0u, rc, lane * 4, 0, 16 /* sc1 */);
    }
    MILLION_STAMP(p, 6);
    if (dbg_on && lane < kStampSlots) {              // copy this wave's LDS stamps out (slots it wrote)
        const unsigned long long v =
            *(volatile __attribute__((address_space(3))) unsigned long long *)(size_t)(kStampOff + (wave * kStampSlots + lane) * 8);
        if (v) p.dbg[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * kStampWaves + wave) * kStampSlots + lane] = v;
    }
#undef STAMP
}

template <bool HAS_CODES, int MS = 64>
__global__ __launch_bounds__(kNW * 64, 2) void attn_mfma_kernel(AttnParams p) {
    typedef typename std::conditional<MS == 64, UnitCodes, UnitCodes32>::type Unit;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int bh = blockIdx.y;
    const int b = bh / p.nh_k, hk = bh % p.nh_k;
    const int G = p.G;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i dl = {p.T, p.r, p.rstart, 0};
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    const bool dbg_on = p.dbg != nullptr;
#define STAMP(i) stamp_lds(dbg_on, lane, wave, i)
    stamp_lds_clear(dbg_on, lane, wave);
    STAMP(0);
    const int q4 = lane >> 4, c16 = lane & 15;

    // B operand of the score MFMA: the query heads (cols), K = 32 dims per step.
    v8f16 qb[4];
    {
        const f16 *qv = p.q + ((long long)b * p.nh + head0(p, hk) + (c16 < G ? c16 : 0)) * 128 + 32 * q4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v4u t = *(const v4u *)(qv + 8 * s);
            if (c16 >= G) t = v4u{0, 0, 0, 0};
            qb[s] = __builtin_bit_cast(v8f16, t);
        }
    }
    // fused append: the new token's K/V row is parked in the window by the last wave of split 0; its two
    // loads are requested here (oldest loads of that wave) and stored after the first score pass
    const bool append_wave = p.k_new && split == 0 && wave == kNW - 1;      // wave-uniform
    h2 new_k = {}, new_v = {};
    if (append_wave) {
        new_k = *(const h2 *)(p.k_new + (long long)bh * 128 + 2 * lane);
        new_v = *(const h2 *)(p.v_new + (long long)bh * 128 + 2 * lane);
    }
    // ---- K codebook (8 x 16 B per thread): requested before anything that depends on a length.  Every
    //      workgroup needs the same 64 KiB at the same time: each starts at its own chunk so that the CUs
    //      do not walk the L2 channels in lockstep. ----
    v4u tabk[8];
    const int rot = (blockIdx.x + 5 * blockIdx.y) & 7;
    {
        const v4u *ks = (const v4u *)p.k_tab;
#pragma unroll
        for (int i = 0; i < 8; ++i) tabk[i] = ks[((i + rot) & 7) * (kNW * 64) + tid];
    }
    if (p.dev_lengths)      // issue + wait in ONE statement (see load_pids4), after q and the table have been requested
        asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dl) : "s"(p.dev_lengths), "s"((unsigned)b * 16u) : "memory");
    int T = dl[0], r_old = dl[1], rstart = dl[2];
    clamp_lengths(p, T, r_old, rstart);            // T <= the host bound the grid was sized for; r, start inside the window
    const int r = r_old + (p.k_new ? 1 : 0);       // fused append: the new token is window row r_old
    // The T tokens that are actually there (device-resident lengths: the host sized the grid for its BOUND on T) are
    // dealt to the nsplit splits in whole 32-token units, as evenly as units allow: the first (units % nsplit) splits
    // carry one unit more.  (Uniform split lengths rounded up to a page left the last splits short or empty and made
    // most waves of the others run a fifth unit as soon as T passed 32 x 1024.)
    const int units_total = (T + 31) >> 5;
    const int units_q = units_total / p.nsplit, units_r = units_total - units_q * p.nsplit;
    const int u_begin = split * units_q + (split < units_r ? split : units_r);
    const int t_begin = u_begin << 5;
    const int t_end = min((u_begin + units_q + (split < units_r ? 1 : 0)) << 5, T);
    const int n_units = (t_end - t_begin + 31) >> 5;
    const int n_mine = n_units > wave ? (n_units - wave + kNW - 1) / kNW : 0;   // units of this wave
    const int n_pass = (n_mine + kRing - 1) / kRing;
    // unit j of this wave starts at token t_begin + 32*(wave + j*kNW); slots past the last unit re-request
    // the unit that holds token T-1 (HAS_CODES guarantees the host bound T >= 1; with device-resident
    // lengths a runtime T of 0 reads page 0, which must be a valid page: million_hip.h)
    const int T_ld = T > 0 ? T : 1;
    const int t_last = (T_ld - 1) & ~31;
#define UNIT_T(j) ((j) < n_mine ? t_begin + 32 * (wave + (j) * kNW) : t_last)

    // ---- residual window rows of this split (list j = split, split + nsplit, ... < r), dealt to the waves
    //      round-robin; this wave's tile is requested BEFORE the code bytes: the counted wait in front of the
    //      K-codebook store then covers these few L2-resident rows, not the HBM-bound code loads behind them ----
    const int rcnt = split < r ? (r - split + p.nsplit - 1) / p.nsplit : 0;
    const bool has_res = kResRows * wave < rcnt;       // wave-uniform
    const f16 *kr = p.k_res + b * p.res_sb + hk * p.res_sh;
    const f16 *vr = p.v_res + b * p.res_sb + hk * p.res_sh;
    ResTile rt;
    if (has_res) load_res_tile<MS>(p, bh, kr, vr, wave, rcnt, split, rstart, r_old, lane, rt);

    // ---- the K bytes of the whole ring, then the K codebook goes to LDS ----
    Unit ring[kRing];
    PidPair pid4[kRing];
    if (HAS_CODES) {
        int pg[kRing];
#pragma unroll
        for (int k = 0; k < kRing; ++k) pg[k] = UNIT_T(k) >> p.ps_shift;
        load_pids4(p, bh, pg, pid4);        // one scalar round trip, after q and the K codebook have been requested
#pragma unroll
        for (int k = 0; k < kRing; ++k) load_unit_k(p, b, hk, pid4[k], UNIT_T(k), T_ld, lane, ring[k]);
    }
    STAMP(7);
    {
        v4u *ld = (v4u *)smem;
#pragma unroll
        for (int i = 0; i < 8; ++i) ld[((i + rot) & 7) * (kNW * 64) + tid] = tabk[i];
    }
    STAMP(8);
    __syncthreads();     // no LDS-DMA in flight: lgkmcnt(0) + s_barrier, the code bytes stay in flight
    STAMP(1);
    tail_mark_xcd(p, bh, split, wave, lane);      // this split's slot of the XCD census (tail)

    // ---- everything else is requested BETWEEN the score units of the first group (the K bytes and the K
    //      codebook are there; a wave that first issued all its remaining loads would sit in a blocked issue
    //      sequence while the LDS pipe idles): after unit 0 the V bytes of units 0-1, after unit 1 those of
    //      units 2-3, after unit 2 the V codebook ----
    v4u tabv[8];
#define ISSUE_AFTER_UNIT(K)                                                                                        \
    if ((K) == 0) {                                                                                                \
        if (HAS_CODES) {                                                                                           \
            _Pragma("unroll") for (int k2 = 0; k2 < kRing / 2; ++k2)                                               \
                load_unit_v(p, pid4[k2], UNIT_T(k2), lane, ring[k2]);                                              \
        }                                                                                                          \
    } else if ((K) == 1) {                                                                                         \
        if (HAS_CODES) {                                                                                           \
            _Pragma("unroll") for (int k2 = kRing / 2; k2 < kRing; ++k2)                                           \
                load_unit_v(p, pid4[k2], UNIT_T(k2), lane, ring[k2]);                                              \
        }                                                                                                          \
    } else if ((K) == 2) {                                                                                         \
        const v4u *vs = (const v4u *)p.v_tab_col;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) tabv[i] = vs[((i + rot) & 7) * (kNW * 64) + tid];            \
    }

    float m_run = -INFINITY, l_run = 0.f;
    v16f32 O[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[n][kk][i] = 0.f;
    STAMP(2);

    const unsigned kbase = (unsigned)q4 * 16u * 1024u;                 // K row image: m = 16*q4 + ...
    // V col image: entry (c, m) at c*256 + m*(2*d_m); the code byte goes to address bits 8..15 by v_perm
    const unsigned vconst0 = (unsigned)kVBase | ((unsigned)(lane & 31) << (MS == 64 ? 2 : 3));        // m = c
    const unsigned vconst1 = (unsigned)kVBase | ((unsigned)((lane & 31) + 32) << 2);  // m = 32 + c (M = 64 only)

    // ---- groups of kRing units: SCORE pass for the whole group (K codebook only), one softmax update per
    //      group, then the VALUE pass.  The first group also carries this wave's residual tile, and the V
    //      codebook goes to LDS between its two passes, so the first scores are computed while V is arriving. ----
#define GROUP(PASS, MASKV, FIRST, REFILL)                                                                          \
    {                                                                                                              \
        float sc[kRing][8], scr[4];                                                                                \
        /* page ids of the units this group refills the ring with: ONE scalar round trip per group, up front (a   */  \
        /* per-unit s_load + lgkmcnt(0) inside the value pass drained the LDS gather queue four times per group)  */  \
        PidPair pidr[kRing];                                                                                       \
        if (HAS_CODES && (REFILL)) {                                                                               \
            int pgr[kRing];                                                                                        \
            _Pragma("unroll") for (int k = 0; k < kRing; ++k) pgr[k] = UNIT_T(((PASS) + 1) * kRing + k) >> p.ps_shift; \
            load_pids4(p, bh, pgr, pidr);                                                                          \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k) {                                                        \
            const int j = (PASS) * kRing + k;                                                                      \
            if (HAS_CODES && j < n_mine)                                                                           \
                score_unit<MASKV>(ring[k].k, qb, t_begin + 32 * (wave + j * kNW), t_end, p.scale_log2e, lane,      \
                                  kbase, sc[k]);                                                                   \
            else                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) sc[k][i] = -INFINITY;                                \
            if (FIRST) {                                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                STAMP(16 + 2 * k);                                                                                 \
                ISSUE_AFTER_UNIT(k)                                                                                \
                STAMP(17 + 2 * k);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }                                                                                                      \
        }                                                                                                          \
        if ((FIRST) && has_res) score_res_tile(rt, qb, p.scale_log2e, wave, rcnt, lane, scr);                      \
        else _Pragma("unroll") for (int i = 0; i < 4; ++i) scr[i] = -INFINITY;                                     \
        float mx = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));                                            \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k)                                                          \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) mx = fmaxf(mx, sc[k][i]);                                \
        mx = rows_max(mx);                                                                                         \
        const float m_new = fmaxf(m_run, mx);                                                                      \
        const float m_safe = m_new > -INFINITY ? m_new : 0.f;                                                      \
        const float alpha = fast_exp2(m_run - m_safe);                                                             \
        if (!(FIRST) && __any(m_new > m_run && m_run > -INFINITY)) {                                               \
            rescale_heads(O, alpha, G, lane);                                                                      \
        }                                                                                                          \
        float ls = 0.f;                                                                                            \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k)                                                          \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
                sc[k][i] = fast_exp2(sc[k][i] - m_safe);                                                           \
                ls += sc[k][i];                                                                                    \
            }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            scr[i] = fast_exp2(scr[i] - m_safe);                                                                   \
            ls += scr[i];                                                                                          \
        }                                                                                                          \
        l_run = l_run * alpha + ls;                                                                                \
        m_run = m_new;                                                                                             \
        if (FIRST) {                                                                                               \
            STAMP(24);                                                                                             \
            if (append_wave) {                                                                                     \
                int row_n = rstart + r_old;                                                                        \
                row_n = row_n >= p.rcap ? row_n - p.rcap : row_n;                                                  \
                const long long o = b * p.res_sb + hk * p.res_sh + (long long)row_n * 128 + 2 * lane;              \
                *(h2 *)(p.k_res_w + o) = new_k;                                                                    \
                *(h2 *)(p.v_res_w + o) = new_v;                                                                    \
            }                                                                                                      \
            v4u *ld = (v4u *)(smem + kVBase);                                                                      \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) ld[((i + rot) & 7) * (kNW * 64) + tid] = tabv[i];        \
            STAMP(12);                                                                                             \
            __syncthreads();                                                                                       \
            STAMP(13);                                                                                             \
            if (has_res) value_res_tile(rt, scr, O);                                                               \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k) {                                                        \
            const int j = (PASS) * kRing + k;                                                                      \
            if (HAS_CODES && j < n_mine) value_unit(ring[k].v, sc[k], vconst0, vconst1, O);                        \
            if (HAS_CODES && (REFILL)) load_unit_pid(p, b, hk, pidr[k], UNIT_T(j + kRing), T_ld, lane, ring[k]);   \
        }                                                                                                          \
    }

    // group 0 (every wave, also one without units: it carries the V-codebook barrier); masked because it
    // may be the last; refills only if more groups follow
    GROUP(0, true, true, n_pass > 1)
    // middle groups: full units, unconditional refills (counted waits, see the note above load_unit)
    for (int pass = 1; pass + 1 < n_pass; ++pass) GROUP(pass, false, false, true)
    // last group: masked, no refills
    if (n_pass > 1) GROUP(n_pass - 1, true, false, false)
#undef GROUP
#undef ISSUE_AFTER_UNIT
#undef UNIT_T
    STAMP(3);
    TailReq treq;
    treq.idx = 0; treq.gen = 0; treq.cen = 0; treq.base = 0; treq.done = false;
    merge_and_publish<MS>(p, smem, b, hk, split, G, tid, lane, wave, dbg_on, O, m_run, l_run, treq);
#undef STAMP
}

// ---- value side of a 32-token unit in pieces, for the pipelined kernel ---------------------------------
// The A operand of the value MFMA (rows = heads, K = 16 tokens) wants, in lane (h, head), the probabilities of tokens
// 16h + 8s + j (token step s, j = 0..7).  The score MFMAs leave pr[4*g2 + rho] = row 4q' + rho of score tile g2 in lane
// (q', head), and WHICH token a tile row is, is the K gather's choice.  With tile g2, row i = token 8*(i >> 2) + 4*g2 +
// (i & 3) (stream_token_of_row), lane rows 0 and 2 - the lanes the value MFMA reads for h = 0 / 1 - already hold the
// eight tokens of step s = 0 in operand order: value_prep is four cvt_pk and nothing else.  Step s = 1 wants what lane
// rows 1 and 3 hold; value_next_step brings it over IN PLACE with four v_permlane16_swap, once the s = 0 steps have
// issued.  (Round 1-2 form: tiles of 16 consecutive tokens, 2 permlane32_swap + 4 copies + 4 permlane16_swap per unit.)
__device__ __forceinline__ int stream_token_of_row(int g2, int i) { return 8 * (i >> 2) + 4 * g2 + (i & 3); }
__device__ __forceinline__ void value_prep(const float (&pr)[8], unsigned (&P)[4]) {
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            h2 t = {(f16)pr[g2 * 4 + 2 * i], (f16)pr[g2 * 4 + 2 * i + 1]};
            P[2 * g2 + i] = __builtin_bit_cast(unsigned, t);
        }
}
// P (token step 0) -> P (token step 1): lane rows 0 / 2 receive what lane rows 1 / 3 held.  v_permlane16_swap(a, b)
// returns {a with its odd rows replaced by b's even rows, b with its even rows replaced by a's odd rows}.
__device__ __forceinline__ void value_next_step(unsigned (&P)[4]) {
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        const v2u t = __builtin_amdgcn_permlane16_swap(P[2 * g2], P[2 * g2 + 1], false, false);
        const unsigned v = t[0], s = t[1];      // s rows 0 / 2 = P[2 g2] rows 1 / 3;  s rows 1 / 3 = P[2 g2 + 1] rows 1 / 3
        const v2u u = __builtin_amdgcn_permlane16_swap(s, v, false, false);
        const unsigned u0 = u[0], u1 = u[1];
        P[2 * g2] = u0;                         // rows 0 / 2 = s rows 0 / 2 (kept)
        P[2 * g2 + 1] = u1;                     // rows 0 / 2 = s rows 1 / 3
    }
}
// the 8 centroid gathers of value step st = 2n + s (subspaces 32n.., tokens 16h + 8s + j)
__device__ __forceinline__ void v_gather(const v4u (&vc)[2], int st, unsigned vconst0, unsigned vconst1, unsigned (&e)[8]) {
    const unsigned vconst = (st >> 1) ? vconst1 : vconst0;
    const unsigned w0 = vc[st >> 1][2 * (st & 1)], w1 = vc[st >> 1][2 * (st & 1) + 1];
    e[0] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020400u));
    e[1] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020500u));
    e[2] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020600u));
    e[3] = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020700u));
    e[4] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020400u));
    e[5] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020500u));
    e[6] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020600u));
    e[7] = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020700u));
}
// pack the gathered centroids (even dims -> B0, odd dims -> B1) and accumulate
__device__ __forceinline__ void v_step(const unsigned (&e)[8], const unsigned (&Ps)[4], v16f32 (&On)[2]) {
    const v8f16 B0 = as_v8f16(__builtin_amdgcn_perm(e[1], e[0], 0x05040100u), __builtin_amdgcn_perm(e[3], e[2], 0x05040100u),
                              __builtin_amdgcn_perm(e[5], e[4], 0x05040100u), __builtin_amdgcn_perm(e[7], e[6], 0x05040100u));
    const v8f16 B1 = as_v8f16(__builtin_amdgcn_perm(e[1], e[0], 0x07060302u), __builtin_amdgcn_perm(e[3], e[2], 0x07060302u),
                              __builtin_amdgcn_perm(e[5], e[4], 0x07060302u), __builtin_amdgcn_perm(e[7], e[6], 0x07060302u));
    const v8f16 A = as_v8f16(Ps[0], Ps[1], Ps[2], Ps[3]);
    On[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B0, On[0], 0, 0, 0);
    On[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B1, On[1], 0, 0, 0);
}

// =====================================================================================================
// Pieces of the pipelined schedule (used by the streaming kernel below).
//
// In the grouped kernel above a wave runs "score pass of 4 units" (LDS-bound: ~7 LDS cycles per random K gather), then
// "value pass of 4 units" (issue-bound: v_perm address + pack work and the 32x32x16 MFMAs), one after the other, and
// with one workgroup per CU nothing else fills the idle pipe.  In the pipelined schedule both codebooks are in LDS
// before the loop, the softmax is online PER UNIT, and the value steps of unit u are interleaved instruction by
// instruction with the score stages of unit u + 1, so the LDS pipe and the vector/matrix issue work at the same time.
// =====================================================================================================
// online softmax over N new scores of this lane's column (head): updates (m_run, l_run), rescales O when a
// running maximum moves, turns the scores into probabilities in place
template <int N, bool PV = false, class ACC>
__device__ __forceinline__ void softmax_online(float (&sc)[N], float &m_run, float &l_run, ACC &O, int G, int lane) {
    float mx = sc[0];
#pragma unroll
    for (int i = 1; i < N; ++i) mx = fmaxf(mx, sc[i]);
    mx = rows_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = m_new > -INFINITY ? m_new : 0.f;
    const float alpha = fast_exp2(m_run - m_safe);
    if (__any(m_new > m_run && m_run > -INFINITY)) {
        rescale_acc(O, alpha, G, lane);
    }
    float ls = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        sc[i] = fast_exp2(sc[i] - m_safe);
        ls += sc[i];
    }
    l_run = l_run * alpha + ls;
    m_run = m_new;
}

// same on RAW scores (q.k, masked to -inf where needed): the 1/sqrt(d)*log2(e) factor c > 0 is folded into the exp2
// argument (one fma per score instead of a multiply and a subtract), the running maximum stays in the scaled domain
// v_max3_f32 without the two canonicalising v_max x, x that fmaxf() of an MFMA result costs under IEEE mode (the operands
// are never signalling NaNs); plain asm, not volatile: the scheduler may move it
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Softmax state of this lane's head in the streaming kernel.  m is the softmax REFERENCE, not necessarily the running
// maximum: it moves (cross-lane maximum, alpha, rescale of O and l) only when a raw score of the unit exceeds thr_raw =
// (m + 2^3) / c, i.e. when a probability would exceed 2^8 (fp16 operand of the value MFMA: exact up to 2^15).  The common
// unit needs no cross-lane reduction: four v_max3, one compare, a wave-uniform branch.  Any reference gives the same
// softmax; the merges downstream only need (m, l, O) to be consistent.  neg_ref = -(m, or 0 while m = -inf) and thr_raw
// are kept in registers so that the common path recomputes neither.
struct SoftRef {
    float m, l, neg_ref, thr_raw;
    float idle = 0.f;      // -inf in the lanes of score columns >= G (no query head): their probabilities come out as exact zeros, so
                           // the idle rows of the value MFMA's A operand multiply zeros (round 4: the MFMAs set the chip's clock -
                           // tools/micro/core_micro.hip "MFMA -> 1 VALU": 1.80 -> 2.31 GHz - and zero operands draw less)
    __device__ __forceinline__ void set(float m_, float l_, float inv_c) {
        m = m_; l = l_;
        neg_ref = (m_ > -INFINITY ? -m_ : 0.f) + idle;
        thr_raw = (m_ + 8.0f) * inv_c;          // -inf while nothing has been seen: the first finite score moves it
    }
};
template <int N, bool PV = false, class ACC>
__device__ __forceinline__ void softmax_online_raw(float (&sc)[N], float c, float inv_c, SoftRef &st, ACC &O, int G, int lane) {
    static_assert(N == 8, "one 32-token unit: 8 scores per lane");
    float mx = max3_raw(sc[0], sc[1], sc[2]);
    mx = max3_raw(mx, sc[3], sc[4]);
    mx = max3_raw(mx, sc[5], sc[6]);
    mx = max3_raw(mx, sc[7], sc[7]);
    if (__any(mx > st.thr_raw)) {
        const float m_new = fmaxf(st.m, rows_max(mx) * c);
        const float m_safe = m_new > -INFINITY ? m_new : 0.f;
        const float alpha = fast_exp2(st.m - m_safe);
        if (__any(m_new > st.m && st.m > -INFINITY)) {
            rescale_acc(O, alpha, G, lane);
        }
        st.set(m_new, st.l * alpha, inv_c);
    }
    float ls = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        sc[i] = fast_exp2(fmaf(sc[i], c, st.neg_ref));
        ls += sc[i];
    }
    st.l += ls;
}

// =====================================================================================================
// Streaming kernel: the pipelined schedule for ANY split length, M = 64, M = 32 and (up to 4 query heads per kv head) M = 16.
//
// What changed against round 1's 4-unit pipelined kernel (which it replaced):
//   * units are dealt to (split, wave) by PAGE, strided: page (j * ppr + wave / upp) * nsplit + split goes to round j of
//     the wave (upp = units per page, ppr = 8 / upp pages per workgroup and round), so where a wave reads does not depend
//     on the context length T - only HOW MANY units it has does.  The page ids of a wave's first 64 rounds are ONE
//     vector load (lane = round) issued as the first instruction; each unit's id is then a v_readlane.  No scalar-cache
//     round trip sits between the kernel start and the first code request, device-resident lengths included
//     (round 1: lengths -> split range -> page ids -> codes, two dependent s_load round trips);
//   * the block "values of unit j | scores of unit j + 1" repeats for as many units as the wave has: whole rounds of
//     four units run in a loop (ring slot = unit & 3, the slot of unit j refilled with unit j + 4 right behind it, never
//     in a conditional: rounds past the last unit re-read it), the up to three units beyond the whole rounds run one by
//     one from the ring slots the last round refilled.  (hipcc keeps the loop at ~235 VGPRs only as long as no branch
//     leaves it with the pipeline state live: early exits, a switch over the slot, or a remainder chain of blocks each
//     spilled hundreds of registers; a self-contained unit behind a branch does not.)
//   * whole pages belong to one wave pair, so rows past T - 1 of the last unit stay inside an allocated page: no row
//     clamping on paged K.
// MODE 0: K and V paged, int32 ids (PagedPQCache).  MODE 1: row-major K, V in dense scratch pages (the reference's
// 10-arg layout after the transpose).  MODE 2: anything else, by run-time flags.
// =====================================================================================================
template <int MS> struct StreamTypes;
template <> struct StreamTypes<64> { typedef UnitCodes Unit; typedef unsigned E[8]; };
template <> struct StreamTypes<32> { typedef UnitCodes32 Unit; typedef unsigned E[2][8]; };
template <> struct StreamTypes<16> { typedef UnitCodes16 Unit; typedef unsigned E[8]; };
template <> struct StreamTypes<320> { typedef UnitCodes32D Unit; typedef unsigned E[8]; };      // M = 32, d_m = 4 form

// the K gathers of score stage st (0..7) of a unit; CL2 = log2 of the centroids per subspace (8: C = 256, 7: C = 128):
// a subspace's row of the K row image is (4 << CL2) bytes at M = 64 and (8 << CL2) at M = 32, a stage covers 16 << CL2
template <int CL2>
__device__ __forceinline__ void st_kgather(const UnitCodes &u, int st, unsigned kbase, unsigned (&a)[4]) {
    const unsigned w = u.k[st >> 2][st & 3], base = kbase + (st & 3) * (16u << CL2);
    a[0] = lds32(base + 0 * (4u << CL2) + ((w & 0xffu) << 2));
    a[1] = lds32(base + 1 * (4u << CL2) + (((w >> 8) & 0xffu) << 2));
    a[2] = lds32(base + 2 * (4u << CL2) + (((w >> 16) & 0xffu) << 2));
    a[3] = lds32(base + 3 * (4u << CL2) + ((w >> 24) << 2));
}
template <int CL2>
__device__ __forceinline__ void st_kgather(const UnitCodes32 &u, int st, unsigned kbase, unsigned (&a)[4]) {
    const unsigned w = u.k[st >> 2][(st & 3) >> 1], base = kbase + (st & 3) * (16u << CL2);
    const unsigned sh = 16 * (st & 1);
    const v2u lo = lds64(base + 0 * (8u << CL2) + (((w >> sh) & 0xffu) << 3));
    const v2u hi = lds64(base + 1 * (8u << CL2) + (((w >> (sh + 8)) & 0xffu) << 3));
    a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
}
template <int CL2>
__device__ __forceinline__ void st_kgather(const UnitCodes32D &u, int st, unsigned kbase, unsigned (&a)[4]) {      // as UnitCodes32
    const unsigned w = u.k[st >> 2][(st & 3) >> 1], base = kbase + (st & 3) * (16u << CL2);
    const unsigned sh = 16 * (st & 1);
    const v2u lo = lds64(base + 0 * (8u << CL2) + (((w >> sh) & 0xffu) << 3));
    const v2u hi = lds64(base + 1 * (8u << CL2) + (((w >> (sh + 8)) & 0xffu) << 3));
    a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
}
// M = 16: stage st = (tile st >> 2, k-step st & 3): lane quarter q4 covers subspace 4 q4 + (st & 3) - byte (st & 3) of its code
// word - whose 16-byte entry (8 dims) IS the lane's half-row of the A operand; kbase = 4 q4 subspace rows as for the others
template <int CL2>
__device__ __forceinline__ void st_kgather(const UnitCodes16 &u, int st, unsigned kbase, unsigned (&a)[4]) {
    const unsigned code = (u.k[st >> 2] >> (8 * (st & 3))) & 0xffu;
    const v4u x = lds128(kbase + (st & 3) * (16u << CL2) + (code << 4));
    a[0] = x[0]; a[1] = x[1]; a[2] = x[2]; a[3] = x[3];
}
// the V gathers of value step i (M = 64: 4 steps of 8 four-byte gathers; M = 32: 2 steps of 8 eight-byte gathers)
__device__ __forceinline__ void st_vgather(const UnitCodes &u, int i, unsigned vconst0, unsigned vconst1, unsigned (&e)[8]) {
    v_gather(u.v, i, vconst0, vconst1, e);
}
__device__ __forceinline__ void st_vgather(const UnitCodes32 &u, int i, unsigned vconst0, unsigned, unsigned (&e)[2][8]) {
    const unsigned w0 = u.v[0][2 * i], w1 = u.v[0][2 * i + 1];
    const unsigned sel[4] = {0x03020400u, 0x03020500u, 0x03020600u, 0x03020700u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const v2u t = lds64(__builtin_amdgcn_perm(j < 4 ? w0 : w1, vconst0, sel[j & 3]));
        e[0][j] = t[0];
        e[1][j] = t[1];
    }
}
// value step i: pack the gathered centroids and accumulate
__device__ __forceinline__ void st_vstep(const unsigned (&e)[8], const unsigned (&Ps)[4], int i, v16f32 (&O)[2][2]) {
    v_step(e, Ps, O[i >> 1]);
}
__device__ __forceinline__ void st_vstep(const unsigned (&e)[2][8], const unsigned (&Ps)[4], int, v16f32 (&O)[2][2]) {
    v_step(e[0], Ps, O[0]);      // dims 4m + 0, 1
    v_step(e[1], Ps, O[1]);      // dims 4m + 2, 3
}

template <int MSX, int MODE, int CL2 = 8>      // MSX = M, or 320 = M 32 in the d_m = 4 form
__global__ __launch_bounds__(kNW * 64, 2) void attn_stream_kernel(AttnParams p) {
    constexpr int MS = MSX == 320 ? 32 : MSX;
    constexpr bool D4 = MSX == 320;            // M = 32, d_m = 4 form (see "d_m = 4 form" above): G <= 4, replicated query heads
    typedef typename StreamTypes<MSX>::Unit Unit;
    typedef typename StreamTypes<MSX>::E EBuf;
    constexpr int kLog2M = MS == 64 ? 6 : MS == 32 ? 5 : 4;
    constexpr bool PV = MS == 64;              // parity-V value product (see "parity-V" above); M = 32 keeps the packed form
    constexpr bool D8 = MS == 16;              // d_m = 8 form (see "d_m = 8 form" above): G <= 4, query heads replicated over the column groups
    constexpr int NV = PV ? 8 : (D8 || D4) ? 4 : 2;      // value steps per unit (PV: token step s = i >> 1, subspace half n = i & 1; D8 / D4: tile i >> 1,
                                               // k-step i & 1)
    constexpr int SPV = 8 / NV;                // score stages that ride along with one value step
    constexpr int VD = PV ? 2 : 1;     // value steps the V gathers run ahead of their MFMA (a parity-V step is 4 gathers +
                                               // 1 MFMA, ~100 cycles of issue: one step ahead does not cover an LDS round trip)
    constexpr int NT = 8 >> (8 - CL2);         // 16-byte pieces of a codebook image per thread (C = 256: 64 KiB, C = 128: 32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup i of a grid lands on XCD i % 8 (tools/micro/xcc_map.hip).  With a multiple of 8 (b, kv head) pairs the
    // pairs are dealt over the linear id first, so ALL splits of a pair run on one XCD and the last arriver can read
    // the partials through its own L2 (common.h: ticket_and_merge checks the placement at run time, it is never assumed).
    int split = blockIdx.x, bh = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        bh = id % (int)gridDim.y;
        split = id / (int)gridDim.y;
    }
    const int b = bh / p.nh_k, hk = bh % p.nh_k;      // hk, bh: VIRTUAL when the launch splits the query heads of a kv head into parts
    // (AttnParams::nhk_real): the real kv head / pair index codes, page ids, window rows and the new rows
    constexpr bool PARTS = MSX == 16;      // only the d_m = 8 form runs as query-head parts: the other instances do not carry the code
    const int part = PARTS ? head_part(p, hk) : 0, hkr = hk - part * p.nhk_mul;
    const int bhr = PARTS ? bh - (b * p.hparts_m1 + part) * p.nhk_mul : bh;
    const int G = PARTS && p.nhk_mul ? min(p.G, p.G_all - part * p.G) : p.G;      // (the last part of an odd head group holds fewer)
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
    const int q4 = lane >> 4, c16 = lane & 15;

    // ---- where this wave reads: page pg0 + j * pg_step in round j, tokens [tin, tin + 32) of it ----
    const int ups = p.ps_shift - 5;                       // log2(units per page)
    const int wp = wave >> ups, uw = wave & ((1 << ups) - 1);
    const int pg0 = wp * p.nsplit + split;
    const int pg_step = p.nsplit << (3 - ups);
    const int tin = uw << 5;
    // page ids of rounds 0..63 (lane = round): the oldest loads of the wave
    int vpk = 0, vpv = 0;
    {
        int pgl = pg0 + lane * pg_step;
        pgl = pgl < p.n_pages_cap ? pgl : p.n_pages_cap - 1;
        const long long idx = (long long)bhr * p.n_pages_cap + pgl;
        if (k_paged) vpk = ids64 ? (int)p.k_ids64[idx] : p.k_ids32[idx];
        if (v_ident) vpv = (int)idx;
        else vpv = ids64 ? (int)p.v_ids64[idx] : p.v_ids32[idx];
#ifdef MILLION_DEBUG_CHECK_IDS
        {      // lane = round: entries of pages beyond the context (host bound) are preloaded but never used
            const bool live = pg0 + lane * pg_step < p.n_pages_cap && ((long long)(pg0 + lane * pg_step) << p.ps_shift) < p.T;
            if (k_paged) vpk = MILLION_CHECK_KID(p, ids64 ? (long long)p.k_ids64[idx] : (long long)vpk, live);
            if (!v_ident) vpv = MILLION_CHECK_VID(p, ids64 ? (long long)p.v_ids64[idx] : (long long)vpv, live);
        }
#endif
    }
    v8f16 qb[4];
    {
        const int hq = (D8 || D4) ? (c16 & 3) : c16;      // D8: column 4 dq + g holds head g (four copies of every head)
        const f16 *qv = p.q + ((long long)b * p.nh + head0(p, hk) + (hq < G ? hq : 0)) * 128 + 32 * q4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v4u t = *(const v4u *)(qv + 8 * s);
            if (hq >= G) t = v4u{0, 0, 0, 0};
            qb[s] = __builtin_bit_cast(v8f16, t);
        }
    }
    const bool append_wave = p.k_new && split == 0 && wave == kNW - 1 && part == 0;      // wave-uniform
    h2 new_k = {}, new_v = {};
    if (append_wave) {
        new_k = *(const h2 *)(p.k_new + (long long)bhr * 128 + 2 * lane);
        new_v = *(const h2 *)(p.v_new + (long long)bhr * 128 + 2 * lane);
    }
    // both codebooks go out before anything that depends on a length or a page id (the CU's load path takes ~28 cycles
    // per 1-KiB wave request, in order: what is requested first is there first).  (Through round 2 the V codebook was
    // requested during the prologue and had a barrier of its own: 18.6 -> 18.2 us at one request with it up here.)
    v4u tabk[NT], tabv[NT];
    const int rot = (blockIdx.x + 5 * blockIdx.y) & 7;
    {
        const v4u *ks = (const v4u *)p.k_tab;
#pragma unroll
        for (int i = 0; i < NT; ++i) tabk[i] = ks[((i + rot) & (NT - 1)) * (kNW * 64) + tid];
        const v4u *vs = (const v4u *)p.v_tab_col;      // the V codebook right behind it: one barrier serves both
#pragma unroll
        for (int i = 0; i < NT; ++i) tabv[i] = vs[((i + rot) & (NT - 1)) * (kNW * 64) + tid];
    }
    if (p.dev_lengths)      // issue + wait in ONE statement (see load_pids4); only the masks and the window depend on it
        asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dl) : "s"(p.dev_lengths), "s"((unsigned)b * 16u) : "memory");
    int T = dl[0], r_old = dl[1], rstart = dl[2];
    clamp_lengths(p, T, r_old, rstart);
    const int r = r_old + (p.k_new ? 1 : 0);
    const int t0 = (pg0 << p.ps_shift) + tin;             // first token of round 0
    const int t_step = pg_step << p.ps_shift;             // tokens between rounds
    const int n_mine = T > t0 ? (T - t0 + t_step - 1) / t_step : 0;      // rounds (= units) of this wave; host: <= 64
    const int j_last = n_mine > 0 ? n_mine - 1 : 0;
    const int T_ld = T > 0 ? T : 1;

    // ---- residual window rows of this split, dealt to the waves round-robin (see load_res_tile) ----
    const int rcnt = split < r ? (r - split + p.nsplit - 1) / p.nsplit : 0;
    const bool has_res = kResRows * wave < rcnt;
    const f16 *kr = p.k_res + b * p.res_sb + hkr * p.res_sh;
    const f16 *vr = p.v_res + b * p.res_sb + hkr * p.res_sh;
    ResTile rt;
    if (has_res) load_res_tile<MSX>(p, bhr, kr, vr, wave, rcnt, split, rstart, r_old, lane, rt);

    // ---- one unit's 16-byte requests into ring slot SL; J is wave-uniform; rounds past the wave's last unit
    //      re-request that unit (L2 hits, never consumed), so that no code load sits in a conditional ----
    Unit ring[kRing];
    typedef const __attribute__((address_space(1))) unsigned *gptr_u32;
    typedef typename std::conditional<D8 || D4, gptr_u32, gptr_v4u>::type VPtr;
    typedef typename std::conditional<MS == 64, gptr_v4u, typename std::conditional<MS == 32, gptr_v2u, gptr_u32>::type>::type KPtr;      // global address space: no FLAT loads
    const int krow0 = stream_token_of_row(0, c16);                                      // token of tile row c16 (tile 1: + 4)
    const unsigned k_lane_off = ((unsigned)krow0 << kLog2M) + (unsigned)(MS / 4) * q4;  // that token's code row, quarter q4
    const unsigned v_lane_off = (D8 || D4) ? ((unsigned)(lane & 15) << p.ps_shift) + 8u * (lane >> 4)     // subspace row n, tile rows 4 t ..: tokens 8 t + 4 g2 + 0..3
                                   : ((unsigned)(lane & 31) << p.ps_shift) + 16u * (lane >> 5);   // subspace row, 16-token half
#define UNIT_REQ_K(SL, J)                                                                                          \
    {                                                                                                              \
        const int jc_ = (J) < n_mine ? (J) : j_last;                                                               \
        gptr_u8 kb_;                                                                                               \
        if (k_paged) {                                                                                             \
            const long long pk_ = (long long)__builtin_amdgcn_readlane(vpk, jc_);                                  \
            kb_ = uniform_ptr(p.k_codes + (((pk_ << p.ps_shift) + tin) << kLog2M));                                \
            _Pragma("unroll") for (int g2 = 0; g2 < 2; ++g2)                                                       \
                ring[SL].k[g2] = *(KPtr)(kb_ + k_lane_off + ((4u * g2) << kLog2M));                                \
        } else {      /* row-major K: absolute row per lane, rows past T - 1 re-read it (masked later) */          \
            const int tu_ = t0 + jc_ * t_step;                                                                     \
            kb_ = uniform_ptr(p.k_codes + b * p.k_sb + hkr * p.k_sh);                                               \
            _Pragma("unroll") for (int g2 = 0; g2 < 2; ++g2)                                                       \
                ring[SL].k[g2] = *(KPtr)(kb_ + (((unsigned)min(tu_ + krow0 + 4 * g2, T_ld - 1) << kLog2M) +        \
                                                (unsigned)(MS / 4) * q4));                                         \
        }                                                                                                          \
    }
#define UNIT_REQ_V(SL, J)                                                                                          \
    {                                                                                                              \
        const int jc_ = (J) < n_mine ? (J) : j_last;                                                               \
        const long long pv_ = (long long)__builtin_amdgcn_readlane(vpv, jc_);                                      \
        const gptr_u8 vb_ = uniform_ptr(p.v_codes + (pv_ << (kLog2M + p.ps_shift)) + tin);                         \
        if constexpr (D8) {      /* the two tiles' rows of a lane are 8 consecutive token bytes: one 8-byte load */  \
            const v2u w_ = *(gptr_v2u)(vb_ + v_lane_off);                                                          \
            ring[SL].v[0] = w_[0];                                                                                 \
            ring[SL].v[1] = w_[1];                                                                                 \
        } else if constexpr (D4) {                                                                                 \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                     \
                const v2u w_ = *(gptr_v2u)(vb_ + v_lane_off + ((16u * j_) << p.ps_shift));                         \
                ring[SL].v[0][j_] = w_[0];                                                                         \
                ring[SL].v[1][j_] = w_[1];                                                                         \
            }                                                                                                      \
        } else {                                                                                                   \
            ring[SL].v[0] = *(VPtr)(vb_ + v_lane_off);                                                             \
            if (MS == 64) ring[SL].v[MS == 64 ? 1 : 0] = *(VPtr)(vb_ + v_lane_off + (32u << p.ps_shift));          \
        }                                                                                                          \
    }
#define UNIT_REQ(SL, J) { UNIT_REQ_K(SL, J) UNIT_REQ_V(SL, J) }
    UNIT_REQ(0, 0)
    UNIT_REQ(1, 1)
#if (MILLION_EXP & 1)
    if constexpr (!D8 && !D4)
    // A/B: TOUCH the lines of rounds 2 and 3 (one dword per 128-byte line, result never read): the real requests of those
    // rounds go out ~2-3 us later and should then find their lines on the way or in L2
    if (p.ps_shift == 6 && k_paged) {
        typedef const volatile __attribute__((address_space(1))) unsigned *gptr_vu;
        constexpr int NK = (32 << kLog2M) / 128, NVL = (MS * 64) / 128;
#pragma unroll
        for (int jj = 2; jj < 4; ++jj) {
            const int jc_ = jj < n_mine ? jj : j_last;
            const long long pk_ = (long long)__builtin_amdgcn_readlane(vpk, jc_), pv_ = (long long)__builtin_amdgcn_readlane(vpv, jc_);
            const gptr_u8 kb_ = uniform_ptr(p.k_codes + (((pk_ << 6) + tin) << kLog2M));
            const gptr_u8 vb_ = uniform_ptr(p.v_codes + (pv_ << (kLog2M + 6)));
            const bool is_k = lane < NK;
            const int li = is_k ? lane : (lane - NK < NVL ? lane - NK : 0);
            const gptr_u8 a_ = (is_k ? kb_ : vb_) + 128u * li;
            (void)*(gptr_vu)a_;      // volatile: the load is issued, nothing ever waits for its data
        }
    }
#endif
    // (Round 3, tools/ab_build.py: units 2 and 3 requested here too - all four ring slots up front - 19.5 us instead of 16.8 at
    // one request, 25.2 vs 24.0 at two; right behind the codebook barrier: 18.1 / 23.9.  The CU's request queue is in order:
    // what is asked for before the codebooks are in LDS delays the barrier every wave waits at.  Also without effect (+-0.15 us
    // at 1 and 2 requests and at 128K): the query rows through LDS (one request instead of 32 per workgroup), the V bytes of
    // units 0-1 requested behind the barrier, s_setprio 1 for waves 4-7 over the last one, two or three blocks.)
    STAMP(7);
    {
        v4u *ld = (v4u *)smem;
        v4u *ldv = (v4u *)(smem + kVBase);
#pragma unroll
        for (int i = 0; i < NT; ++i) ld[((i + rot) & (NT - 1)) * (kNW * 64) + tid] = tabk[i];
#pragma unroll
        for (int i = 0; i < NT; ++i) ldv[((i + rot) & (NT - 1)) * (kNW * 64) + tid] = tabv[i];
    }
    STAMP(8);
    __syncthreads();
    STAMP(1);

    float m_run = -INFINITY, l_run = 0.f;
    // parity-V: one 32 x 32 tile per subspace half; d_m = 8 form: two 16 x 16 row tiles
    typename std::conditional<D8 || D4, Acc8, v16f32[2][PV ? 1 : 2]>::type O;
    if constexpr (D8 || D4) {
        O.t[0] = v4f32{0.f, 0.f, 0.f, 0.f};
        O.t[1] = v4f32{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int kk = 0; kk < (PV ? 1 : 2); ++kk)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[n][kk][i] = 0.f;
    }
    if (append_wave) {
        int row_n = rstart + r_old;
        row_n = row_n >= p.rcap ? row_n - p.rcap : row_n;
        const long long o = b * p.res_sb + hkr * p.res_sh + (long long)row_n * 128 + 2 * lane;
        *(h2 *)(p.k_res_w + o) = new_k;
        *(h2 *)(p.v_res_w + o) = new_v;
    }
    unsigned sel_lo, sel_hi;      // parity-V: where a probability goes in this lane's A-operand registers
    par_selectors(lane, sel_lo, sel_hi);
    unsigned d8mx, d8my;          // d_m = 8 form: the half of the A-operand registers this lane's rows take their probability in
    d8_masks(lane, d8mx, d8my);
    if (has_res) {      // residual tile of this wave first: it needs neither codebook
        float scr[4];
        score_res_tile(rt, qb, p.scale_log2e, wave, rcnt, lane, scr);
        softmax_online<4, PV>(scr, m_run, l_run, O, G, lane);
        if constexpr (D8 || D4) value_res_tile_d4(rt, scr, d8mx, d8my, O);
        else if constexpr (PV) value_res_tile_par(rt, scr, sel_lo, sel_hi, O);
        else value_res_tile(rt, scr, O);
    }
    STAMP(2);
    const float inv_c = 1.0f / p.scale_log2e;
    SoftRef sr;
    sr.idle = ((D8 || D4) ? (c16 & 3) : c16) < G ? 0.f : -INFINITY;
    sr.set(m_run, l_run, inv_c);

    const unsigned kbase = (unsigned)q4 * (64u << CL2);      // quarter q4 of the K row image: its 16 (M = 64) / 8 (M = 32) subspaces
    const unsigned vconst0 = D4 ? ((unsigned)kVBase | ((unsigned)(lane & 15) << 3))      // V col image base | 8 n (entries of 8 bytes)
                           : D8 ? ((unsigned)kVBase | ((unsigned)(lane & 15) << 4))      // V col image base | 16 n (entries of 16 bytes)
                                : ((unsigned)kVBase | ((unsigned)(lane & 31) << (MS == 64 ? 2 : 3)));
    const unsigned vconst1 = (unsigned)kVBase | ((unsigned)((lane & 31) + 32) << 2);      // M = 64 only

    unsigned a[2][4], P[4];
    EBuf e[2];          // packed form (M = 32)
    unsigned e4[4][4];  // parity-V (M = 64): the gathers of value step i sit in e4[i & 3], two steps ahead of their MFMA
    ParA pa;
    v8f16 Acur;
    float sc[8];
    unsigned e5[2][8];  // d_m = 4 form: the four gathered entries of a value step (two column tiles x two tokens), one step ahead
#if MILLION_EXP & 32
    // development build "the launch without arithmetic" (tools/ab_build.py 32): every request, wait, barrier and the whole tail
    // stay; a unit's bytes are xor-ed into a sink instead of gathered, multiplied and soft-maxed.  What this build takes at a
    // shape is what that shape costs before the first instruction of the attention arithmetic (profiles/r04_launch_floor.txt).
    unsigned sink = 0;
#define SINK_V4(x) sink ^= (x)[0] ^ (x)[1] ^ (x)[2] ^ (x)[3]
#define KG(SL, ST)                                                                                                 \
    do {                                                                                                           \
        if ((ST) == 0) {                                                                                           \
            if constexpr (MS == 64) { SINK_V4(ring[SL].k[0]); SINK_V4(ring[SL].k[1]); }                            \
            else if constexpr (MS == 32) sink ^= ring[SL].k[0][0] ^ ring[SL].k[0][1] ^ ring[SL].k[1][0] ^ ring[SL].k[1][1]; \
            else sink ^= ring[SL].k[0] ^ ring[SL].k[1];                                                            \
        }                                                                                                          \
    } while (0)
#define KM(ST) (void)0
#define VG(SL, I)                                                                                                  \
    do {                                                                                                           \
        if ((I) == 0) {                                                                                            \
            if constexpr (D8) sink ^= ring[SL].v[0] ^ ring[SL].v[1];                                               \
            else if constexpr (D4) sink ^= ring[SL].v[0][0] ^ ring[SL].v[0][1] ^ ring[SL].v[1][0] ^ ring[SL].v[1][1]; \
            else {                                                                                                 \
                SINK_V4(ring[SL].v[0]);                                                                            \
                if constexpr (MS == 64) SINK_V4(ring[SL].v[MS == 64 ? 1 : 0]);                                     \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
#define VS(I) {}
#define VPREP() {}
#define SCORES_OUT(J) { _Pragma("unroll") for (int i = 0; i < 8; ++i) sc[i] = 0.f; (void)D; }
#define SOFTMAX_RAW() (void)0
#else
#define SOFTMAX_RAW() softmax_online_raw<8, PV>(sc, p.scale_log2e, inv_c, sr, O, G, lane)
#define KG(SL, ST) st_kgather<CL2>(ring[SL], ST, kbase, a[(ST) & 1])
    // (the first k-step of a tile takes a literal zero accumulator - an inline constant of the MFMA - instead of a zeroed D: 8 v_mov
    // per unit less)
#define KM(ST) D[(ST) >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                              \
        as_v8f16(a[(ST) & 1][0], a[(ST) & 1][1], a[(ST) & 1][2], a[(ST) & 1][3]), qb[(ST) & 3],                   \
        ((ST) & 3) == 0 ? v4f32{0.f, 0.f, 0.f, 0.f} : D[(ST) >> 2], 0, 0, 0)
    // value steps run token-step major: i -> st = 2n + s with s = i / (NV / 2), so that P serves both s = 0 steps, is
    // moved on in place (value_next_step), and then serves both s = 1 steps
#define VG(SL, I)                                                                                                  \
    do {                                                                                                           \
        if constexpr (D8) d8_vgather(ring[SL].v, (I), vconst0, e5[(I) & 1]);                                       \
        else if constexpr (D4) d4_vgather(ring[SL].v, (I), vconst0, vconst0 + 128u, e5[(I) & 1]);                  \
        else if constexpr (PV) v_gather_par(ring[SL].v, (I) >> 1, (I) & 1, vconst0, vconst1, e4[(I) & 3]);         \
        else st_vgather(ring[SL], (I), vconst0, vconst1, e[(I) & 1]);                                              \
    } while (0)
#define VS(I)                                                                                                      \
    {                                                                                                              \
        if constexpr (D8 || D4) {                                                                                 \
            d4_vstep(sc[2 * (I)], sc[2 * (I) + 1], e5[(I) & 1], d8mx, d8my, O);      /* sc: the unit's probabilities until the next SCORES_OUT */ \
        } else if constexpr (PV) {                                                                                 \
            if (((I) & 1) == 0) Acur = value_A_par(pa, (I) >> 1, sel_lo, sel_hi);                                  \
            O[(I) & 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                                \
                Acur, as_v8f16(e4[(I) & 3][0], e4[(I) & 3][1], e4[(I) & 3][2], e4[(I) & 3][3]), O[(I) & 1][0], 0, 0, 0); \
        } else {                                                                                                   \
            if ((I) == NV / 2) value_next_step(P);                                                                 \
            st_vstep(e[(I) & 1], P, (I), O);                                                                       \
        }                                                                                                          \
    }
#define VPREP()                                                                                                    \
    {                                                                                                              \
        if constexpr (D8 || D4) { }                                                                                \
        else if constexpr (PV) value_prep_par(sc, pa);                                                             \
        else value_prep(sc, P);                                                                                    \
    }
    // raw scores of round J out of the accumulators; only the unit that holds token T - 1 (wave-uniform) is masked; a
    // round whose first token is past T - 1 (only the prologue of a wave without whole rounds meets one) gives -inf
#define SCORES_OUT(J)                                                                                              \
    {                                                                                                              \
        const int t_u = t0 + (J) * t_step;                                                                         \
        if (t_u + 32 <= T) {                                                                                       \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) sc[i] = D[i >> 2][i & 3];                                \
        } else {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                          \
                sc[i] = t_u + 8 * q4 + 4 * (i >> 2) + (i & 3) < T ? D[i >> 2][i & 3] : -INFINITY;                  \
        }                                                                                                          \
    }
#endif
    // BLOCK: the value steps of the unit in slot U4 (round J) interleaved with the 8 score stages of the unit in slot
    // U4 + 1 (round J + 1); then the first gathers of the next block, the refill of slot U4 with round J + 4 and the
    // online softmax of round J + 1.
#define BLOCK(U4, J)                                                                                               \
    {                                                                                                              \
        v4f32 D[2];                                                 \
        UNIT_REQ_K(U4, (J) + 4)      /* the K bytes of slot U4 (round J) were consumed by the previous block */    \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                                           \
            VS(i)                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            if (i + VD < NV) VG(U4, i + VD); else VG(((U4) + 1) & 3, i + VD - NV);                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            _Pragma("unroll") for (int k = 0; k < SPV; ++k) {                                                      \
                KM(SPV * i + k);                                                                                   \
                if (SPV * i + k + 2 < 8) KG(((U4) + 1) & 3, SPV * i + k + 2);                                      \
                else KG(((U4) + 2) & 3, SPV * i + k + 2 - 8);                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }                                                                                                      \
        }                                                                                                          \
        UNIT_REQ_V(U4, (J) + 4)                                                                                    \
        SCORES_OUT((J) + 1)                                                                                        \
        SOFTMAX_RAW();                                       \
        VPREP()                                                                                                    \
    }
#define VALUE_ALONE(U4)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                                               \
        if (i + VD < NV) VG(U4, i + VD);                                                                           \
        VS(i)                                                                                                      \
    }
    // One unit on its own (the up to three units a wave has beyond its whole rounds of four): scores, softmax, values,
    // self-contained, so that the branch around it carries no pipeline state.
#define SINGLE(SL, J)                                                                                              \
    {                                                                                                              \
        v4f32 D[2];                                                 \
        KG(SL, 0);                                                                                                 \
        KG(SL, 1);                                                                                                 \
        _Pragma("unroll") for (int st = 0; st < 8; ++st) {                                                         \
            KM(st);                                                                                                \
            if (st + 2 < 8) KG(SL, st + 2);                                                                        \
        }                                                                                                          \
        SCORES_OUT(J)                                                                                              \
        SOFTMAX_RAW();                                       \
        VPREP()                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < VD; ++k) VG(SL, k);                                                  \
        VALUE_ALONE(SL)                                                                                            \
    }
    const int n_whole = n_mine >> 2, n_rem = n_mine & 3;      // whole rounds of four units + up to three more
    TailReq treq;
    treq.idx = 0; treq.gen = 0; treq.cen = 0; treq.base = 0; treq.done = false;
    {
        // prologue: the 8 score stages of round 0 (masked out when the wave has no whole round: its units are all
        // handled as single units below); round 2 is requested in between
        {
            v4f32 D[2];
            KG(0, 0);
            KG(0, 1);
            tail_mark_xcd(p, bh, split, wave, lane);      // this split's slot of the XCD census (see there for the placement)
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                KM(st);
                if (st + 2 < 8) KG(0, st + 2);
                if (st == 4) UNIT_REQ(2, 2)
                __builtin_amdgcn_sched_barrier(0);
            }
            SCORES_OUT(0)
            if (n_whole == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) sc[i] = -INFINITY;
            }
        }
        SOFTMAX_RAW();
        VPREP()
        STAMP(16);
        UNIT_REQ(3, 3)
        if (n_whole > 0) {
#pragma unroll
            for (int k = 0; k < VD; ++k) VG(0, k);
            KG(1, 0);
            KG(1, 1);
            // The first round's three blocks, then the loop ROTATED by three (round 4): the path of a wave with ONE whole round
            // (the headline shape at one request) joins the loop's exit with only ring slot 3 and the softmax state live.  With
            // the loop in front of these three blocks (rounds 2-3) the whole ring was live across it on that path, and the
            // parity-V build spilled 53 registers around the loop - scratch, which alone cost ~9 us per launch.
            BLOCK(0, 0)
            BLOCK(1, 1)
            BLOCK(2, 2)
            int j = 3;
            for (int w = 1; w < n_whole; ++w) {
                BLOCK(3, j)
                ++j;
                BLOCK(0, j)
                ++j;
                BLOCK(1, j)
                ++j;
                BLOCK(2, j)
                ++j;
                if (w == 1) STAMP(17);
            }
            tail_request(p, bh, p.nslots, wave, lane, treq);      // ~3 us ahead of the point where the tail needs the answers
            STAMP(19);
            VALUE_ALONE(3)
        }
        // the units beyond the whole rounds sit in ring slots 0..2 (requested by the last round's refills, or up front)
        if (n_rem > 0) SINGLE(0, 4 * n_whole)
        if (n_rem > 1) SINGLE(1, 4 * n_whole + 1)
        if (n_rem > 2) SINGLE(2, 4 * n_whole + 2)
    }
#undef SINGLE
#undef KG
#undef KM
#undef VG
#undef VS
#undef VPREP
#undef SCORES_OUT
#undef SOFTMAX_RAW
#undef BLOCK
#undef VALUE_ALONE
#undef UNIT_REQ
#undef UNIT_REQ_K
#undef UNIT_REQ_V
#if MILLION_EXP & 32
    if (sink == 0x9e3779b9u) sr.l += 1.f;      // never: keeps the sink (and the loads behind it) alive
#endif
    STAMP(3);
    merge_and_publish<MSX, PV>(p, smem, b, hk, split, G, tid, lane, wave, dbg_on, O, sr.m, sr.l, treq);
#undef STAMP
}

#include "attn_lean.h"      // attn_lean_kernel (round 5): the MFMA-lean core on this file's launch skeleton and tail

// Self-check of the row-swap reductions (tests/test_gpu_parity.py): one wave, in[64] -> max / sum over the
// four 16-lane rows per column.
__global__ void rows_reduce_check_kernel(const float *in, float *out_max, float *out_sum) {
    const float x = in[threadIdx.x];
    out_max[threadIdx.x] = rows_max(x);
    out_sum[threadIdx.x] = rows_sum(x);
}
int launch_rows_reduce_check(const float *in, float *out_max, float *out_sum, hipStream_t s) {
    hipLaunchKernelGGL(rows_reduce_check_kernel, dim3(1), dim3(64), 0, s, in, out_max, out_sum);
    return hipGetLastError() == hipSuccess ? MILLION_OK : MILLION_ERR_LAUNCH;
}

// A split takes every nsplit-th window row and has kNW * kResRows = 128 slots for them in its waves' residual tiles:
// windows of up to 128 rows work with any split count, longer ones (extended_residual_size 256, the reference's
// flash_decoding_paged_v_*_Lt256 names) get at least ceil(rcap / 128) splits (launch_attn_mfma).
// d = 64 with M = 32 / 16 (d_m = 2 / 4) and M = 64 (d_m = 1: run as d_m = 2 with every odd dim zero, attn_lean.h): the lean kernel
// only (round 5; before: the tile kernel) - 256 centroids, up to 4 query heads per kv head
static int g_mfma_policy = 0, g_tail_test = 0, g_lean_off = 0;      // A/B and test knobs: see set_mfma_policy below
// 5 .. 16 query heads per kv head: the launch runs ceil(G / 4) VIRTUAL kv heads of ceil(G / parts) heads per real one (AttnParams::nhk_real;
// the parts re-read the codes - from the XCD's L2 when they run together: the parts of a real head sit on one XCD).  The workspace
// head is laid out for max(2048, bs * nh_k) pairs (million_api.hip): the virtual pairs must fit it.
static int mfma_hparts(const AttnParams &p) {      // (the lean kernel's d = 64 forms and the streaming kernel's d = 128 / M = 16 form)
    if (!(p.d == 64 || (p.d == 128 && p.M == 16)) || p.nhk_mul || g_lean_off) return 1;      // (policy 16: no parts either - the tile kernel)
    const int P = p.G > 4 && p.G <= 16 ? (p.G + 3) / 4 : 1;      // 5 .. 8 heads: 2 parts, 9 .. 12: 3, 13 .. 16: 4
    return (P > 1 && (long long)p.bs * p.nh_k * P <= 2048) ? P : 1;
}
static AttnParams mfma_virtual(const AttnParams &p_in) {      // the call as the lean kernel sees it
    AttnParams p = p_in;
    const int P = mfma_hparts(p);
    if (P > 1) {
        p.nhk_real = p.nhk_mul = p.nh_k;
        p.hparts_m1 = P - 1;
        p.nh_k *= P;
        p.G_all = p.G;
        p.G = (p.G + P - 1) / P;      // 3 or 4 heads per part; the last part: G_all - (P - 1) G >= 1
        p.slot_floats = (p.G * p.d + 2 * p.G + 31) / 32 * 32;      // = slot_floats_for (million_api.hip)
    }
    return p;
}
static bool lean_d64_shape(const AttnParams &p) {
    return p.d == 64 && (p.M == 64 || p.M == 32 || p.M == 16) && (p.C == 256 || p.C == 128) && (p.G <= 4 || mfma_hparts(p) > 1) && p.rcap <= 4 * kNW * kResRows;
}
bool attn_mfma_shape_ok(const AttnParams &p) {
    if (lean_d64_shape(p)) return true;
    if (p.d == 128 && p.M == 16)      // d_m = 8 form of the streaming kernel (round 4): up to 4 query heads per kv head (6 .. 16: as parts)
        return (p.C == 256 || p.C == 128) && (p.G <= 4 || mfma_hparts(p) > 1) && p.rcap <= 4 * kNW * kResRows;
    return p.d == 128 && (p.M == 64 || p.M == 32) && (p.C == 256 || p.C == 128) && p.G <= kMaxGMfma && p.rcap <= 4 * kNW * kResRows;
}

bool attn_mfma_supported(const AttnParams &p) {
    return attn_mfma_shape_ok(p) && p.v_paged && (p.page_size == 32 || p.page_size == 64 || p.page_size == 128);
}

// host side of g_tail_faults: waits for the device, returns and clears the count (-1: the runtime refused)
int read_tail_faults() {
    unsigned n = 0, zero = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tail_faults), sizeof(n)) != hipSuccess) return -1;
    if (n && hipMemcpyToSymbol(HIP_SYMBOL(g_tail_faults), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}

// A/B knob (million_set_force_generic 2): 0 = auto (streaming kernel wherever it applies), 1 = grouped kernel only
// g_tail_test (million_set_force_generic 4 / 8): the merge helpers give up at once - the last arriver's take-over path, for
// tests: 1 = every give-up bit is set in the prologue, 2 = the helpers give up through the real path (no polls, then the atomic)
// g_lean_off (million_set_force_generic 16): the lean kernel's shapes stay on the streaming kernel (A/B, tests of the parity-V form)
// development A/B (dev_switches.h; environment MILLION_M32_PACKED=1 in a MILLION_DEV_BUILD): M = 32 keeps the packed form at G <= 4
// too.  The constant 0 in the product build.
static const int g_mfma_form = MILLION_DEV_M32_PACKED();
void set_mfma_policy(int policy) { g_mfma_policy = policy & 1; g_tail_test = (policy >> 1) & 3; g_lean_off = (policy >> 3) & 1; }

// split policy: about one workgroup per CU; a split is at least 512 tokens long
static int mfma_splits(const AttnParams &p) {
    const int cus = device_cus();
    const int bh = p.bs * p.nh_k;
    int ns = (cus + bh - 1) / bh;
    if (ns > kMaxSplits) ns = kMaxSplits;
    const int by_len = p.T > 0 ? (p.T + 511) / 512 : 1;
    if (ns > by_len) ns = by_len;
    if (ns < 1) ns = 1;
    // the kernels deal whole 32-token units to the splits (first units % ns splits carry one more): len = the longest
    const int units = p.T > 0 ? (p.T + 31) / 32 : 1;
    if (ns > units) ns = units;
    const int ns_window = (p.rcap + kNW * kResRows - 1) / (kNW * kResRows);      // splits the residual window needs
    if (ns < ns_window) ns = ns_window;      // (a split beyond the last unit just has no code units)
    // The streaming kernel preloads the page ids of a wave's first 64 rounds (one vector load, lane = round): a call with more
    // rounds per wave - many (b, kv head) pairs AND a long context, e.g. 16 requests x 8 kv heads at 40K tokens - gets more
    // splits instead of the grouped kernel (round 4; rounds 2-3 dropped such calls to the grouped kernel, C = 128 even to the
    // scalar one).  The grid then holds more workgroups than CUs; the tail's single merger waits only for workgroups that
    // have started, so any dispatch order is fine.  64 splits x 64 rounds x 256 tokens = 1M tokens per (b, kv head).
    if (p.T > 0 && (p.T + ns * 256 - 1) / (ns * 256) > 64) {
        while (ns < kMaxSplits && (p.T + ns * 256 - 1) / (ns * 256) > 64) ++ns;
        // more workgroups than CUs now: prefer a grid that is a whole number of chip-fulls (128 pairs: 4 splits = 2 x 256
        // workgroups of 40 rounds instead of 3 splits = 256 + 128 of 53), if that costs at most twice the splits
        for (int n2 = ns; n2 <= 2 * ns && n2 <= kMaxSplits; ++n2)
            if ((long long)bh * n2 % cus == 0) { ns = n2; break; }
    }
    return ns;
}
// streaming kernel: rounds per wave = ceil(T / (ns * 256 tokens)) must fit the 64 page ids a wave preloads
static bool mfma_stream_ok(const AttnParams &p, int ns) { return p.T > 0 && (p.T + ns * 256 - 1) / (ns * 256) <= 64; }

// C = 128 runs on the streaming kernel only: without it (T = 0, more than 1M tokens) the call goes back to the caller
// the lean kernel takes the call (launch_attn_mfma): pages of 64 / 128 tokens, streaming policy
static bool lean_takes(const AttnParams &p_in) {
    if (!attn_mfma_supported(p_in)) return false;
    const AttnParams p = mfma_virtual(p_in);
    return (p.C == 256 || p.C == 128) && p.G <= 4 && p.page_size >= 64 && !g_lean_off && g_mfma_policy == 0 &&
           mfma_stream_ok(p, mfma_splits(p)) && (p.d == 64 || (p.M == 64 || (p.M == 32 && !(g_mfma_form & 1))));
}
bool attn_mfma_handles(const AttnParams &p) {
    if (p.d == 64) return lean_takes(p);      // no other MFMA kernel of this file takes d = 64: the caller goes on to the tile kernel
    if (p.M == 16) {      // streaming kernel or not at all
        if (!attn_mfma_supported(p) || g_mfma_policy != 0) return false;
        const AttnParams pv = mfma_virtual(p);
        return mfma_stream_ok(pv, mfma_splits(pv));
    }
    return attn_mfma_supported(p) && (p.C != 128 || mfma_stream_ok(p, mfma_splits(p)));
}
// the call will run the STREAMING kernel (not the grouped fallback): million_attn_kernel_kind
bool attn_mfma_streams(const AttnParams &p) {
    if (p.d == 64) return lean_takes(p);
    if (!attn_mfma_supported(p) || g_mfma_policy != 0) return false;
    const AttnParams pv = mfma_virtual(p);
    return mfma_stream_ok(pv, mfma_splits(pv));
}

int launch_attn_mfma(const AttnParams &p_in, hipStream_t s) {
    AttnParams p = mfma_virtual(p_in);      // (the identity unless the shape runs as head parts)
    const int bh = p.bs * p.nh_k;
    const int ns = mfma_splits(p);
    const int units = p.T > 0 ? (p.T + 31) / 32 : 1;
    int len = 32 * ((units + ns - 1) / ns);
    p.nsplit = ns;
    p.nslots = ns;
    p.split_len = len;
    // mergers per (b, kv head): helpers only when every workgroup of the launch is resident at once (one per CU); a helper
    // that still cannot see its flags gives up and the last arriver takes over (merge_and_publish)
    {
        int nm = p.G < ns ? p.G : ns;
        if (nm > 8) nm = 8;
        p.nmerge = (long long)bh * ns <= device_cus() ? (nm > 0 ? nm : 1) : 1;
        p.tail_test = g_tail_test;
#if MILLION_EXP & 2
        p.nmerge = 1;      // A/B: the primary alone
#endif
    }
    if (device_once(1)) {
        (void)hipFuncSetAttribute((const void *)attn_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_mfma_kernel<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_mfma_kernel<false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<64, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<32, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<320, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<320, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<320, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<16, 2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<64, 2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_stream_kernel<32, 2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<1, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<2, 64, 128, 128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<0, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<1, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<2, 32, 128, 128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<0, 32, 64, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<1, 32, 64, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<2, 32, 64, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<0, 16, 64, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<1, 16, 64, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<2, 16, 64, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<0, 64, 128, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<1, 64, 128, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_lean_kernel<2, 64, 128, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    }
    const bool stream_ok = mfma_stream_ok(p, ns);
    const int mode = (p.k_paged && !p.v_identity && !p.ids64) ? 0 : (!p.k_paged && p.v_identity) ? 1 : 2;
    const dim3 grid(ns, bh), block(kNW * 64);
    if (p.d == 64) {       // lean kernel or nothing of this file (the caller's next choice: the tile kernel)
        if (!lean_takes(p_in)) return kAttnNotHandled;
        if (p.M == 64) {
            if (mode == 0) hipLaunchKernelGGL((attn_lean_kernel<0, 64, 128, 64, 3>), grid, block, kLdsBytes, s, p);
            else if (mode == 1) hipLaunchKernelGGL((attn_lean_kernel<1, 64, 128, 64, 3>), grid, block, kLdsBytes, s, p);
            else hipLaunchKernelGGL((attn_lean_kernel<2, 64, 128, 64, 3>), grid, block, kLdsBytes, s, p);
        } else if (p.M == 32) {
            if (mode == 0) hipLaunchKernelGGL((attn_lean_kernel<0, 32, 64, 64, 3>), grid, block, kLdsBytes, s, p);
            else if (mode == 1) hipLaunchKernelGGL((attn_lean_kernel<1, 32, 64, 64, 3>), grid, block, kLdsBytes, s, p);
            else hipLaunchKernelGGL((attn_lean_kernel<2, 32, 64, 64, 3>), grid, block, kLdsBytes, s, p);
        } else {
            if (mode == 0) hipLaunchKernelGGL((attn_lean_kernel<0, 16, 64, 64, 3>), grid, block, kLdsBytes, s, p);
            else if (mode == 1) hipLaunchKernelGGL((attn_lean_kernel<1, 16, 64, 64, 3>), grid, block, kLdsBytes, s, p);
            else hipLaunchKernelGGL((attn_lean_kernel<2, 16, 64, 64, 3>), grid, block, kLdsBytes, s, p);
        }
    } else
    if (p.M == 16) {       // d_m = 8 form: the streaming kernel or the tile kernel (the caller's next choice)
        if (!stream_ok || g_mfma_policy != 0) return kAttnNotHandled;
        if (p.C == 128) hipLaunchKernelGGL((attn_stream_kernel<16, 2, 7>), grid, block, kLdsBytes, s, p);
        else if (mode == 0) hipLaunchKernelGGL((attn_stream_kernel<16, 0>), grid, block, kLdsBytes, s, p);
        else if (mode == 1) hipLaunchKernelGGL((attn_stream_kernel<16, 1>), grid, block, kLdsBytes, s, p);
        else hipLaunchKernelGGL((attn_stream_kernel<16, 2>), grid, block, kLdsBytes, s, p);
    } else
    // lean kernel (round 5): 64-token units, lane = token; C = 256 and (its table copy spreads the K rows) C = 128
    if (g_mfma_policy == 0 && stream_ok && p.M == 64 && p.G <= 4 && p.page_size >= 64 && !g_lean_off) {
        if (p.C == 128) hipLaunchKernelGGL((attn_lean_kernel<2, 64, 128, 128, 1>), grid, block, kLdsBytes, s, p);
        else if (mode == 0) hipLaunchKernelGGL((attn_lean_kernel<0>), grid, block, kLdsBytes, s, p);
        else if (mode == 1) hipLaunchKernelGGL((attn_lean_kernel<1>), grid, block, kLdsBytes, s, p);
        else hipLaunchKernelGGL((attn_lean_kernel<2>), grid, block, kLdsBytes, s, p);
    } else if (g_mfma_policy == 0 && stream_ok && p.M == 32 && p.G <= 4 && p.page_size >= 64 && !g_lean_off && !(g_mfma_form & 1)) {      // d_m = 4
        if (p.C == 128) hipLaunchKernelGGL((attn_lean_kernel<2, 32, 128, 128, 1>), grid, block, kLdsBytes, s, p);
        else if (mode == 0) hipLaunchKernelGGL((attn_lean_kernel<0, 32>), grid, block, kLdsBytes, s, p);
        else if (mode == 1) hipLaunchKernelGGL((attn_lean_kernel<1, 32>), grid, block, kLdsBytes, s, p);
        else hipLaunchKernelGGL((attn_lean_kernel<2, 32>), grid, block, kLdsBytes, s, p);
    } else
    if (p.C == 128) {      // 128 centroids per subspace (reference setup.py:15): streaming kernel by run-time layout flags only
        if (!stream_ok) return kAttnNotHandled;      // T = 0 or more than 64 rounds per wave: the caller takes the generic kernel
        if (p.M == 64) hipLaunchKernelGGL((attn_stream_kernel<64, 2, 7>), grid, block, kLdsBytes, s, p);
        else hipLaunchKernelGGL((attn_stream_kernel<32, 2, 7>), grid, block, kLdsBytes, s, p);
    } else if (g_mfma_policy == 0 && stream_ok) {
        if (p.M == 64) {
            if (mode == 0) hipLaunchKernelGGL((attn_stream_kernel<64, 0>), grid, block, kLdsBytes, s, p);
            else if (mode == 1) hipLaunchKernelGGL((attn_stream_kernel<64, 1>), grid, block, kLdsBytes, s, p);
            else hipLaunchKernelGGL((attn_stream_kernel<64, 2>), grid, block, kLdsBytes, s, p);
        } else if (p.G <= 4 && !(g_mfma_form & 1)) {      // d_m = 4 form: query heads replicated over the column groups of the score tile
            if (mode == 0) hipLaunchKernelGGL((attn_stream_kernel<320, 0>), grid, block, kLdsBytes, s, p);
            else if (mode == 1) hipLaunchKernelGGL((attn_stream_kernel<320, 1>), grid, block, kLdsBytes, s, p);
            else hipLaunchKernelGGL((attn_stream_kernel<320, 2>), grid, block, kLdsBytes, s, p);
        } else {
            if (mode == 0) hipLaunchKernelGGL((attn_stream_kernel<32, 0>), grid, block, kLdsBytes, s, p);
            else if (mode == 1) hipLaunchKernelGGL((attn_stream_kernel<32, 1>), grid, block, kLdsBytes, s, p);
            else hipLaunchKernelGGL((attn_stream_kernel<32, 2>), grid, block, kLdsBytes, s, p);
        }
    } else if (p.M == 32) {
        if (p.T > 0) hipLaunchKernelGGL((attn_mfma_kernel<true, 32>), dim3(ns, bh), dim3(kNW * 64), kLdsBytes, s, p);
        else hipLaunchKernelGGL((attn_mfma_kernel<false, 32>), dim3(ns, bh), dim3(kNW * 64), kLdsBytes, s, p);
    } else if (p.T > 0)
        hipLaunchKernelGGL(attn_mfma_kernel<true>, dim3(ns, bh), dim3(kNW * 64), kLdsBytes, s, p);
    else
        hipLaunchKernelGGL(attn_mfma_kernel<false>, dim3(ns, bh), dim3(kNW * 64), kLdsBytes, s, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("attn_mfma launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million
