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
//   * the residual window (r <= 128 fp16 rows) is dealt round-robin to the splits and goes through the
//     same MFMA score path (A = the fp16 K rows themselves); its V rows are accumulated with fp32 FMAs.
//     Its rows are requested first thing in the kernel and consumed while the code ring is in flight.
#include "common.h"

namespace million {

typedef _Float16 v8f16 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int kNW = 8;                       // waves per workgroup
static_assert(true, "");
constexpr int kRing = 4;                     // 32-token units in flight per wave (16 VGPRs each)
constexpr int kTabBytes = 64 * 1024;         // one codebook image (M*C*dm*2)
constexpr int kVBase = kTabBytes;            // V col image behind the K row image
constexpr int kPartOff = 2 * kTabBytes;      // [128K,136K): final partial, flag, residual (m, l)
constexpr int kResWaves = 2;                 // residual groups (16 rows each) staged in LDS per workgroup
constexpr int kResOut = kPartOff + 8192;     // [136K,144K): per group O_res[G][128] fp32
constexpr int kResML = kPartOff + 6144;      // per group m[8], l[8]
constexpr int kLdsBytes = kResOut + kResWaves * 4096;  // 144 KiB

struct UnitCodes {
    v4u k[2];   // K code bytes: group g2 (16 tokens), lane (q, c): token c, bytes [16q, 16q+16)
    v4u v[2];   // V code bytes: half n (32 subspaces), lane (h, c): m = 32n + c, tokens [16h, 16h+16)
};

// LDS by absolute byte address: the dynamic LDS segment of this kernel starts at 0 (no static LDS; the
// kernel traps otherwise), so a lookup address needs no base add.
__device__ __forceinline__ unsigned lds32(unsigned addr) {
    return *(const __attribute__((address_space(3))) unsigned *)(size_t)addr;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the gfx950 row swaps:
// pure VALU.  (ds_bpermute-based __shfl_xor goes through the LDS queue, which the code-byte gathers of the
// other waves keep saturated: a dependent shuffle then costs a microsecond.)
// NOTE (hipcc, ROCm 7.2): `__builtin_bit_cast(float, v[1])` on an element of an ext_vector (here the pair a
// permlane swap builtin returns) is mis-lowered to element 0 — seen in the ISA: the row maximum became
// "row 0" and the row sum 4 x row 0.  The elements are therefore copied to scalars and converted with
// __uint_as_float.  tests: test_rows_reduce_selfcheck.
__device__ __forceinline__ unsigned opaque_copy(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ v2u swap16_self(unsigned x) {      // rows {0,0,2,2} of x / rows {1,1,3,3} of x
    return __builtin_amdgcn_permlane16_swap(x, opaque_copy(x), false, false);
}
__device__ __forceinline__ v2u swap32_self(unsigned x) {      // lower half twice / upper half twice
    return __builtin_amdgcn_permlane32_swap(x, opaque_copy(x), false, false);
}
__device__ __forceinline__ float rows_max(float x) {
    const v2u a = swap16_self(__float_as_uint(x));
    const unsigned a0 = a[0], a1 = a[1];
    const float m1 = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
    const v2u b = swap32_self(__float_as_uint(m1));
    const unsigned b0 = b[0], b1 = b[1];
    return fmaxf(__uint_as_float(b0), __uint_as_float(b1));
}
__device__ __forceinline__ float rows_sum(float x) {
    const v2u a = swap16_self(__float_as_uint(x));
    const unsigned a0 = a[0], a1 = a[1];
    const float s1 = __uint_as_float(a0) + __uint_as_float(a1);
    const v2u b = swap32_self(__float_as_uint(s1));
    const unsigned b0 = b[0], b1 = b[1];
    return __uint_as_float(b0) + __uint_as_float(b1);
}
__device__ __forceinline__ float lane_bcast(float x, int lane_const) {       // v_readlane -> SGPR operand
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane_const));
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
}

// Request the 4 x 16-byte loads of one 32-token unit (see UnitCodes).  t_unit: multiple of 32, < T.
__device__ __forceinline__ void load_unit_pid(const AttnParams &p, int b, int hk, const PidPair &pid, int t_unit, int T,
                                              int lane, UnitCodes &u) {
    const int q4 = lane >> 4, c16 = lane & 15, h2i = lane >> 5, c32 = lane & 31;
    const int page = t_unit >> p.ps_shift;                           // a unit never straddles a page
    const int page0 = page << p.ps_shift;
    const int inpage = t_unit - page0;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        int tok = t_unit + 16 * g2 + c16;
        tok = tok < T ? tok : T - 1;                 // stay inside the store; masked later
        const uint8_t *src = p.k_paged
            ? p.k_codes + (((pid.k << p.ps_shift) + (tok - page0)) << 6) + 16 * q4
            : p.k_codes + b * p.k_sb + hk * p.k_sh + ((long long)tok << 6) + 16 * q4;
        u.k[g2] = *(const v4u *)src;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint8_t *src = p.v_codes + (((pid.v << 6) + 32 * n + c32) << p.ps_shift) + inpage + 16 * h2i;
        u.v[n] = *(const v4u *)src;
    }
}
__device__ __forceinline__ void load_unit(const AttnParams &p, int b, int hk, int bh, int t_unit, int T,
                                          int lane, UnitCodes &u) {
    load_unit_pid(p, b, hk, load_pids(p, bh, t_unit >> p.ps_shift), t_unit, T, lane, u);
}

// Scores of one 32-token unit (MFMA 16x16x32): sc[g2*4 + rho] = scaled score (exp2 domain) of token
// 16*g2 + 4*q' + rho for the head of this lane's column (lane & 15); -inf beyond the split's last token.
template <bool MASK>
__device__ __forceinline__ void score_unit(const v4u (&kc)[2], const v8f16 (&qb)[4], int t_unit, int t_end,
                                           float scale_log2e, int lane, unsigned kbase, float (&sc)[8]) {
    const int q4 = lane >> 4;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        v4f32 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned w = kc[g2][s];
            const unsigned a0 = lds32(kbase + (4 * s + 0) * 1024 + ((w & 0xffu) << 2));
            const unsigned a1 = lds32(kbase + (4 * s + 1) * 1024 + (((w >> 8) & 0xffu) << 2));
            const unsigned a2 = lds32(kbase + (4 * s + 2) * 1024 + (((w >> 16) & 0xffu) << 2));
            const unsigned a3 = lds32(kbase + (4 * s + 3) * 1024 + ((w >> 24) << 2));
            D = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_v8f16(a0, a1, a2, a3), qb[s], D, 0, 0, 0);
        }
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
            const float v = D[rho] * scale_log2e;
            if (MASK) {
                const int tok = t_unit + 16 * g2 + 4 * q4 + rho;
                sc[g2 * 4 + rho] = tok < t_end ? v : -INFINITY;
            } else {
                sc[g2 * 4 + rho] = v;
            }
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
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const unsigned vconst = n ? vconst1 : vconst0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned w0 = vc[n][2 * s], w1 = vc[n][2 * s + 1];
            const unsigned e0 = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020400u));
            const unsigned e1 = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020500u));
            const unsigned e2 = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020600u));
            const unsigned e3 = lds32(__builtin_amdgcn_perm(w0, vconst, 0x03020700u));
            const unsigned e4 = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020400u));
            const unsigned e5 = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020500u));
            const unsigned e6 = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020600u));
            const unsigned e7 = lds32(__builtin_amdgcn_perm(w1, vconst, 0x03020700u));
            const v8f16 B0 = as_v8f16(__builtin_amdgcn_perm(e1, e0, 0x05040100u), __builtin_amdgcn_perm(e3, e2, 0x05040100u),
                                      __builtin_amdgcn_perm(e5, e4, 0x05040100u), __builtin_amdgcn_perm(e7, e6, 0x05040100u));
            const v8f16 B1 = as_v8f16(__builtin_amdgcn_perm(e1, e0, 0x07060302u), __builtin_amdgcn_perm(e3, e2, 0x07060302u),
                                      __builtin_amdgcn_perm(e5, e4, 0x07060302u), __builtin_amdgcn_perm(e7, e6, 0x07060302u));
            const v8f16 A = as_v8f16(P[s][0], P[s][1], P[s][2], P[s][3]);
            O[n][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B0, O[n][0], 0, 0, 0);
            O[n][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B1, O[n][1], 0, 0, 0);
        }
    }
}

// One group of <= 16 residual-window rows as an independent softmax partial (m, l, O_res): kraw = this
// lane's K row slice (A operand of the score MFMA, lane (q4, row c16): dims 32*q4 + 8*s ..), vrow[i] = dims
// (2*lane, 2*lane+1) of the i-th row's V.  nvalid = rows of the group that exist.  On return lane g < G
// holds (m, l) of head g, ores[g] = sum_i p[g][i] * V[i][2*lane .. 2*lane+1].
__device__ __forceinline__ void resid_group(const v4u (&kraw)[4], const h2 (&vrow)[16], int nvalid,
                                            const v8f16 (&qb)[4], float scale_log2e, int G, int lane,
                                            float &m_out, float &l_out, float (&ores)[kMaxG][2]) {
    const int q4 = lane >> 4;
    v4f32 D = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
        D = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8f16, kraw[s]), qb[s], D, 0, 0, 0);
    float sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
        sc[rho] = (4 * q4 + rho) < nvalid ? D[rho] * scale_log2e : -INFINITY;
        mx = fmaxf(mx, sc[rho]);
    }
    mx = rows_max(mx);
    const float m_safe = mx > -INFINITY ? mx : 0.f;
    float ls = 0.f;
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
        sc[rho] = fast_exp2(sc[rho] - m_safe);
        ls += sc[rho];
    }
    ls = rows_sum(ls);
    m_out = mx;
    l_out = ls;
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) ores[g][0] = ores[g][1] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i >= nvalid) break;                 // wave-uniform: typically <= 4 rows per split
        const float v0 = (float)vrow[i][0], v1 = (float)vrow[i][1];
#pragma unroll
        for (int g = 0; g < kMaxG; ++g)
            if (g < G) {
                // p of (row i, head g) sits in lane 16*(i/4) + g, register i%4: v_readlane -> SGPR operand
                const float pg = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                    __builtin_bit_cast(int, sc[i & 3]), (i >> 2) * 16 + g));
                ores[g][0] = fmaf(pg, v0, ores[g][0]);
                ores[g][1] = fmaf(pg, v1, ores[g][1]);
            }
    }
}

template <bool HAS_CODES>
__global__ __launch_bounds__(kNW * 64, 2) void attn_mfma_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int bh = blockIdx.y;
    const int b = bh / p.nh_k, hk = bh % p.nh_k;
    const int G = p.G;
    // ---- lengths: device-resident (graph replay) -> one scalar load, issued now and waited for after the
    //      codebook and q loads (which need no length) have been issued ----
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i dl = {p.T, p.r, p.rstart, 0};
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    MILLION_STAMP(p, 0);
    const int q4 = lane >> 4, c16 = lane & 15;

    // B operand of the score MFMA: the query heads (cols), K = 32 dims per step.  Oldest loads of the
    // wave: the residual partial below needs them before the codebooks are here.
    v8f16 qb[4];
    {
        const f16 *qv = p.q + ((long long)b * p.nh + hk * G + (c16 < G ? c16 : 0)) * 128 + 32 * q4;
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
    //      do not walk the L2 channels in lockstep.  The V codebook is requested AFTER the code ring: the
    //      score pass needs only K, so V may land behind the first codes. ----
    v4u tabk[8];
    const int rot = (blockIdx.x + 5 * blockIdx.y) & 7;
    {
        const v4u *ks = (const v4u *)p.k_tab;
#pragma unroll
        for (int i = 0; i < 8; ++i) tabk[i] = ks[((i + rot) & 7) * (kNW * 64) + tid];
    }
    if (p.dev_lengths)      // issue + wait in ONE statement (see load_pids4), after q and the tables have been requested
        asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dl) : "s"(p.dev_lengths), "s"((unsigned)b * 16u) : "memory");
    int T = dl[0] < p.T ? dl[0] : p.T;     // the host value is the bound the grid was sized for
    const int r_old = dl[1], rstart = dl[2];
    const int r = r_old + (p.k_new ? 1 : 0);       // fused append: the new token is window row r_old
    if (T < 1) T = 0;
    const int t_begin = min(split * p.split_len, T);
    const int t_end = min(t_begin + p.split_len, T);
    const int n_units = (t_end - t_begin + 31) >> 5;
    const int n_mine = n_units > wave ? (n_units - wave + kNW - 1) / kNW : 0;   // units of this wave
    const int n_pass = (n_mine + kRing - 1) / kRing;
    // unit j of this wave starts at token t_begin + 32*(wave + j*kNW); slots past the last unit re-request
    // the unit that holds token T-1 (HAS_CODES guarantees the host bound T >= 1; with device-resident
    // lengths a runtime T of 0 reads page 0, which must be a valid page: million_hip.h)
    const int T_ld = T > 0 ? T : 1;
    const int t_last = (T_ld - 1) & ~31;
#define UNIT_T(j) ((j) < n_mine ? t_begin + 32 * (wave + (j) * kNW) : t_last)

    // ---- residual window rows of this split: j = split, split + nsplit, ... < r; 16 rows per group.
    //      Groups 0..kResWaves-1 (all of them unless a split holds > 32 window rows) are requested by waves
    //      0..kResWaves-1 before everything else (OLDER loads never make a later counted wait over-wait),
    //      parked in LDS next to the tables, and turned into their own softmax partial right after the
    //      barrier, in the shadow of the code loads.  Later groups take the slow path at the end. ----
    const int rcnt = split < r ? (r - split + p.nsplit - 1) / p.nsplit : 0;
    const int rgroups = (rcnt + 15) >> 4;
    const f16 *kr = p.k_res + b * p.res_sb + hk * p.res_sh;
    const f16 *vr = p.v_res + b * p.res_sb + hk * p.res_sh;
    const bool res_wave = wave < kResWaves && wave < rgroups;      // wave-uniform

    // ---- the first code unit of this wave, then the K codebook goes to LDS.  Only these 16 loads (q, K
    //      codebook, unit 0) are issued before the first barrier: the CU's load path takes a few microseconds
    //      to accept everything this workgroup requests (64 B/clk), and a wave cannot write its share of the
    //      codebook while it is still stuck issuing loads. ----
    UnitCodes ring[kRing];
    PidPair pid4[kRing];
    if (HAS_CODES) {
        int pg[kRing];
#pragma unroll
        for (int k = 0; k < kRing; ++k) pg[k] = UNIT_T(k) >> p.ps_shift;
        load_pids4(p, bh, pg, pid4);        // one scalar round trip, after q and the K codebook have been requested
        load_unit_pid(p, b, hk, pid4[0], UNIT_T(0), T_ld, lane, ring[0]);
    }
    MILLION_STAMP(p, 7);
    {
        v4u *ld = (v4u *)smem;
#pragma unroll
        for (int i = 0; i < 8; ++i) ld[((i + rot) & 7) * (kNW * 64) + tid] = tabk[i];
    }
    MILLION_STAMP(p, 8);
    __syncthreads();     // no LDS-DMA in flight: lgkmcnt(0) + s_barrier, unit 0 stays in flight
    MILLION_STAMP(p, 1);

    // ---- everything else is requested now and lands while the first scores are computed: the residual rows
    //      of this wave's group, code units 1..3, the V codebook ----
    v4u rk[4];
    h2 rv[16];
    if (res_wave) {
        const int i_lane = wave * 16 + c16;
        const int j_lane = split + (i_lane < rcnt ? i_lane : 0) * p.nsplit;
        int row_l = rstart + j_lane;
        row_l = row_l >= p.rcap ? row_l - p.rcap : row_l;      // rstart, j < rcap: one wrap at most
        const f16 *kp = (p.k_new && j_lane == r_old ? p.k_new + (long long)bh * 128 : kr + (long long)row_l * 128) + 32 * q4;
#pragma unroll
        for (int s = 0; s < 4; ++s) rk[s] = *(const v4u *)(kp + 8 * s);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ii = wave * 16 + i;
            const int j = split + (ii < rcnt ? ii : 0) * p.nsplit;        // wave-uniform
            int row_i = rstart + j;
            row_i = row_i >= p.rcap ? row_i - p.rcap : row_i;
            rv[i] = *(const h2 *)((p.v_new && j == r_old ? p.v_new + (long long)bh * 128 : vr + (long long)row_i * 128) + 2 * lane);
        }
    }
    if (HAS_CODES) {
#pragma unroll
        for (int k = 1; k < kRing; ++k) load_unit_pid(p, b, hk, pid4[k], UNIT_T(k), T_ld, lane, ring[k]);
    }
    v4u tabv[8];
    {
        const v4u *vs = (const v4u *)p.v_tab_col;
#pragma unroll
        for (int i = 0; i < 8; ++i) tabv[i] = vs[((i + rot) & 7) * (kNW * 64) + tid];
    }
    MILLION_STAMP(p, 9);
    float m_run = -INFINITY, l_run = 0.f;
    v16f32 O[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[n][kk][i] = 0.f;
    MILLION_STAMP(p, 2);

    // this wave's residual group -> its own softmax partial (m, l, O_res) in LDS; runs between the score pass
    // and the V-codebook barrier of the first group, when its rows have long arrived
#define RESID_PARTIAL()                                                                                            \
    if (res_wave) {                                                                                                \
        /* opaque use: keeps hipcc from hoisting the fp16->fp32 conversions (and with them the wait for */        \
        /* these rows) above the loads issued after them */                                                       \
        asm volatile("" : "+v"(rk[0]), "+v"(rk[1]), "+v"(rk[2]), "+v"(rk[3]), "+v"(rv[0]), "+v"(rv[1]),          \
                          "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4]), "+v"(rv[5]), "+v"(rv[6]), "+v"(rv[7]),          \
                          "+v"(rv[8]), "+v"(rv[9]), "+v"(rv[10]), "+v"(rv[11]), "+v"(rv[12]), "+v"(rv[13]),      \
                          "+v"(rv[14]), "+v"(rv[15]));                                                             \
        const int nv = rcnt - wave * 16;                                                                           \
        float mr, lr, ores[kMaxG][2];                                                                              \
        resid_group(rk, rv, nv < 16 ? nv : 16, qb, p.scale_log2e, G, lane, mr, lr, ores);                          \
        float *ro = (float *)(smem + kResOut + wave * 4096);                                                       \
        _Pragma("unroll") for (int g = 0; g < kMaxG; ++g)                                                          \
            if (g < G) *(float2 *)(ro + g * 128 + 2 * lane) = float2{ores[g][0], ores[g][1]};                      \
        if (lane < G) {                                                                                            \
            float *ml = (float *)(smem + kResML) + wave * 16;                                                      \
            ml[lane] = mr;                                                                                         \
            ml[8 + lane] = lr;                                                                                     \
        }                                                                                                          \
    }

    const unsigned kbase = (unsigned)q4 * 16u * 1024u;                 // K row image: m = 16*q4 + ...
    const unsigned vconst0 = (unsigned)kVBase | ((unsigned)(lane & 31) << 2);        // m = c
    const unsigned vconst1 = (unsigned)kVBase | ((unsigned)((lane & 31) + 32) << 2);  // m = 32 + c

    // ---- groups of kRing units: SCORE pass for the whole group (K codebook only), one softmax update per
    //      group, then the VALUE pass.  The V codebook goes to LDS between the two passes of the first group,
    //      so the first scores are computed while it is still arriving. ----
#define GROUP(PASS, MASKV, FIRST, REFILL)                                                                          \
    {                                                                                                              \
        float sc[kRing][8];                                                                                        \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k) {                                                        \
            const int j = (PASS) * kRing + k;                                                                      \
            if (HAS_CODES && j < n_mine)                                                                           \
                score_unit<MASKV>(ring[k].k, qb, t_begin + 32 * (wave + j * kNW), t_end, p.scale_log2e, lane,      \
                                  kbase, sc[k]);                                                                   \
            else                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) sc[k][i] = -INFINITY;                                \
        }                                                                                                          \
        float mx = sc[0][0];                                                                                       \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k)                                                          \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) mx = fmaxf(mx, sc[k][i]);                                \
        mx = rows_max(mx);                                                                                         \
        const float m_new = fmaxf(m_run, mx);                                                                      \
        const float m_safe = m_new > -INFINITY ? m_new : 0.f;                                                      \
        const float alpha = fast_exp2(m_run - m_safe);                                                             \
        if (!(FIRST) && __any(m_new > m_run && m_run > -INFINITY)) {                                               \
            /* alpha of head g sits in lane g; O rows: lanes < 32 hold heads rho, lanes >= 32 heads 4 + rho */     \
            _Pragma("unroll") for (int rho = 0; rho < 4; ++rho) {                                                  \
                const float flo = rho < G ? lane_bcast(alpha, rho) : 1.0f;                                         \
                const float fhi = 4 + rho < G ? lane_bcast(alpha, 4 + rho) : 1.0f;                                 \
                const float f = lane < 32 ? flo : fhi;                                                             \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                      \
                    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) O[n][kk][rho] *= f;                           \
            }                                                                                                      \
        }                                                                                                          \
        float ls = 0.f;                                                                                            \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k)                                                          \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
                sc[k][i] = fast_exp2(sc[k][i] - m_safe);                                                           \
                ls += sc[k][i];                                                                                    \
            }                                                                                                      \
        l_run = l_run * alpha + ls;                                                                                \
        m_run = m_new;                                                                                             \
        if (FIRST) {                                                                                               \
            RESID_PARTIAL()                                                                                        \
            if (append_wave) {                                                                                     \
                int row_n = rstart + r_old;                                                                        \
                row_n = row_n >= p.rcap ? row_n - p.rcap : row_n;                                                  \
                const long long o = b * p.res_sb + hk * p.res_sh + (long long)row_n * 128 + 2 * lane;              \
                *(h2 *)(p.k_res_w + o) = new_k;                                                                    \
                *(h2 *)(p.v_res_w + o) = new_v;                                                                    \
            }                                                                                                      \
            v4u *ld = (v4u *)(smem + kVBase);                                                                      \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) ld[((i + rot) & 7) * (kNW * 64) + tid] = tabv[i];        \
            __syncthreads();                                                                                       \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < kRing; ++k) {                                                        \
            const int j = (PASS) * kRing + k;                                                                      \
            if (HAS_CODES && j < n_mine) value_unit(ring[k].v, sc[k], vconst0, vconst1, O);                        \
            if (HAS_CODES && (REFILL)) load_unit(p, b, hk, bh, UNIT_T(j + kRing), T_ld, lane, ring[k]);            \
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
#undef RESID_PARTIAL
#undef UNIT_T
    // residual groups beyond the LDS-staged ones (only when a split holds more than 32 window rows):
    // slow path, loads inside; each becomes a partial merged online into (m_late, l_late, olate)
    float m_late = -INFINITY, l_late = 0.f, olate[kMaxG][2];
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) olate[g][0] = olate[g][1] = 0.f;
    for (int gi = kResWaves + wave; gi < rgroups; gi += kNW) {
        const int i_lane = gi * 16 + c16;
        const int j_lane = split + (i_lane < rcnt ? i_lane : 0) * p.nsplit;
        int row_l = rstart + j_lane;
        row_l = row_l >= p.rcap ? row_l - p.rcap : row_l;      // rstart, j < rcap: one wrap at most
        const f16 *kp = (p.k_new && j_lane == r_old ? p.k_new + (long long)bh * 128 : kr + (long long)row_l * 128) + 32 * q4;
#pragma unroll
        for (int s = 0; s < 4; ++s) rk[s] = *(const v4u *)(kp + 8 * s);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ii = gi * 16 + i;
            const int j = split + (ii < rcnt ? ii : 0) * p.nsplit;
            int row_i = rstart + j;
            row_i = row_i >= p.rcap ? row_i - p.rcap : row_i;
            rv[i] = *(const h2 *)((p.v_new && j == r_old ? p.v_new + (long long)bh * 128 : vr + (long long)row_i * 128) + 2 * lane);
        }
        const int nv = rcnt - gi * 16;
        float mr, lr, ores[kMaxG][2];
        resid_group(rk, rv, nv < 16 ? nv : 16, qb, p.scale_log2e, G, lane, mr, lr, ores);
        // (m, l) of head g live in lane g; the O_res rows live per lane -> broadcast the scale factors
#pragma unroll
        for (int g = 0; g < kMaxG; ++g)
            if (g < G) {
                const float mg = __shfl(mr, g, 64), ml = __shfl(m_late, g, 64);
                const float mn = fmaxf(mg, ml), ms = mn > -INFINITY ? mn : 0.f;
                const float fa = fast_exp2(ml - ms), fb = fast_exp2(mg - ms);
                olate[g][0] = olate[g][0] * fa + ores[g][0] * fb;
                olate[g][1] = olate[g][1] * fa + ores[g][1] * fb;
            }
        {
            const float mn = fmaxf(mr, m_late), ms = mn > -INFINITY ? mn : 0.f;
            l_late = l_late * fast_exp2(m_late - ms) + lr * fast_exp2(mr - ms);
            m_late = mn;
        }
    }
    MILLION_STAMP(p, 3);
    // ---- merge the waves of this workgroup through LDS (tables are dead after the barrier) ----
    l_run = rows_sum(l_run);
    __syncthreads();
    MILLION_STAMP(p, 4);
    const int wstride = G * 128 + 2 * kMaxG;              // floats per wave
    float *scr = (float *)smem;
    float *mine = scr + wave * wstride;
    {
        const bool hi = lane >= 32;
        const int c32 = lane & 31;
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
            const int g = hi ? 4 + rho : rho;
            if (g < G) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) mine[g * 128 + 2 * (32 * n + c32) + kk] = O[n][kk][rho];
            }
        }
        if (lane < G) {                                  // lane g: row q' = 0, col g
            mine[G * 128 + lane] = m_run;
            mine[G * 128 + kMaxG + lane] = l_run;
        }
    }
    // late residual partial of this wave: fold into the wave's entry (same wave: LDS ops are ordered)
    if (kResWaves + wave < rgroups) {
#pragma unroll
        for (int g = 0; g < kMaxG; ++g)
            if (g < G) {
                const float mw_ = mine[G * 128 + g], ml = __shfl(m_late, g, 64), ll = __shfl(l_late, g, 64);
                const float mn = fmaxf(mw_, ml), ms = mn > -INFINITY ? mn : 0.f;
                const float fa = fast_exp2(mw_ - ms), fb = fast_exp2(ml - ms);
                mine[g * 128 + 2 * lane] = mine[g * 128 + 2 * lane] * fa + olate[g][0] * fb;
                mine[g * 128 + 2 * lane + 1] = mine[g * 128 + 2 * lane + 1] * fa + olate[g][1] * fb;
                if (lane == 0) {
                    mine[G * 128 + kMaxG + g] = mine[G * 128 + kMaxG + g] * fa + ll * fb;
                    mine[G * 128 + g] = mn;
                }
            }
    }
    __syncthreads();
    float *part = (float *)(smem + kPartOff);
    int *flag = (int *)(part + G * 128 + 2 * G + 4);
    for (int e = tid; e < G * 128; e += kNW * 64) {
        const int g = e >> 7;
        float mw[kNW], vw[kNW], lw[kNW];
#pragma unroll
        for (int w = 0; w < kNW; ++w) {
            mw[w] = scr[w * wstride + G * 128 + g];
            vw[w] = scr[w * wstride + e];
            lw[w] = scr[w * wstride + G * 128 + kMaxG + g];
        }
        // LDS-staged residual partials (groups 0..kResWaves-1)
        float mr[kResWaves], vr_[kResWaves], lr[kResWaves];
#pragma unroll
        for (int w = 0; w < kResWaves; ++w) {
            const bool on = w < rgroups;
            const float *ml = (const float *)(smem + kResML) + w * 16;
            mr[w] = on ? ml[g] : -INFINITY;
            lr[w] = on ? ml[8 + g] : 0.f;
            vr_[w] = on ? ((const float *)(smem + kResOut + w * 4096))[e] : 0.f;
        }
        float Mx = mw[0];
#pragma unroll
        for (int w = 1; w < kNW; ++w) Mx = fmaxf(Mx, mw[w]);
#pragma unroll
        for (int w = 0; w < kResWaves; ++w) Mx = fmaxf(Mx, mr[w]);
        const float Ms = Mx > -INFINITY ? Mx : 0.f;
        float acc = 0.f, lsum = 0.f;
#pragma unroll
        for (int w = 0; w < kNW; ++w) {
            const float f = fast_exp2(mw[w] - Ms);      // -inf -> 0
            acc = fmaf(f, vw[w], acc);
            lsum = fmaf(f, lw[w], lsum);
        }
#pragma unroll
        for (int w = 0; w < kResWaves; ++w) {
            const float f = fast_exp2(mr[w] - Ms);
            acc = fmaf(f, vr_[w], acc);
            lsum = fmaf(f, lr[w], lsum);
        }
        part[e] = acc;
        if ((e & 127) == 0) {
            part[G * 128 + g] = Mx;
            part[G * 128 + G + g] = lsum;
        }
    }
    __syncthreads();
    MILLION_STAMP(p, 5);
    publish_and_merge(p, b, hk, split, part, scr, flag);
    MILLION_STAMP(p, 6);   // wave scratch is dead after the barrier above
}

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

bool attn_mfma_shape_ok(const AttnParams &p) { return p.d == 128 && p.M == 64 && p.C == 256 && p.G <= kMaxG; }

bool attn_mfma_supported(const AttnParams &p) {
    return attn_mfma_shape_ok(p) && p.v_paged && (p.page_size == 32 || p.page_size == 64 || p.page_size == 128);
}

int launch_attn_mfma(const AttnParams &p_in, hipStream_t s) {
    AttnParams p = p_in;
    // split policy: about one workgroup per CU; a split is a multiple of 64 tokens, at least 512 long
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int bh = p.bs * p.nh_k;
    int ns = (cus + bh - 1) / bh;
    if (ns > kMaxSplits) ns = kMaxSplits;
    const int by_len = p.T > 0 ? (p.T + 511) / 512 : 1;
    if (ns > by_len) ns = by_len;
    if (ns < 1) ns = 1;
    int len = p.T > 0 ? (p.T + ns - 1) / ns : 64;
    len = (len + 63) / 64 * 64;
    ns = p.T > 0 ? (p.T + len - 1) / len : 1;
    p.nsplit = ns;
    p.nslots = ns;
    p.split_len = len;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)attn_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        (void)hipFuncSetAttribute((const void *)attn_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        attr_set = true;
    }
    if (p.T > 0)
        hipLaunchKernelGGL(attn_mfma_kernel<true>, dim3(ns, bh), dim3(kNW * 64), kLdsBytes, s, p);
    else
        hipLaunchKernelGGL(attn_mfma_kernel<false>, dim3(ns, bh), dim3(kNW * 64), kLdsBytes, s, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("attn_mfma launch: %s", hipGetErrorString(e)); return MILLION_ERR_LAUNCH; }
    return MILLION_OK;
}

}  // namespace million
