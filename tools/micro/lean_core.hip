// Microbenchmark (round 5): the "MFMA-lean" decode core on the harness of core_micro.hip.
//
// What is different from the shipped core (attn_mfma.hip: scores by v_mfma_f32_16x16x32_f16 with 4 of 16 head columns in
// use, lane = (token row, quarter of the code row); values by v_mfma_f32_32x32x16_f16 with 8 of 32 rows in use):
//   * a unit is 64 tokens (one page) and lane = token on the K side: a lane owns its token's whole 64-byte code row;
//   * scores: v_mfma_f32_4x4x4_16B_f16 - 16 independent blocks of 4 tokens x 4 heads x 4 dims, every MAC useful at G = 4.  A = the
//     two gathered centroid words of subspaces (2 sigma, 2 sigma + 1) AS GATHERED; the subspace is wave-uniform, so the table base
//     rides in the ds_read offset field and a K lookup address is ONE instruction (an SDWA shift of the code byte) instead of
//     two.  B = the query heads: 8 register pairs, k-step sigma = 4 u + lambda sits in lane group lambda of pair u and is
//     broadcast to all blocks by the MFMA's blgp field.  The scores land in lane (token quad b, head j): 4 registers = the quad's
//     4 tokens: all 64 lanes carry useful scores (the 16 x 16 tile: 16 of 64), so the softmax is 4 exponentials per 64 tokens
//     and lane instead of 16;
//   * values: v_mfma_f32_16x16x32_f16, rows = (parity of the dim, head < 8), reduction = (token, parity): the gathered V word is
//     the B operand as it stands (as in parity-V), 16 accumulator registers (32 in parity-V), half the matrix cycles.  The
//     probabilities a lane needs for its A operand - 4 tokens of ONE head - are exactly what a score lane holds; they move inside
//     their 16-lane row by two v_mov_dpp per register (row shifts with a bank mask), no LDS, no permlane.
// The micro checks its own arithmetic against the host (normalised outputs of two waves) before it reports any time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lean_core tools/micro/lean_core.hip && ./lean_core
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef f16 v4f16 __attribute__((ext_vector_type(4)));
typedef f16 v8f16 __attribute__((ext_vector_type(8)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kVBase = 64 * 1024;          // V col image [c][m] behind the K row image [m][c]
constexpr int kWinPagesMax = 1024;
constexpr int G = 4;

__device__ __forceinline__ unsigned lds32(unsigned addr) { return *(const __attribute__((address_space(3))) unsigned *)(size_t)addr; }
typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;
typedef const __attribute__((address_space(1))) v4u *gptr_v4u;
__device__ __forceinline__ gptr_u8 uniform_ptr(const uint8_t *q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gptr_u8)(((unsigned long long)hi << 32) | lo);
}

struct MP {
    const uint8_t *kwin, *vwin;      // (workgroups, win_pages, 4096): K page = [token 64][m 64], V page = [m 64][token 64]
    const v4u *ktab, *vtab;          // K row image [m][c] (4-byte entries), V col image [c][m]
    const f16 *q;                    // (4 heads, 128)
    float *out;                      // (workgroups, waves, 64 lanes, 24): O (16), l, m
    unsigned long long *cyc;
    int n_units;                     // 64-token units per wave
    int win_pages;
    float scale_log2e;
};

struct Unit64 { v4u k[4], v[4]; };      // K: lane t, bytes [16 q, 16 q + 16) of its row; V: tile jt, lane (kg, n): subspace 16 jt + n, tokens 16 kg .. + 15

// VAR 0: the lean core, value rows = (parity, head < 8): lanes (kg, n) and (kg + 1, n) of a 32-lane LDS group gather the SAME
//        subspace for different tokens: every V gather is a 2-way bank conflict.
// VAR 1: VAR 0 with every MFMA replaced by one v_xor that keeps its operands alive (what do the matrix instructions cost?).
// VAR 2: "z-rows" (G <= 4): value rows = (z, parity, head < 4) - all 16 rows in use.  Row (z, p, g) is fed by the token groups kg
//        with (kg & 1) == z ^ phi in the MFMA of phase phi, and those lanes gather subspace 32 pi + 16 z + n: the two token groups of
//        an LDS half read two different 16-subspace column sets (32 distinct banks: conflict-free), phase 1 swaps the roles, both
//        phases accumulate into the same 4 registers (8 accumulator registers in all).  The A operand: (P, P) pairs broadcast from
//        lane bank s to the whole 16-lane row by ONE ds_swizzle per register and token step, then ANDed with a lane-constant mask
//        per phase (half of the dword by parity, zero where the lane's token group does not feed the row).
// VAR 3: VAR 2 with the MFMAs replaced by one v_xor each.
template <int VAR, int NW, int RING>
__global__ __launch_bounds__(NW * 64, NW / 4) void lean_kernel(MP p) {
    constexpr bool ZR = VAR >= 2, NOMFMA = (VAR & 1) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, n16 = lane & 15, hj = lane & 3;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    {
        v4u *ld = (v4u *)smem;
        for (int i = tid; i < 4096; i += NW * 64) { ld[i] = p.ktab[i]; ld[4096 + i] = p.vtab[i]; }
    }
    // query operand: pair u, lane group kg: k-step sigma = 4 u + kg = dims 4 sigma .. 4 sigma + 3 of head (lane & 3)
    v2u Q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int sg = 4 * u + kg;
        v2u t = *(const v2u *)(p.q + hj * 128 + 4 * sg);
        if (hj >= G) t = v2u{0, 0};
        Q[u] = t;
    }
    __syncthreads();
    const uint8_t *kw = p.kwin + (long long)blockIdx.x * p.win_pages * 4096;
    const uint8_t *vw = p.vwin + (long long)blockIdx.x * p.win_pages * 4096;
    const int pg_mask = p.win_pages - 1;
    Unit64 ring[RING];
    const unsigned k_lane_off = (unsigned)lane << 6;
    // ZR: ring.v[x], x = 2 pi + t, holds subspace 32 pi + 16 (t ^ (kg & 1)) + n (phase phi gathers from t = phi)
    const unsigned v_lane_off = ((unsigned)n16 << 6) + 16u * kg;
    const unsigned v_lane_off1 = ZR ? (((unsigned)(n16 + 16 * (kg & 1)) << 6) + 16u * kg) : 0u;      // x even; x odd: the other 16
    const unsigned v_lane_off2 = ZR ? (((unsigned)(n16 + 16 * (1 - (kg & 1))) << 6) + 16u * kg) : 0u;
#define UNIT_REQ_K(SL, J)                                                                                          \
    {                                                                                                              \
        const int pg_ = ((J) * NW + wave) & pg_mask;                                                               \
        const gptr_u8 kb_ = uniform_ptr(kw + pg_ * 4096);                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) ring[SL].k[q] = *(gptr_v4u)(kb_ + k_lane_off + 16u * q);     \
    }
#define UNIT_REQ_V(SL, J)                                                                                          \
    {                                                                                                              \
        const int pg_ = ((J) * NW + wave) & pg_mask;                                                               \
        const gptr_u8 vb_ = uniform_ptr(vw + pg_ * 4096);                                                          \
        if (ZR) {                                                                                                  \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                          \
                ring[SL].v[x] = *(gptr_v4u)(vb_ + ((x & 1) ? v_lane_off2 : v_lane_off1) + 2048u * (x >> 1));       \
        } else {                                                                                                   \
            _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) ring[SL].v[jt] = *(gptr_v4u)(vb_ + v_lane_off + 1024u * jt); \
        }                                                                                                          \
    }
#pragma unroll
    for (int s = 0; s < RING; ++s) { UNIT_REQ_K(s, s) UNIT_REQ_V(s, s) }

    const unsigned vconst = (unsigned)kVBase | ((unsigned)n16 << 2);
    const unsigned vcz0 = (unsigned)kVBase | ((unsigned)(n16 + 16 * (kg & 1)) << 2), vcz1 = (unsigned)kVBase | ((unsigned)(n16 + 16 * (1 - (kg & 1))) << 2);
    // z-rows masks: lane (kg, y = 8 z + 4 p + g); phase phi keeps the lane's (P, P) pair iff (kg & 1) == z ^ phi, in half p
    const unsigned zr_half = ((lane >> 2) & 1) ? 0xffff0000u : 0x0000ffffu;
    const unsigned zmask0 = ((kg & 1) == ((lane >> 3) & 1)) ? zr_half : 0u, zmask1 = ((kg & 1) != ((lane >> 3) & 1)) ? zr_half : 0u;
    unsigned sw[2][4];              // z-rows: swizzled (P, P) pairs of token step s in sw[s & 1]
    unsigned Az[2][4];              // z-rows: A operands of the two phases
    float mref = -INFINITY, lsum = 0.f, lprev = 0.f;
    v4f32 O[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) O[jt] = v4f32{0.f, 0.f, 0.f, 0.f};
    unsigned L[4], H[4];            // probabilities of this lane's (token quad, head): fp16 in the low / in the high half
    unsigned A[4];                  // value A operand of the current token step
    v2u a[4];                       // K gathers in flight (k-step sigma sits in a[sigma & 3])
    unsigned e[4][4];               // V gathers in flight (value step i sits in e[i & 3])
    v4f32 D[2];

    // ---- K side: k-step sigma = subspaces 2 sigma, 2 sigma + 1 (wave-uniform: the table base is the read's offset field) ----
#define KBYTE(SL, B) ((ring[SL].k[(B) >> 4][((B) >> 2) & 3] >> (8 * ((B) & 3))) & 0xffu)
#define KG(SL, SG)                                                                                                 \
    {                                                                                                              \
        a[(SG) & 3][0] = lds32((KBYTE(SL, 2 * (SG)) << 2) + (2 * (SG)) * 1024u);                                   \
        a[(SG) & 3][1] = lds32((KBYTE(SL, 2 * (SG) + 1) << 2) + (2 * (SG) + 1) * 1024u);                           \
    }
#define KM(SG)                                                                                                     \
    {                                                                                                              \
        if (NOMFMA) asm volatile("v_xor_b32 %0, %1, %2" : "+v"(D[(SG) & 1][0]) : "v"(a[(SG) & 3][0]), "v"(a[(SG) & 3][1]), "v"(Q[(SG) >> 2][0]), "v"(Q[(SG) >> 2][1])); \
        else D[(SG) & 1] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4f16, a[(SG) & 3]), __builtin_bit_cast(v4f16, Q[(SG) >> 2]), \
                                                              D[(SG) & 1], 0, 0, 4 + ((SG) & 3));                  \
    }
    // ---- V side: value step i = 4 s + jt: token step s (tokens 16 kg + 4 s + 0..3 = dword s of the lane's 16 bytes), column tile jt ----
#define VGATHER(E, W, VC, IMM)                                                                                     \
    {                                                                                                              \
        E[0] = lds32(__builtin_amdgcn_perm(W, VC, 0x03020400u) + (IMM));                                           \
        E[1] = lds32(__builtin_amdgcn_perm(W, VC, 0x03020500u) + (IMM));                                           \
        E[2] = lds32(__builtin_amdgcn_perm(W, VC, 0x03020600u) + (IMM));                                           \
        E[3] = lds32(__builtin_amdgcn_perm(W, VC, 0x03020700u) + (IMM));                                           \
    }
    // VAR 0: step i = 4 s + jt.  z-rows: step i = 4 s + 2 pi + phi: gathers from ring.v[2 pi + phi], dword s
#define VG(SL, I)                                                                                                  \
    {                                                                                                              \
        const unsigned w_ = ring[SL].v[(I) & 3][(I) >> 2];                                                         \
        if (ZR) VGATHER(e[(I) & 3], w_, (((I) & 1) ? vcz1 : vcz0), 128u * (((I) >> 1) & 1))                        \
        else VGATHER(e[(I) & 3], w_, vconst, 64u * ((I) & 3))                                                      \
    }
    // A operand of token step s: rows (p, g): lanes 0-3 of every 16-lane row take L of lanes 4 s + g, lanes 8-11 take H of them
#define DPP_ID 0xE4
#define DPP_SHL(N) (0x100 + (N))
#define DPP_SHR(N) (0x110 + (N))
#define A_BUILD(S)                                                                                                 \
    {                                                                                                              \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                         \
            unsigned t_ = __builtin_amdgcn_mov_dpp(L[ii], (S) == 0 ? DPP_ID : DPP_SHL(4 * (S)), 0xf, 0x1, false);  \
            A[ii] = __builtin_amdgcn_update_dpp(t_, H[ii], (S) == 0 ? DPP_SHR(8) : (S) == 1 ? DPP_SHR(4) : (S) == 2 ? DPP_ID : DPP_SHL(4), \
                                                0xf, 0x4, false);                                                  \
        }                                                                                                          \
    }
    // z-rows: (P, P) of lane bank s -> every lane of the 16-lane row (bit mode: lane' = (lane & 0x13) | (s << 2))
#define ZSWZ(S) { _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) sw[(S) & 1][ii] = __builtin_amdgcn_ds_swizzle(L[ii], 0x13 | (((S) << 2) << 5)); }
#define ZA_BUILD(S)                                                                                                \
    {                                                                                                              \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) { Az[0][ii] = sw[(S) & 1][ii] & zmask0; Az[1][ii] = sw[(S) & 1][ii] & zmask1; } \
    }
#define VS(I)                                                                                                      \
    {                                                                                                              \
        if (ZR) {                                                                                                  \
            if (((I) & 3) == 0) { ZA_BUILD((I) >> 2) if ((I) < 12) ZSWZ(((I) >> 2) + 1) }                          \
            const v4u av_ = {Az[(I) & 1][0], Az[(I) & 1][1], Az[(I) & 1][2], Az[(I) & 1][3]};                      \
            const v4u bv_ = {e[(I) & 3][0], e[(I) & 3][1], e[(I) & 3][2], e[(I) & 3][3]};                          \
            if (NOMFMA) asm volatile("v_xor_b32 %0, %1, %2" : "+v"(O[((I) >> 1) & 1][0]) : "v"(av_[0]), "v"(bv_[0]), "v"(av_[1]), "v"(av_[2]), "v"(av_[3]), "v"(bv_[1]), "v"(bv_[2]), "v"(bv_[3])); \
            else O[((I) >> 1) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8f16, av_), __builtin_bit_cast(v8f16, bv_), O[((I) >> 1) & 1], 0, 0, 0); \
        } else {                                                                                                   \
            if (((I) & 3) == 0) A_BUILD((I) >> 2)                                                                  \
            const v4u av_ = {A[0], A[1], A[2], A[3]};                                                              \
            const v4u bv_ = {e[(I) & 3][0], e[(I) & 3][1], e[(I) & 3][2], e[(I) & 3][3]};                          \
            if (NOMFMA) asm volatile("v_xor_b32 %0, %1, %2" : "+v"(O[(I) & 3][0]) : "v"(av_[0]), "v"(bv_[0]), "v"(av_[1]), "v"(av_[2]), "v"(av_[3]), "v"(bv_[1]), "v"(bv_[2]), "v"(bv_[3])); \
            else O[(I) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8f16, av_), __builtin_bit_cast(v8f16, bv_), O[(I) & 3], 0, 0, 0); \
        }                                                                                                          \
    }
    // ---- online softmax of the unit whose scores are in D: lane (quad, head j) holds its quad's 4 tokens ----
    // lazy reference (as the shipped core): mref moves only when a score exceeds it by more than 2^8 in the exp2 domain
#define SOFTMAX()                                                                                                  \
    {                                                                                                              \
        float x_[4];                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) x_[i] = (D[0][i] + D[1][i]) * p.scale_log2e;                 \
        const float mx_ = fmaxf(fmaxf(x_[0], x_[1]), fmaxf(x_[2], x_[3]));                                         \
        if (__builtin_amdgcn_ballot_w64(mx_ > mref + 8.f)) {                                                       \
            float m_ = fmaxf(mx_, mref);                          /* new reference per head: max over the lanes of that head */ \
            _Pragma("unroll") for (int sh = 4; sh < 64; sh <<= 1)                                                  \
                m_ = fmaxf(m_, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ sh) << 2, __builtin_bit_cast(int, m_)))); \
            const float al_ = __builtin_amdgcn_exp2f(mref - m_);                                                   \
            lsum *= al_;                                                                                           \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                        \
                const float f_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, al_), g)); \
                _Pragma("unroll") for (int jt = 0; jt < (ZR ? 2 : 4); ++jt) O[jt][g] *= f_;                        \
            }                                                                                                      \
            mref = m_;                                                                                             \
        }                                                                                                          \
        float ps_ = 0.f;                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            const float pr_ = __builtin_amdgcn_exp2f(x_[i] - mref);                                                \
            ps_ += pr_;                                                                                            \
            const f16 h_ = (f16)pr_;                                                                               \
            L[i] = (unsigned)__builtin_bit_cast(unsigned short, h_);                                               \
            H[i] = L[i] << 16;                                                                                     \
            if (ZR) L[i] |= H[i];                                                                                  \
        }                                                                                                          \
        lprev = lsum;                                                                                              \
        lsum += ps_;                                                                                               \
        if (ZR) ZSWZ(0)                                                                                            \
    }
#define FOR16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define FOR32(X) FOR16(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
    // (every step index below is a literal: the MFMA's blgp field and the DPP controls are immediates)
#define SA_STEP(SG) { KM(SG) if ((SG) + 3 < 32) KG(SA_SL, ((SG) + 3) & 31) __builtin_amdgcn_sched_barrier(0); }
    // one block: the 16 value steps of unit U (slot SL) interleaved with the 32 score steps of unit U + 1 (slot SLN; its first three
    // gathers are in flight, the first three of unit U + 2 - slot SLN2 - go out at the end); the K bytes of unit U + RING are
    // requested at the start into slot SL (whose K bytes were used up by the previous block), its V bytes once the last value
    // gather of unit U has been issued
#define BL_STEP(I)                                                                                                 \
    {                                                                                                              \
        VS(I)                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        VG(((I) + 2 < 16 ? BL_SL : BL_SLN), ((I) + 2) & 15)                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        KM(2 * (I))                                                                                                \
        KG((2 * (I) + 3 < 32 ? BL_SLN : BL_SLN2), (2 * (I) + 3) & 31)                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        KM(2 * (I) + 1)                                                                                            \
        KG((2 * (I) + 4 < 32 ? BL_SLN : BL_SLN2), (2 * (I) + 4) & 31)                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if ((I) == 13) UNIT_REQ_V(BL_SL, bl_j + RING)                                                              \
    }

    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    D[0] = v4f32{0.f, 0.f, 0.f, 0.f}; D[1] = v4f32{0.f, 0.f, 0.f, 0.f};
#define SA_SL 0
    KG(0, 0) KG(0, 1) KG(0, 2)
    FOR32(SA_STEP)
#undef SA_SL
    SOFTMAX()
    VG(0, 0) VG(0, 1)
    KG(1 % RING, 0) KG(1 % RING, 1) KG(1 % RING, 2)
    // units 0 .. n_units - 1 get their values; scores run one unit ahead (the last block scores a unit nobody uses: the steady
    // state is what is timed; lprev = the row sums of the units whose values are in O)
    int bl_j = 0;
#define BLOCK_BEGIN() { D[0] = v4f32{0.f, 0.f, 0.f, 0.f}; D[1] = v4f32{0.f, 0.f, 0.f, 0.f}; }
    if (RING == 3) {
        for (; bl_j < p.n_units; ) {
#define BL_SL 0
#define BL_SLN 1
#define BL_SLN2 2
            BLOCK_BEGIN() UNIT_REQ_K(BL_SL, bl_j + RING) FOR16(BL_STEP) SOFTMAX() ++bl_j;
#undef BL_SL
#undef BL_SLN
#undef BL_SLN2
#define BL_SL 1
#define BL_SLN 2
#define BL_SLN2 0
            BLOCK_BEGIN() UNIT_REQ_K(BL_SL, bl_j + RING) FOR16(BL_STEP) SOFTMAX() ++bl_j;
#undef BL_SL
#undef BL_SLN
#undef BL_SLN2
#define BL_SL 2
#define BL_SLN 0
#define BL_SLN2 1
            BLOCK_BEGIN() UNIT_REQ_K(BL_SL, bl_j + RING) FOR16(BL_STEP) SOFTMAX() ++bl_j;
#undef BL_SL
#undef BL_SLN
#undef BL_SLN2
        }
    } else {
        for (; bl_j < p.n_units; ) {
#define BL_SL 0
#define BL_SLN (1 % RING)
#define BL_SLN2 0
            BLOCK_BEGIN() UNIT_REQ_K(BL_SL, bl_j + RING) FOR16(BL_STEP) SOFTMAX() ++bl_j;
#undef BL_SL
#undef BL_SLN
#undef BL_SLN2
#define BL_SL (1 % RING)
#define BL_SLN 0
#define BL_SLN2 (1 % RING)
            BLOCK_BEGIN() UNIT_REQ_K(BL_SL, bl_j + RING) FOR16(BL_STEP) SOFTMAX() ++bl_j;
#undef BL_SL
#undef BL_SLN
#undef BL_SLN2
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { p.cyc[blockIdx.x * 16 + wave] = t1 - t0; p.cyc[4096 + blockIdx.x * 16 + wave] = r1 - r0; }
    float *o = p.out + (((long long)blockIdx.x * NW + wave) * 64 + lane) * 24;
#pragma unroll
    for (int jt = 0; jt < (ZR ? 2 : 4); ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) o[4 * jt + g] = O[jt][g];
    o[16] = lprev; o[17] = mref; o[18] = __uint_as_float((ZR ? Az[0][0] ^ Az[1][1] : A[0]) ^ e[0][0] ^ a[0][0]); o[19] = D[0][0];
}

template <int VAR, int NW, int RING>
static double run(const char *name, MP p, int n_units, bool check, const std::vector<uint8_t> &hk, const std::vector<uint8_t> &hv,
                  const std::vector<f16> &tab, const std::vector<f16> &hq) {
    p.n_units = n_units;
    const int lds = 2 * 64 * 1024 + 4096;
    CK(hipFuncSetAttribute((const void *)lean_kernel<VAR, NW, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lean_kernel<VAR, NW, RING>), dim3(256), dim3(NW * 64), lds, 0, p);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lean_kernel<VAR, NW, RING>), dim3(256), dim3(NW * 64), lds, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * 256 * 16);
    CK(hipMemcpy(h.data(), p.cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> per, clk;
    for (int b = 0; b < 256; ++b) {
        double s = 0;
        for (int w = 0; w < NW; ++w) s += (double)h[b * 16 + w];
        per.push_back(s / NW);
        clk.push_back((double)h[b * 16] / (double)h[4096 + b * 16] * 0.1);
    }
    std::sort(per.begin(), per.end());
    std::sort(clk.begin(), clk.end());
    const double us = ms * 1e3 / reps;
    const double units32 = 256.0 * NW * n_units * 2.0;      // in the 32-token units of core_micro.hip, for comparison
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)lean_kernel<VAR, NW, RING>));
    double worst = 0.0;
    if (check && (VAR & 1) == 0) {
        // host reference for workgroup 3, waves 0 and NW - 1: normalised output of the units 0 .. n_units - 1
        std::vector<float> ho((size_t)256 * NW * 64 * 24);
        CK(hipMemcpy(ho.data(), p.out, ho.size() * 4, hipMemcpyDeviceToHost));
        const int wg = 3;
        for (int w : {0, NW - 1}) {
            std::vector<double> sc((size_t)n_units * 64 * G);
            double mx[G];
            for (int g = 0; g < G; ++g) mx[g] = -1e300;
            for (int j = 0; j < n_units; ++j) {
                const int pg = (j * NW + w) & (p.win_pages - 1);
                const uint8_t *kp = hk.data() + ((size_t)wg * p.win_pages + pg) * 4096;
                for (int t = 0; t < 64; ++t)
                    for (int g = 0; g < G; ++g) {
                        double s = 0;
                        for (int m = 0; m < 64; ++m) {
                            const int c = kp[t * 64 + m];
                            s += (double)(float)hq[g * 128 + 2 * m] * (float)tab[(m * 256 + c) * 2] + (double)(float)hq[g * 128 + 2 * m + 1] * (float)tab[(m * 256 + c) * 2 + 1];
                        }
                        s *= p.scale_log2e;
                        sc[((size_t)j * 64 + t) * G + g] = s;
                        mx[g] = std::max(mx[g], s);
                    }
            }
            std::vector<double> num((size_t)G * 128, 0.0);
            double den[G] = {0, 0, 0, 0};
            for (int j = 0; j < n_units; ++j) {
                const int pg = (j * NW + w) & (p.win_pages - 1);
                const uint8_t *vp = hv.data() + ((size_t)wg * p.win_pages + pg) * 4096;
                for (int t = 0; t < 64; ++t)
                    for (int g = 0; g < G; ++g) {
                        const double pr = std::exp2(sc[((size_t)j * 64 + t) * G + g] - mx[g]);
                        den[g] += pr;
                        for (int m = 0; m < 64; ++m) {
                            const int c = vp[m * 64 + t];
                            num[g * 128 + 2 * m] += pr * (float)tab[32768 + (c * 64 + m) * 2];
                            num[g * 128 + 2 * m + 1] += pr * (float)tab[32768 + (c * 64 + m) * 2 + 1];
                        }
                    }
            }
            const float *o = ho.data() + ((size_t)wg * NW + w) * 64 * 24;
            double l[G] = {0, 0, 0, 0};
            for (int ln = 0; ln < 64; ++ln) l[ln & 3] += o[ln * 24 + 16];
            double e2 = 0, r2 = 0;
            for (int g = 0; g < G; ++g)
                for (int jt = 0; jt < 4; ++jt)
                    for (int n = 0; n < 16; ++n)
                        for (int par = 0; par < 2; ++par) {
                            // VAR 0: rows 0-3 (parity 0) live in lanes 0-15, rows 8-11 (parity 1) in lanes 32-47; tile jt = subspaces 16 jt + n
                            // z-rows: row 8 z + 4 p + g -> lane group rg = 2 z + p, register g; accumulator pi: subspace 32 pi + 16 z + n
                            const int ln = VAR >= 2 ? 16 * (2 * (jt & 1) + par) + n : (par ? 32 : 0) + n;
                            const int slot = VAR >= 2 ? 4 * (jt >> 1) + g : 4 * jt + g;
                            const double got = o[ln * 24 + slot] / l[g], want = num[g * 128 + 2 * (16 * jt + n) + par] / den[g];
                            e2 += (got - want) * (got - want);
                            r2 += want * want;
                        }
            const double rel = std::sqrt(e2 / r2);
            if (!(rel == rel)) worst = 1e9; else worst = std::max(worst, rel);
            printf("   check wave %d: rel %.3e  l = %.4g %.4g %.4g %.4g (host %.4g %.4g %.4g %.4g)  O[0] got %.5f want %.5f\n", w, rel, l[0], l[1], l[2], l[3], den[0], den[1], den[2], den[3], o[0] / l[0], num[0] / den[0]);
        }
    }
    printf("%-26s ring %d x 64 tok  %2d waves/CU  VGPRs %3d  %8.1f us/launch  %6.3f units32/us/SIMD  %7.0f wave-cycles/unit32 (median WG)  = %.2f TB/s of codes  clock %.2f GHz",
           name, RING, NW, fa.numRegs, us, units32 / us / 1024.0, per[128] / (n_units * 2.0), units32 * 4096.0 / us * 1e-6, clk[128]);
    if (check && (VAR & 1) == 0) printf("  host check rel-L2 %.1e %s", worst, worst < 2e-3 ? "ok" : "MISMATCH");
    printf("\n");
    return worst;
}

int main() {
    MP p{};
    const size_t win = 256ull * kWinPagesMax * 4096;
    std::vector<uint8_t> hk(win), hv(win);
    srand(3);
    {
        unsigned long long x = 88172645463325252ull, *pk = (unsigned long long *)hk.data(), *pv = (unsigned long long *)hv.data();
        for (size_t i = 0; i < win / 8; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17; pk[i] = x;
            x ^= x << 13; x ^= x >> 7; x ^= x << 17; pv[i] = x;
        }
    }
    uint8_t *dk, *dv;
    CK(hipMalloc(&dk, win)); CK(hipMalloc(&dv, win));
    CK(hipMemcpy(dk, hk.data(), win, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), win, hipMemcpyHostToDevice));
    std::vector<f16> tab(2 * 32768), hq(4 * 128);
    for (auto &x : tab) x = (f16)((rand() % 2001 - 1000) * 0.001f);
    for (auto &x : hq) x = (f16)((rand() % 2001 - 1000) * 0.004f);
    f16 *dtab, *dq;
    CK(hipMalloc(&dtab, tab.size() * 2)); CK(hipMalloc(&dq, hq.size() * 2));
    CK(hipMemcpy(dtab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    float *dout;
    unsigned long long *dcyc;
    CK(hipMalloc(&dout, 256ull * 16 * 64 * 24 * 4));
    CK(hipMalloc(&dcyc, 2 * 256 * 16 * 8));
    p.kwin = dk; p.vwin = dv; p.ktab = (const v4u *)dtab; p.vtab = (const v4u *)(dtab + 32768); p.q = dq; p.out = dout; p.cyc = dcyc;
    p.scale_log2e = 1.4426950408889634f / sqrtf(128.f);
    printf("lean core micro: 256 workgroups (1 per CU), G = 4, both codebooks in LDS; units of 64 tokens, reported in 32-token units (core_micro.hip's)\n");
    // arithmetic first: a short run whose outputs are compared with the host
    p.win_pages = 8;
    double bad = 0;
    bad = std::max(bad, run<0, 8, 3>("lean core (check)", p, 6, true, hk, hv, tab, hq));
    bad = std::max(bad, run<0, 8, 2>("lean core (check)", p, 6, true, hk, hv, tab, hq));
    bad = std::max(bad, run<2, 8, 3>("lean z-rows (check)", p, 6, true, hk, hv, tab, hq));
    bad = std::max(bad, run<2, 8, 2>("lean z-rows (check)", p, 6, true, hk, hv, tab, hq));
    if (!(bad < 2e-3)) { printf("ARITHMETIC MISMATCH - no timings\n"); return 1; }
    for (int hbm = 0; hbm < 2; ++hbm) {
        p.win_pages = hbm ? kWinPagesMax : 8;
        printf("-- codes %s\n", hbm ? "streamed from HBM (8 MiB per workgroup per launch, 2 GiB in rotation)" : "L2-resident (64 KiB window per workgroup)");
        run<0, 8, 3>("lean core", p, 126, false, hk, hv, tab, hq);
        run<0, 8, 2>("lean core", p, 128, false, hk, hv, tab, hq);
        run<0, 12, 2>("lean core", p, 86, false, hk, hv, tab, hq);
        run<1, 8, 3>("lean, MFMA -> 1 VALU", p, 126, false, hk, hv, tab, hq);
        run<2, 8, 3>("lean z-rows", p, 126, false, hk, hv, tab, hq);
        run<2, 8, 2>("lean z-rows", p, 128, false, hk, hv, tab, hq);
        run<2, 12, 2>("lean z-rows", p, 86, false, hk, hv, tab, hq);
        run<3, 8, 3>("z-rows, MFMA -> 1 VALU", p, 126, false, hk, hv, tab, hq);
    }
    p.win_pages = 8;
    CK(hipMemset(dv, 0x5a, win));
    printf("-- V codes all equal (no V bank conflict), L2-resident\n");
    run<0, 8, 3>("lean, equal V codes", p, 126, false, hk, hv, tab, hq);
    run<2, 8, 3>("z-rows, equal V codes", p, 126, false, hk, hv, tab, hq);
    CK(hipMemset(dk, 0x5a, win));
    printf("-- K and V codes all equal, L2-resident\n");
    run<0, 8, 3>("lean, equal K, V codes", p, 126, false, hk, hv, tab, hq);
    run<2, 8, 3>("z-rows, equal K, V codes", p, 126, false, hk, hv, tab, hq);
    CK(hipMemcpy(dv, hv.data(), win, hipMemcpyHostToDevice));
    printf("-- K codes all equal, L2-resident\n");
    run<0, 8, 3>("lean, equal K codes", p, 126, false, hk, hv, tab, hq);
    run<2, 8, 3>("z-rows, equal K codes", p, 126, false, hk, hv, tab, hq);
    return 0;
}
