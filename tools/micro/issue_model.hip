// Microbenchmark: what ONE SIMD of a gfx950 CU issues per cycle, by instruction class and by waves per SIMD.
//
// The decode-attention core (attn_mfma.hip) is ~400 instructions per wave and 32-token unit and runs at two waves per SIMD;
// round 2-3 measured "5.2 cycles per instruction per SIMD with two waves, 6.3 with one" and concluded the SIMD's issue port
// is what is used up.  This micro prices the pieces in isolation so that a candidate core can be costed before it is
// written: vector-ALU forms the core uses (v_perm_b32, v_lshl_or_b32, v_bfe_u32, SDWA shifts, v_exp_f32), LDS gathers
// (conflict-free and random over a 64-KiB table), the two MFMA shapes, and interleaved mixes, at 1 / 2 / 3 / 4 waves per SIMD
// (one workgroup of 256 / 512 / 768 / 1024 threads per CU: 66 KiB of LDS keeps a second workgroup off the CU).
// Output: shader cycles (s_memtime) per instruction per SIMD = (last wave out - first wave in) / (instructions per wave x
// waves per SIMD), median over workgroups; in brackets the same from the MEAN wave time (a SIMD serves its older waves first:
// they finish early, so the mean understates what the SIMD needed).
//   hipcc --offload-arch=gfx950 -O3 -o issue_model issue_model.hip && ./issue_model
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kIters = 200;
constexpr int kLds = 66 * 1024;

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)

// 64 vector instructions per iteration on 8 independent registers
#define VALU_KERNEL(NAME, ASM8)                                                                                    \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *cyc, unsigned *sink, unsigned seed) {         \
        extern __shared__ unsigned lds[];                                                                          \
        unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 + 77, a3 = a1 + 99,          \
                 a4 = a0 * 3, a5 = a1 * 5, a6 = a2 * 7, a7 = a3 * 11, b = seed | 0x03020100u, c = 0x00010203u;     \
        __syncthreads();                                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int it = 0; it < kIters; ++it) {                                                                      \
            asm volatile(R8(ASM8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c));                                                                        \
        }                                                                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 32 + (threadIdx.x >> 6)] = t0; cyc[blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = t1; }                          \
        sink[blockIdx.x * 1024 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                             \
    }

VALU_KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                    "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")
VALU_KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %8\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %8\n"
                      "v_lshl_or_b32 %4, %4, 2, %8\n v_lshl_or_b32 %5, %5, 2, %8\n v_lshl_or_b32 %6, %6, 2, %8\n v_lshl_or_b32 %7, %7, 2, %8\n")
VALU_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n"
                   "v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %5, %5, 8, 8\n v_bfe_u32 %6, %6, 8, 8\n v_bfe_u32 %7, %7, 8, 8\n")
VALU_KERNEL(k_sdwa, "v_lshlrev_b32_sdwa %0, %9, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
                    "v_lshlrev_b32_sdwa %1, %9, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
                    "v_lshlrev_b32_sdwa %2, %9, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
                    "v_lshlrev_b32_sdwa %3, %9, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
                    "v_lshlrev_b32_sdwa %4, %9, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
                    "v_lshlrev_b32_sdwa %5, %9, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
                    "v_lshlrev_b32_sdwa %6, %9, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
                    "v_lshlrev_b32_sdwa %7, %9, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n")
VALU_KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
VALU_KERNEL(k_fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
VALU_KERNEL(k_cvtpk, "v_cvt_pk_f16_f32 %0, %0, %8\n v_cvt_pk_f16_f32 %1, %1, %8\n v_cvt_pk_f16_f32 %2, %2, %8\n v_cvt_pk_f16_f32 %3, %3, %8\n"
                     "v_cvt_pk_f16_f32 %4, %4, %8\n v_cvt_pk_f16_f32 %5, %5, %8\n v_cvt_pk_f16_f32 %6, %6, %8\n v_cvt_pk_f16_f32 %7, %7, %8\n")

// LDS gathers: 64 ds_read_b32 per iteration, 16 in flight; RANDOM = addresses spread over a 64-KiB table like K code bytes
// (a subspace's 1-KiB row, bank = code), else conflict-free (lane-consecutive dwords); each result feeds the next address of
// its own chain only through an AND with 0 (keeps the dependency, not the value)
template <bool RANDOM>
__global__ __launch_bounds__(1024) void k_lds(unsigned long long *cyc, unsigned *sink, unsigned seed) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const unsigned h = (threadIdx.x * 40503u + j * 9973u + seed) * 2654435761u;
        addr[j] = RANDOM ? (((lane >> 4) * 16384u + (j & 15) * 1024u + ((h >> 20) & 0xffu) * 4u)) : (unsigned)((lane + 64 * j) * 4);
    }
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = *(volatile __attribute__((address_space(3))) unsigned *)(size_t)addr[j];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc ^= v[j];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cyc[blockIdx.x * 32 + (threadIdx.x >> 6)] = t0; cyc[blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = t1; }
    sink[blockIdx.x * 1024 + threadIdx.x] = acc;
}

// MFMA alone: 16 per iteration on 4 accumulators
template <int SHAPE>
__global__ __launch_bounds__(1024) void k_mfma(unsigned long long *cyc, unsigned *sink, unsigned seed) {
    extern __shared__ unsigned lds[];
    v8h a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * ((threadIdx.x + i + seed) & 15)); b[i] = (_Float16)(0.02f * ((threadIdx.x * 3 + i) & 7)); }
    v4f d4[4] = {};
    v16f d16[4] = {};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (SHAPE == 16) d4[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d4[j & 3], 0, 0, 0);
            else d16[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d16[j & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 32 + (threadIdx.x >> 6)] = t0; cyc[blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = t1; }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s += d4[j][0] + d4[j][3]; s += d16[j][0] + d16[j][15]; }
    sink[blockIdx.x * 1024 + threadIdx.x] = __float_as_uint(s);
}

// The core's mix, per "quarter unit" (x 4 = one 32-token unit of attn_stream_kernel): 8 V address v_perm + 8 gathers
// (conflict-free), 8 K address ops (bfe + lshl_or) + 4 random gathers ... i.e. per iteration of this loop:
//   16 ds_read_b32 (8 conflict-free, 8 random) | VADDR: 8 v_perm | KADDR: 16 (bfe / lshl_or) | PACK: 8 v_perm |
//   2 MFMA 16x16x32 + 2 MFMA 32x32x16 | SOFT: 2 fma + 2 exp + 2 add + 1 cvt_pk
// MASK selects the pieces: 1 gathers, 2 V address, 4 K address, 8 pack, 16 MFMA, 32 softmax.  Instructions per iteration
// are counted by the host from the mask.
template <int MASK>
__global__ __launch_bounds__(1024) void k_mix(unsigned long long *cyc, unsigned *sink, unsigned seed) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned w0 = (threadIdx.x * 40503u + seed) * 2654435761u, w1 = w0 * 7u + 3u;
    const unsigned kbase = (lane >> 4) * 16384u, vconst = (lane & 31) << 2;
    v8h qa, pb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { qa[i] = (_Float16)(0.01f * ((threadIdx.x + i) & 15)); pb[i] = (_Float16)(0.02f * ((threadIdx.x * 3 + i) & 7)); }
    v4f D = {};
    v16f O0 = {}, O1 = {};
    float sc0 = 0.3f, sc1 = 0.7f, lsum = 0.f;
    unsigned pk = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        unsigned av[8], ak[8], ev[8], ek[8];
        // V side: address = v_perm(code word, lane const) (conflict-free col image), 8 gathers
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MASK & 2) av[j] = __builtin_amdgcn_perm(j < 4 ? w0 : w1, vconst, 0x03020400u + ((j & 3) << 8));
            else av[j] = vconst + 256u * j;
            if (MASK & 1) ev[j] = *(volatile __attribute__((address_space(3))) unsigned *)(size_t)(av[j] & 0xffffu);
            else ev[j] = av[j];
        }
        // K side: address = ((byte) << 2) | lane/subspace base (random rows of the row image), 8 gathers
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MASK & 4) {
                unsigned t;
                asm volatile("v_bfe_u32 %0, %1, %2, 8" : "=v"(t) : "v"(j < 4 ? w1 : w0), "v"(8u * (j & 3)));
                asm volatile("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(ak[j]) : "v"(t), "v"(kbase + 1024u * j));
            } else ak[j] = kbase + 1024u * j + 4u * lane;
            if (MASK & 1) ek[j] = *(volatile __attribute__((address_space(3))) unsigned *)(size_t)ak[j];
            else ek[j] = ak[j];
        }
        // pack: even / odd halves of token pairs -> B operands
        unsigned b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MASK & 8) {
                b0[j] = __builtin_amdgcn_perm(ev[2 * j + 1], ev[2 * j], 0x05040100u);
                b1[j] = __builtin_amdgcn_perm(ev[2 * j + 1], ev[2 * j], 0x07060302u);
            } else { b0[j] = ev[2 * j]; b1[j] = ev[2 * j + 1]; }
        }
        if (MASK & 16) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u ka0 = {ek[0], ek[1], ek[2], ek[3]}, ka1 = {ek[4], ek[5], ek[6], ek[7]};
            const v4u vb0 = {b0[0], b0[1], b0[2], b0[3]}, vb1 = {b1[0], b1[1], b1[2], b1[3]};
            D = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, ka0), qa, D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, ka1), qa, D, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb, __builtin_bit_cast(v8h, vb0), O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb, __builtin_bit_cast(v8h, vb1), O1, 0, 0, 0);
        } else {
            w0 ^= ek[0] ^ ek[1] ^ ek[2] ^ ek[3] ^ ek[4] ^ ek[5] ^ ek[6] ^ ek[7];
            w1 ^= b0[0] ^ b0[1] ^ b0[2] ^ b0[3] ^ b1[0] ^ b1[1] ^ b1[2] ^ b1[3];
        }
        if (MASK & 32) {
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc0, 0.5f, -1.f)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc1, 0.5f, -1.f));
            lsum += e0;
            lsum += e1;
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 t = {(_Float16)e0, (_Float16)e1};
            pk ^= __builtin_bit_cast(unsigned, t);
            sc0 = e1; sc1 = e0;
        }
        w0 = w0 * 1664525u + 1013904223u;      // 2 VALU of bookkeeping per iteration (counted)
        w1 ^= w0;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cyc[blockIdx.x * 32 + (threadIdx.x >> 6)] = t0; cyc[blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = t1; }
    float s = D[0] + D[3] + O0[0] + O0[15] + O1[0] + O1[15] + lsum;
    sink[blockIdx.x * 1024 + threadIdx.x] = __float_as_uint(s) ^ w0 ^ w1 ^ pk;
}

// Do two instruction classes OVERLAP on one SIMD when they come from different waves?  512-thread workgroup: waves 0-3 (one per
// SIMD) run role RA, waves 4-7 role RB.  Roles: 0 idle, 1 VALU (v_perm_b32), 2 LDS gathers conflict-free, 3 LDS gathers random
// rows, 4 MFMA 16x16x32, 5 MFMA 32x32x16, 6 v_exp_f32.  Every role issues 64 instructions per iteration with nothing else in
// the loop (gathers: 16 in flight, then one wait).  Output per role: cycles per instruction of that wave.
template <int ROLE>
__device__ __forceinline__ unsigned role_body(int lane, unsigned seed, unsigned long long &t0, unsigned long long &t1) {
    unsigned acc = 0;
    if (ROLE == 1 || ROLE == 6) {
        unsigned a0 = lane * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 + 77, a3 = a1 + 99, a4 = a0 * 3, a5 = a1 * 5, a6 = a2 * 7,
                 a7 = a3 * 11, b = seed | 0x03020100u, c = 0x00010203u;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < kIters; ++it) {
            if (ROLE == 1)
                asm volatile(R8("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                                "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            else
                asm volatile(R8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                                "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        t1 = __builtin_amdgcn_s_memtime();
        acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    } else if (ROLE == 2 || ROLE == 3) {
        unsigned addr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned h = (lane * 40503u + j * 9973u + seed) * 2654435761u;
            addr[j] = ROLE == 3 ? (((lane >> 4) * 16384u + (j & 15) * 1024u + ((h >> 20) & 0xffu) * 4u)) : (unsigned)((lane + 64 * j) * 4);
        }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)      // the result register is overwritten by the next group: no VALU in the loop
                    asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"(addr[j]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" :: "v"(v[0]), "v"(v[5]), "v"(v[15]));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    } else if (ROLE == 4 || ROLE == 5) {
        v8h a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * ((lane + i + seed) & 15)); b[i] = (_Float16)(0.02f * ((lane * 3 + i) & 7)); }
        v4f d4[4] = {};
        v16f d16[4] = {};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (ROLE == 4) d4[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d4[j & 3], 0, 0, 0);
                else d16[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d16[j & 3], 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        float sres = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sres += d4[j][0] + d16[j][0];
        acc = __float_as_uint(sres);
    } else {
        t0 = t1 = __builtin_amdgcn_s_memtime();
    }
    return acc;
}
template <int RA, int RB>
__global__ __launch_bounds__(512) void k_roles(unsigned long long *cyc, unsigned *sink, unsigned seed) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0, t1;
    unsigned acc;
    if (wave < 4) acc = role_body<RA>(lane, seed, t0, t1);
    else acc = role_body<RB>(lane, seed, t0, t1);
    if (lane == 0) cyc[blockIdx.x * 32 + wave] = t1 - t0;
    sink[blockIdx.x * 1024 + threadIdx.x] = acc;
}
template <int RA, int RB>
static void run_roles(const char *name, unsigned long long *d_cyc, unsigned *d_sink) {
    CK(hipFuncSetAttribute((const void *)k_roles<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipMemset(d_cyc, 0, 256 * 32 * 8));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_roles<RA, RB>), dim3(256), dim3(512), kLds, 0, d_cyc, d_sink, 17u + rep);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 32);
    CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ta, tb;
    for (int b = 0; b < 256; ++b) {
        double sa = 0, sb = 0;
        for (int w = 0; w < 4; ++w) { sa += (double)h[b * 32 + w]; sb += (double)h[b * 32 + 4 + w]; }
        ta.push_back(sa / 4);
        tb.push_back(sb / 4);
    }
    std::sort(ta.begin(), ta.end());
    std::sort(tb.begin(), tb.end());
    printf("%-44s waves 0-3: %6.2f cyc/instr   waves 4-7: %6.2f cyc/instr\n", name, ta[128] / (64.0 * kIters), tb[128] / (64.0 * kIters));
}

static int mix_instr(int mask) {      // instructions per loop iteration (vector + LDS + MFMA; scalar loop control not counted)
    int n = 3;                                       // bookkeeping (mul-add, xor) ~3 VALU
    if (mask & 1) n += 16;
    if (mask & 2) n += 8;
    if (mask & 4) n += 16;
    if (mask & 8) n += 8;
    if (mask & 16) n += 4; else n += 16;             // xor reductions stand in when the MFMAs are off
    if (mask & 32) n += 7;
    return n;
}

template <class K>
static void run(const char *name, K kern, int instr_per_iter, unsigned long long *d_cyc, unsigned *d_sink) {
    printf("%-34s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 256 * wps, nwaves = 4 * wps;
        CK(hipMemset(d_cyc, 0, 256 * 32 * 8));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), kLds, 0, d_cyc, d_sink, 17u + rep);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(256 * 32);
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> span, mean;
        for (int b = 0; b < 256; ++b) {
            unsigned long long lo = ~0ull, hi = 0;
            double s = 0;
            for (int w = 0; w < nwaves; ++w) {
                lo = std::min(lo, h[b * 32 + w]);
                hi = std::max(hi, h[b * 32 + 16 + w]);
                s += (double)(h[b * 32 + 16 + w] - h[b * 32 + w]);
            }
            span.push_back((double)(hi - lo));      // first wave in -> last wave out: what the SIMDs needed for all their waves
            mean.push_back(s / nwaves);
        }
        std::sort(span.begin(), span.end());
        std::sort(mean.begin(), mean.end());
        // a SIMD hosts wps waves, each issuing instr_per_iter * kIters instructions, all done after `span` cycles
        printf("  %dw: %5.2f (%5.2f)", wps, span[128] / ((double)instr_per_iter * kIters * wps), mean[128] / ((double)instr_per_iter * kIters * wps));
    }
    printf("   (%d instr/iter/wave)\n", instr_per_iter);
}

int main() {
    unsigned long long *d_cyc;
    unsigned *d_sink;
    CK(hipMalloc(&d_cyc, 256 * 32 * 8));
    CK(hipMalloc(&d_sink, 256 * 1024 * 4));
#define ATTR(k) CK(hipFuncSetAttribute((const void *)(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds))
    ATTR(k_perm); ATTR(k_lshlor); ATTR(k_bfe); ATTR(k_sdwa); ATTR(k_exp); ATTR(k_fma); ATTR(k_cvtpk);
    ATTR(k_lds<false>); ATTR(k_lds<true>); ATTR(k_mfma<16>); ATTR(k_mfma<32>);
    ATTR(k_mix<63>); ATTR(k_mix<62>); ATTR(k_mix<47>); ATTR(k_mix<31>); ATTR(k_mix<14>); ATTR(k_mix<1>); ATTR(k_mix<17>); ATTR(k_mix<55>); ATTR(k_mix<59>); ATTR(k_mix<61>);
    printf("shader cycles per instruction per SIMD at 1-4 waves per SIMD: workgroup makespan (and, in brackets, mean wave time: lower when the\n older waves of a SIMD finish first), median workgroup, one workgroup per CU, 256 workgroups\n");
    run("v_perm_b32", k_perm, 64, d_cyc, d_sink);
    run("v_lshl_or_b32", k_lshlor, 64, d_cyc, d_sink);
    run("v_bfe_u32", k_bfe, 64, d_cyc, d_sink);
    run("v_lshlrev_b32_sdwa (byte select)", k_sdwa, 64, d_cyc, d_sink);
    run("v_fma_f32", k_fma, 64, d_cyc, d_sink);
    run("v_cvt_pk_f16_f32", k_cvtpk, 64, d_cyc, d_sink);
    run("v_exp_f32", k_exp, 64, d_cyc, d_sink);
    run("ds_read_b32 conflict-free", k_lds<false>, 128, d_cyc, d_sink);      // 64 reads + 64 xor per iteration
    run("ds_read_b32 random 64 KiB rows", k_lds<true>, 128, d_cyc, d_sink);
    run("v_mfma_f32_16x16x32_f16", k_mfma<16>, 16, d_cyc, d_sink);
    run("v_mfma_f32_32x32x16_f16", k_mfma<32>, 16, d_cyc, d_sink);
    run("mix: all pieces", k_mix<63>, mix_instr(63), d_cyc, d_sink);
    run("mix: no gathers", k_mix<62>, mix_instr(62), d_cyc, d_sink);
    run("mix: no MFMA", k_mix<47>, mix_instr(47), d_cyc, d_sink);
    run("mix: no softmax", k_mix<31>, mix_instr(31), d_cyc, d_sink);
    run("mix: no pack", k_mix<55>, mix_instr(55), d_cyc, d_sink);
    run("mix: no K address", k_mix<59>, mix_instr(59), d_cyc, d_sink);
    run("mix: no V address", k_mix<61>, mix_instr(61), d_cyc, d_sink);
    run("mix: address + pack VALU only", k_mix<14>, mix_instr(14), d_cyc, d_sink);
    run("mix: gathers only", k_mix<1>, mix_instr(1), d_cyc, d_sink);
    run("mix: gathers + MFMA", k_mix<17>, mix_instr(17), d_cyc, d_sink);
    printf("\ntwo roles on one SIMD (8 waves per CU: waves 0-3 role A, waves 4-7 role B; 64 instructions per iteration each)\n");
    run_roles<1, 0>("VALU | idle", d_cyc, d_sink);
    run_roles<2, 0>("LDS conflict-free | idle", d_cyc, d_sink);
    run_roles<3, 0>("LDS random | idle", d_cyc, d_sink);
    run_roles<4, 0>("MFMA 16x16x32 | idle", d_cyc, d_sink);
    run_roles<1, 1>("VALU | VALU", d_cyc, d_sink);
    run_roles<1, 2>("VALU | LDS conflict-free", d_cyc, d_sink);
    run_roles<1, 3>("VALU | LDS random", d_cyc, d_sink);
    run_roles<2, 1>("LDS conflict-free | VALU", d_cyc, d_sink);
    run_roles<3, 1>("LDS random | VALU", d_cyc, d_sink);
    run_roles<2, 2>("LDS conflict-free | LDS conflict-free", d_cyc, d_sink);
    run_roles<3, 3>("LDS random | LDS random", d_cyc, d_sink);
    run_roles<1, 4>("VALU | MFMA 16x16x32", d_cyc, d_sink);
    run_roles<1, 5>("VALU | MFMA 32x32x16", d_cyc, d_sink);
    run_roles<4, 1>("MFMA 16x16x32 | VALU", d_cyc, d_sink);
    run_roles<2, 4>("LDS conflict-free | MFMA 16x16x32", d_cyc, d_sink);
    run_roles<3, 5>("LDS random | MFMA 32x32x16", d_cyc, d_sink);
    run_roles<6, 1>("v_exp | VALU", d_cyc, d_sink);
    run_roles<6, 3>("v_exp | LDS random", d_cyc, d_sink);
    return 0;
}
