// Microbenchmark: the score / softmax / value CORE of the fused decode-attention kernel on cache-resident data.
//
// "a standalone micro of the candidate core on LDS-resident data ... ISA instruction counts and
// VGPR count ... go to the kernel only if the micro shows >= 1.5x units/us per SIMD".  Both codebooks sit in LDS exactly as in
// attn_stream_kernel; the code bytes come from a 64-KiB window per workgroup (2 MiB per XCD: L2-resident), through the same
// 16-byte loads into the same 4-slot ring; there is no front, no page table, no tail: what is timed is the steady state of the
// core, in units (32 tokens x 128 code bytes of one kv head, G = 4 query heads) per microsecond and SIMD.
//   VAR 0: the shipped core (attn_mfma.hip: BLOCK = value steps of unit j | score stages of unit j + 1), 8 waves per CU
//   VAR 1: parity-V core (attn_mfma.hip "parity-V", shipped since round 4 for M = 64): the gathered V word IS the B operand (k = token x parity-of-dim), the probabilities carry
//          the zero pattern (rows = head x parity): no pack v_perm, 32 accumulator registers instead of 64
// The helper functions of VAR 0 are the kernel's own (the source is included, not restated).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I million_amd/csrc -o core_micro tools/micro/core_micro.hip && ./core_micro
#include "../../million_amd/csrc/attn_mfma.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace million {      // the three host symbols attn_mfma.hip expects from million_api.hip
int device_cus() { return 256; }
bool device_once(int) { return true; }
void set_error(const char *, ...) {}
}  // namespace million
using namespace million;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kWinPagesMax = 1024; // K pages and V pages per workgroup window (4 KiB each): 8 = L2-resident, 1024 = streamed from HBM

struct MP {
    const uint8_t *kwin, *vwin;      // (workgroups, kWinPages, 4096)
    const v4u *ktab, *vtab;          // prepared images: K row image, V col image (64 KiB each)
    const f16 *q;                    // (4 heads, 128)
    float *out;                      // (workgroups, waves, 64 lanes, 8)
    unsigned long long *cyc;         // (workgroups, 16 waves)
    int n_whole;                     // rounds of four units per wave
    int win_pages;                   // power of two <= kWinPagesMax
    float scale_log2e;
};

template <int VAR, int NW, int RING>
__global__ __launch_bounds__(NW * 64, NW / 4) void core_kernel(MP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef UnitCodes Unit;
    constexpr int CL2 = 8, NV = 4, SPV = 2, NT = 4096 / (NW * 64) * 2;      // 16-byte pieces of one image per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, c16 = lane & 15;
    const int G = 4;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    // tables -> LDS
    {
        v4u *ld = (v4u *)smem;
        for (int i = tid; i < 4096; i += NW * 64) { ld[i] = p.ktab[i]; ld[4096 + i] = p.vtab[i]; }
    }
    v8f16 qb[4];
    {
        const f16 *qv = p.q + (c16 < G ? c16 : 0) * 128 + 32 * q4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v4u t = *(const v4u *)(qv + 8 * s);
            if (c16 >= G) t = v4u{0, 0, 0, 0};
            qb[s] = __builtin_bit_cast(v8f16, t);
        }
    }
    __syncthreads();
    // where this wave reads: round j -> window page (j * (NW / 2) + wave / 2) % win_pages, half wave & 1
    const int tin = (wave & 1) << 5;
    const uint8_t *kw = p.kwin + (long long)blockIdx.x * p.win_pages * 4096;
    const uint8_t *vw = p.vwin + (long long)blockIdx.x * p.win_pages * 4096;
    const int pg_mask = p.win_pages - 1;
    Unit ring[RING];
    const int krow0 = stream_token_of_row(0, c16);
    const unsigned k_lane_off = ((unsigned)krow0 << 6) + 16u * q4;
    const unsigned v_lane_off = ((unsigned)(lane & 31) << 6) + 16u * (lane >> 5);
#define UNIT_REQ_K(SL, J)                                                                                          \
    {                                                                                                              \
        const int pg_ = ((J) * (NW / 2) + (wave >> 1)) & pg_mask;                                          \
        const gptr_u8 kb_ = uniform_ptr(kw + pg_ * 4096 + (tin << 6));                                             \
        _Pragma("unroll") for (int g2 = 0; g2 < 2; ++g2)                                                           \
            ring[SL].k[g2] = *(gptr_v4u)(kb_ + k_lane_off + ((4u * g2) << 6));                                     \
    }
#define UNIT_REQ_V(SL, J)                                                                                          \
    {                                                                                                              \
        const int pg_ = ((J) * (NW / 2) + (wave >> 1)) & pg_mask;                                          \
        const gptr_u8 vb_ = uniform_ptr(vw + pg_ * 4096 + tin);                                                    \
        ring[SL].v[0] = *(gptr_v4u)(vb_ + v_lane_off);                                                             \
        ring[SL].v[1] = *(gptr_v4u)(vb_ + v_lane_off + (32u << 6));                                                \
    }
#define UNIT_REQ(SL, J) { UNIT_REQ_K(SL, J) UNIT_REQ_V(SL, J) }
    UNIT_REQ(0, 0)
    UNIT_REQ(1, 1)
    UNIT_REQ(2, 2)
    if (RING == 4) UNIT_REQ(RING - 1, 3)

    const float inv_c = 1.0f / p.scale_log2e;
    SoftRef sr;
    sr.idle = (p.n_whole & 1) ? 0.f : (c16 < G ? 0.f : -INFINITY);      // odd n_whole: the round-3 behaviour (idle columns carry garbage probabilities)
    sr.set(-INFINITY, 0.f, inv_c);
    const unsigned kbase = (unsigned)q4 * (64u << CL2);
    const unsigned vconst0 = (unsigned)kVBase | ((unsigned)(lane & 31) << 2);
    const unsigned vconst1 = (unsigned)kVBase | ((unsigned)((lane & 31) + 32) << 2);
    unsigned a[2][4];
    float sc[8];
#define KG(SL, ST) st_kgather<CL2>(ring[SL], ST, kbase, a[(ST) & 1])
#define KM(ST) D[(ST) >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                              \
        as_v8f16(a[(ST) & 1][0], a[(ST) & 1][1], a[(ST) & 1][2], a[(ST) & 1][3]), qb[(ST) & 3], D[(ST) >> 2], 0, 0, 0)
#define SCORES_OUT() { _Pragma("unroll") for (int i = 0; i < 8; ++i) sc[i] = D[i >> 2][i & 3]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    if (VAR == 0) {
        v16f32 O[2][2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[n][kk][i] = 0.f;
        unsigned P[4];
        unsigned e[2][8];
#define VST(I) ((((I) & 1) << 1) | ((I) >> 1))
#define VG(SL, I) st_vgather(ring[SL], VST(I), vconst0, vconst1, e[(I) & 1])
#define VS(I)                                                                                                      \
    {                                                                                                              \
        if ((I) == NV / 2) value_next_step(P);                                                                     \
        st_vstep(e[(I) & 1], P, VST(I), O);                                                                        \
    }
#define BLOCK(U4, J)                                                                                               \
    {                                                                                                              \
        v4f32 D[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};                                                 \
        UNIT_REQ_K(U4, (J) + 4)                                                                                    \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                                           \
            VS(i)                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            if (i + 1 < NV) VG(U4, i + 1); else VG(((U4) + 1) & 3, 0);                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            _Pragma("unroll") for (int k = 0; k < SPV; ++k) {                                                      \
                KM(SPV * i + k);                                                                                   \
                if (SPV * i + k + 2 < 8) KG(((U4) + 1) & 3, SPV * i + k + 2);                                      \
                else KG(((U4) + 2) & 3, SPV * i + k + 2 - 8);                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }                                                                                                      \
        }                                                                                                          \
        UNIT_REQ_V(U4, (J) + 4)                                                                                    \
        SCORES_OUT()                                                                                               \
        softmax_online_raw<8>(sc, p.scale_log2e, inv_c, sr, O, G, lane);                                           \
        value_prep(sc, P);                                                                                         \
    }
        {
            v4f32 D[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            KG(0, 0);
            KG(0, 1);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                KM(st);
                if (st + 2 < 8) KG(0, st + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            SCORES_OUT()
        }
        softmax_online_raw<8>(sc, p.scale_log2e, inv_c, sr, O, G, lane);
        value_prep(sc, P);
        VG(0, 0);
        KG(1, 0);
        KG(1, 1);
        int j = 0;
        for (int w = 0; w < p.n_whole; ++w) {
            BLOCK(0, j)
            ++j;
            BLOCK(1, j)
            ++j;
            BLOCK(2, j)
            ++j;
            BLOCK(3, j)
            ++j;
        }
#undef BLOCK
#undef VS
#undef VG
#undef VST
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { p.cyc[blockIdx.x * 16 + wave] = t1 - t0; p.cyc[4096 + blockIdx.x * 16 + wave] = r1 - r0; }
        float *o = p.out + (((long long)blockIdx.x * NW + wave) * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = O[0][0][i] + O[1][0][i] + O[0][1][i + 4] + O[1][1][i + 8] + e[0][i] + a[0][i]; }
        o[4] = sr.l; o[5] = sr.m; o[6] = __uint_as_float(P[0]); o[7] = sc[0];
    } else {
        // ---- parity-V: O[n] = one 32 x 32 tile per subspace half: rows (parity, head), cols subspaces ----
        // (VAR 2: the same with every MFMA replaced by ONE v_xor that keeps the operand registers alive: what do the MFMAs cost
        //  in time and in clock?)
        v16f32 O[2][1];      // parity-V: one tile per subspace half
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[n][0][i] = 0.f;
        unsigned sel_lo, sel_hi;
        par_selectors(lane, sel_lo, sel_hi);
        ParA pa;
#undef KM
#define KM(ST)                                                                                                     \
    {                                                                                                              \
        if (VAR == 2) {                                                                                            \
            asm volatile("v_xor_b32 %0, %1, %2" : "+v"(D[(ST) >> 2][0]) : "v"(a[(ST) & 1][0]), "v"(a[(ST) & 1][1]), \
                         "v"(a[(ST) & 1][2]), "v"(a[(ST) & 1][3]));                                                \
        } else {                                                                                                   \
            D[(ST) >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_v8f16(a[(ST) & 1][0], a[(ST) & 1][1], a[(ST) & 1][2], a[(ST) & 1][3]), \
                                                                  qb[(ST) & 3], D[(ST) >> 2], 0, 0, 0);            \
        }                                                                                                          \
    }
        unsigned e[4][4];
        // value steps i = 0..7: token step s = i >> 1, subspace half n = i & 1
#define VG(SL, I) v_gather_par(ring[SL].v, (I) >> 1, (I) & 1, vconst0, vconst1, e[(I) & 3])
#define VS(I)                                                                                                      \
    {                                                                                                              \
        if (((I) & 1) == 0) Acur = value_A_par(pa, (I) >> 1, sel_lo, sel_hi);                                      \
        if (VAR == 2) {                                                                                            \
            asm volatile("v_xor_b32 %0, %1, %2" : "+v"(O[(I) & 1][0][0]) : "v"(e[(I) & 3][0]), "v"(e[(I) & 3][1]), "v"(e[(I) & 3][2]), \
                         "v"(e[(I) & 3][3]), "v"(Acur));                                                           \
        } else                                                                                                     \
        O[(I) & 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Acur, as_v8f16(e[(I) & 3][0], e[(I) & 3][1], e[(I) & 3][2], e[(I) & 3][3]), \
                                                            O[(I) & 1][0], 0, 0, 0);                                  \
    }
#define BLOCK(U4, J)                                                                                               \
    {                                                                                                              \
        v4f32 D[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};                                                 \
        v8f16 Acur;                                                                                                \
        UNIT_REQ_K(U4, (J) + RING)                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                            \
            VS(i)                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            if (i + 2 < 8) VG(U4, i + 2); else VG(((U4) + 1) % RING, i + 2 - 8);                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            KM(i);                                                                                                 \
            if (i + 2 < 8) KG(((U4) + 1) % RING, i + 2);                                                           \
            else KG(((U4) + 2) % RING, i + 2 - 8);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        }                                                                                                          \
        UNIT_REQ_V(U4, (J) + RING)                                                                                 \
        SCORES_OUT()                                                                                               \
        softmax_online_raw<8, true>(sc, p.scale_log2e, inv_c, sr, O, G, lane);                                          \
        value_prep_par(sc, pa);                                                                                    \
    }
        {
            v4f32 D[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            KG(0, 0);
            KG(0, 1);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                KM(st);
                if (st + 2 < 8) KG(0, st + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            SCORES_OUT()
        }
        softmax_online_raw<8, true>(sc, p.scale_log2e, inv_c, sr, O, G, lane);
        value_prep_par(sc, pa);
        VG(0, 0);
        VG(0, 1);
        KG(1, 0);
        KG(1, 1);
        int j = 0;
        for (int w = 0; w < p.n_whole; ++w) {
            BLOCK(0, j)
            ++j;
            BLOCK(1, j)
            ++j;
            BLOCK(2, j)
            ++j;
            if (RING == 4) {
                BLOCK(RING - 1, j)
                ++j;
            }
        }
#undef BLOCK
#undef VS
#undef VG
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { p.cyc[blockIdx.x * 16 + wave] = t1 - t0; p.cyc[4096 + blockIdx.x * 16 + wave] = r1 - r0; }
        float *o = p.out + (((long long)blockIdx.x * NW + wave) * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = O[0][0][i] + O[1][0][i] + O[0][0][i + 8] + O[1][0][i + 8] + e[0][i] + a[0][i]; }
        o[4] = sr.l; o[5] = sr.m; o[6] = __uint_as_float(pa.E[0]); o[7] = sc[0];
    }
}

template <int VAR, int NW, int RING = 4>
static void run(const char *name, MP p, int n_whole) {
    p.n_whole = n_whole;
    const int lds = 2 * 64 * 1024 + 4096;
    CK(hipFuncSetAttribute((const void *)core_kernel<VAR, NW, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((core_kernel<VAR, NW, RING>), dim3(256), dim3(NW * 64), lds, 0, p);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((core_kernel<VAR, NW, RING>), dim3(256), dim3(NW * 64), lds, 0, p);
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
        clk.push_back((double)h[b * 16] / (double)h[4096 + b * 16] * 0.1);      // shader cycles per 10 ns tick -> GHz
    }
    std::sort(per.begin(), per.end());
    std::sort(clk.begin(), clk.end());
    const double units_wave = (double)RING * n_whole;
    const double us = ms * 1e3 / reps;
    const double units = 256.0 * NW * units_wave;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)core_kernel<VAR, NW, RING>));
    printf("%-28s ring %d  %2d waves/CU  VGPRs %3d  %8.1f us/launch  %6.3f units/us/SIMD  %7.0f wave-cycles/unit (median WG)  = %.2f TB/s of codes chip-wide  in-kernel clock %.2f GHz\n",
           name, RING, NW, fa.numRegs, us, units / us / 1024.0, per[128] / units_wave, units * 4096.0 / us * 1e-6, clk[128]);
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
    for (auto &x : hq) x = (f16)((rand() % 2001 - 1000) * 0.001f);
    f16 *dtab, *dq;
    CK(hipMalloc(&dtab, tab.size() * 2)); CK(hipMalloc(&dq, hq.size() * 2));
    CK(hipMemcpy(dtab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    float *dout;
    unsigned long long *dcyc;
    CK(hipMalloc(&dout, 256ull * 16 * 64 * 8 * 4));
    CK(hipMalloc(&dcyc, 2 * 256 * 16 * 8));
    p.kwin = dk; p.vwin = dv; p.ktab = (const v4u *)dtab; p.vtab = (const v4u *)(dtab + 32768); p.q = dq; p.out = dout; p.cyc = dcyc;
    p.scale_log2e = 1.4426950408889634f / sqrtf(128.f);
    printf("core micro: 256 workgroups (1 per CU), G = 4, codes L2-resident, both codebooks in LDS; a unit = 32 tokens x 128 code bytes\n");
    for (int hbm = 0; hbm < 2; ++hbm) {
        p.win_pages = hbm ? kWinPagesMax : 8;
        printf("-- codes %s\n", hbm ? "streamed from HBM (8 MiB per workgroup per launch, 2 GiB in rotation)" : "L2-resident (64 KiB window per workgroup)");
        run<0, 8>("shipped core", p, 64);
        run<1, 8>("parity-V core", p, 64);
        run<1, 8, 3>("parity-V core", p, 86);
        run<1, 12, 3>("parity-V core", p, 58);
        run<1, 12, 4>("parity-V core", p, 44);
        run<2, 8>("parity-V, MFMA -> 1 VALU", p, 64);
        run<1, 8>("parity-V, idle rows NOT zero", p, 63);
        run<0, 8>("shipped, idle rows NOT zero", p, 63);
    }
    // every code byte the same value: K gathers of a wave instruction hit one address (broadcast, no bank conflict) - what a
    // conflict-free K layout could buy at most
    p.win_pages = 8;
    CK(hipMemset(dk, 0x5a, win));
    printf("-- K codes all equal, L2-resident\n");
    run<0, 8>("shipped, equal K codes", p, 64);
    run<1, 12, 3>("parity-V, equal K codes", p, 58);
    return 0;
}
