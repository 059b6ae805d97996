// Microbenchmark: the fused decode-attention launch with everything that COMPUTES taken out.
//
// What is left is the launch's dependency structure at BASELINE configs[2] (one request, 8 kv heads x 32 splits = 256
// workgroups of 512 threads, 128 KiB of codes + 128 KiB of codebooks per workgroup):
//   lengths (scalar load) -> page ids (one vector load per wave, lane = round) -> code pages (K rows 16 B per lane, V
//   transposed pages 16 B per lane at a 64-byte stride: two dependent loaded round trips), both codebooks from L2 into LDS
//   behind one barrier, one pass over the loaded bytes (xor: no gathers, no MFMA, no softmax), then the tail of
//   attn_mfma.hip: wave-merge barrier, the split's partial (4 heads x 128 floats) stored plain into the XCD's L2, drain,
//   barrier, flag; an arrival counter picks the 4 last workgroups of each (b, kv head), which poll the 32 flags (sc1), load
//   a quarter of every slot per wave and write one head's output each.
// It answers: how much of the 16.8 us launch is the chip moving these bytes through these dependencies, and how much is the
// kernel's own arithmetic (the rest).  Variants switch pieces off to price them.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kNS = 32, kBH = 8, kNT = 512, kSlotFloats = 544;      // 4 heads x 128 + 8 (m, l), padded to 128-byte lines

struct P {
    const int *lengths;          // {T, r, start, 0}
    const int *page_ids;         // (bh, n_pages_cap)
    const unsigned char *kpool;  // (pages, 64 tokens, 64 bytes)
    const unsigned char *vpool;  // (pages, 64 subspaces, 64 tokens)
    const u32x4 *tables;         // 128 KiB, L2-resident
    float *part;                 // (bh, ns, kSlotFloats)
    unsigned *flags;             // (bh, 64)
    unsigned *cnt;               // (bh, 32): arrival counter in word 0
    unsigned short *out;         // (bh, 4 heads, 128) fp16
    unsigned *sink;
    int n_pages_cap;
    unsigned launch_id;          // flags of this launch carry it: nothing to reset
};

// FLAGS: 1 page ids through memory (else computed), 2 codebooks -> LDS, 4 tail (publish + flags + merge), 8 code loads
template <int FLAGS>
__device__ __forceinline__ void floor_body(P p);
template <int FLAGS>
__global__ __launch_bounds__(kNT) void floor_kernel(P p) { floor_body<FLAGS>(p); }
// the same kernel with its arguments passed one by one (pointers first): compiled with
// -mllvm -amdgpu-kernarg-preload-count=16 the first 16 dwords arrive in SGPRs with the wave (no s_load round trip in front of
// the page-id load); a struct passed by value is not preloaded
template <int FLAGS>
__global__ __launch_bounds__(kNT) void floor_kernel_args(const int *lengths, const int *page_ids, const unsigned char *kpool,
                                                         const unsigned char *vpool, const u32x4 *tables, float *part, unsigned *flags,
                                                         unsigned *cnt, unsigned short *out, unsigned *sink, int n_pages_cap,
                                                         unsigned launch_id) {
    P p;
    p.lengths = lengths; p.page_ids = page_ids; p.kpool = kpool; p.vpool = vpool; p.tables = tables; p.part = part;
    p.flags = flags; p.cnt = cnt; p.out = out; p.sink = sink; p.n_pages_cap = n_pages_cap; p.launch_id = launch_id;
    floor_body<FLAGS>(p);
}
template <int FLAGS>
__device__ __forceinline__ void floor_body(P p) {
    extern __shared__ u32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int bh = id % kBH, split = id / kBH;      // workgroup i runs on XCD i % 8: all splits of a pair share an XCD
    // page ids of this wave's 4 units (round u reads page (4 u + wave / 2) * ns + split, half wave & 1 of it)
    int pid = 0;
    {
        const int page = (4 * (lane & 3) + (wave >> 1)) * kNS + split;
        if (FLAGS & 1) pid = p.page_ids[bh * p.n_pages_cap + page];
        else pid = bh * p.n_pages_cap + page;
    }
    u32x4 t[16];
    if (FLAGS & 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = p.tables[((i + id) & 15) * kNT + tid];
    }
    const int T = p.lengths[0];
    u32x4 c[16];
    if (FLAGS & 8) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long pg = __builtin_amdgcn_readlane(pid, u);
            const unsigned char *kp = p.kpool + (pg * 64 + 32 * (wave & 1)) * 64;             // 32 tokens x 64 bytes
            const unsigned char *vp = p.vpool + (pg * 64 + lane) * 64 + 32 * (wave & 1);      // lane = subspace, 32 tokens
            c[4 * u + 0] = *(const u32x4 *)(kp + 16 * lane);
            c[4 * u + 1] = *(const u32x4 *)(kp + 1024 + 16 * lane);
            c[4 * u + 2] = *(const u32x4 *)(vp);
            c[4 * u + 3] = *(const u32x4 *)(vp + 16);
        }
    }
    unsigned acc = (unsigned)T;
    if (FLAGS & 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[((i + id) & 15) * kNT + tid] = t[i];
        __syncthreads();
        acc += lds[(tid * 7) & 8191].x;
    }
    if (FLAGS & 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += c[i].x ^ c[i].y ^ c[i].z ^ c[i].w;
    }
    if (!(FLAGS & 4)) {
        if (acc == 0x12345678u) p.sink[id * kNT + tid] = acc;
        return;
    }
    // ---- tail ----
    __syncthreads();                                   // wave merge (its LDS traffic is not modelled)
    float *slot = p.part + ((long long)bh * kNS + split) * kSlotFloats;
    if (tid < kSlotFloats / 4) {
        const float f = __uint_as_float(acc & 0x3fffffffu);
        ((float4 *)slot)[tid] = make_float4(f, f, f, f);                     // plain: stays in this XCD's L2
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int idx = 0;
    if (tid == 0) {
        __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)(p.flags + bh * 64), 0, 256, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(p.launch_id, rf, split * 4, 0, 0);
        idx = __hip_atomic_fetch_add(p.cnt + bh * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ((volatile int *)lds)[0] = idx;
    }
    __syncthreads();
    idx = ((volatile int *)lds)[0];
    const int j = idx - (kNS - 4);
    if (j >= 0 && wave < 4) {                          // merger of head j: waves 0-3 take a quarter of the head each
        __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)(p.flags + bh * 64), 0, 256, 0x00020000);
        for (int spin = 0; spin < (1 << 20); ++spin) {
            const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(rf, (lane < kNS ? lane : 0) * 4, 0, 16 /* sc1 */);
            if (__all(f == p.launch_id)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.part + (long long)bh * kNS * kSlotFloats), 0, 0x7fffffff, 0x00020000);
        const int q8 = lane & 7, h = lane >> 3;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (((8 * k + h) * kSlotFloats) + j * 128 + 32 * wave + 4 * q8) * 4, 0, 16);
            s += __uint_as_float(v.x) + __uint_as_float(v.y) + __uint_as_float(v.z) + __uint_as_float(v.w);
        }
        for (int o = 8; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        if (lane < 8) p.out[(bh * 4 + j) * 128 + 32 * wave + 4 * q8] = (unsigned short)(int)s;
    }
    if (idx == kNS - 1 && tid == 0) __hip_atomic_store(p.cnt + bh * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int FLAGS, bool ARGS = false>
static double run(const char *name, P p, const unsigned char *kbase, const unsigned char *vbase, size_t pool_bytes, int iters) {
    auto k = floor_kernel<FLAGS>;
    auto ka = floor_kernel_args<FLAGS>;
    const size_t lds = 128 * 1024;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned launch = 1;
    auto go = [&](int i) {
        P q = p;
        q.kpool = kbase + (size_t)(i % 32) * pool_bytes;      // 32 "layers": 1 GB of codes in rotation beats the Infinity Cache
        q.vpool = vbase + (size_t)(i % 32) * pool_bytes;
        q.launch_id = launch++;
        if (ARGS) hipLaunchKernelGGL(ka, dim3(kNS * kBH), dim3(kNT), lds, 0, q.lengths, q.page_ids, q.kpool, q.vpool, q.tables, q.part, q.flags,
                                     q.cnt, q.out, q.sink, q.n_pages_cap, q.launch_id);
        else hipLaunchKernelGGL(k, dim3(kNS * kBH), dim3(kNT), lds, 0, q);
    };
    for (int i = 0; i < 16; ++i) go(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) go(i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters;
    printf("%-98s %6.2f us / launch\n", name, us);
    return us;
}

int main() {
    const int n_pages_cap = 512;                          // 32768 tokens / 64
    const size_t pool_bytes = (size_t)kBH * n_pages_cap * 4096;      // 16.8 MB per side and layer
    unsigned char *kpool, *vpool;
    CK(hipMalloc(&kpool, 32 * pool_bytes)); CK(hipMemset(kpool, 1, 32 * pool_bytes));
    CK(hipMalloc(&vpool, 32 * pool_bytes)); CK(hipMemset(vpool, 2, 32 * pool_bytes));
    P p = {};
    int *lengths, *ids;
    CK(hipMalloc(&lengths, 16));
    const int hl[4] = {32768, 100, 0, 0};
    CK(hipMemcpy(lengths, hl, 16, hipMemcpyHostToDevice));
    std::vector<int> hid(kBH * n_pages_cap);
    for (int i = 0; i < kBH * n_pages_cap; ++i) hid[i] = i;      // identity table: same pages as the computed form
    CK(hipMalloc(&ids, hid.size() * 4));
    CK(hipMemcpy(ids, hid.data(), hid.size() * 4, hipMemcpyHostToDevice));
    u32x4 *tables;
    CK(hipMalloc(&tables, 128 * 1024)); CK(hipMemset(tables, 3, 128 * 1024));
    CK(hipMalloc(&p.part, (size_t)kBH * kNS * kSlotFloats * 4)); CK(hipMemset(p.part, 0, (size_t)kBH * kNS * kSlotFloats * 4));
    CK(hipMalloc(&p.flags, kBH * 64 * 4)); CK(hipMemset(p.flags, 0, kBH * 64 * 4));
    CK(hipMalloc(&p.cnt, kBH * 32 * 4)); CK(hipMemset(p.cnt, 0, kBH * 32 * 4));
    CK(hipMalloc(&p.out, kBH * 4 * 128 * 2));
    CK(hipMalloc(&p.sink, (size_t)kNS * kBH * kNT * 4));
    p.lengths = lengths; p.page_ids = ids; p.tables = tables; p.n_pages_cap = n_pages_cap;
    const int it = 400;
    printf("256 workgroups x 512 threads, 33.5 MB of codes per launch out of 1 GB in rotation, 128 KiB of codebooks per workgroup\n");
    run<0>("launch alone (lengths load, nothing else)", p, kpool, vpool, pool_bytes, it);
    run<8>("+ codes, page ids computed (one loaded round trip)", p, kpool, vpool, pool_bytes, it);
    run<8 | 1>("+ page ids through memory (two dependent round trips)", p, kpool, vpool, pool_bytes, it);
    run<8 | 1 | 2>("+ both codebooks L2 -> LDS, one barrier", p, kpool, vpool, pool_bytes, it);
    run<8 | 1 | 2 | 4>("+ tail: partial -> L2, flags, arrival counter, 4 mergers per (b, kv head)   = the launch without arithmetic", p, kpool, vpool, pool_bytes, it);
    run<2 | 4>("codebooks + tail, no codes", p, kpool, vpool, pool_bytes, it);
    run<4>("tail alone", p, kpool, vpool, pool_bytes, it);
    // arguments one by one instead of one struct by value (see floor_kernel_args): preloaded into SGPRs when built with
    // -mllvm -amdgpu-kernarg-preload-count=16, plain s_load otherwise
    run<0, true>("args one by one: launch alone", p, kpool, vpool, pool_bytes, it);
    run<8 | 1, true>("args one by one: + codes, page ids through memory", p, kpool, vpool, pool_bytes, it);
    run<8 | 1 | 2 | 4, true>("args one by one: the launch without arithmetic", p, kpool, vpool, pool_bytes, it);
    run<8 | 1, false>("struct again: + codes, page ids through memory", p, kpool, vpool, pool_bytes, it);
    run<8 | 1 | 2 | 4, false>("struct again: the launch without arithmetic", p, kpool, vpool, pool_bytes, it);
    return 0;
}
