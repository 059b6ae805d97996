// Microbenchmark: what does "launch 256 workgroups, pull 128 KiB of codes each from HBM (+ optionally 128 KiB
// of L2-resident tables into LDS)" cost on this chip?  Floors for the fused decode-attention launch.
//   hipcc --offload-arch=gfx950 -O3 -o stream_floor stream_floor.hip && ./stream_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, int TABLE_KB, int NT, int VPAT = 0>
__global__ __launch_bounds__(NT) void stream_kernel(const u32x4* __restrict__ codes, const u32x4* __restrict__ table,
                                                    unsigned* __restrict__ out, size_t wg_stride_vec,
                                                    unsigned long long* stamps) {
    extern __shared__ u32x4 lds[];
    const int tid = threadIdx.x;
    if (stamps && tid == 0) stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    const u32x4* src = codes + (size_t)blockIdx.x * wg_stride_vec;
    u32x4 v[U];
    u32x4 t[TABLE_KB > 0 ? TABLE_KB * 1024 / 16 / NT : 1];
    constexpr int TV = TABLE_KB * 1024 / 16 / NT;
    if (TABLE_KB > 0) {
#pragma unroll
        for (int i = 0; i < TV; ++i) t[i] = table[((i + blockIdx.x) % TV) * NT + tid];
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        if (VPAT && (i & 1)) {   // V-page pattern: 16 B per lane at a 64 B stride (4 consecutive loads fill the lines)
            const int wave = tid >> 6, lane = tid & 63;
            v[i] = src[((size_t)(i >> 3) * (NT / 64) + wave) * 512 + (size_t)((i >> 1) & 3) + lane * 4 + 256 * 0];
        } else {
            v[i] = src[i * NT + tid];
        }
    }
    unsigned acc = 0;
    if (TABLE_KB > 0) {
#pragma unroll
        for (int i = 0; i < TV; ++i) lds[((i + blockIdx.x) % TV) * NT + tid] = t[i];
        __syncthreads();
        if (stamps && tid == 0) stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        acc = lds[(tid * 7) % (TV * NT)].x;
    }
#pragma unroll
    for (int i = 0; i < U; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u) out[blockIdx.x * NT + tid] = acc;
    __syncthreads();
    if (stamps && tid == 0) stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
}

template <int U, int TABLE_KB, int NT, int VPAT = 0>
void run(const char* name, const u32x4* codes, const u32x4* table, unsigned* out, int wgs, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    size_t stride = (size_t)U * NT;
    size_t lds = (size_t)TABLE_KB * 1024;
    auto k = stream_kernel<U, TABLE_KB, NT, VPAT>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nbuf = 32;                                  // rotate over 32 distinct code buffers (beats the MALL)
    size_t buf_vec = stride * wgs;
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(NT), lds, 0, codes + (i % nbuf) * buf_vec, table, out, stride, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(NT), lds, 0, codes + (i % nbuf) * buf_vec, table, out, stride, nullptr);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double us = ms * 1e3 / iters;
    double mb = (double)buf_vec * 16 / 1e6;
    static unsigned long long* st = nullptr;
    if (!st) CK(hipMalloc(&st, 4096 * 4 * 8));
    CK(hipMemset(st, 0, 4096 * 4 * 8));
    hipLaunchKernelGGL(k, dim3(wgs), dim3(NT), lds, 0, codes + 5 * buf_vec, table, out, stride, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(wgs * 4);
    CK(hipMemcpy(h.data(), st, wgs * 4 * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0, tmaxs = 0; double bar = 0, endm = 0;
    for (int w = 0; w < wgs; ++w) { if (h[w*4] < t0) t0 = h[w*4]; if (h[w*4] > tmaxs) tmaxs = h[w*4]; if (h[w*4+2] > t1) t1 = h[w*4+2]; }
    for (int w = 0; w < wgs; ++w) { bar += (double)(h[w*4+1] > h[w*4] ? h[w*4+1] - h[w*4] : 0) / wgs; endm += (double)(h[w*4+2] - h[w*4]) / wgs; }
    printf("%-44s wgs=%4d  %7.2f us/launch  %6.1f MB  %7.1f GB/s | in-kernel span %.2f us, start skew %.2f, mean WG: table barrier +%.2f, end +%.2f\n",
           name, wgs, us, mb, mb / us * 1e3, (t1 - t0) / 100.0, (tmaxs - t0) / 100.0, bar / 100.0, endm / 100.0);
}

int main() {
    size_t max_bytes = (size_t)32 * 128 * 1024 * 1024;       // 32 buffers x up to 128 MB
    u32x4 *codes, *table; unsigned* out;
    CK(hipMalloc(&codes, max_bytes)); CK(hipMemset(codes, 1, max_bytes));
    CK(hipMalloc(&table, 256 * 1024)); CK(hipMemset(table, 2, 256 * 1024));
    CK(hipMalloc(&out, 64 << 20));
    const int it = 200;
    run<1, 0, 512>("empty-ish: 8 KB/WG", codes, table, out, 256, it);
    run<16, 0, 512>("stream 128 KB/WG, 16 loads upfront", codes, table, out, 256, it);
    run<16, 64, 512>("stream 128 KB/WG + 64 KB table->LDS", codes, table, out, 256, it);
    run<16, 128, 512>("stream 128 KB/WG + 128 KB table->LDS", codes, table, out, 256, it);
    run<16, 128, 512, 1>("same, odd loads 16B @ 64B stride (V pages)", codes, table, out, 256, it);
    run<8, 128, 512>("stream 64 KB/WG + 128 KB table->LDS", codes, table, out, 512, it);
    run<16, 0, 512>("stream 128 KB/WG, 2 WG/CU", codes, table, out, 512, it);
    run<16, 0, 256>("stream 64 KB/WG (256 thr), 1024 WGs", codes, table, out, 1024, it);
    run<8, 0, 256>("stream 32 KB/WG (256 thr), 1024 WGs", codes, table, out, 1024, it);
    run<16, 0, 512>("stream 128 KB/WG, 1024 WGs (128 MB)", codes, table, out, 1024, it);
    run<16, 128, 512>("128 KB/WG + 128 KB table, 1024 WGs (128 MB)", codes, table, out, 1024, it);
    run<0 + 1, 128, 512>("table only 128 KB ->LDS", codes, table, out, 256, it);
    return 0;
}
