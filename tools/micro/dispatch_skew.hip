// Microbenchmark: how long does the dispatcher take to start 256 workgroups of 512 threads, as a function of the
// registers per lane and the LDS per workgroup the kernel declares?  (The fused decode-attention kernel declares
// ~234 VGPRs and 138 KiB: its workgroups start 1.6-1.8 us apart, first to last.)
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_skew dispatch_skew.hip && ./dispatch_skew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Big { unsigned long long *stamps; long long pad[40]; };      // ~330-byte kernarg block like AttnParams

template <int VG>
__global__ __launch_bounds__(512, 2) void probe(Big p) {
    extern __shared__ char lds[];
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if (VG >= 64) asm volatile("v_mov_b32 v60, 0" ::: "v60");
    if (VG >= 128) asm volatile("v_mov_b32 v124, 0" ::: "v124");
    if (VG >= 232) asm volatile("v_mov_b32 v230, 0" ::: "v230");
    if (threadIdx.x == 0) {
        p.stamps[blockIdx.x * 2] = t;
        p.stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (p.pad[3] == 12345) lds[threadIdx.x] = 1;
}

template <int VG>
void run(int lds_kb, unsigned long long *st, int wgs) {
    auto k = probe<VG>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    Big b{};
    b.stamps = st;
    std::vector<double> skews, b2b;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipMemset(st, 0, wgs * 16));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds_kb * 1024, 0, b);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(wgs * 2);
        CK(hipMemcpy(h.data(), st, wgs * 16, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int w = 0; w < wgs; ++w) { t0 = std::min(t0, h[2 * w]); t1 = std::max(t1, h[2 * w]); }
        if (rep >= 2) skews.push_back((t1 - t0) / 100.0);
    }
    hipEvent_t a, ev;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&ev));
    CK(hipEventRecord(a));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds_kb * 1024, 0, b);
    CK(hipEventRecord(ev));
    CK(hipEventSynchronize(ev));
    float ms; CK(hipEventElapsedTime(&ms, a, ev));
    std::sort(skews.begin(), skews.end());
    printf("vgprs>=%3d  lds %3d KiB  wgs %4d : start skew first->last  median %.2f us (min %.2f max %.2f) | back-to-back %.2f us/launch\n",
           VG, lds_kb, wgs, skews[skews.size() / 2], skews.front(), skews.back(), ms * 1e3 / 200);
}

int main() {
    unsigned long long *st;
    CK(hipMalloc(&st, 4096 * 16));
    for (int wgs : {256, 512}) {
        run<32>(0, st, wgs);
        run<32>(64, st, wgs);
        run<32>(138, st, wgs);
        run<128>(0, st, wgs);
        run<128>(138, st, wgs);
        run<232>(0, st, wgs);
        run<232>(138, st, wgs);
    }
    return 0;
}
