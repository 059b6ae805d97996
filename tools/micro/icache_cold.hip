// Microbenchmark: what does it cost a wave to run through code that is NOT in the instruction cache?
//
// The decode-attention launch at one request executes ~25 KB of straight-line code exactly once per wave (front, prologue,
// three unrolled blocks, tail), and every launch starts with a cold instruction cache.  Round 3 saw the split merge (1.5 KB of
// code, once per workgroup) take 1.25 us cold and 0.8 us after a dry run.  This micro prices it directly: the SAME number of
// the same instructions, once as N KB of straight-line code (.rept), once as a loop over 1 KB of it; one or eight waves per
// CU; in-kernel time by s_memrealtime around the sequence (100 MHz), 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o icache_cold icache_cold.hip && ./icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// 256 instructions of 4 bytes = 1 KB: independent v_add chains on 8 registers (no dependency stalls, one issue per 4+ cycles)
#define KB1                                                                                             \
    ".rept 32\n"                                                                                        \
    "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"       \
    "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"       \
    ".endr\n"
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one)

template <int KB, bool LOOP>
__global__ void k(unsigned long long *stamps, unsigned *sink) {
    unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, one = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (LOOP) {
        for (int i = 0; i < KB; ++i) asm volatile(KB1 : OPS);
    } else {
        // KB copies of the 1-KB sequence, straight-line
        if (KB >= 1) asm volatile(KB1 : OPS);
        if (KB >= 2) asm volatile(KB1 : OPS);
        if (KB >= 4) { asm volatile(KB1 : OPS); asm volatile(KB1 : OPS); }
        if (KB >= 8) { asm volatile(KB1 KB1 KB1 KB1 : OPS); }
        if (KB >= 16) { asm volatile(KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 : OPS); }
        if (KB >= 32) { asm volatile(KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 KB1 : OPS); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345u) sink[0] = a0;
}

template <int KB, bool LOOP>
static void run(int threads, unsigned long long *stamps, unsigned *sink) {
    const int waves = 256 * threads / 64;
    std::vector<unsigned long long> h(waves);
    std::vector<double> med;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<KB, LOOP>), dim3(256), dim3(threads), 0, 0, stamps, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h.data(), stamps, waves * 8, hipMemcpyDeviceToHost));
        double s = 0, mx = 0;
        for (auto v : h) { s += v; mx = std::max(mx, (double)v); }
        if (rep >= 2) med.push_back(s / waves / 100.0);
        if (rep == 5)
            printf("%2d KB %-13s %d wave(s)/CU: in-kernel %6.2f us mean %6.2f us max per wave | %6.2f us per launch (20 back to back)\n", KB,
                   LOOP ? "loop over 1KB" : "straight-line", threads / 64, s / waves / 100.0, mx / 100.0, ms * 1e3 / 20);
    }
}

int main() {
    unsigned long long *stamps;
    unsigned *sink;
    CK(hipMalloc(&stamps, 256 * 8 * 8));
    CK(hipMalloc(&sink, 64));
    printf("same instruction count, straight-line (cold instruction cache every launch) vs a loop over 1 KB (warm after the first pass)\n");
    for (int threads : {64, 512}) {
        run<4, false>(threads, stamps, sink);
        run<4, true>(threads, stamps, sink);
        run<16, false>(threads, stamps, sink);
        run<16, true>(threads, stamps, sink);
        run<32, false>(threads, stamps, sink);
        run<32, true>(threads, stamps, sink);
    }
    return 0;
}
