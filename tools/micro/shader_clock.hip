// What the shader clock is while a kernel runs: s_memtime (shader clock ticks) against s_memrealtime (100 MHz) over a busy
// loop, one workgroup per CU, and over a loop of dependent v_fma (4 cycles each on a 16-lane SIMD).
//   hipcc --offload-arch=gfx950 -O2 -o build/micro/shader_clock tools/micro/shader_clock.hip && build/micro/shader_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long *o, float *sink, int iters) {
    float x = threadIdx.x * 1e-9f, y = 1.000001f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, y, 1e-7f);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[2 * blockIdx.x] = c1 - c0; o[2 * blockIdx.x + 1] = r1 - r0; }
    if (x == 12345.f) sink[0] = x;
}
int main() {
    const int nb = 256, iters = 4096;
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, nb * 16); (void)hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(2 * nb);
    for (int waves = 1; waves <= 8; waves *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(nb), dim3(64 * waves), 0, 0, d, sink, iters);
            (void)hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
        }
        double c = 0, r = 0;
        for (int i = 0; i < nb; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
        c /= nb; r /= nb;
        printf("%d wave(s)/workgroup: %.0f s_memtime ticks in %.2f us -> %.3f ticks/ns; %.2f ticks per dependent v_fma (%d of them)\n",
               waves, c, r / 100.0, c / (r * 10.0), c / (iters * 64.0), iters * 64);
    }
    return 0;
}
