// What the shader clock is while a kernel runs, and what a vector instruction costs: s_memtime (shader clock ticks)
// against s_memrealtime (100 MHz) over loops of v_fma_f32 / v_perm_b32, dependent or in four independent chains, with
// one or two waves per SIMD, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O2 -o build/micro/shader_clock tools/micro/shader_clock.hip && build/micro/shader_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void probe(unsigned long long *o, float *sink, int iters) {
    float x0 = threadIdx.x * 1e-9f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, y = 1.000001f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, s = 0x03020400u + blockIdx.x;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) { x0 = __builtin_fmaf(x0, y, 1e-7f); x0 = __builtin_fmaf(x0, y, 1e-7f); x0 = __builtin_fmaf(x0, y, 1e-7f); x0 = __builtin_fmaf(x0, y, 1e-7f); }
            if (KIND == 1) { x0 = __builtin_fmaf(x0, y, 1e-7f); x1 = __builtin_fmaf(x1, y, 1e-7f); x2 = __builtin_fmaf(x2, y, 1e-7f); x3 = __builtin_fmaf(x3, y, 1e-7f); }
            if (KIND == 2) { u0 = __builtin_amdgcn_perm(u0, s, u0); u0 = __builtin_amdgcn_perm(u0, s, u0); u0 = __builtin_amdgcn_perm(u0, s, u0); u0 = __builtin_amdgcn_perm(u0, s, u0); }
            if (KIND == 3) { u0 = __builtin_amdgcn_perm(u0, s, u0); u1 = __builtin_amdgcn_perm(u1, s, u1); u2 = __builtin_amdgcn_perm(u2, s, u2); u3 = __builtin_amdgcn_perm(u3, s, u3); }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[2 * blockIdx.x] = c1 - c0; o[2 * blockIdx.x + 1] = r1 - r0; }
    if (x0 + x1 + x2 + x3 == 12345.f || (u0 ^ u1 ^ u2 ^ u3) == 0x12345u) sink[0] = x0;
}
int main() {
    const int nb = 256, iters = 4096;
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, nb * 16); (void)hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(2 * nb);
    const char *names[4] = {"v_fma_f32, one dependent chain", "v_fma_f32, four independent chains", "v_perm_b32, one dependent chain", "v_perm_b32, four independent chains"};
    for (int kind = 0; kind < 4; ++kind)
        for (int waves = 4; waves <= 8; waves *= 2) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(64 * waves), 0, 0, d, sink, iters);
                if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(64 * waves), 0, 0, d, sink, iters);
                if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(nb), dim3(64 * waves), 0, 0, d, sink, iters);
                if (kind == 3) hipLaunchKernelGGL(probe<3>, dim3(nb), dim3(64 * waves), 0, 0, d, sink, iters);
                (void)hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
            }
            double c = 0, r = 0;
            for (int i = 0; i < nb; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
            c /= nb; r /= nb;
            printf("%-38s %d wave(s) per SIMD: %.3f ticks/ns; %.2f ticks per instruction of a wave -> %.2f per instruction and SIMD\n",
                   names[kind], waves / 4, c / (r * 10.0), c / (iters * 64.0), c / (iters * 64.0) / (waves / 4));
        }
    return 0;
}
