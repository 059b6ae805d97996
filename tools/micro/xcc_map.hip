// Which XCD does workgroup i of a 2-D grid land on?  (s_getreg HW_REG_XCC_ID.)  Development probe behind the
// "all splits of one (b, kv head) on one XCD" mapping of the fused attention kernel.
//   hipcc --offload-arch=gfx950 -O2 -o build/micro/xcc_map tools/micro/xcc_map.hip && build/micro/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *o) {
    if (threadIdx.x == 0) o[blockIdx.y * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_getreg(6164) & 15;   // hwreg(HW_REG_XCC_ID, 0, 4)
    __builtin_amdgcn_s_sleep(64);
}
int main() {
    const int gx = 32, gy = 8;
    int *d;
    hipMalloc(&d, gx * gy * sizeof(int));
    std::vector<int> h(gx * gy);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(512), 64 * 1024, 0, d);
        hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < gx * gy; ++i) bad += h[i] != (i % 8);
        printf("launch %d: workgroups whose XCC_ID != linear id %% 8: %d of %d; first 16:", rep, bad, gx * gy);
        for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
        printf("\n");
    }
    return 0;
}
