// Probe of v_permlane32_swap / v_permlane16_swap lane semantics on gfx950 (hipcc --offload-arch=gfx950 permlane.hip -o permlane && ./permlane)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"p32 ret0", "p32 ret1", "p16 ret0", "p16 ret1"};
    for (int q = 0; q < 4; ++q) { printf("%s:", nm[q]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[q * 64 + i]); printf("\n"); }
    return 0;
}
