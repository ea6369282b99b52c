// Stream time per back-to-back launch of an EMPTY kernel as a function of grid size, block size, dynamic LDS and register footprint.
// hipcc --offload-arch=gfx950 -O3 tools/probe/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NV> __global__ __launch_bounds__(256) void empty_k(int* p, int flag) {
    extern __shared__ char smem[];
    if (flag) {          // never taken: keeps NV registers allocated
        float v[NV];
        for (int i = 0; i < NV; ++i) v[i] = p[i + threadIdx.x];
        for (int r = 0; r < 100; ++r) for (int i = 0; i < NV; ++i) v[i] = v[i] * v[(i + 1) % NV] + 1.f;
        float s = 0; for (int i = 0; i < NV; ++i) s += v[i];
        p[threadIdx.x] = (int)s + smem[threadIdx.x];
    }
}
template <int NV> __global__ __launch_bounds__(512) void empty_k512(int* p, int flag) {
    extern __shared__ char smem[];
    if (flag) {
        float v[NV];
        for (int i = 0; i < NV; ++i) v[i] = p[i + threadIdx.x];
        for (int r = 0; r < 100; ++r) for (int i = 0; i < NV; ++i) v[i] = v[i] * v[(i + 1) % NV] + 1.f;
        float s = 0; for (int i = 0; i < NV; ++i) s += v[i];
        p[threadIdx.x] = (int)s + smem[threadIdx.x];
    }
}
template <typename K> float run(K k, int grid, int block, int lds, int* d) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, d, 0);
    hipEventRecord(a, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, d, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 200.f;
}
int main() {
    int* d; hipMalloc(&d, 1 << 20);
    const int grids[] = {1, 64, 256, 480, 512, 1024, 2048};
    const int ldss[] = {0, 16 * 1024, 40 * 1024, 70 * 1024, 80 * 1024};
    printf("us per launch (200 back-to-back empty launches on the null stream)\n");
    for (int lds : ldss) for (int g : grids) {
        printf("block 256 lds %3dK grid %4d : small-reg %5.2f  big-reg %5.2f | block 512: %5.2f\n", lds / 1024, g,
               run(empty_k<8>, g, 256, lds, d), run(empty_k<200>, g, 256, lds, d), run(empty_k512<100>, g, 512, lds, d));
    }
    return 0;
}
