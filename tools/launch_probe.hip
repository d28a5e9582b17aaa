// Development probe: cost of a back-to-back dependent kernel launch vs dynamic LDS size / VGPR budget.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float smem[];
__global__ void __launch_bounds__(128) small_k(float* p, int n) { if (n < 0) p[threadIdx.x] = smem[threadIdx.x]; }
__global__ void __launch_bounds__(128) big_k(float* p, int n) {
    float a[200];
#pragma unroll
    for (int i = 0; i < 200; i++) a[i] = p[i];
    if (n < 0) { float s = 0; for (int i = 0; i < 200; i++) s += a[i] * a[(i * 7) % 200]; p[threadIdx.x] = s + smem[threadIdx.x]; }
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)small_k, hipFuncAttributeMaxDynamicSharedMemorySize, 126000);
    hipFuncSetAttribute((const void*)big_k, hipFuncAttributeMaxDynamicSharedMemorySize, 126000);
    for (int lds : {0, 32768, 126000}) for (int grid : {1, 64, 240}) {
        for (int which = 0; which < 2; which++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                for (int i = 0; i < 200; i++) {
                    if (which == 0) hipLaunchKernelGGL(small_k, dim3(grid), dim3(128), lds, 0, d, 1);
                    else hipLaunchKernelGGL(big_k, dim3(grid), dim3(128), lds, 0, d, 1);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) printf("%s lds=%6d grid=%3d: %.2f us per launch\n", which ? "big_vgpr " : "small    ", lds, grid, ms * 1000 / 200);
            }
        }
    }
    return 0;
}
