// Development probe: effective shader clock and single-wave issue rate on a nearly idle GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int n, unsigned long long* stamps) {
    float a = out[threadIdx.x], b = 1.0001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = a * b + 0.5f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
int main() {
    float* d; unsigned long long* s; unsigned long long h[2];
    hipMalloc(&d, 256 * 1024); hipMalloc(&s, 16); hipMemset(d, 0, 256 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 64, 256, 1024}) {
        for (int rep = 0; rep < 3; rep++) {
            int n = 20000;
            hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, n, s); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, s, 16, hipMemcpyDeviceToHost);
            double instr = (double)n * 64;
            printf("blocks=%4d: %.3f ms, %.2f ns/instr, memtime cycles/instr %.2f, clock %.0f MHz\n", blocks, ms, ms * 1e6 / instr,
                   (double)h[0] / instr, (double)h[0] / ((double)h[1] / 100.0));
        }
    }
    return 0;
}
