// Exhaustive check (all 2^32 bit patterns) of candidate fast reciprocals against IEEE 1.0f / x on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/rcp_probe tools/rcp_probe.hip   (built here, run under gpurun)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ float rcp_a(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_b(float x)
{
    float r = rcp_a(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__global__ void probe(unsigned long long *out, unsigned *first)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad_a = 0, bad_b = 0, in_range = 0, bad_raw = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const unsigned bits = (unsigned)i;
        const unsigned ex = (bits >> 23) & 0xff;
        if (ex < 1 || ex > 252) continue; // normal x with a normal reciprocal
        const float x = __uint_as_float(bits);
        const float ref = 1.0f / x;
        const float a = rcp_a(x), b = rcp_b(x), raw = __builtin_amdgcn_rcpf(x);
        in_range++;
        if (__float_as_uint(raw) != __float_as_uint(ref)) bad_raw++;
        if (__float_as_uint(a) != __float_as_uint(ref)) {
            bad_a++;
            if (atomicAdd(first, 1u) < 8) printf("A differs: x=%08x ref=%08x a=%08x\n", bits, __float_as_uint(ref), __float_as_uint(a));
        }
        if (__float_as_uint(b) != __float_as_uint(ref)) bad_b++;
    }
    atomicAdd(out + 0, in_range);
    atomicAdd(out + 1, bad_a);
    atomicAdd(out + 2, bad_b);
    atomicAdd(out + 3, bad_raw);
}
int main()
{
    unsigned long long *d, h[4];
    unsigned *f;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&f, 4);
    hipMemset(d, 0, sizeof(h));
    hipMemset(f, 0, 4);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d, f);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("inputs in range %llu  mismatches: rcp+1 Newton step %llu, rcp+2 steps %llu, bare v_rcp_f32 %llu\n", h[0], h[1], h[2], h[3]);
    return 0;
}
