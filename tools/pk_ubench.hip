// Developer microbenchmark: does v_pk_fma_f32 buy anything at THREE waves per SIMD (the fused
// kernels' occupancy), where a wave issues at most one VALU instruction per 4 cycles?
// 16 independent fma chains per lane as 16 v_fma_f32 or as 8 v_pk_fma_f32 per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_plain(float *out, int iters, float s)
{
    extern __shared__ char pad[];
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(s));
    }
    float t = 0;
    for (int i = 0; i < 16; ++i) t += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t + pad[0];
}
__global__ __launch_bounds__(256) void k_packed(float *out, int iters, float s)
{
    extern __shared__ char pad[];
    f2 x[8], ss = {s, s};
    for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(ss), "v"(ss));
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = t + pad[0];
}
template <class K> float run(K k, int wgs, size_t lds)
{
    float *d; (void)hipMalloc(&d, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, d, 10, 1.0001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, d, 20000, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}
int main()
{
    (void)hipFuncSetAttribute((const void *)k_plain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_packed, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int wpc[] = {1, 2, 3, 4, 8};
    for (int w : wpc) { // w workgroups (of 4 waves) per CU = w waves per SIMD, forced by the LDS each takes
        const size_t lds = w == 8 ? 16 * 1024 : (size_t)(160 / w - 1) * 1024;
        const float a = run(k_plain, 256 * w, lds), b = run(k_packed, 256 * w, lds);
        printf("%d waves/SIMD: 16 v_fma_f32 per step %.3f ms, 8 v_pk_fma_f32 per step %.3f ms (ratio %.2f); per wave-instruction %.2f / %.2f cycles at 2.4 GHz\n",
               w, a, b, a / b, a * 1e-3 * 2.4e9 / (20000.0 * 64 * w), b * 1e-3 * 2.4e9 / (20000.0 * 32 * w));
    }
    return 0;
}
