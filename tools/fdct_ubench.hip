// Developer microbenchmark: the transform stage alone (block_fdct_quant vs block_fdct_quant_pk),
// lane = block on register data, three waves per SIMD like the fused encoder.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "pk_fdct_experiment.h"
template <int kKind>
__global__ __launch_bounds__(256, 3) void k(const float *in, float *out, int iters, unsigned long long *clk)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ char pad[];
    __shared__ __attribute__((aligned(16))) float qtab[64];
    if (threadIdx.x < 64)
        qtab[threadIdx.x] = 0.01f / (1 + (threadIdx.x & 15));
    __syncthreads();
    float v[64], acc = 0;
    const PkBank bank = pk_bank();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i)
            v[i] = in[(threadIdx.x + i + it) & 1023];
        if (kKind == 0)
            block_fdct_quant(v, qtab);
        else
            block_fdct_quant_pk(v, qtab, bank);
#pragma unroll
        for (int i = 0; i < 64; ++i)
            acc += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + pad[0];
    if (blockIdx.x == 5 && threadIdx.x == 0)
        clk[0] = __builtin_amdgcn_s_memtime() - c0, clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
}
int main()
{
    float *in, *out;
    (void)hipMalloc(&in, 4096);
    (void)hipMemset(in, 0x3f, 4096);
    (void)hipMalloc(&out, 768 * 256 * 4);
    unsigned long long *clk, hclk[2];
    (void)hipMalloc(&clk, 16);
    for (int kind = 0; kind < 2; ++kind) {
        auto kern = kind ? k<1> : k<0>;
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(768), dim3(256), 52 * 1024, 0, in, out, 10, clk);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(768), dim3(256), 52 * 1024, 0, in, out, 2000, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms for 2000 blocks per lane = %.0f cycles per block per SIMD at 2.4 GHz (3 waves per SIMD)\n",
               kind ? "packed" : "scalar", ms, ms * 1e-3 * 2.4e9 / (2000.0 * 3));
        (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        printf("  one workgroup: %llu shader clocks in %llu ticks of the 100 MHz counter = %.0f MHz\n", hclk[0], hclk[1],
               hclk[0] / (hclk[1] / 100.0));
    }
    return 0;
}
