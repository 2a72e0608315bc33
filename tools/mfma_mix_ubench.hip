// Developer microbenchmark (round 3): incremental cost of v_mfma_f32_16x16x32_f16 (and 32x32x16)
// inside waves that are otherwise VALU-bound, three waves per SIMD all running the SAME mix
// (bursts of NM matrix instructions with distinct operand registers between blocks of 256 fma),
// the situation of the fused encoder with its row pass on the matrix pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NM, int SHAPE>
__global__ __launch_bounds__(768) void k(float *out, int iters)
{
    float x = threadIdx.x * 1e-3f;
    float v[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    f16x8 a[6], b[4];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 8; ++j)
            a[i][j] = (_Float16)(i + j + (threadIdx.x & 3));
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j)
            b[i][j] = (_Float16)(i * j + (threadIdx.x & 7));
    f32x4 acc[8];
    f32x16 big[2];
    for (int i = 0; i < 8; ++i)
        acc[i] = f32x4{x, x, x, x};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 16; ++j)
            big[i][j] = x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(x), "v"(0.25f));
        if (SHAPE == 0) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
                acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m % 6], b[m & 3], acc[m & 7], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < NM; ++m)
                big[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % 6], b[m & 3], big[m & 1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < 8; ++c)
        s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + v[c];
    for (int j = 0; j < 16; ++j)
        s += big[0][j] + big[1][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int SHAPE>
static float run(float *d, int iters, int threads)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<NM, SHAPE><<<256, threads>>>(d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NM, SHAPE><<<256, threads>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 768 * sizeof(float));
    const int iters = 2000; // 256 fma per iteration and wave
    for (int threads : {256, 768}) {
        const float base = run<0, 0>(d, iters, threads);
        printf("%d waves/SIMD: 256 fma per iteration alone: %.1f us (%.2f cycles per fma and SIMD at 2.4 GHz)\n", threads / 256, base,
               base * 2400.0 / (iters * 256.0 * (threads / 256)));
        const float t8 = run<8, 0>(d, iters, threads), t16 = run<16, 0>(d, iters, threads), t24 = run<24, 0>(d, iters, threads);
        printf("  + 8 / 16 / 24 mfma 16x16x32 per iteration: %.1f %.1f %.1f us -> %.1f %.1f %.1f cycles per mfma and SIMD\n", t8, t16, t24,
               (t8 - base) * 2400.0 / (iters * 8.0 * (threads / 256)), (t16 - base) * 2400.0 / (iters * 16.0 * (threads / 256)),
               (t24 - base) * 2400.0 / (iters * 24.0 * (threads / 256)));
        const float u4 = run<4, 1>(d, iters, threads), u8 = run<8, 1>(d, iters, threads);
        printf("  + 4 / 8 mfma 32x32x16 per iteration: %.1f %.1f us -> %.1f %.1f cycles per mfma and SIMD\n", u4, u8,
               (u4 - base) * 2400.0 / (iters * 4.0 * (threads / 256)), (u8 - base) * 2400.0 / (iters * 8.0 * (threads / 256)));
    }
    return 0;
}
