// Developer microbenchmark (round 3): does the cost of a v_mfma_f32_16x16x32_f16 to its co-resident VALU work
// depend on WHICH VALU instructions run beside it?  Three waves per SIMD, every wave: 256 instructions of one
// class per iteration, with and without a burst of 24 matrix instructions (results consumed by the VALU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CLS, int NM>
__global__ __launch_bounds__(768) void k(float *out, int iters)
{
    __shared__ float lds[768 * 2];
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
    lds[threadIdx.x] = x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (CLS == 0)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(x), "v"(0.25f));
                else if (CLS == 1)
                    asm volatile("v_fmac_f32 %0, 0x3e99999a, %1" : "+v"(v[c]) : "v"(x)); // literal constant
                else if (CLS == 2)
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(x), "v"(0x05040100u));
                else if (CLS == 3)
                    asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v[c]));
                else if (CLS == 4)
                    asm volatile("v_rndne_f32 %0, %0" : "+v"(v[c]));
                else if (CLS == 5)
                    asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(x), "v"(0.25f));
                else if (CLS == 6)
                    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(x) : "vcc");
                else if (CLS == 7)
                    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[c]) : "v"(x));
                else if (CLS == 8)
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(x));
            }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m)
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m % 6], b[m & 3], acc[m & 7], 0, 0, 0);
        if (NM)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                v[c] += acc[c][0] * 1e-30f;
    }
    float s = lds[(threadIdx.x + 1) % 768];
    for (int c = 0; c < 8; ++c)
        s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CLS, int NM>
static float run(float *d, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<CLS, NM><<<256, 768>>>(d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<CLS, NM><<<256, 768>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

template <int CLS>
static void line(float *d, const char *name, int per_iter)
{
    const int iters = 2000;
    const float t0 = run<CLS, 0>(d, iters), t1 = run<CLS, 24>(d, iters);
    printf("%-22s alone %7.1f us = %.2f cycles per instruction and SIMD; + 24 mfma %7.1f us -> %.1f cycles per mfma and SIMD\n", name, t0,
           t0 * 2400.0 / (iters * (double)per_iter * 3), t1, (t1 - t0) * 2400.0 / (iters * 24.0 * 3));
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 768 * sizeof(float));
    line<0>(d, "v_fma_f32 (3 vgpr)", 256);
    line<8>(d, "v_add_f32", 256);
    line<1>(d, "v_fmac_f32 literal", 256);
    line<2>(d, "v_perm_b32", 256);
    line<3>(d, "v_cvt_f32_ubyte1", 256);
    line<4>(d, "v_rndne_f32", 256);
    line<5>(d, "v_min3_f32", 256);
    line<6>(d, "v_cmp + v_cndmask", 512);
    line<7>(d, "v_pk_add_u16", 256);
    return 0;
}
