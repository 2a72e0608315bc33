// Developer microbenchmark (round 3): why does a matrix instruction cost ~50 cycles inside the fused encoder
// when it costs 8-9 in tools/mfma_mix_ubench.hip?  Same mix (three waves per SIMD, 256 fma + a burst of 24
// v_mfma_f32_16x16x32_f16 per iteration), plus the kernel's two dependencies, switchable:
//   DEP_IN   the B operands are produced by VALU instructions (v_perm) right before the burst
//   DEP_OUT  VALU instructions consume the accumulators right after the burst
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NM, bool DEP_IN, bool DEP_OUT>
__global__ __launch_bounds__(768) void k(float *out, int iters)
{
    float x = threadIdx.x * 1e-3f;
    float v[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    f16x8 a[6];
    u32x4 braw[4];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 8; ++j)
            a[i][j] = (_Float16)(i + j + (threadIdx.x & 3));
    for (int i = 0; i < 4; ++i)
        braw[i] = u32x4{threadIdx.x * 3u + i, threadIdx.x * 5u + i, threadIdx.x * 7u + i, threadIdx.x * 11u + i};
    f32x4 acc[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(x), "v"(0.25f));
        f16x8 b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 w = braw[i];
            if (DEP_IN) {
                const uint32_t s = __builtin_bit_cast(uint32_t, v[i]);
                w = u32x4{__builtin_amdgcn_perm(0u, w.x ^ s, 0x0C010C00u), __builtin_amdgcn_perm(0u, w.y ^ s, 0x0C030C02u),
                          __builtin_amdgcn_perm(0u, w.z ^ s, 0x0C010C00u), __builtin_amdgcn_perm(0u, w.w ^ s, 0x0C030C02u)};
            }
            b[i] = __builtin_bit_cast(f16x8, w);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m)
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m % 6], b[m & 3], acc[m & 7], 0, 0, 0);
        if (DEP_OUT) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                v[c] += acc[c][0] * 1e-30f + acc[c][3] * 1e-30f;
        } else if (NM) {
            if (it == iters - 1)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    v[c] += acc[c][0] + acc[c][3];
        }
    }
    float s = 0;
    for (int c = 0; c < 8; ++c)
        s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, bool DI, bool DO>
static float run(float *d, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<NM, DI, DO><<<256, 768>>>(d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NM, DI, DO><<<256, 768>>>(d, iters);
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
    const int iters = 2000;
    const float base = run<0, false, false>(d, iters), base_in = run<0, true, false>(d, iters);
    printf("3 waves/SIMD, 256 fma per iteration: %.1f us (with the 16 perms: %.1f us)\n", base, base_in);
    auto per = [&](float t, float b) { return (t - b) * 2400.0 / (iters * 24.0 * 3); };
    const float t00 = run<24, false, false>(d, iters), t10 = run<24, true, false>(d, iters);
    const float t01 = run<24, false, true>(d, iters), t11 = run<24, true, true>(d, iters);
    printf("+ 24 mfma, no dependencies      : %.1f us -> %.1f cycles per mfma and SIMD\n", t00, per(t00, base));
    printf("+ 24 mfma, operands from VALU   : %.1f us -> %.1f\n", t10, per(t10, base_in));
    printf("+ 24 mfma, results into VALU    : %.1f us -> %.1f\n", t01, per(t01, base));
    printf("+ 24 mfma, both                 : %.1f us -> %.1f\n", t11, per(t11, base_in));
    return 0;
}
