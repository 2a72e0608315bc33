// Developer microbenchmark (round 3): does a scalar instruction take a SIMD's issue slot like a vector one?
// Three waves per SIMD (256 workgroups of 768 threads); every wave runs 256 v_fmac_f32 per iteration with
// kS scalar instructions (s_or_b64 on dead registers, or a never-taken s_cbranch_scc1 after s_cmp) spread
// evenly between them.  If scalar and vector instructions issue independently the time stays that of the 256 fmac.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>

template <int kS, int kKind, int I>
__device__ __forceinline__ void step(float (&v)[8], float x)
{
    asm volatile("v_fmac_f32 %0, 0x3e99999a, %1" : "+v"(v[I & 7]) : "v"(x));
    if constexpr (kS > 0 && (I * kS) / 256 != ((I + 1) * kS) / 256) {
        if constexpr (kKind == 0)
            asm volatile("s_or_b64 s[20:21], s[22:23], s[24:25]" ::: "s20", "s21", "scc");
        else if constexpr (kKind == 1)
            asm volatile("s_nop 0");
        else
            asm volatile("s_and_saveexec_b64 s[20:21], -1\n\ts_or_b64 exec, exec, s[20:21]" ::: "s20", "s21", "scc");
    }
}
template <int kS, int kKind, int... I>
__device__ __forceinline__ void body(float (&v)[8], float x, std::integer_sequence<int, I...>)
{
    (step<kS, kKind, I>(v, x), ...);
}

template <int kS, int kKind>
__global__ __launch_bounds__(768) void k(float *out, int iters)
{
    float x = threadIdx.x * 1e-3f;
    float v[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    for (int it = 0; it < iters; ++it)
        body<kS, kKind>(v, x, std::make_integer_sequence<int, 256>{});
    float s = 0;
    for (int c = 0; c < 8; ++c)
        s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kS, int kKind>
static void run(float *d, const char *name)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<kS, kKind><<<256, 768>>>(d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<kS, kKind><<<256, 768>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int n_s = kKind == 2 ? 2 * kS : kS;
    printf("256 fmac + %3d %-22s: %8.1f us = %.2f cycles per fmac, %.2f per instruction of any kind (SIMD, 2.4 GHz)\n", n_s, name, ms * 1e3,
           ms * 1e3 * 2400.0 / (iters * 256.0 * 3), ms * 1e3 * 2400.0 / (iters * (256.0 + n_s) * 3));
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 768 * sizeof(float));
    run<0, 0>(d, "-");
    run<32, 0>(d, "s_or_b64");
    run<64, 0>(d, "s_or_b64");
    run<128, 0>(d, "s_or_b64");
    run<256, 0>(d, "s_or_b64");
    run<64, 1>(d, "s_nop 0");
    run<256, 1>(d, "s_nop 0");
    run<32, 2>(d, "saveexec + or exec");
    run<64, 2>(d, "saveexec + or exec");
    run<128, 2>(d, "saveexec + or exec");
    return 0;
}
