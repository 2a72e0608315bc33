// Developer microbenchmark: packed f32 instruction forms at three waves per SIMD -- cost per
// instruction for different operand kinds and dependency distances.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERN(NAME, NCH, ASM)                                                             \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float s, unsigned long long k) \
    {                                                                                    \
        extern __shared__ char pad[];                                                    \
        f2 x[NCH], ss = {s, s};                                                          \
        for (int i = 0; i < NCH; ++i) x[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i}; \
        for (int it = 0; it < iters; ++it) {                                             \
            _Pragma("unroll") for (int r = 0; r < 64 / NCH; ++r)                         \
            _Pragma("unroll") for (int i = 0; i < NCH; ++i)                              \
                asm volatile(ASM : "+v"(x[i]) : "v"(ss), "s"(k));                        \
        }                                                                                \
        float t = 0;                                                                     \
        for (int i = 0; i < NCH; ++i) t += x[i].x + x[i].y;                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = t + pad[0];                         \
    }
KERN(k_fma8, 8, "v_pk_fma_f32 %0, %0, %1, %1")
KERN(k_fma4, 4, "v_pk_fma_f32 %0, %0, %1, %1")
KERN(k_fma3, 3, "v_pk_fma_f32 %0, %0, %1, %1")
KERN(k_fma2, 2, "v_pk_fma_f32 %0, %0, %1, %1")
KERN(k_fma1, 1, "v_pk_fma_f32 %0, %0, %1, %1")
KERN(k_add8, 8, "v_pk_add_f32 %0, %0, %1")
KERN(k_mul8, 8, "v_pk_mul_f32 %0, %0, %1")
KERN(k_fmas8, 8, "v_pk_fma_f32 %0, %0, %2, %1")
KERN(k_addsw8, 8, "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]")
KERN(k_addneg8, 8, "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]")
KERN(k_fmabc8, 8, "v_pk_fma_f32 %0, %0, %2, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]")
#define SKERN(NAME, NCH, ASM)                                                            \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float s, unsigned long long k) \
    {                                                                                    \
        extern __shared__ char pad[];                                                    \
        float x[NCH];                                                                    \
        for (int i = 0; i < NCH; ++i) x[i] = (float)threadIdx.x + i;                     \
        for (int it = 0; it < iters; ++it) {                                             \
            _Pragma("unroll") for (int r = 0; r < 64 / NCH; ++r)                         \
            _Pragma("unroll") for (int i = 0; i < NCH; ++i)                              \
                asm volatile(ASM : "+v"(x[i]) : "v"(s), "s"(k));                         \
        }                                                                                \
        float t = 0;                                                                     \
        for (int i = 0; i < NCH; ++i) t += x[i];                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = t + pad[0];                         \
    }
SKERN(k_sfma8, 8, "v_fma_f32 %0, %0, %1, %1")
SKERN(k_sfma2, 2, "v_fma_f32 %0, %0, %1, %1")
SKERN(k_sfma1, 1, "v_fma_f32 %0, %0, %1, %1")
SKERN(k_scvt8, 8, "v_cvt_f32_ubyte1 %0, %0")
SKERN(k_smin38, 8, "v_min3_f32 %0, %0, %1, %1")
SKERN(k_sperm8, 8, "v_perm_b32 %0, %0, %1, %1")
typedef void (*kern_t)(float *, int, float, unsigned long long);
void run(const char *name, kern_t k)
{
    const int wgs = 768, iters = 4000;
    const size_t lds = 52 * 1024;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float *d; (void)hipMalloc(&d, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, d, 10, 1.0001f, 0x3f8000003f800000ull);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, d, iters, 1.0001f, 0x3f8000003f800000ull);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s %.3f ms  = %.2f cycles per wave-instruction per SIMD at 2.4 GHz (3 waves/SIMD)\n", name, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 3));
    (void)hipFree(d);
}
int main()
{
#define R(n) run(#n, n);
    R(k_sfma8) R(k_sfma2) R(k_sfma1) R(k_scvt8) R(k_smin38) R(k_sperm8) R(k_fma8) R(k_fma4) R(k_fma3) R(k_fma2) R(k_fma1) R(k_add8) R(k_mul8) R(k_fmas8) R(k_addsw8) R(k_addneg8) R(k_fmabc8)
    return 0;
}
