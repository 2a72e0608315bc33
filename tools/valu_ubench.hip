// Developer microbenchmark: VALU issue rate of scalar vs packed f32 FMA on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[8];
    v2f y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = v2f{x[i], x[i] + 1}; }
    v2f aa = {a, a}, bb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], aa, bb);
                if (MODE == 2) x[i] = __builtin_truncf(x[i] + a);
                if (MODE == 3) { int t = (int)x[i]; x[i] = (float)(t * 3 + 1); } // cvt + int mad + cvt
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int per_iter)
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<4096, 256>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<4096, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = 4096.0 * 4 * iters * 64 * per_iter; // wave-instructions
    printf("%-28s %.3f ms  %.3f T wave-instr/s  (%.2f cycles/instr/SIMD at 2.4GHz)\n", name, ms,
           winstr / ms / 1e9, 1024 * 2.4e9 / (winstr / (ms * 1e-3)));
    hipFree(d);
}
int main()
{
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 1);
    run<2>("v_add+v_trunc", 2);
    run<3>("cvt_i32+mad+cvt_f32", 3);
    return 0;
}
