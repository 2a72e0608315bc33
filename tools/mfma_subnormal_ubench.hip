// Developer microbenchmark (round 3): what does a v_mfma_f32_16x16x32_f16 cost when one operand holds
// f16 SUBNORMALS (the byte-as-subnormal operand trick) instead of normal numbers -- alone, and beside
// two VALU waves on the same SIMD?  768-thread workgroups: waves 0-3 role A, 4-7 role B, 8-11 role C.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// role: 0 idle, 1 mfma f16 normal operands, 2 valu fma, 3 mfma f16 with subnormal B, 4 mfma f16 with zero B
__global__ __launch_bounds__(768) void k(float *out, int iters, int role_a, int role_b, int role_c)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? role_a : wave < 8 ? role_b : role_c;
    float x = threadIdx.x * 1e-3f;
    f32x4 acc[4] = {{x, x, x, x}, {x, x, x, x}, {x, x, x, x}, {x, x, x, x}};
    float v[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    if (role == 1 || role == 3 || role == 4) {
        f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8};
        const unsigned short lo = (unsigned short)(threadIdx.x & 255);
        u16x8 bits = {lo, (unsigned short)(lo ^ 1), (unsigned short)(lo ^ 2), 3, 4, 5, 6, (unsigned short)(lo ^ 7)};
        f16x8 b = role == 3 ? __builtin_bit_cast(f16x8, bits) : role == 4 ? f16x8{0, 0, 0, 0, 0, 0, 0, 0} : a;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(x), "v"(0.25f));
        }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c)
        s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int c = 0; c < 8; ++c)
        s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(float *d, int iters, int a, int b, int c)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<<<256, 768>>>(d, iters, a, b, c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 768>>>(d, iters, a, b, c);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 768 * sizeof(float));
    const int iters = 4000; // 64 k mfma (16 per iter) or 512 k fma (128 per iter) per wave
    const char *names[] = {"idle", "mfma_norm", "valu_fma", "mfma_subn", "mfma_zero"};
    const int cases[][3] = {{1, 0, 0}, {3, 0, 0}, {4, 0, 0}, {2, 2, 0}, {2, 2, 1}, {2, 2, 3}, {2, 2, 4}, {1, 1, 1}, {3, 3, 3}, {2, 1, 1}, {2, 3, 3}};
    for (auto &c : cases)
        printf("%-10s | %-10s | %-10s : %8.1f us\n", names[c[0]], names[c[1]], names[c[2]], run(d, iters, c[0], c[1], c[2]));
    return 0;
}
