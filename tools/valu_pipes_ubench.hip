// Developer microbenchmark (round 3): do the two VALU instruction classes of profiles/r03_a section 5 (plain f32
// add / mul / fma at ~3 cycles, everything else at ~4.3) share ONE issue resource, or do instructions of different
// classes from different waves overlap?  Three waves per SIMD; each wave runs 256 instructions per iteration of
// its own class.  If the classes were separate pipes, a mixed population would finish in about the time of its
// largest class; if they share one, in about the sum.
#include <hip/hip_runtime.h>
#include <cstdio>

// class: 0 idle, 1 v_fmac_f32 literal (fast), 2 v_cvt_f32_ubyte1 (slow), 3 v_perm_b32 (slow), 4 v_add_f32 (fast)
__global__ __launch_bounds__(768) void k(float *out, int iters, int ca, int cb, int cc)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cls = wave < 4 ? ca : wave < 8 ? cb : cc;
    float x = threadIdx.x * 1e-3f;
    float v[8] = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
    for (int it = 0; it < iters; ++it) {
        if (cls == 1) {
#pragma unroll
            for (int r = 0; r < 32; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    asm volatile("v_fmac_f32 %0, 0x3e99999a, %1" : "+v"(v[c]) : "v"(x));
        } else if (cls == 2) {
#pragma unroll
            for (int r = 0; r < 32; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v[c]));
        } else if (cls == 3) {
#pragma unroll
            for (int r = 0; r < 32; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(x), "v"(0x05040100u));
        } else if (cls == 4) {
#pragma unroll
            for (int r = 0; r < 32; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(x));
        }
    }
    float s = 0;
    for (int c = 0; c < 8; ++c)
        s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(float *d, int iters, int a, int b, int c)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    k<<<256, 768>>>(d, iters, a, b, c);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<256, 768>>>(d, iters, a, b, c);
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
    const char *names[] = {"idle", "fmac", "cvt", "perm", "add"};
    const int cases[][3] = {{1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {1, 1, 0}, {2, 2, 0}, {1, 2, 0}, {1, 3, 0}, {2, 3, 0}, {1, 1, 1}, {2, 2, 2},
                            {1, 1, 2}, {1, 2, 2}, {1, 1, 3}, {4, 4, 2}, {1, 2, 3}};
    for (auto &c : cases) {
        const float t = run(d, iters, c[0], c[1], c[2]);
        int n = (c[0] != 0) + (c[1] != 0) + (c[2] != 0);
        printf("%-5s | %-5s | %-5s : %8.1f us = %.2f cycles per instruction and SIMD\n", names[c[0]], names[c[1]], names[c[2]], t,
               t * 2400.0 / (iters * 256.0 * n));
    }
    return 0;
}
