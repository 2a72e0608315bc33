// Developer microbenchmark: device-to-device copy of 201 MB in / 201 MB out (the byte counts of
// one 8192^2 encode or decode) with plain and with nontemporal stores, 16 B per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT_STORE, bool NT_LOAD>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = NT_LOAD ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT_STORE)
            __builtin_nontemporal_store(v, dst + i);
        else
            dst[i] = v;
    }
}
template <bool S, bool L> static void run(const char *name, const u32x4 *src, u32x4 *dst, size_t n, int grid)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        copy_kernel<S, L><<<grid, 256>>>(src, dst, n);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i)
        copy_kernel<S, L><<<grid, 256>>>(src, dst, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20;
    printf("%-28s grid %6d : %7.1f us  %6.2f TB/s read+write\n", name, grid, us, 2.0 * n * 16 / us / 1e6);
}
int main()
{
    const size_t bytes = 201326592, n = bytes / 16; // 8192 * 8192 * 3
    u32x4 *src, *dst;
    (void)hipMalloc(&src, bytes), (void)hipMalloc(&dst, bytes);
    (void)hipMemset(src, 1, bytes);
    for (int grid : {2048, 8192, 49152}) {
        run<false, false>("plain", src, dst, n, grid);
        run<true, false>("nontemporal stores", src, dst, n, grid);
        run<true, true>("nontemporal loads + stores", src, dst, n, grid);
    }
    return 0;
}
