// Developer microbenchmark: the fused decoder's memory access pattern without its arithmetic.
// One wave per 512x16 tile: reads 24 KiB of coefficients as 1 KiB wave loads, writes 16 rows x
// 1536 B as 768-byte wave stores (12 B per lane, nontemporal) or as 16-byte-per-lane stores.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3), aligned(4)));
template <int MODE> // 0: 12 B per lane (the decoder's), 1: 16 B per lane on 48 lanes, same bytes
__global__ __launch_bounds__(256, 3) void pattern_kernel(const u32x4 *__restrict__ coef, unsigned char *__restrict__ out,
                                                         int tiles_x, int n_tiles, size_t stride)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles)
        return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const u32x4 *src = coef + (size_t)tile * 1536; // 24 KiB per tile
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < 3; ++r) {
        u32x4 d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            d[k] = src[(r * 8 + k) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            acc ^= d[k];
        if (r == 0)
            continue;
        for (int t = 0; t < 8; ++t) {
            const int row = ty * 16 + (r - 1) * 8 + 2 * (t >> 1);
            unsigned char *p0 = out + (size_t)row * stride + (size_t)tx * 1536 + (t & 1) * 768;
            if (MODE == 0) {
                u32x3 v = {acc.x + t, acc.y, acc.z};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x3 *>(p0 + lane * 12));
                __builtin_nontemporal_store(v, reinterpret_cast<u32x3 *>(p0 + stride + lane * 12));
            } else if (lane < 48) {
                u32x4 v = {acc.x + t, acc.y, acc.z, acc.w};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p0 + lane * 16));
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p0 + stride + lane * 16));
            }
        }
    }
}
template <int MODE> static void run(const char *name, const u32x4 *coef, unsigned char *out)
{
    const int tiles_x = 16, n_tiles = 8192;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        pattern_kernel<MODE><<<n_tiles / 4, 256>>>(coef, out, tiles_x, n_tiles, 8192 * 3);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i)
        pattern_kernel<MODE><<<n_tiles / 4, 256>>>(coef, out, tiles_x, n_tiles, 8192 * 3);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %7.1f us\n", name, ms * 1e3 / 20);
}
int main()
{
    const size_t bytes = 201326592;
    u32x4 *coef;
    unsigned char *out;
    (void)hipMalloc(&coef, bytes), (void)hipMalloc(&out, bytes);
    (void)hipMemset(coef, 1, bytes);
    run<0>("12 B per lane, 64 lanes", coef, out);
    run<1>("16 B per lane, 48 lanes", coef, out);
    run<0>("12 B per lane, 64 lanes", coef, out);
    return 0;
}
