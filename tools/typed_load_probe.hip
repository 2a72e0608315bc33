#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// buffer resource (V#) for gfx9: base, stride 0, num_records = bytes, word3 = dst_sel xyzw + formats
__device__ __forceinline__ u32x4 make_rsrc(const void *p, uint32_t bytes, uint32_t word3)
{
    uint64_t a = (uint64_t)p;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, word3};
}

__global__ void k(const uint8_t *src, float *out, uint32_t bytes, uint32_t word3)
{
    u32x4 r = make_rsrc(src, bytes, word3);
    uint32_t off = threadIdx.x * 4;
    f32x4 v;
    asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n\ts_waitcnt vmcnt(0)"
                 : "=v"(v) : "v"(off), "s"(r) : "memory");
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main()
{
    uint8_t h[256]; for (int i = 0; i < 256; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; float *o; hipMalloc(&d, 256); hipMalloc(&o, 256 * 4); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    // word3: DST_SEL_X=4(X),Y=5,Z=6,W=7 in bits 0..11
    uint32_t word3 = (4u) | (5u << 3) | (6u << 6) | (7u << 9) | (2u << 12) | (10u << 15); // + NUM_FORMAT USCALED, DATA_FORMAT 8_8_8_8
    k<<<1, 64>>>(d, o, 256, word3);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) if (r[i] != (float)h[i]) ++bad;
    printf("bad %d  first: %g %g %g %g (want %d %d %d %d)\n", bad, r[0], r[1], r[2], r[3], h[0], h[1], h[2], h[3]);
    return 0;
}
