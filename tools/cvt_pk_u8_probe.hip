// Developer probe: rounding and saturation of v_cvt_pk_u8_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        unsigned r;
        asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(in[i]));
        out[i] = r;
    }
}
int main()
{
    const float h[] = {-3.7f, -0.5f, -0.49f, 0.0f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 127.5f, 128.5f, 254.5f, 254.51f,
                       255.0f, 255.49f, 255.5f, 300.0f, 1e9f};
    const int n = sizeof h / sizeof *h;
    float *d;
    unsigned *o, ho[32];
    (void)hipMalloc(&d, sizeof h), (void)hipMalloc(&o, sizeof ho);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    (void)hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("%10.2f -> %u\n", h[i], ho[i]);
    return 0;
}
