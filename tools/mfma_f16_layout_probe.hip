// Developer probe: operand/result layout of v_mfma_f32_16x16x32_f16 on gfx950, and whether
// f16 subnormal inputs are flushed.  Assumed: lane l holds A[m = l%16][k = 8*(l/16) + i],
// B[k = 8*(l/16) + i][n = l%16], i = 0..7, and D[m = 4*(l/16) + r][n = l%16], r = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16 *A, const _Float16 *B, float *D)
{
    const int l = threadIdx.x, g = l >> 4, x = l & 15;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = A[x * 32 + 8 * g + i];
        b[i] = B[(8 * g + i) * 16 + x];
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r)
        D[(4 * g + r) * 16 + x] = c[r];
}
int main()
{
    _Float16 hA[16 * 32], hB[32 * 16];
    float hD[256], ref[256];
    srand(1);
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 512; ++i) {
            hA[i] = (_Float16)(float)(rand() % 255 - 127);
            hB[i] = pass == 0 ? (_Float16)(float)(rand() % 15 - 7) : (_Float16)(ldexpf((float)(rand() % 15 - 7), -22)); // subnormals
        }
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double s = 0;
                for (int kk = 0; kk < 32; ++kk)
                    s += (double)(float)hA[m * 32 + kk] * (double)(float)hB[kk * 16 + n];
                ref[m * 16 + n] = (float)s;
            }
        _Float16 *dA, *dB;
        float *dD;
        hipMalloc(&dA, sizeof hA), hipMalloc(&dB, sizeof hB), hipMalloc(&dD, sizeof hD);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i)
            bad += hD[i] != ref[i];
        printf("%s: %d of 256 differ from the assumed layout (D[0]=%g ref %g)\n",
               pass == 0 ? "integers" : "subnormal B", bad, hD[0], ref[0]);
    }
    return 0;
}
