// encode_variants.hip -- the opt-in encoder whose transform runs as f32 matrix instructions
// (JPEGAMD_TRANSFORM=mfma).  Same tile, colour stage, parity tests and C ABI as encode_image_kernel
// in jpeg_kernels.hip; kept for measurement.  The other opt-in form, the row pass on the
// 16-bit matrix pipe (JPEGAMD_TRANSFORM=xdl), is a template option of encode_image_kernel itself.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "host_common.h"
#include "jpeg_device_common.h"

namespace {

// =============================================================================
// Fused kernels with the 8x8 transforms on the matrix cores.
//
// Same tile as above (one wave = 512 x 16 pixels), but the DCT runs as
// v_mfma_f32_16x16x4_f32 products over groups of 16 pixel columns x 16 rows
// (= 2 x 2 luma blocks, or 2 chroma blocks of Cb and of Cr), which leaves the VALU
// for colour conversion and quantisation only.  Per group, with g = lane >> 4 and
// xl = lane & 15:
//   pass one   D1[x][(s,u)]       = sum_(s',y) P[(s',y)][x]   * [s'=s]   m(u,y)     (4 MFMAs, K = 16)
//   pass two   D2[(s,u)][(b,v)]   = sum_x      D1[x][(s,u)]   * [x/8=b]  100 m(v,x%8) (4 MFMAs)
// s = block row (or chroma plane), b = block column.  D1's accumulator register r
// of lane l holds x = 4g + r for (s,u) = xl -- exactly the A-operand layout of
// K-step r of pass two once the constant B operand is permuted the same way -- so
// nothing moves between the passes.  Lane l ends up with the four coefficients
// (s,u) = 4g + r, (b,v) = xl of its group: quantised, staged in LDS at their zigzag
// slots (32 blocks per round) and stored as 1 KiB wave-contiguous pieces.  The level
// shift is the accumulator's initial value.
// Numerics: an MFMA is a k-ordered fma chain, replayed on the host by
// jm_mfma_fdct_block / jm_mfma_idct_block.
// =============================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

// natural (row-major) index -> zigzag scan position: the inverse of kZZ
__device__ constexpr uint8_t kZZInv[64] = {
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42,
    3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// Operand images.  Sample (k, X) -- k = contraction index (luma: tile row; chroma:
// 8*plane + row), X = column -- lives in group c = X >> 4, dword (k & 3) * 16 + (X & 15),
// byte k >> 2: lane (g, xl) of the wave finds its four K-steps k = 4t + g of group c in
// ONE dword, at lane-linear (conflict-free) addresses.  Groups are 68 dwords apart and
// group c starts (c & 3) dwords in, so the colour stage's byte stores (lane = four pixel
// columns, i.e. a quarter group) spread over 16 banks instead of 4.
constexpr int kOpPitch = 68;
constexpr int kYOpBytes = 32 * kOpPitch * 4;     // 8704: 16 luma rows x 512
constexpr int kCOpBytes = 16 * kOpPitch * 4;     // 4352: (Cb, Cr) x 8 rows x 256
constexpr int kStageHalf = 2048 + 64;            // 16 staged blocks of one strip / plane (+ 16 banks)
constexpr int kWaveLdsM = kYOpBytes + kCOpBytes; // 13056; the staging area aliases the chroma operands
static_assert(2 * kStageHalf <= kCOpBytes, "staging must fit in the chroma operand area");

__device__ __forceinline__ constexpr int operand_group(int c) // byte offset of group c's dword 0
{
    return (c * kOpPitch + (c & 3)) * 4;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// NG groups at once: operand dwords (four bytes = K-steps) -> four quantised coefficients
// (magic floats) per group.  The groups' MFMA chains are interleaved so that every
// accumulator is revisited only after NG - 1 independent products (a dependent
// 16x16x4 needs 40 cycles, the pipe accepts one every 32).
template <int NG>
__device__ __forceinline__ void groups_fdct_quant(const uint32_t (&a4)[NG], const float (&b1)[4],
                                                  const float (&b2)[4], f32x4 c1, const float (&qm)[4],
                                                  float (&q)[NG][4])
{
    f32x4 d1[NG], d2[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i)
        d1[i] = mfma4((float)((a4[i] >> 0) & 0xFFu), b1[0], c1);
#pragma unroll
    for (int i = 0; i < NG; ++i)
        d1[i] = mfma4((float)((a4[i] >> 8) & 0xFFu), b1[1], d1[i]);
#pragma unroll
    for (int i = 0; i < NG; ++i)
        d1[i] = mfma4((float)((a4[i] >> 16) & 0xFFu), b1[2], d1[i]);
#pragma unroll
    for (int i = 0; i < NG; ++i)
        d1[i] = mfma4((float)((a4[i] >> 24) & 0xFFu), b1[3], d1[i]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < NG; ++i)
            d2[i] = mfma4(d1[i][r], b2[r], r == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : d2[i]);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = (d2[i][r] + JM_MAGIC) - JM_MAGIC; // round(100 c), :284
            q[i][r] = jm_fma(x, qm[r], JM_MAGIC);       // roundf(c / Q), :306 (see jm_quantise_block)
        }
    }
}

// One round = 8 groups = 16 block columns of both strips (luma) or both planes (chroma):
// transform, quantise, stage the 32 blocks in zigzag order (2 halves x 16 blocks x 128 B)
// and store them as four 1 KiB wave-contiguous pieces.  n_ok: blocks of this round's run
// that lie inside the plane.
__device__ __forceinline__ void mfma_round(const uint32_t (&a4)[8], const float (&b1)[4], const float (&b2)[4],
                                           f32x4 c1, const float (&qm)[4], const int (&zoff)[4],
                                           uint8_t *stage, int lane, uint8_t *dst0, uint8_t *dst1, int n_ok)
{
    float q[8][4];
    groups_fdct_quant<8>(a4, b1, b2, c1, qm, q);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<uint16_t *>(stage + zoff[r] + i * 256) = (uint16_t)jm_bits(q[i][r]);
    }
    wave_lds_fence();
    const int blk = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((i & 1) * 8 + blk < n_ok) {
            uint4 d = *reinterpret_cast<const uint4 *>(stage + (i >> 1) * kStageHalf + (i & 1) * 1024 + lane * 16);
            *reinterpret_cast<uint4 *>(((i >> 1) ? dst1 : dst0) + (i & 1) * 1024 + lane * 16) = d;
        }
    }
    wave_lds_fence(); // staged blocks are read (in-order LDS) before the next round overwrites them
}

__global__ __launch_bounds__(256, 3) void encode_image_mfma_kernel(
    const uint8_t *__restrict__ bgr, int rows, int cols, size_t stride, QuantPair qp,
    int16_t *__restrict__ coef_y, int16_t *__restrict__ coef_cb, int16_t *__restrict__ coef_cr,
    int tiles_x, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerWg * kWaveLdsM];
    __shared__ __attribute__((aligned(16))) float qtab[128];
    if (threadIdx.x < 128)
        qtab[threadIdx.x] = qp.t[threadIdx.x >> 6].inv100q[threadIdx.x & 63];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // uniform: tile arithmetic stays scalar
    const int tile = blockIdx.x * kWavesPerWg + wave;
    if (tile >= n_tiles)
        return; // whole wave leaves together; nothing else is shared between waves
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    uint8_t *ybuf = lds + wave * kWaveLdsM; // luma operand image, 16 rows x 512
    uint8_t *cbuf = ybuf + kYOpBytes;       // chroma operand image, (Cb, Cr) x 8 rows x 256; later the staging area
    const int x0 = tx * kWaveTileW;
    const int bw_y = cols >> 3, bw_c = cols >> 4;

    // ---- colour + 4:2:0 for the 16 tile rows: the task scheme of encode_image_kernel ------
    // (task T covers tile rows 2*(T>>1), +1 and pixel group lane + 64*(T&1); four tasks of
    // register-resident BGR rows are converted while the next four are in flight)
    constexpr int kRing = 4;
    auto task_load = [&](int T, U3 &a, U3 &b) {
        T = T < 15 ? T : 15;
        const int row = ty * 16 + 2 * (T >> 1);
        int x = x0 + 4 * (lane + 64 * (T & 1));
        x = x < cols - 4 ? x : cols - 4;
        const uint8_t *p = bgr + (size_t)row * stride + (size_t)x * 3;
        a = *reinterpret_cast<const U3 *>(p);
        b = *reinterpret_cast<const U3 *>(p + stride);
    };
    U3 ring_a[kRing], ring_b[kRing];
#pragma unroll
    for (int i = 0; i < kRing; ++i)
        task_load(i, ring_a[i], ring_b[i]);
    // this lane's quarter group (luma: columns 4G .. 4G+3) / eighth group (chroma: 2G, 2G+1)
    const int ylane = operand_group(lane >> 2) + 16 * (lane & 3);
    const int clane = operand_group(lane >> 3) + 8 * (lane & 7);
#pragma unroll 1
    for (int T0 = 0; T0 < 16; T0 += kRing) {
        U3 next_a[kRing], next_b[kRing];
#pragma unroll
        for (int i = 0; i < kRing; ++i)
            task_load(T0 + kRing + i, next_a[i], next_b[i]);
        // rows T0 + (slot & 2), +1: k & 3 = (slot & 2), +1 and k >> 2 = T0 >> 2;
        // chroma row kc = T0/2 + (slot >> 1): kc & 3 = (T0/2 & 2) + (slot >> 1), kc >> 2 = T0 >> 3
        uint8_t *yq = ybuf + ylane + (T0 >> 2);
        uint8_t *cq = cbuf + clane + ((T0 >> 1) & 2) * 64 + (T0 >> 3);
#pragma unroll
        for (int slot = 0; slot < kRing; ++slot) {
            const int half = slot & 1; // pixel groups 64 .. 127: 16 luma / 8 chroma operand groups further on
            const int g = lane + 64 * half;
            if (x0 + 4 * g < cols) {
                float ya[4], yb[4], sum_b[2], sum_r[2];
                colour_task(ring_a[slot], ring_b[slot], ya, yb, sum_b, sum_r);
                uint8_t *yp = yq + half * operand_group(16) + (slot & 2) * 64;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    yp[4 * j] = (uint8_t)jm_bits(ya[j]);
                    yp[64 + 4 * j] = (uint8_t)jm_bits(yb[j]);
                }
                // chromaSubsampling: truncating mean of the quad (:139-149); Cr is 8 contraction steps on
                uint8_t *cp = cq + half * operand_group(8) + (slot >> 1) * 64;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    cp[4 * jj] = (uint8_t)jm_bits(jm_quad_mean_magic(sum_b[jj]));
                    cp[4 * jj + 2] = (uint8_t)jm_bits(jm_quad_mean_magic(sum_r[jj]));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kRing; ++i)
            ring_a[i] = next_a[i], ring_b[i] = next_b[i];
    }
    wave_lds_fence();

    // ---- transforms on the matrix cores ------------------------------------------------
    const int g = lane >> 4, xl = lane & 15;
    float b1[4], b2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = 4 * t + g; // pass one: k = (s', y), column xl = (s, u)
        b1[t] = (k >> 3) == (xl >> 3) ? jm_dct_basis(xl & 7, k & 7, 1.0) : 0.0f;
        const int x = 4 * g + t; // pass two, K-step t: x = 4g + t, column xl = (b, v)
        b2[t] = (x >> 3) == (xl >> 3) ? jm_dct_basis(xl & 7, x & 7, 100.0) : 0.0f;
    }
    const float lv = (xl & 7) == 0 ? JM_LEVEL_DC : 0.0f; // recenterAroundZero (:289-298) reaches u = 0 only
    const f32x4 c1 = {lv, lv, lv, lv};
    // this lane's four outputs of a group: (s, u) = 4g + r, (b, v) = xl -> half s, block b of the
    // group's two, zigzag slot of (u, v)
    float qml[4], qmc[4];
    int zoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nat = (4 * (g & 1) + r) * 8 + (xl & 7);
        qml[r] = qtab[nat];
        qmc[r] = qtab[64 + nat];
        zoff[r] = (g >> 1) * kStageHalf + (xl >> 3) * 128 + 2 * kZZInv[nat];
    }
    uint8_t *stage = cbuf;
    {
        // chroma first: all 16 operand dwords go to registers, which frees their area for staging
        uint32_t ca[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            ca[i] = *reinterpret_cast<const uint32_t *>(cbuf + operand_group(i) + lane * 4);
        wave_lds_fence();
        const int n_ok = min(32, bw_c - tx * 32);
        uint8_t *pb = reinterpret_cast<uint8_t *>(coef_cb + ((size_t)ty * bw_c + tx * 32) * 64);
        uint8_t *pr = reinterpret_cast<uint8_t *>(coef_cr + ((size_t)ty * bw_c + tx * 32) * 64);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            uint32_t a4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                a4[i] = ca[8 * f + i];
            mfma_round(a4, b1, b2, c1, qmc, zoff, stage, lane, pb + f * 2048, pr + f * 2048, n_ok - 16 * f);
        }
    }
    {
        // luma: strip s = block row of the tile
        const int n_ok = min(64, bw_y - tx * 64);
        uint8_t *p0 = reinterpret_cast<uint8_t *>(coef_y + ((size_t)(ty * 2) * bw_y + tx * 64) * 64);
        uint8_t *p1 = p0 + (size_t)bw_y * 128;
#pragma unroll 1
        for (int f = 0; f < 4; ++f) {
            if (16 * f >= n_ok)
                break; // groups beyond the image were never written by the colour stage
            uint32_t a4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                a4[i] = *reinterpret_cast<const uint32_t *>(ybuf + f * operand_group(8) + operand_group(i) + lane * 4);
            mfma_round(a4, b1, b2, c1, qml, zoff, stage, lane, p0 + f * 2048, p1 + f * 2048, n_ok - 16 * f);
        }
    }
}

} // namespace

int jpegamd_launch_encode_variant(int variant, int grid, void *stream, hipEvent_t start, hipEvent_t stop,
                                  const uint8_t *d_bgr, int rows, int cols, size_t stride, const QuantPair &qp,
                                  int16_t *d_coef_y, int16_t *d_coef_cb, int16_t *d_coef_cr, int tiles_x, int n_tiles)
{
    if (variant != 1)
        return jpegamd_fail("encode variant %d is not in this translation unit", variant);
    auto kernel = encode_image_mfma_kernel;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, start, stop, 0, d_bgr, rows,
                          cols, stride, qp, d_coef_y, d_coef_cb, d_coef_cr, tiles_x, n_tiles);
    JPEGAMD_HIPCHK(hipGetLastError());
    return 0;
}
