// jpeg_device_common.h -- device-side pieces shared by the translation units that hold fused
// kernels (jpeg_kernels.hip, encode_variants.hip): tile constants, the 12-byte pixel group,
// streaming stores, wave-level LDS fences, the transform stage of one block and the colour
// task.  Everything is internal to its translation unit (anonymous namespace).
#ifndef JPEGAMD_DEVICE_COMMON_H
#define JPEGAMD_DEVICE_COMMON_H
#include <hip/hip_runtime.h>

#include <cstdint>

#include "jpeg_math.h"

typedef JmQuantTab QuantConsts; // inv100q (biased) and q, natural order (jpeg_math.h)
struct JpegamdQuantPair {       // (a named type: it crosses translation units as a kernel argument)
    QuantConsts t[2];           // [0] luma, [1] chroma
};
typedef JpegamdQuantPair QuantPair;

namespace {

// ------------------------------------------------------------------ constants

// zz position i holds natural index kZZ[i] (encode_decode.cpp:312-348)
__device__ constexpr uint8_t kZZ[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int kWaveTileW = 512;      // pixels per wave tile, horizontally
constexpr int kStagePitch = 144;     // bytes per staged block (128 + 16 pad: conflict-free b128)
constexpr int kStageBytes = 64 * kStagePitch; // 9216
constexpr int kChromaBytes = 2 * 8 * 256;     // Cb + Cr, 8 rows x 256 samples
constexpr int kWaveLds = kStageBytes + kChromaBytes; // 13312
constexpr int kWavesPerWg = 4;

typedef __attribute__((address_space(3))) uint8_t lds_u8; // a byte in LDS, for stores that must stay byte stores

struct __attribute__((aligned(4))) U3 {
    uint32_t a, b, c;
};

// Streaming stores.  The decoder's pixels are written once and consumed outside the GPU, so
// its stores carry the nontemporal hint: the lines are not kept in the caches and do not evict
// the coefficients the decoder is still reading.  The ENCODER's coefficients are read next by
// another kernel (run-length stage or decoder) and 201 MB of them fit the 256 MB last-level
// cache, so its stores stay plain.  Measured in the real sequence encode, decode, encode, ...
// at 8192^2 (tools/lib_step_bench.py): 182.6 us per step with plain stores everywhere, 178.4
// with the hint in both kernels, 165.3 with the hint in the decoder only (decoder 78 -> 67 us).
// The same hint on the LOADS only costs (decoder +10 us).
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void store8_stream(void *p, uint2 v)
{
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(u32x2_t{v.x, v.y}, static_cast<u32x2_t *>(p));
}
__device__ __forceinline__ void store12_stream(void *p, const U3 &v)
{
    __builtin_nontemporal_store(u32x3_t{v.a, v.b, v.c}, static_cast<u32x3_t *>(p));
}
__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave are issued and retired in order; this only
    // stops the compiler from moving accesses across the hand-over point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// byte J of w as a float, opaque to the optimiser: left to itself LLVM turns the first
// butterfly level of the DCT into SDWA integer adds plus one conversion per SUM, which
// costs more slow-rate instructions than converting the eight inputs
template <int J>
__device__ __forceinline__ float ubyte_f32(uint32_t w)
{
    float f;
    if (J == 0)
        asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w));
    else if (J == 1)
        asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w));
    else if (J == 2)
        asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w));
    else
        asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w));
    return f;
}

__device__ __forceinline__ int u3_byte(const U3 &v, int idx)
{
    uint32_t w = idx < 4 ? v.a : (idx < 8 ? v.b : v.c);
    return (int)((w >> ((idx & 3) * 8)) & 0xFFu);
}

// ------------------------------------------------------------ transform stage
// v: 64 pixel values (row-major) on entry; on exit v[i] = MAGIC + quantised
// coefficient i (natural order): the int16 value sits in the low 16 bits.
// Straight-line code on purpose: with no branch in it the scheduler keeps the
// whole transform near 80 VGPRs.
__device__ __forceinline__ void block_fdct_quant(float (&v)[64], const float *inv100q)
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
        jm_fdct8<1, true>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3],
                          v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
        jm_fdct8<100, false>(v[0 * 8 + c], v[1 * 8 + c], v[2 * 8 + c], v[3 * 8 + c],
                             v[4 * 8 + c], v[5 * 8 + c], v[6 * 8 + c], v[7 * 8 + c]);
    jm_quantise_block(v, inv100q);
}

// v: 64 dequantised coefficients (natural order) on entry; on exit v[i] = MAGIC +
// 128 + pixel i (unclamped): the int16 value sits in the low 16 bits.
__device__ __forceinline__ void block_idct(float (&v)[64])
{
#pragma unroll
    for (int c = 0; c < 8; ++c)
        jm_idct8<false>(v[0 * 8 + c], v[1 * 8 + c], v[2 * 8 + c], v[3 * 8 + c],
                 v[4 * 8 + c], v[5 * 8 + c], v[6 * 8 + c], v[7 * 8 + c]);
#pragma unroll
    for (int r = 0; r < 8; ++r)
        jm_idct8<true>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3],
                 v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
#pragma unroll
    for (int i = 0; i < 64; ++i)
        v[i] = jm_pixel_magic_preshrunk(v[i]);
}

// quantised block -> 128 bytes in zigzag order at stage + lane*144
__device__ __forceinline__ void stage_write_zigzag(const float (&qv)[64], uint8_t *stage, int lane)
{
    uint4 *dst = reinterpret_cast<uint4 *>(stage + lane * kStagePitch);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int i = (k * 4 + j) * 2;
            w[j] = jm_pack_lo16(jm_bits(qv[kZZ[i]]), jm_bits(qv[kZZ[i + 1]]));
        }
        dst[k] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// 128 staged bytes (zigzag int16) -> dequantised floats in natural order
__device__ __forceinline__ void stage_read_dequant(const uint8_t *stage, int lane, float (&v)[64],
                                                   const float *q)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(stage + lane * kStagePitch);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint4 d = src[k];
        uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int i = (k * 4 + j) * 2;
            int lo = (int)(int16_t)(w[j] & 0xFFFFu);
            int hi = (int)w[j] >> 16;
            v[kZZ[i]] = (float)lo * q[kZZ[i]];         // encode_decode.cpp:564 / kernel.cu:175
            v[kZZ[i + 1]] = (float)hi * q[kZZ[i + 1]];
        }
    }
}

// Bit i of the result is set where position i is the LAST of its run in the block's zigzag
// sequence: value i differs from value i + 1, or i == 63 (encode_decode.cpp:357-368).  The
// number of (value, run) pairs is the popcount; the runs are the distances between set bits.
__device__ __forceinline__ unsigned long long run_ends(const uint32_t (&w)[32])
{
    uint32_t half[2] = {0u, 0u};
#pragma unroll
    for (int k = 31; k >= 0; --k) { // MSB first: each step doubles the half-mask and adds one bit
        // low half: value 2k ^ value 2k+1, high half: value 2k+1 ^ value 2k+2
        const uint32_t x = w[k] ^ __builtin_amdgcn_alignbit(k < 31 ? w[k + 1] : 0u, w[k], 16);
        const uint32_t odd = k == 31 ? 1u : (uint32_t)(x > 0xFFFFu);
        const uint32_t even = (uint32_t)((x & 0xFFFFu) != 0u);
        uint32_t &h = half[k >> 4];
        h = h + h + odd;
        h = h + h + even;
    }
    return ((unsigned long long)half[1] << 32) | half[0];
}

// The popcount of run_ends() without the mask: differing neighbours counted two per packed
// instruction (16-bit halves), four instructions per dword instead of nine.
__device__ __forceinline__ uint32_t count_pairs(const uint32_t (&w)[32])
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 one = {1, 1};
    u16x2 acc = {0, 0};
#pragma unroll
    for (int k = 0; k < 31; ++k) { // low half: value 2k != value 2k+1, high half: value 2k+1 != value 2k+2
        const uint32_t x = w[k] ^ __builtin_amdgcn_alignbit(w[k + 1], w[k], 16);
        acc += __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), one);
    }
    const uint32_t last = (w[31] ^ (w[31] >> 16)) & 0xFFFFu; // value 62 != value 63; position 63 always ends a run
    return (uint32_t)acc.x + (uint32_t)acc.y + (last ? 2u : 1u);
}

// One task of the colour stage for one lane: 2 rows x 4 pixels of BGR -> the eight luma
// values as magic floats (byte in the low 8 bits) and the two quads' sums of centred Cb, Cr
// (RGB2YCbCr, encode_decode.cpp:39-75, and the sums of chromaSubsampling, :139-149).
// The two rows of a column go through one straight-line block with ONE tie test for the
// pair: two independent fma chains interleave, and half as many branches sit in the hot
// path (measured: 158 -> 143 us for the kernel).
__device__ __forceinline__ void colour_task(const U3 &row_a, const U3 &row_b, float (&ya)[4], float (&yb)[4],
                                            float (&sum_b)[2], float (&sum_r)[2])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float uy[2], cb[2], cr[2], p[2][3];
        JmFwdTies tie[2];
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const U3 &px = row ? row_b : row_a;
            p[row][0] = (float)u3_byte(px, 3 * j), p[row][1] = (float)u3_byte(px, 3 * j + 1);
            p[row][2] = (float)u3_byte(px, 3 * j + 2);
            jm_fwd_colour_fast(p[row][0], p[row][1], p[row][2], uy[row], cb[row], cr[row], tie[row]);
        }
        const float worst_a = jm_min2(jm_min3(tie[0].ry, tie[0].rr, tie[1].ry), tie[1].rr);
        const float worst_b = jm_min2(tie[0].rb, tie[1].rb);
        if (__builtin_expect(worst_a < JM_TIE_A || worst_b < JM_TIE_B, 0)) { // ask the doubles
#pragma unroll
            for (int row = 0; row < 2; ++row)
                if (jm_fwd_any_tie(tie[row]))
                    jm_fwd_colour_fix(p[row][0], p[row][1], p[row][2], tie[row], uy[row], cb[row], cr[row]);
        }
        ya[j] = uy[0], yb[j] = uy[1];
        if (j & 1) { // second column of the quad
            sum_b[j >> 1] += cb[0] + cb[1];
            sum_r[j >> 1] += cr[0] + cr[1];
        } else {
            sum_b[j >> 1] = cb[0] + cb[1];
            sum_r[j >> 1] = cr[0] + cr[1];
        }
    }
}

} // namespace
#endif
