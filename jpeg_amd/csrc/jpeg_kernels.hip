// jpeg_kernels.hip -- gfx950 (MI355X, CDNA4) kernels and the C ABI of
// include/jpeg_amd.h.  Written for wave64 / 160 KiB LDS; there is no other
// target and no CPU fallback.
//
// Work decomposition (DESIGN.md has the long form):
//   * one LANE owns one 8x8 block for the transform stage: 64 samples live in
//     VGPRs, both 1-D passes and the zigzag are pure register renaming -- no
//     cross-lane traffic, no transposes.
//   * one WAVE owns a 512 x 16 pixel tile (32 MCUs: 128 Y + 32 Cb + 32 Cr
//     blocks = three 64-lane transform rounds) and shares nothing with other
//     waves but the quantiser table (one barrier after its fill): from then on
//     only wave-level LDS ordering.
//   * global traffic is always lane-contiguous: 12 B/lane BGR rows (768 B per
//     wave instruction) and 16 B/lane coefficient chunks (1 KiB per wave
//     instruction, staged through LDS so that each block's 128 B stay whole).
//
// Arithmetic lives in jpeg_math.h (shared with the CPU model the tests check
// against the oracle).  Build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/jpeg_amd.h"
#include "jpeg_math.h"
#include "host_common.h"

namespace {

// ------------------------------------------------------------------ constants
typedef JmQuantTab QuantConsts; // inv100q (biased) and q, natural order (jpeg_math.h)
struct QuantPair {
    QuantConsts t[2]; // [0] luma, [1] chroma
};

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

// ------------------------------------------------------------- fused encoder
// grid: one wave per 512x16 tile, 4 waves per workgroup.
__global__ __launch_bounds__(256, 3) void encode_image_kernel(
    const uint8_t *__restrict__ bgr, int rows, int cols, size_t stride, QuantPair qp,
    int16_t *__restrict__ coef_y, int16_t *__restrict__ coef_cb, int16_t *__restrict__ coef_cr,
    int tiles_x, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerWg * kWaveLds];
    // quantiser multipliers as an LDS table (broadcast b128 reads): 128 SGPRs of
    // kernel arguments would not fit next to the transform
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
    uint8_t *stage = lds + wave * kWaveLds; // Y rows (8 x 512) alias the staging area
    uint8_t *cbuf = stage + kStageBytes;    // Cb: 8 x 256, Cr: 8 x 256

    const int x0 = tx * kWaveTileW;
    const int bw_y = cols >> 3, bw_c = cols >> 4;

    // BGR rows are double-buffered in registers four tasks at a time: while the four tasks
    // of one buffer are converted, the loads of the next four are in flight (HBM latency
    // under load is several tasks long).  The inner four tasks are unrolled so that every
    // slot is a fixed register; the buffers swap by moves once per four tasks, by which
    // time the loads issued four tasks earlier have landed.  Task T = 8*it + t covers
    // 2 rows x 4 pixels per lane: row pair rp = t>>1, pixel group g = lane + 64*(t&1).
    // Loads are unconditional: lanes beyond the image edge re-read its last pixel group,
    // tasks beyond 15 re-read tasks 14 and 15.
    constexpr int kRing = 4;
    // The row base is uniform (scalar registers) and the lane's byte offset within a row
    // takes only two values, so a load costs no vector arithmetic: saddr + 32-bit voffset.
    uint32_t lane_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int x = x0 + 4 * (lane + 64 * h);
        x = x < cols - 4 ? x : cols - 4;
        lane_off[h] = 3u * (uint32_t)x;
    }
    auto task_load = [&](int T, int odd, U3 &a, U3 &b) { // odd == T & 1, a compile-time constant at every call
        T = T < 16 ? T : 14 + odd;
        const uint8_t *rowp = bgr + (size_t)(ty * 16 + 2 * (T >> 1)) * stride;
        a = *reinterpret_cast<const U3 *>(rowp + lane_off[odd]);
        b = *reinterpret_cast<const U3 *>(rowp + stride + lane_off[odd]);
    };
    U3 ring_a[kRing], ring_b[kRing];
#pragma unroll
    for (int i = 0; i < kRing; ++i)
        task_load(i, i & 1, ring_a[i], ring_b[i]);

#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        if (it < 2) {
            // ---- colour + 4:2:0 for tile rows 8*it .. 8*it+7 -----------------
#pragma unroll 1
            for (int t0 = 0; t0 < 8; t0 += kRing) {
            U3 next_a[kRing], next_b[kRing];
#pragma unroll
            for (int i = 0; i < kRing; ++i)
                task_load(it * 8 + t0 + kRing + i, i & 1, next_a[i], next_b[i]);
#pragma unroll
            for (int slot = 0; slot < kRing; ++slot) {
                const int t = t0 + slot;
                const int rp = t >> 1;
                const int g = lane + 64 * (t & 1);
                if (x0 + 4 * g < cols) {
                    float ya[4], yb[4], sum_b[2], sum_r[2];
                    // colour_task(), written out: as a call the compiler schedules this kernel
                    // 5 % slower (109.8 vs 104.5 us at 8192^2, same instructions, other order)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // The two rows of this column in one straight-line block, ONE tie test for
                        // the pair: two independent fma chains interleave, and half as many
                        // branches sit in the hot path (measured: 158 -> 143 us for the kernel).
                        float uy[2], cb[2], cr[2], p[2][3];
                        JmFwdTies tie[2];
#pragma unroll
                        for (int row = 0; row < 2; ++row) {
                            const U3 &px = row ? ring_b[slot] : ring_a[slot];
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
                    // bytes go to LDS one by one (the low byte of each magic float): byte stores
                    // ride the LDS queue, packing them first costs 16 VALU instructions per task
                    // (volatile: left alone, the compiler merges the stores and packs)
                    volatile lds_u8 *y0 = (volatile lds_u8 *)(stage + (2 * rp) * 512 + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        y0[j] = (uint8_t)jm_bits(ya[j]);
                        y0[512 + j] = (uint8_t)jm_bits(yb[j]);
                    }
                    // chromaSubsampling: truncating mean of the quad (:139-149)
                    const int crow = it * 4 + rp;
                    volatile lds_u8 *c0 = (volatile lds_u8 *)(cbuf + crow * 256 + 2 * g);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        c0[q] = (uint8_t)jm_bits(jm_quad_mean_magic(sum_b[q]));
                        c0[2048 + q] = (uint8_t)jm_bits(jm_quad_mean_magic(sum_r[q]));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kRing; ++i)
                ring_a[i] = next_a[i], ring_b[i] = next_b[i];
            }
        }
        wave_lds_fence();

        // ---- transform: lane = block ---------------------------------------
        const bool luma = it < 2;
        const uint8_t *src = luma ? stage + lane * 8 : cbuf + (lane >> 5) * 2048 + (lane & 31) * 8;
        const int pitch = luma ? 512 : 256;
        const int bcol = luma ? tx * 64 + lane : tx * 32 + (lane & 31);
        const bool valid = bcol < (luma ? bw_y : bw_c);
        if (valid) {
            float v[64];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint2 d = *reinterpret_cast<const uint2 *>(src + r * pitch);
                v[r * 8 + 0] = ubyte_f32<0>(d.x), v[r * 8 + 1] = ubyte_f32<1>(d.x);
                v[r * 8 + 2] = ubyte_f32<2>(d.x), v[r * 8 + 3] = ubyte_f32<3>(d.x);
                v[r * 8 + 4] = ubyte_f32<0>(d.y), v[r * 8 + 5] = ubyte_f32<1>(d.y);
                v[r * 8 + 6] = ubyte_f32<2>(d.y), v[r * 8 + 7] = ubyte_f32<3>(d.y);
            }
            block_fdct_quant(v, qtab + (luma ? 0 : 64));
            // The staging area aliases the Y rows.  LDS instructions of one wave
            // retire in program order, so every lane's reads above are done before
            // the writes below; the fence only pins the compiler's order.
            wave_lds_fence();
            stage_write_zigzag(v, stage, lane);
        }
        wave_lds_fence();

        // ---- coalesced store: 64 blocks x 128 B = 8 x (64 lanes x 16 B) ------
        // piece k of lane l is bytes [16 l, 16 l + 16) of blocks 8k .. 8k+7: one
        // uniform base per plane plus constant offsets
        {
            const int n_ok = luma ? bw_y - tx * 64 : bw_c - tx * 32; // blocks of this tile inside the plane
            uint8_t *base_a, *base_b; // pieces 0-3 / 4-7
            if (luma) {
                base_a = reinterpret_cast<uint8_t *>(coef_y + ((size_t)(ty * 2 + it) * bw_y + tx * 64) * 64);
                base_b = base_a + 4096;
            } else {
                base_a = reinterpret_cast<uint8_t *>(coef_cb + ((size_t)ty * bw_c + tx * 32) * 64);
                base_b = reinterpret_cast<uint8_t *>(coef_cr + ((size_t)ty * bw_c + tx * 32) * 64);
            }
            const int blk = lane >> 3;
            const uint8_t *sp = stage + blk * kStagePitch + (lane & 7) * 16;
            if (n_ok >= (luma ? 64 : 32)) {
                // the whole round lies inside the plane (every tile but the last of a row): all
                // eight LDS reads in flight at once, then the eight stores
                uint4 d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    d[k] = *reinterpret_cast<const uint4 *>(sp + k * 8 * kStagePitch);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    *reinterpret_cast<uint4 *>((k < 4 ? base_a : base_b) + (k & 3) * 1024 + lane * 16) = d[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int b = (luma ? k * 8 : (k & 3) * 8) + blk; // block index within its plane's run
                    if (b < n_ok) {
                        uint4 d = *reinterpret_cast<const uint4 *>(sp + k * 8 * kStagePitch);
                        *reinterpret_cast<uint4 *>((k < 4 ? base_a : base_b) + (k & 3) * 1024 + lane * 16) = d;
                    }
                }
            }
        }
        wave_lds_fence(); // staging area is free again
    }
}

// ------------------------------------------------------------- fused decoder
__device__ __forceinline__ void fetch_coefs(int r, const int16_t *__restrict__ coef_y,
                                            const int16_t *__restrict__ coef_cb,
                                            const int16_t *__restrict__ coef_cr, int tx, int ty, int bw_y,
                                            int bw_c, int lane, uint4 (&d)[8])
{
    const bool lum = r > 0;
    const int blk = lane >> 3;
    const int n_ok = lum ? bw_y - tx * 64 : bw_c - tx * 32; // blocks of this tile inside the plane
    const uint8_t *base_a, *base_b;                          // pieces 0-3 / 4-7
    if (lum) {
        base_a = reinterpret_cast<const uint8_t *>(coef_y + ((size_t)(ty * 2 + r - 1) * bw_y + tx * 64) * 64);
        base_b = base_a + 4096;
    } else {
        base_a = reinterpret_cast<const uint8_t *>(coef_cb + ((size_t)ty * bw_c + tx * 32) * 64);
        base_b = reinterpret_cast<const uint8_t *>(coef_cr + ((size_t)ty * bw_c + tx * 32) * 64);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int b = (lum ? k * 8 : (k & 3) * 8) + blk;
        if (b < n_ok)
            d[k] = *reinterpret_cast<const uint4 *>((k < 4 ? base_a : base_b) + (k & 3) * 1024 + lane * 16);
    }
}

__global__ __launch_bounds__(256, 3) void decode_image_kernel(
    const int16_t *__restrict__ coef_y, const int16_t *__restrict__ coef_cb,
    const int16_t *__restrict__ coef_cr, int rows, int cols, QuantPair qp,
    uint8_t *__restrict__ bgr, size_t stride, int tiles_x, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerWg * kWaveLds];
    __shared__ __attribute__((aligned(16))) float qtab[128]; // Q as floats: luma, chroma
    if (threadIdx.x < 128)
        qtab[threadIdx.x] = qp.t[threadIdx.x >> 6].q[threadIdx.x & 63];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // uniform: tile arithmetic stays scalar
    const int tile = blockIdx.x * kWavesPerWg + wave;
    if (tile >= n_tiles)
        return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    uint8_t *stage = lds + wave * kWaveLds;
    uint8_t *cbuf = stage + kStageBytes;
    const int x0 = tx * kWaveTileW;
    const int bw_y = cols >> 3, bw_c = cols >> 4;

    // Coefficients of round `r` (r = 0: the tile's 32 Cb + 32 Cr blocks; r = 1, 2: its two
    // rows of 64 Y blocks) as 8 pieces of 16 B per lane: piece k of lane l is bytes
    // [16 l, 16 l + 16) of blocks 8k .. 8k+7 -- one uniform base per plane plus constant
    // offsets.  They are fetched one round ahead, so the HBM latency hides behind the
    // previous round's transform.
    const int blk = lane >> 3;
    uint4 pre[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        pre[k] = make_uint4(0, 0, 0, 0);
    fetch_coefs(0, coef_y, coef_cb, coef_cr, tx, ty, bw_y, bw_c, lane, pre);

#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        const bool luma = it > 0;
        // ---- staged coefficients: 64 blocks x 128 B at a 144 B pitch -----------
        {
            uint8_t *sp = stage + blk * kStagePitch + (lane & 7) * 16;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<uint4 *>(sp + k * 8 * kStagePitch) = pre[k];
        }
        if (it < 2)
            fetch_coefs(it + 1, coef_y, coef_cb, coef_cr, tx, ty, bw_y, bw_c, lane, pre);
        wave_lds_fence();

        // ---- transform: lane = block ---------------------------------------
        const int bcol = luma ? tx * 64 + lane : tx * 32 + (lane & 31);
        const bool valid = bcol < (luma ? bw_y : bw_c);
        if (valid) {
            float v[64];
            stage_read_dequant(stage, lane, v, qtab + (luma ? 0 : 64));
            block_idct(v);
            wave_lds_fence(); // staged coefficients consumed (in-order LDS): Y rows may overwrite them
            uint8_t *dst = luma ? stage + lane * 8 : cbuf + (lane >> 5) * 2048 + (lane & 31) * 8;
            const int pitch = luma ? 512 : 256;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                // std::clamp(block[i][j], 0, 255), encode_decode.cpp:808
                uint32_t lo = jm_pack4_u8(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]);
                uint32_t hi = jm_pack4_u8(v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
                *reinterpret_cast<uint2 *>(dst + r * pitch) = make_uint2(lo, hi);
            }
        }
        wave_lds_fence();

        if (luma) {
            // ---- upsample + YCbCr2RGB for tile rows 8*(it-1) .. +7 ------------
            const int row_base = ty * 16 + (it - 1) * 8;
#pragma unroll 2
            for (int t = 0; t < 8; ++t) {
                const int rp = t >> 1;
                const int g = lane + 64 * (t & 1);
                const int x = x0 + 4 * g;
                if (x < cols) {
                    uint32_t ya = *reinterpret_cast<const uint32_t *>(stage + (2 * rp) * 512 + 4 * g);
                    uint32_t yb = *reinterpret_cast<const uint32_t *>(stage + (2 * rp + 1) * 512 + 4 * g);
                    const int crow = (it - 1) * 4 + rp;
                    uint32_t cbp = *reinterpret_cast<const uint16_t *>(cbuf + crow * 256 + 2 * g);
                    uint32_t crp = *reinterpret_cast<const uint16_t *>(cbuf + 2048 + crow * 256 + 2 * g);
                    // upsampleChroma: (i/2, j/2), encode_decode.cpp:165-170 -- each chroma
                    // sample serves a 2x2 quad, i.e. two pixels in each of the two rows
                    const int cb0 = (int)(cbp & 0xFFu), cb1 = (int)(cbp >> 8);
                    const int cr0 = (int)(crp & 0xFFu), cr1 = (int)(crp >> 8);
                    const JmChromaPk c0 = jm_inv_chroma_pk_fast<0>(cbp, crp), c1 = jm_inv_chroma_pk_fast<1>(cbp, crp);
                    U3 wa, wb; // both rows in one straight-line block, one tie test for the task
                    jm_inv_row4_fast(ya, c0, c1, wa.a, wa.b, wa.c);
                    jm_inv_row4_fast(yb, c0, c1, wb.a, wb.b, wb.c);
                    if (__builtin_expect((c0.tie | c1.tie) < 0, 0)) {
                        jm_inv_row4_exact(ya, cb0, cr0, cb1, cr1, wa.a, wa.b, wa.c);
                        jm_inv_row4_exact(yb, cb0, cr0, cb1, cr1, wb.a, wb.b, wb.c);
                    }
                    uint8_t *p0 = bgr + (size_t)(row_base + 2 * rp) * stride + (size_t)x * 3;
                    store12_stream(p0, wa);
                    store12_stream(p0 + stride, wb);
                }
            }
            wave_lds_fence(); // Y rows consumed before the next round stages over them
        }
    }
}

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

// =============================================================================
// Fused encoder with the ROW pass of the DCT on the 16-bit matrix pipe.
//
// f32 MFMAs execute on the VALU's own lanes (see above), the 16-bit forms do not: they
// run beside other waves' VALU work.  The pixels are integers, exact in f16, so the row
// pass  D1[y][v] = sum_x P[y][x] a_v(x)  can run as v_mfma_f32_16x16x32_f16 without
// losing anything: each constant is split into three f16 terms (33 bits), all products are
// exact, and the sum is accumulated in f32 like the VALU chain's.  The column pass needs
// its real-valued input at f32 precision, so it stays on the VALU (jm_fdct8 in-lane).
//
// One MFMA "unit" = 16 blocks = 128 px x 8 rows:
//   A[m][k]  m = 4*quad + r (row 4h + r of the 32-px quad `quad`), k = 8*blk + x: lane
//            (ga, m) supplies 8 consecutive bytes of one row.  A byte b in the low half of
//            an f16 with a zero exponent IS the subnormal b * 2^-24, which the matrix pipe
//            honours (tools/mfma_f16_layout_probe.hip): one v_perm per two samples.
//   B[k][n]  n = 8*bsel + v: term t of 2^14 a_v(x) where blk == 2*nv + bsel, else 0.
//   D[m][n]  lane (quad, n) gets rows 4h + 0..3 of block 4*quad + 2*nv + bsel, frequency v:
//            after 2 (h) x 2 (nv) x 3 (terms) MFMAs it holds two whole columns D1[0..7][v],
//            runs the column pass on them and owns coefficients (0..7, v) of two blocks.
// Scale: 2^-24 * 2^14 = 2^-10, undone by the column pass's constants (100 * 1024; a power
// of two, so they are the usual constants with another exponent).  The level shift
// (-128 per sample) reaches v = 0 only and is the accumulator's initial value.
// The accumulation order inside a 16-bit MFMA is not documented, so this kernel has no
// bit-exact host replay: it is held to the oracle's bar only (+-1 LSB, <= 8 per million).
// =============================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct XdlTable {
    _Float16 c[3][8][8]; // [term][v][x]: 2^14 a_v(x) = c[0] + c[1] + c[2]
};
constexpr XdlTable make_xdl_table()
{
    XdlTable t{};
    for (int v = 0; v < 8; ++v)
        for (int x = 0; x < 8; ++x) {
            const double a = jm_dct_basis_f64(v, x) * 16384.0;
            const _Float16 h = (_Float16)a;
            const double r1 = a - (double)h;
            const _Float16 m = (_Float16)r1;
            const _Float16 l = (_Float16)(r1 - (double)m);
            t.c[0][v][x] = h, t.c[1][v][x] = m, t.c[2][v][x] = l;
        }
    return t;
}
__device__ const XdlTable kXdlTable = make_xdl_table();

constexpr int kXPitchY = 640;                  // 160 dwords: a row starts 32 banks after the one above
constexpr int kXPitchC = 384;                  // 96 dwords: likewise for the 256-sample chroma rows
constexpr int kXYBytes = 8 * kXPitchY;         // 5120: one strip of 8 luma rows; later the staging area
constexpr int kXCBytes = 16 * kXPitchC;        // 6144: Cb rows 0-7, Cr rows 8-15
constexpr int kWaveLdsX = kXYBytes + kXCBytes; // 11264
static_assert(32 * kStagePitch <= kXYBytes, "staging (32 blocks at the 144 B pitch) must fit in the luma strip");

struct XdlLane {
    f16x8 b[3][2]; // B operands [term][nv]
    f32x4 c0;      // accumulator start: the level shift for v = 0
    int zoff[8];   // staging byte offset of coefficient (u, v) of this lane's block pair
};

// eight bytes -> eight f16 subnormals (byte * 2^-24)
__device__ __forceinline__ f16x8 bytes_as_f16(uint2 raw)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = {__builtin_amdgcn_perm(0u, raw.x, 0x0C010C00u), __builtin_amdgcn_perm(0u, raw.x, 0x0C030C02u),
               __builtin_amdgcn_perm(0u, raw.y, 0x0C010C00u), __builtin_amdgcn_perm(0u, raw.y, 0x0C030C02u)};
    return __builtin_bit_cast(f16x8, w);
}

// NU units (16 blocks each) of one 8-row strip at `src`: row pass on the matrix pipe, column
// pass + quantiser in-lane, 32 blocks at a time staged in zigzag order (144 B pitch) and stored as four
// 1 KiB wave-contiguous pieces.  The staging area may alias `src`: every operand byte is
// in registers before the first staged coefficient is written.
template <int NU>
__device__ __forceinline__ void xdl_transform_strip(const uint8_t *src, int pitch, uint8_t *stage, int lane,
                                                    const XdlLane &k, const float (&qm)[8], uint8_t *dst, int n_ok)
{
    uint2 raw[NU][2];
    const uint8_t *p = src + (lane & 3) * pitch + ((lane >> 2) & 3) * 32 + (lane >> 4) * 8;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            raw[u][h] = *reinterpret_cast<const uint2 *>(p + h * 4 * pitch + u * 128);
    }
    wave_lds_fence();
    const int blk = lane >> 3;
#pragma unroll
    for (int round = 0; round < NU / 2; ++round) {
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            f32x4 acc[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f16x8 a = bytes_as_f16(raw[2 * round + uu][h]);
#pragma unroll
                for (int nv = 0; nv < 2; ++nv) {
                    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, k.b[0][nv], k.c0, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, k.b[1][nv], d, 0, 0, 0);
                    acc[h][nv] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, k.b[2][nv], d, 0, 0, 0);
                }
            }
#pragma unroll
            for (int nv = 0; nv < 2; ++nv) {
                float c[8] = {acc[0][nv][0], acc[0][nv][1], acc[0][nv][2], acc[0][nv][3],
                              acc[1][nv][0], acc[1][nv][1], acc[1][nv][2], acc[1][nv][3]};
                jm_fdct8<102400, false>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float x = (c[u] + JM_MAGIC) - JM_MAGIC;  // round(100 c), :284
                    const float q = jm_fma(x, qm[u], JM_MAGIC);     // roundf(c / Q), :306 (see jm_quantise_block)
                    *reinterpret_cast<uint16_t *>(stage + k.zoff[u] + (2 * nv + 16 * uu) * kStagePitch) = (uint16_t)jm_bits(q);
                }
            }
        }
        wave_lds_fence();
        if (32 * round + 32 <= n_ok) { // the whole round lies inside the plane: reads first, then stores
            uint4 d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                d[i] = *reinterpret_cast<const uint4 *>(stage + (i * 8 + blk) * kStagePitch + (lane & 7) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<uint4 *>(dst + round * 4096 + i * 1024 + lane * 16) = d[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (32 * round + i * 8 + blk < n_ok) {
                    uint4 d = *reinterpret_cast<const uint4 *>(stage + (i * 8 + blk) * kStagePitch + (lane & 7) * 16);
                    *reinterpret_cast<uint4 *>(dst + round * 4096 + i * 1024 + lane * 16) = d;
                }
            }
        }
        wave_lds_fence(); // staged blocks are read (in-order LDS) before the next round overwrites them
    }
}

__global__ __launch_bounds__(256, 3) void encode_image_xdl_kernel(
    const uint8_t *__restrict__ bgr, int rows, int cols, size_t stride, QuantPair qp,
    int16_t *__restrict__ coef_y, int16_t *__restrict__ coef_cb, int16_t *__restrict__ coef_cr,
    int tiles_x, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerWg * kWaveLdsX];
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
    uint8_t *ybuf = lds + wave * kWaveLdsX; // 8 luma rows (one strip); staging area during the transforms
    uint8_t *cbuf = ybuf + kXYBytes;        // Cb rows 0-7, Cr rows 8-15
    const int x0 = tx * kWaveTileW;
    const int bw_y = cols >> 3, bw_c = cols >> 4;

    // matrix-pipe constants of this lane: column n = lane & 15 = 8*bsel + v, K-group lane >> 4 = blk
    XdlLane k;
    {
        const int v = lane & 7, bsel = (lane >> 3) & 1, kg = lane >> 4;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f16x8 term = *reinterpret_cast<const f16x8 *>(kXdlTable.c[t][v]);
            const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int nv = 0; nv < 2; ++nv)
                k.b[t][nv] = kg == 2 * nv + bsel ? term : zero;
        }
        const float lv = v == 0 ? -(float)JM_A0 : 0.0f; // recenterAroundZero (:289-298): -128 * 8 a_0 * 2^-10
        k.c0 = f32x4{lv, lv, lv, lv};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            k.zoff[u] = (4 * kg + bsel) * kStagePitch + 2 * kZZInv[u * 8 + v];
    }

    // ---- colour + 4:2:0: the task scheme of encode_image_kernel ---------------------------
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

#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
#pragma unroll 1
        for (int t0 = 0; t0 < 8; t0 += kRing) {
            U3 next_a[kRing], next_b[kRing];
#pragma unroll
            for (int i = 0; i < kRing; ++i)
                task_load(it * 8 + t0 + kRing + i, next_a[i], next_b[i]);
#pragma unroll
            for (int slot = 0; slot < kRing; ++slot) {
                const int t = t0 + slot;
                const int rp = t >> 1;
                const int g = lane + 64 * (t & 1);
                if (x0 + 4 * g < cols) {
                    float ya[4], yb[4], sum_b[2], sum_r[2];
                    colour_task(ring_a[slot], ring_b[slot], ya, yb, sum_b, sum_r);
                    uint8_t *y0 = ybuf + (2 * rp) * kXPitchY + 4 * g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        y0[j] = (uint8_t)jm_bits(ya[j]);
                        y0[kXPitchY + j] = (uint8_t)jm_bits(yb[j]);
                    }
                    // chromaSubsampling: truncating mean of the quad (:139-149)
                    uint8_t *c0 = cbuf + (it * 4 + rp) * kXPitchC + 2 * g;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        c0[q] = (uint8_t)jm_bits(jm_quad_mean_magic(sum_b[q]));
                        c0[8 * kXPitchC + q] = (uint8_t)jm_bits(jm_quad_mean_magic(sum_r[q]));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kRing; ++i)
                ring_a[i] = next_a[i], ring_b[i] = next_b[i];
        }
        wave_lds_fence();

        // ---- luma strip `it`: 64 blocks ------------------------------------------------------
        float qm[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            qm[u] = qtab[u * 8 + (lane & 7)];
        uint8_t *dst = reinterpret_cast<uint8_t *>(coef_y + ((size_t)(ty * 2 + it) * bw_y + tx * 64) * 64);
        xdl_transform_strip<4>(ybuf, kXPitchY, ybuf, lane, k, qm, dst, bw_y - tx * 64);
    }
    {
        // ---- chroma: 32 blocks of Cb, 32 of Cr -------------------------------------------------
        float qm[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            qm[u] = qtab[64 + u * 8 + (lane & 7)];
        const size_t first = ((size_t)ty * bw_c + tx * 32) * 64;
        xdl_transform_strip<2>(cbuf, kXPitchC, ybuf, lane, k, qm, reinterpret_cast<uint8_t *>(coef_cb + first),
                               bw_c - tx * 32);
        xdl_transform_strip<2>(cbuf + 8 * kXPitchC, kXPitchC, ybuf, lane, k, qm,
                               reinterpret_cast<uint8_t *>(coef_cr + first), bw_c - tx * 32);
    }
}

// --------------------------------------------------------------- plane level
// lane = block, wave = 64 consecutive blocks of one block-row.  Any rows/cols:
// samples outside the plane read as 0 before the level shift (extractBlocks,
// encode_decode.cpp:242-251).
__global__ __launch_bounds__(256) void encode_plane_kernel(
    const uint8_t *__restrict__ plane, int rows, int cols, size_t stride, QuantConsts qc,
    int16_t *__restrict__ coef, int bw, int groups_x, int n_groups)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerWg * kStageBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // uniform: tile arithmetic stays scalar
    const int grp = blockIdx.x * kWavesPerWg + wave;
    if (grp >= n_groups)
        return;
    const int gx = grp % groups_x, by = grp / groups_x;
    uint8_t *stage = lds + wave * kStageBytes;
    const int bx = gx * 64 + lane;
    const bool valid = bx < bw;
    if (valid) {
        float v[64];
        const int r0 = by * 8, c0 = bx * 8;
        const bool interior = (r0 + 8 <= rows) && (c0 + 8 <= cols) && ((stride & 7) == 0) &&
                              ((reinterpret_cast<uintptr_t>(plane) & 7) == 0);
        if (interior) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint2 d = *reinterpret_cast<const uint2 *>(plane + (size_t)(r0 + r) * stride + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[r * 8 + j] = (float)((d.x >> (8 * j)) & 0xFFu);
                    v[r * 8 + 4 + j] = (float)((d.y >> (8 * j)) & 0xFFu);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    v[r * 8 + c] = (r0 + r < rows && c0 + c < cols)
                                       ? (float)plane[(size_t)(r0 + r) * stride + c0 + c]
                                       : 0.0f;
        }
        block_fdct_quant(v, qc.inv100q);
        stage_write_zigzag(v, stage, lane);
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = k * 64 + lane;
        const int b = j >> 3, part = j & 7;
        if (gx * 64 + b < bw) {
            uint4 d = *reinterpret_cast<const uint4 *>(stage + b * kStagePitch + part * 16);
            uint8_t *dst = reinterpret_cast<uint8_t *>(coef + ((size_t)by * bw + gx * 64 + b) * 64);
            *reinterpret_cast<uint4 *>(dst + part * 16) = d;
        }
    }
}

__global__ __launch_bounds__(256) void decode_plane_kernel(
    const int16_t *__restrict__ coef, int rows, int cols, QuantConsts qc,
    uint8_t *__restrict__ plane, size_t stride, int bw, int groups_x, int n_groups)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kWavesPerWg * kStageBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // uniform: tile arithmetic stays scalar
    const int grp = blockIdx.x * kWavesPerWg + wave;
    if (grp >= n_groups)
        return;
    const int gx = grp % groups_x, by = grp / groups_x;
    uint8_t *stage = lds + wave * kStageBytes;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = k * 64 + lane;
        const int b = j >> 3, part = j & 7;
        if (gx * 64 + b < bw) {
            const uint8_t *src = reinterpret_cast<const uint8_t *>(coef + ((size_t)by * bw + gx * 64 + b) * 64);
            *reinterpret_cast<uint4 *>(stage + b * kStagePitch + part * 16) =
                *reinterpret_cast<const uint4 *>(src + part * 16);
        }
    }
    wave_lds_fence();
    const int bx = gx * 64 + lane;
    if (bx < bw) {
        float v[64];
        stage_read_dequant(stage, lane, v, qc.q);
        block_idct(v);
        const int r0 = by * 8, c0 = bx * 8;
        const bool interior = (r0 + 8 <= rows) && (c0 + 8 <= cols) && ((stride & 7) == 0) &&
                              ((reinterpret_cast<uintptr_t>(plane) & 7) == 0);
        if (interior) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint32_t lo = jm_pack4_u8(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]);
                uint32_t hi = jm_pack4_u8(v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
                store8_stream(plane + (size_t)(r0 + r) * stride + c0, make_uint2(lo, hi));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (r0 + r < rows && c0 + c < cols) // encode_decode.cpp:805
                        plane[(size_t)(r0 + r) * stride + c0 + c] = (uint8_t)jm_clamp255(jm_magic_to_int(v[r * 8 + c]));
        }
    }
}

// -------------------------------------------------------- legacy block level
// kernel.cuh boundary: 32-bit in, 32-bit out, natural order.  lane = block.
__global__ __launch_bounds__(256) void fdct_quant_blocks_kernel(
    const int32_t *__restrict__ blocks, size_t nb, QuantConsts qc, float *__restrict__ out)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb)
        return;
    const int4 *src = reinterpret_cast<const int4 *>(blocks + b * 64);
    float v[64];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int4 d = src[k];
        v[4 * k] = (float)d.x, v[4 * k + 1] = (float)d.y, v[4 * k + 2] = (float)d.z, v[4 * k + 3] = (float)d.w;
    }
    block_fdct_quant(v, qc.inv100q);
    float4 *dst = reinterpret_cast<float4 *>(out + b * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) // MAGIC + q -> q (exact; +0 for 0, never -0: kernel.cu:86)
        dst[k] = make_float4(v[4 * k] - JM_MAGIC, v[4 * k + 1] - JM_MAGIC, v[4 * k + 2] - JM_MAGIC,
                             v[4 * k + 3] - JM_MAGIC);
}

__global__ __launch_bounds__(256) void dequant_idct_blocks_kernel(
    const float *__restrict__ coef, size_t nb, QuantConsts qc, int32_t *__restrict__ out)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb)
        return;
    const float4 *src = reinterpret_cast<const float4 *>(coef + b * 64);
    float v[64];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float4 d = src[k];
        v[4 * k] = d.x * qc.q[4 * k], v[4 * k + 1] = d.y * qc.q[4 * k + 1];
        v[4 * k + 2] = d.z * qc.q[4 * k + 2], v[4 * k + 3] = d.w * qc.q[4 * k + 3];
    }
    block_idct(v);
    int4 *dst = reinterpret_cast<int4 *>(out + b * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) // unclamped (the caller clamps, encode_decode.cpp:808)
        dst[k] = make_int4(jm_magic_to_int(v[4 * k]), jm_magic_to_int(v[4 * k + 1]),
                           jm_magic_to_int(v[4 * k + 2]), jm_magic_to_int(v[4 * k + 3]));
}

// ------------------------------------------------------------------ host side
#define fail(...) jpegamd_fail(__VA_ARGS__)
#define HIPCHK(expr) JPEGAMD_HIPCHK(expr)

int make_consts(const int32_t q[64], QuantConsts &c, const char *what)
{
    if (!q)
        return fail("%s: null quantisation table", what);
    for (int i = 0; i < 64; ++i) {
        if (q[i] < 1 || q[i] > 255)
            return fail("%s: quantisation entry %d = %d outside 1..255", what, i, q[i]);
    }
    jm_make_quant_tab(q, c);
    return 0;
}

int check_image(const void *p, int rows, int cols, size_t stride, const char *what)
{
    if (!p)
        return fail("%s: null image pointer", what);
    if (rows <= 0 || cols <= 0 || (rows & 15) || (cols & 15))
        return fail("%s: rows=%d cols=%d must be positive multiples of 16 "
                    "(the reference resizes before this path, encode_decode.cpp:29-37)",
                    what, rows, cols);
    if (stride < (size_t)cols * 3 || (stride & 3) || (reinterpret_cast<uintptr_t>(p) & 3))
        return fail("%s: stride=%zu must be >= 3*cols and, like the base pointer, 4-byte aligned", what, stride);
    return 0;
}

// JPEGAMD_TRANSFORM=mfma runs the forward transform of the fused encoder on the matrix
// cores (encode_image_mfma_kernel).  Measured slower than the VALU kernel on gfx950
// (148 vs 116 us at 8192^2): v_mfma_f32_*_f32 executes on the VALU's own
// lanes there (tools/mfma_overlap_ubench.hip), so it adds to the VALU time instead of
// hiding behind it.  Read on every call.
enum class Transform { valu, mfma_f32, xdl_f16 };
Transform transform_choice()
{
    const char *e = getenv("JPEGAMD_TRANSFORM");
    if (e && strcmp(e, "mfma") == 0)
        return Transform::mfma_f32;
    if (e && strcmp(e, "xdl") == 0)
        return Transform::xdl_f16;
    return Transform::valu;
}

struct DevBuf { // host-pointer entry points: per-call device memory
    void *p = nullptr;
    ~DevBuf()
    {
        if (p)
            (void)hipFree(p);
    }
};

} // namespace

static thread_local char g_err[512] = "";

int jpegamd_fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" {

int jpegamd_abi_version(void) { return 2; }

const char *jpegamd_last_error(void) { return g_err; }

int jpegamd_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail("no HIP device available (%s); libjpeg_amd has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail("device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return 0;
}

// the two fused launches; with events, the kernel's own start and stop time stamps (the
// dispatch packet's, what rocprofv3 reports too) are recorded into them
static int launch_encode_image(const uint8_t *d_bgr, int rows, int cols, size_t stride, const int32_t q_luma[64],
                               const int32_t q_chroma[64], int16_t *d_coef_y, int16_t *d_coef_cb,
                               int16_t *d_coef_cr, void *stream, hipEvent_t start, hipEvent_t stop)
{
    QuantPair qp;
    if (check_image(d_bgr, rows, cols, stride, "encode_image") || make_consts(q_luma, qp.t[0], "encode_image") ||
        make_consts(q_chroma, qp.t[1], "encode_image"))
        return 1;
    if (!d_coef_y || !d_coef_cb || !d_coef_cr)
        return fail("encode_image: null coefficient pointer");
    const int tiles_x = (cols + kWaveTileW - 1) / kWaveTileW;
    const int n_tiles = tiles_x * (rows / 16);
    const int grid = (n_tiles + kWavesPerWg - 1) / kWavesPerWg;
    const Transform tr = transform_choice();
    auto kernel = tr == Transform::mfma_f32 ? encode_image_mfma_kernel
                  : tr == Transform::xdl_f16 ? encode_image_xdl_kernel
                                             : encode_image_kernel;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, start, stop, 0,
                          d_bgr, rows, cols, stride, qp, d_coef_y, d_coef_cb, d_coef_cr, tiles_x, n_tiles);
    HIPCHK(hipGetLastError());
    return 0;
}

static int launch_decode_image(const int16_t *d_coef_y, const int16_t *d_coef_cb, const int16_t *d_coef_cr, int rows,
                               int cols, const int32_t q_luma[64], const int32_t q_chroma[64], uint8_t *d_bgr,
                               size_t stride, void *stream, hipEvent_t start, hipEvent_t stop)
{
    QuantPair qp;
    if (check_image(d_bgr, rows, cols, stride, "decode_image") || make_consts(q_luma, qp.t[0], "decode_image") ||
        make_consts(q_chroma, qp.t[1], "decode_image"))
        return 1;
    if (!d_coef_y || !d_coef_cb || !d_coef_cr)
        return fail("decode_image: null coefficient pointer");
    const int tiles_x = (cols + kWaveTileW - 1) / kWaveTileW;
    const int n_tiles = tiles_x * (rows / 16);
    const int grid = (n_tiles + kWavesPerWg - 1) / kWavesPerWg;
    hipExtLaunchKernelGGL(decode_image_kernel, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, start, stop,
                          0, d_coef_y, d_coef_cb, d_coef_cr, rows, cols, qp, d_bgr, stride, tiles_x, n_tiles);
    HIPCHK(hipGetLastError());
    return 0;
}

int jpegamd_encode_image_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride,
                             const int32_t q_luma[64], const int32_t q_chroma[64],
                             int16_t *d_coef_y, int16_t *d_coef_cb, int16_t *d_coef_cr, void *stream)
{
    return launch_encode_image(d_bgr, rows, cols, stride, q_luma, q_chroma, d_coef_y, d_coef_cb, d_coef_cr, stream,
                               nullptr, nullptr);
}

int jpegamd_decode_image_dev(const int16_t *d_coef_y, const int16_t *d_coef_cb, const int16_t *d_coef_cr,
                             int rows, int cols, const int32_t q_luma[64], const int32_t q_chroma[64],
                             uint8_t *d_bgr, size_t stride, void *stream)
{
    return launch_decode_image(d_coef_y, d_coef_cb, d_coef_cr, rows, cols, q_luma, q_chroma, d_bgr, stride, stream,
                               nullptr, nullptr);
}

int jpegamd_encode_plane_dev(const uint8_t *d_plane, int rows, int cols, size_t stride,
                             const int32_t q[64], int16_t *d_coef_zz, void *stream)
{
    QuantConsts qc;
    if (make_consts(q, qc, "encode_plane"))
        return 1;
    if (!d_plane || !d_coef_zz || rows <= 0 || cols <= 0 || stride < (size_t)cols)
        return fail("encode_plane: bad arguments (rows=%d cols=%d stride=%zu)", rows, cols, stride);
    const int bw = (cols + 7) / 8, bh = (rows + 7) / 8;
    const int groups_x = (bw + 63) / 64;
    const long n_groups = (long)groups_x * bh;
    const int grid = (int)((n_groups + kWavesPerWg - 1) / kWavesPerWg);
    hipLaunchKernelGGL(encode_plane_kernel, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream,
                       d_plane, rows, cols, stride, qc, d_coef_zz, bw, groups_x, (int)n_groups);
    HIPCHK(hipGetLastError());
    return 0;
}

int jpegamd_decode_plane_dev(const int16_t *d_coef_zz, int rows, int cols, const int32_t q[64],
                             uint8_t *d_plane, size_t stride, void *stream)
{
    QuantConsts qc;
    if (make_consts(q, qc, "decode_plane"))
        return 1;
    if (!d_plane || !d_coef_zz || rows <= 0 || cols <= 0 || stride < (size_t)cols)
        return fail("decode_plane: bad arguments (rows=%d cols=%d stride=%zu)", rows, cols, stride);
    const int bw = (cols + 7) / 8, bh = (rows + 7) / 8;
    const int groups_x = (bw + 63) / 64;
    const long n_groups = (long)groups_x * bh;
    const int grid = (int)((n_groups + kWavesPerWg - 1) / kWavesPerWg);
    hipLaunchKernelGGL(decode_plane_kernel, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream,
                       d_coef_zz, rows, cols, qc, d_plane, stride, bw, groups_x, (int)n_groups);
    HIPCHK(hipGetLastError());
    return 0;
}

int jpegamd_fdct_quant_blocks_dev(const int32_t *d_blocks, size_t nb, const int32_t q[64], float *d_out,
                                  void *stream)
{
    QuantConsts qc;
    if (make_consts(q, qc, "fdct_quant_blocks"))
        return 1;
    if (nb == 0)
        return 0;
    if (!d_blocks || !d_out)
        return fail("fdct_quant_blocks: null pointer");
    if (nb > (size_t)0x7FFFFFFF * 64)
        return fail("fdct_quant_blocks: nb=%zu too large", nb);
    hipLaunchKernelGGL(fdct_quant_blocks_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_blocks, nb, qc, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

int jpegamd_dequant_idct_blocks_dev(const float *d_coef, size_t nb, const int32_t q[64], int32_t *d_out,
                                    void *stream)
{
    QuantConsts qc;
    if (make_consts(q, qc, "dequant_idct_blocks"))
        return 1;
    if (nb == 0)
        return 0;
    if (!d_coef || !d_out)
        return fail("dequant_idct_blocks: null pointer");
    if (nb > (size_t)0x7FFFFFFF * 64)
        return fail("dequant_idct_blocks: nb=%zu too large", nb);
    hipLaunchKernelGGL(dequant_idct_blocks_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_coef, nb, qc, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

int jpegamd_fdct_quant_blocks(const int32_t *blocks, size_t nb, const int32_t q[64], float *out)
{
    if (nb == 0)
        return 0;
    if (!blocks || !out)
        return fail("fdct_quant_blocks: null pointer");
    DevBuf in, res;
    HIPCHK(hipMalloc(&in.p, nb * 64 * sizeof(int32_t)));
    HIPCHK(hipMalloc(&res.p, nb * 64 * sizeof(float)));
    HIPCHK(hipMemcpy(in.p, blocks, nb * 64 * sizeof(int32_t), hipMemcpyHostToDevice));
    if (jpegamd_fdct_quant_blocks_dev((const int32_t *)in.p, nb, q, (float *)res.p, nullptr))
        return 1;
    HIPCHK(hipMemcpy(out, res.p, nb * 64 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int jpegamd_dequant_idct_blocks(const float *coef, size_t nb, const int32_t q[64], int32_t *out)
{
    if (nb == 0)
        return 0;
    if (!coef || !out)
        return fail("dequant_idct_blocks: null pointer");
    DevBuf in, res;
    HIPCHK(hipMalloc(&in.p, nb * 64 * sizeof(float)));
    HIPCHK(hipMalloc(&res.p, nb * 64 * sizeof(int32_t)));
    HIPCHK(hipMemcpy(in.p, coef, nb * 64 * sizeof(float), hipMemcpyHostToDevice));
    if (jpegamd_dequant_idct_blocks_dev((const float *)in.p, nb, q, (int32_t *)res.p, nullptr))
        return 1;
    HIPCHK(hipMemcpy(out, res.p, nb * 64 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int jpegamd_encode_image(const uint8_t *bgr, int rows, int cols, size_t stride,
                         const int32_t q_luma[64], const int32_t q_chroma[64],
                         int16_t *coef_y, int16_t *coef_cb, int16_t *coef_cr)
{
    if (!bgr || !coef_y || !coef_cb || !coef_cr)
        return fail("encode_image: null pointer");
    if (rows <= 0 || cols <= 0 || (rows & 15) || (cols & 15) || stride < (size_t)cols * 3)
        return fail("encode_image: rows=%d cols=%d stride=%zu unsupported", rows, cols, stride);
    const size_t dstride = (size_t)cols * 3; // packed on the device
    const size_t ny = (size_t)rows * cols, nc = ny / 4;
    DevBuf img, cy, cb, cr;
    HIPCHK(hipMalloc(&img.p, dstride * rows));
    HIPCHK(hipMalloc(&cy.p, ny * 2));
    HIPCHK(hipMalloc(&cb.p, nc * 2));
    HIPCHK(hipMalloc(&cr.p, nc * 2));
    if (jpegamd_staged_copy(img.p, const_cast<uint8_t *>(bgr), stride, dstride, rows, true))
        return 1;
    if (jpegamd_encode_image_dev((const uint8_t *)img.p, rows, cols, dstride, q_luma, q_chroma,
                                 (int16_t *)cy.p, (int16_t *)cb.p, (int16_t *)cr.p, nullptr))
        return 1;
    HIPCHK(hipDeviceSynchronize()); // the kernel ran on the null stream, the copier uses its own
    if (jpegamd_staged_copy_linear(cy.p, coef_y, ny * 2, false) || jpegamd_staged_copy_linear(cb.p, coef_cb, nc * 2, false) ||
        jpegamd_staged_copy_linear(cr.p, coef_cr, nc * 2, false))
        return 1;
    return 0;
}

int jpegamd_decode_image(const int16_t *coef_y, const int16_t *coef_cb, const int16_t *coef_cr,
                         int rows, int cols, const int32_t q_luma[64], const int32_t q_chroma[64],
                         uint8_t *bgr, size_t stride)
{
    if (!bgr || !coef_y || !coef_cb || !coef_cr)
        return fail("decode_image: null pointer");
    if (rows <= 0 || cols <= 0 || (rows & 15) || (cols & 15) || stride < (size_t)cols * 3)
        return fail("decode_image: rows=%d cols=%d stride=%zu unsupported", rows, cols, stride);
    const size_t dstride = (size_t)cols * 3;
    const size_t ny = (size_t)rows * cols, nc = ny / 4;
    DevBuf img, cy, cb, cr;
    HIPCHK(hipMalloc(&img.p, dstride * rows));
    HIPCHK(hipMalloc(&cy.p, ny * 2));
    HIPCHK(hipMalloc(&cb.p, nc * 2));
    HIPCHK(hipMalloc(&cr.p, nc * 2));
    if (jpegamd_staged_copy_linear(cy.p, const_cast<int16_t *>(coef_y), ny * 2, true) ||
        jpegamd_staged_copy_linear(cb.p, const_cast<int16_t *>(coef_cb), nc * 2, true) ||
        jpegamd_staged_copy_linear(cr.p, const_cast<int16_t *>(coef_cr), nc * 2, true))
        return 1;
    if (jpegamd_decode_image_dev((const int16_t *)cy.p, (const int16_t *)cb.p, (const int16_t *)cr.p, rows, cols,
                                 q_luma, q_chroma, (uint8_t *)img.p, dstride, nullptr))
        return 1;
    HIPCHK(hipDeviceSynchronize()); // the kernel ran on the null stream, the copier uses its own
    if (jpegamd_staged_copy(img.p, bgr, stride, dstride, rows, false))
        return 1;
    return 0;
}

int jpegamd_time_image_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride,
                           const int32_t q_luma[64], const int32_t q_chroma[64],
                           int16_t *d_coef_y, int16_t *d_coef_cb, int16_t *d_coef_cr,
                           uint8_t *d_bgr_out, int iters, void *stream, float *enc_ms, float *dec_ms)
{
    if (iters <= 0 || iters > 4096 || !enc_ms || !dec_ms)
        return fail("time_image: bad arguments");
    // The kernels are timed IN the sequence they run in (encode, decode, encode, ...) from
    // their own start/stop time stamps, without synchronising until the end.  Timed in
    // separate back-to-back loops each kernel finds part of its 201 MB input still in the
    // 256 MB last-level cache and looks 5-12 % faster than it is in the real sequence.
    std::vector<hipEvent_t> ev(4 * (size_t)iters, nullptr);
    int rc = 0;
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess)
            rc = fail("time_image: hipEventCreate failed");
    for (int i = 0; i < iters && !rc; ++i) {
        rc = launch_encode_image(d_bgr, rows, cols, stride, q_luma, q_chroma, d_coef_y, d_coef_cb, d_coef_cr, stream,
                                 ev[4 * i], ev[4 * i + 1]);
        if (!rc)
            rc = launch_decode_image(d_coef_y, d_coef_cb, d_coef_cr, rows, cols, q_luma, q_chroma, d_bgr_out, stride,
                                     stream, ev[4 * i + 2], ev[4 * i + 3]);
    }
    if (!rc && hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        rc = fail("time_image: hipStreamSynchronize failed");
    double enc = 0, dec = 0;
    for (int i = 0; i < iters && !rc; ++i) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, ev[4 * i], ev[4 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&b, ev[4 * i + 2], ev[4 * i + 3]) != hipSuccess)
            rc = fail("time_image: hipEventElapsedTime failed");
        enc += a, dec += b;
    }
    for (auto &e : ev)
        if (e)
            (void)hipEventDestroy(e);
    if (!rc)
        *enc_ms = (float)(enc / iters), *dec_ms = (float)(dec / iters);
    return rc;
}

} // extern "C"
