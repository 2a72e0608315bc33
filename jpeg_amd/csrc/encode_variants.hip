// encode_variants.hip -- the two opt-in encoders whose transform runs (partly) on the matrix
// pipes (JPEGAMD_TRANSFORM=mfma / xdl).  Same tile, colour stage, parity tests and C ABI as
// encode_image_kernel in jpeg_kernels.hip; kept for measurement.
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
                    const float x = jm_rint(c[u]);                  // round(100 c), :284
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
    int tile = blockIdx.x * kWavesPerWg + wave;
    if (tile >= n_tiles)
        return; // whole wave leaves together; nothing else is shared between waves
    // persistent like encode_image_kernel (jpeg_kernels.hip): tiles tile, tile + tile_step, ..., the first
    // four tasks of the next tile fetched during the current one's last colour chunk
    const int tile_step = gridDim.x * kWavesPerWg;
    const int skew = (tile_step % tiles_x) == 0 ? 1 : 0;
    int round_no = 0;
    int tx = tile % tiles_x, ty = tile / tiles_x;
    uint8_t *ybuf = lds + wave * kWaveLdsX; // 8 luma rows (one strip); staging area during the transforms
    uint8_t *cbuf = ybuf + kXYBytes;        // Cb rows 0-7, Cr rows 8-15
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
    auto lane_offset = [&](int tile_x, int h) {
        int x = tile_x * kWaveTileW + 4 * (lane + 64 * h);
        x = x < cols - 4 ? x : cols - 4;
        return 3u * (uint32_t)x;
    };
    uint32_t lane_off0 = lane_offset(tx, 0), lane_off1 = lane_offset(tx, 1), next_off0 = 0, next_off1 = 0;
    int next_ty = ty;
    bool has_next = false;
    auto task_load = [&](int T, int odd, U3 &a, U3 &b) { // odd == T & 1, a compile-time constant at every call
        const bool ahead = T >= 16 && has_next;
        const int row = ahead ? next_ty * 16 + 2 * ((T - 16) >> 1) : ty * 16 + 2 * ((T < 16 ? T : 14 + odd) >> 1);
        const uint32_t off = odd ? (ahead ? next_off1 : lane_off1) : (ahead ? next_off0 : lane_off0);
        const uint8_t *rowp = bgr + (size_t)row * stride;
        a = *reinterpret_cast<const U3 *>(rowp + off);
        b = *reinterpret_cast<const U3 *>(rowp + stride + off);
    };
    U3 ring_a[kRing], ring_b[kRing];
#pragma unroll
    for (int i = 0; i < kRing; ++i)
        task_load(i, i & 1, ring_a[i], ring_b[i]);

#pragma unroll 1
    for (;;) {
    const int x0 = tx * kWaveTileW;
    const int next_tile = tile + tile_step;
    has_next = next_tile < n_tiles;
    const int next_tx = has_next ? (next_tile % tiles_x + skew * (round_no + 1)) % tiles_x : tx;
    next_ty = has_next ? next_tile / tiles_x : ty;
    next_off0 = lane_offset(next_tx, 0), next_off1 = lane_offset(next_tx, 1);

#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
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
                    colour_task(ring_a[slot], ring_b[slot], ya, yb, sum_b, sum_r);
                    volatile lds_u8 *y0 = (volatile lds_u8 *)(ybuf + (2 * rp) * kXPitchY + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        y0[j] = (uint8_t)jm_bits(ya[j]);
                        y0[kXPitchY + j] = (uint8_t)jm_bits(yb[j]);
                    }
                    // chromaSubsampling: truncating mean of the quad (:139-149)
                    volatile lds_u8 *c0 = (volatile lds_u8 *)(cbuf + (it * 4 + rp) * kXPitchC + 2 * g);
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
    if (!has_next)
        break;
    tile = next_tile, tx = next_tx, ty = next_ty, ++round_no;
    lane_off0 = next_off0, lane_off1 = next_off1;
    } // tiles of this wave
}

} // namespace

int jpegamd_launch_encode_variant(int variant, int grid, void *stream, hipEvent_t start, hipEvent_t stop,
                                  const uint8_t *d_bgr, int rows, int cols, size_t stride, const QuantPair &qp,
                                  int16_t *d_coef_y, int16_t *d_coef_cb, int16_t *d_coef_cr, int tiles_x, int n_tiles)
{
    auto kernel = variant == 1 ? encode_image_mfma_kernel : encode_image_xdl_kernel;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, start, stop, 0, d_bgr, rows,
                          cols, stride, qp, d_coef_y, d_coef_cb, d_coef_cr, tiles_x, n_tiles);
    JPEGAMD_HIPCHK(hipGetLastError());
    return 0;
}
