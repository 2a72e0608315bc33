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
#include <type_traits>
// developer switch, timing only (results wrong at exact ties): 1 = candidates found but not re-evaluated, 2 = no candidate test at all
// (profiles/r03_a_matrix_pipe_rowpass.txt, section 7)
#ifndef JPEGAMD_ABLATE_TIES
#define JPEGAMD_ABLATE_TIES 0
#endif
#include <vector>

#include "../../include/jpeg_amd.h"
#include "host_common.h"

#include "jpeg_device_common.h"

// variant 1: both passes as f32 MFMA, 2: row pass as f16 MFMA (encode_variants.hip)
__attribute__((visibility("hidden"))) int jpegamd_launch_encode_variant(
    int variant, int grid, void *stream, hipEvent_t start, hipEvent_t stop, const uint8_t *d_bgr, int rows, int cols,
    size_t stride, const QuantPair &qp, int16_t *d_coef_y, int16_t *d_coef_cb, int16_t *d_coef_cr, int tiles_x,
    int n_tiles);

namespace {

// ------------------------------------------- row pass on the 16-bit matrix pipe
// v_mfma_f32_16x16x32_f16 runs beside other waves' VALU work (the f32 forms do not: DESIGN.md 3),
// and the encoder is bound by VALU issue, so the ROW pass of the forward DCT can leave the VALU:
//   D[m][n] = sum_k A[m][k] B[k][n]      A: lane (m = l & 15, g = l >> 4) holds k = 8g .. 8g+7
//                                        B: lane (n = l & 15, g = l >> 4) holds k = 8g .. 8g+7
//                                        D: lane (n = l & 15, g = l >> 4) holds m = 4g .. 4g+3
// B is DATA: lane l supplies the eight pixels of row y of ITS OWN block (lane = block, as in the
// VALU transform) -- a byte in the low half of an f16 with a zero exponent is the subnormal
// b * 2^-24, which the matrix pipe honours (tools/mfma_f16_layout_probe.hip), so an operand costs
// one v_perm per two samples instead of one conversion per sample.  A is CONSTANT and block
// diagonal: row m = 4g' + i is a_v(x) for v = 4*vh + i in the K-slots of lane group g' and zero
// elsewhere, so D[4g' + i][n] = sum_x a_v(x) P[block (n, g')][y][x]: lane (n, g') receives four
// frequencies of row y of its own block, and after 8 rows x 2 halves x 3 terms it holds all 64
// row-pass outputs D1[y][v] in the registers the column pass (jm_fdct8, in-lane) works on.
// Nothing crosses lanes and the staging is the VALU kernel's.  Each constant 2^14 a_v(x) is split
// into three f16 terms (33 bits); products are exact, the sum is accumulated in f32.  Scale:
// 2^-24 * 2^14 = 2^-10, undone by the column pass's constants (100 * 1024, a power of two times
// the usual ones).  The level shift reaches v = 0 only and is that accumulator's initial value.
// The accumulation order inside a 16-bit MFMA is not documented: this path is held to the oracle's
// bar (+-1 LSB, <= 8 per million), not to the host model bit for bit.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RowPassTable {
    _Float16 c[3][8][8]; // [term][v][x]: 2^14 a_v(x) = c[0] + c[1] + c[2]
};
constexpr RowPassTable make_row_pass_table()
{
    RowPassTable t{};
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
__device__ const RowPassTable kRowPassTable = make_row_pass_table();

#ifdef JPEGAMD_ROWPASS_A_FROM_MEMORY
// experiment: the A operands are re-read from the table (L2) before every transform instead of being
// held in 24 registers across the colour stage
struct RowPassLane {
    uint32_t off[2]; // byte offset of this lane's rows in kRowPassTable (or of the zero row), per v half
};
__device__ const RowPassTable kRowPassZero{};
__device__ __forceinline__ RowPassLane make_row_pass_lane(int lane)
{
    RowPassLane k;
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int vh = 0; vh < 2; ++vh)
        k.off[vh] = (m >> 2) == g ? (uint32_t)((4 * vh + (m & 3)) * 16) : 0x80000000u;
    return k;
}
__device__ __forceinline__ f16x8 row_pass_operand(const RowPassLane &k, int t, int vh)
{
    uint32_t off = k.off[vh];
    asm volatile("" : "+v"(off)); // (not loop invariant for the compiler: it would hoist the loads and pin the registers again)
    const uint8_t *base = (off & 0x80000000u) ? reinterpret_cast<const uint8_t *>(&kRowPassZero)
                                              : reinterpret_cast<const uint8_t *>(&kRowPassTable) + off;
    return *reinterpret_cast<const f16x8 *>(base + 128 * t);
}
#else
struct RowPassLane {
    f16x8 a[3][2]; // A operands [term][v half]
};

__device__ __forceinline__ RowPassLane make_row_pass_lane(int lane)
{
    RowPassLane k;
    const int m = lane & 15, g = lane >> 4;
    const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh) {
            const f16x8 term = *reinterpret_cast<const f16x8 *>(kRowPassTable.c[t][4 * vh + (m & 3)]);
            k.a[t][vh] = (m >> 2) == g ? term : zero;
        }
    return k;
}
__device__ __forceinline__ f16x8 row_pass_operand(const RowPassLane &k, int t, int vh) { return k.a[t][vh]; }
#endif

// eight bytes -> eight f16 subnormals (byte * 2^-24)
__device__ __forceinline__ f16x8 bytes_as_f16(uint2 raw)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = {__builtin_amdgcn_perm(0u, raw.x, 0x0C010C00u), __builtin_amdgcn_perm(0u, raw.x, 0x0C030C02u),
               __builtin_amdgcn_perm(0u, raw.y, 0x0C010C00u), __builtin_amdgcn_perm(0u, raw.y, 0x0C030C02u)};
    return __builtin_bit_cast(f16x8, w);
}

// the eight rows of the lane's block at src (pitch bytes apart) -> v[y * 8 + f] = 2^-10 * row-pass
// output of row y, horizontal frequency f.  Runs for ALL lanes of the wave (a matrix instruction
// is a wave-level operation); lanes whose block lies outside the image read whatever the LDS
// holds -- bytes, hence finite -- and only their own accumulators see it.
__device__ __forceinline__ void rows_fdct_matrix_pipe(const uint8_t *src, int pitch, const RowPassLane &k, float (&v)[64])
{
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    // Four rows at a time, term by term over their 8 accumulators: a dependent 16x16x32 waits for its
    // predecessor, 7 independent ones in between hide that (and only 4 operands are live).  (Rows in
    // the order 0, 7, 1, 6 | 2, 5, 3, 4, so that the column pass's first butterflies can be issued
    // under the second half's matrix instructions, measured no faster: profiles/r03_a_*.)
    constexpr int kOrder[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    f16x8 a[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int vh = 0; vh < 2; ++vh)
            a[t][vh] = row_pass_operand(k, t, vh);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f16x8 b[4];
#pragma unroll
        for (int y = 0; y < 4; ++y)
            b[y] = bytes_as_f16(*reinterpret_cast<const uint2 *>(src + kOrder[4 * h + y] * pitch));
        f32x4 acc[4][2];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int vh = 0; vh < 2; ++vh)
                    acc[y][vh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][vh], b[y], t == 0 ? zero : acc[y][vh], 0, 0, 0);
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int vh = 0; vh < 2; ++vh)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[kOrder[4 * h + y] * 8 + 4 * vh + i] = acc[y][vh][i];
    }
}

// ------------------------------------------------ typed loads for the colour stage
// The texture path converts for free what the VALU pays 4.2 cycles per byte for: a typed buffer load with
// the format 8_8_8_8 / USCALED returns four bytes as four floats 0.0 .. 255.0 (tools/typed_load_probe.hip),
// so the three v_cvt_f32_ubyte per pixel of the colour stage (384 of a tile's 6300 instructions, and of the
// slow class: profiles/r03_a section 5) disappear.  Floats cost four times the registers of the bytes, so
// only ONE task is loaded ahead where the byte path has four to eight (2 rows x 12 floats per lane and task, two
// buffers of 24 registers; with three waves per SIMD a wave reaches its next task about 2 us later).
// clang has no builtin for format loads: they are inline assembly, the compiler does not know them as
// memory operations and the wait is explicit (`s_waitcnt vmcnt(N)` before a task's floats are read, N = the loads
// of the tasks issued after this one: all older memory operations have completed; coefficient stores issued
// since only make the wait stricter).  Between a
// load and its wait the registers are live asm operands, so the compiler keeps its hands off them; the
// build checks that it has placed no copy of them in between (tools/check_typed_loads.py).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
struct TaskFloats {
    f32x4 a[3], b[3]; // rows a and b of a task: 12 bytes each as floats (B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3)
};
// buffers of 24 registers: 2 = one task ahead (the registers the byte path's ring takes); 4 = three ahead, which
// spills 49-62 registers next to the transform (measured: see DESIGN.md)
#ifndef JPEGAMD_TYPED_BUFFERS
#define JPEGAMD_TYPED_BUFFERS 2
#endif
constexpr int kTypedBufs = JPEGAMD_TYPED_BUFFERS;
__device__ __forceinline__ u32x4_t make_bgr_resource(const uint8_t *base, uint32_t bytes)
{
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    // word 3: dst_sel x, y, z, w = 4, 5, 6, 7; NUM_FORMAT USCALED (2); DATA_FORMAT 8_8_8_8 (10); type 0 = buffer
    return u32x4_t{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 4u | (5u << 3) | (6u << 6) | (7u << 9) | (2u << 12) | (10u << 15)};
}
__device__ __forceinline__ void typed_load_row(f32x4 (&f)[3], uint32_t voff, u32x4_t rsrc, uint32_t soff)
{
    asm volatile("tbuffer_load_format_xyzw %0, %3, %4, %5 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen\n\t"
                 "tbuffer_load_format_xyzw %1, %3, %4, %5 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:4\n\t"
                 "tbuffer_load_format_xyzw %2, %3, %4, %5 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_USCALED] offen offset:8"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2])
                 : "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
// the task's floats may be read behind this
__device__ __forceinline__ void wait_task_floats(TaskFloats &t)
{
    asm volatile("s_waitcnt vmcnt(%6)"
                 : "+v"(t.a[0]), "+v"(t.a[1]), "+v"(t.a[2]), "+v"(t.b[0]), "+v"(t.b[1]), "+v"(t.b[2])
                 : "n"(6 * (kTypedBufs - 2)) // the tasks issued after this one: six loads each
                 : "memory");
}
// channel c of pixel j of a row held as floats
template <int J, int C>
__device__ __forceinline__ float row_float(const f32x4 (&f)[3])
{
    constexpr int idx = 3 * J + C;
    return f[idx >> 2][idx & 3];
}

// ------------------------------------------------------------- fused encoder
// grid: one wave per 512x16 tile, 4 waves per workgroup.
// kCounts: also write, per block, the number of (value, run) pairs run_length_encode
// (encode_decode.cpp:350-373) will make of it -- the lane has the 64 quantised values in
// registers, so the run-length stage need not read the planes a second time for its offsets.
struct CountPlanes {
    uint8_t *y, *cb, *cr; // one byte per block, raster order like the coefficient planes
};
// kDump: stop after the colour stage and write the Y, Cb, Cr planes (RGB2YCbCr +
// chromaSubsampling on their own: jpegamd_colour_planes_dev); `counts` then names the planes.
// kPipe: 0 = both DCT passes on the VALU; 1 = the row pass on the 16-bit matrix pipe (above)
// kTyped: the colour stage reads its pixels as floats through typed buffer loads (above)
template <bool kCounts, bool kDump, int kPipe = 0, bool kTyped = false>
__global__ __launch_bounds__(256, 3) void encode_image_kernel(
    const uint8_t *__restrict__ bgr, int rows, int cols, size_t stride, QuantPair qp,
    int16_t *__restrict__ coef_y, int16_t *__restrict__ coef_cb, int16_t *__restrict__ coef_cr,
    int tiles_x, int n_tiles, CountPlanes counts)
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
    int tile = blockIdx.x * kWavesPerWg + wave;
    if (tile >= n_tiles)
        return; // whole wave leaves together; nothing else is shared between waves
    // The grid is PERSISTENT when the host launches fewer waves than tiles: a wave then takes
    // tiles tile, tile + tile_step, ... and fetches the first rows of its next tile while the
    // current one is in its chroma round, so neither a workgroup launch nor a cold first fetch
    // sits between two tiles.  Every wave leaves the loop when its next tile is beyond the image.
    const int tile_step = gridDim.x * kWavesPerWg;
    // When the wave count is a multiple of the tiles per row a wave would meet the same tile
    // column every time -- and the right-most column may be a partial, cheaper tile (3840 = 7.5
    // tiles): the column is therefore rotated by the round number (a rotation within a tile row
    // keeps the assignment one to one).
    const int skew = (tile_step % tiles_x) == 0 ? 1 : 0;
    int round_no = 0;
    int tx = tile % tiles_x, ty = tile / tiles_x;
    uint8_t *stage = lds + wave * kWaveLds; // Y rows (8 x 512) alias the staging area
    uint8_t *cbuf = stage + kStageBytes;    // Cb: 8 x 256, Cr: 8 x 256
    const int bw_y = cols >> 3, bw_c = cols >> 4;

    // BGR rows are double-buffered in registers four tasks at a time: while the four tasks
    // of one buffer are converted, the loads of the next four are in flight (HBM latency
    // under load is several tasks long).  The inner four tasks are unrolled so that every
    // slot is a fixed register; the buffers swap by moves once per four tasks, by which
    // time the loads issued four tasks earlier have landed.  Task T = 8*it + t covers
    // 2 rows x 4 pixels per lane: row pair rp = t>>1, pixel group g = lane + 64*(t&1).
    // Loads are unconditional: lanes beyond the image edge re-read its last pixel group;
    // tasks 16..19 are tasks 0..3 of the wave's NEXT tile (or, without one, tasks 14 and 15 again).
    constexpr int kRing = 4;
    // The row base is uniform (scalar registers) and the lane's byte offset within a row
    // takes only two values, so a load costs no vector arithmetic: saddr + 32-bit voffset.
    auto lane_offset = [&](int tile_x, int h) { // (separate scalars: arrays of them end up in scratch)
        int x = tile_x * kWaveTileW + 4 * (lane + 64 * h);
        x = x < cols - 4 ? x : cols - 4;
        return 3u * (uint32_t)x;
    };
    uint32_t lane_off0 = lane_offset(tx, 0), lane_off1 = lane_offset(tx, 1), next_off0 = 0, next_off1 = 0;
    int next_ty = ty; // (set per tile below)
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
    if constexpr (!kTyped) {
#pragma unroll
        for (int i = 0; i < kRing; ++i)
            task_load(i, i & 1, ring_a[i], ring_b[i]);
    }
    // typed path: byte offset of task T's row pair from the image base (the buffer's soffset) and the lane's
    // offset within the row; tasks 16.. are the next tile's (or, without one, tasks 14 and 15 again)
    u32x4_t rsrc{};
    TaskFloats tf[kTypedBufs];
    auto task_row_bytes = [&](int T, int odd) {
        const bool ahead = T >= 16 && has_next;
        const int row = ahead ? next_ty * 16 + 2 * ((T - 16) >> 1) : ty * 16 + 2 * ((T < 16 ? T : 14 + odd) >> 1);
        return (uint32_t)row * (uint32_t)stride;
    };
    auto task_lane_off = [&](int T, int odd) {
        const bool ahead = T >= 16 && has_next;
        return odd ? (ahead ? next_off1 : lane_off1) : (ahead ? next_off0 : lane_off0);
    };
    auto typed_issue = [&](int T, int odd, TaskFloats &t) {
        const uint32_t so = task_row_bytes(T, odd), vo = task_lane_off(T, odd);
        typed_load_row(t.a, vo, rsrc, so);
        typed_load_row(t.b, vo, rsrc, so + (uint32_t)stride);
    };
    if constexpr (kTyped) {
        rsrc = make_bgr_resource(bgr, (uint32_t)((size_t)rows * stride)); // (the host takes this path below 4 GiB only)
#pragma unroll
        for (int i = 0; i + 1 < kTypedBufs; ++i)
            typed_issue(i, i & 1, tf[i]);
    }
    RowPassLane rowpass;
    if constexpr (kPipe == 1)
        rowpass = make_row_pass_lane(lane);

#pragma unroll 1
    for (;;) {
    const int x0 = tx * kWaveTileW;
    const int next_tile = tile + tile_step;
    has_next = next_tile < n_tiles;
    const int next_tx = has_next ? (next_tile % tiles_x + skew * (round_no + 1)) % tiles_x : tx;
    next_ty = has_next ? next_tile / tiles_x : ty;
    next_off0 = lane_offset(next_tx, 0), next_off1 = lane_offset(next_tx, 1);

#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        if (it < 2 && kTyped) {
            // ---- colour + 4:2:0 for tile rows 8*it .. 8*it+7, pixels arriving as floats ------------
#pragma unroll 1
            for (int t0 = 0; t0 < 8; t0 += kTypedBufs) {
#pragma unroll
            for (int slot = 0; slot < kTypedBufs; ++slot) {
                const int half = slot & 1;
                const int T = it * 8 + t0 + slot; // (slot: the buffer; half == T & 1: the pixel-group half)
                TaskFloats &cur = tf[slot], &nxt = tf[(slot + kTypedBufs - 1) % kTypedBufs];
                wait_task_floats(cur); // all but the loads of the tasks issued after this one have landed
                typed_issue(T + kTypedBufs - 1, (half + kTypedBufs - 1) & 1, nxt);
                const int rp = (t0 + slot) >> 1;
                const int g = lane + 64 * half;
                if (x0 + 4 * g < cols) {
                    float ya[4], yb[4], sum_b[2], sum_r[2];
                    auto pixel = [&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        float uy[2], cb[2], cr[2], p[2][3];
                        JmFwdTies tie[2];
                        p[0][0] = row_float<j, 0>(cur.a), p[0][1] = row_float<j, 1>(cur.a), p[0][2] = row_float<j, 2>(cur.a);
                        p[1][0] = row_float<j, 0>(cur.b), p[1][1] = row_float<j, 1>(cur.b), p[1][2] = row_float<j, 2>(cur.b);
#pragma unroll
                        for (int row = 0; row < 2; ++row)
                            jm_fwd_colour_fast(p[row][0], p[row][1], p[row][2], uy[row], cb[row], cr[row], tie[row]);
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
                    };
                    pixel(std::integral_constant<int, 0>{}), pixel(std::integral_constant<int, 1>{});
                    pixel(std::integral_constant<int, 2>{}), pixel(std::integral_constant<int, 3>{});
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
            }
        } else if (it < 2) {
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
#if JPEGAMD_ABLATE_TIES == 2
                        (void)tie;
#else
                        const float worst_a = jm_min2(jm_min3(tie[0].ry, tie[0].rr, tie[1].ry), tie[1].rr);
                        const float worst_b = jm_min2(tie[0].rb, tie[1].rb);
                        if (__builtin_expect(worst_a < JM_TIE_A || worst_b < JM_TIE_B, 0)) { // ask the doubles
#if JPEGAMD_ABLATE_TIES == 1
                            asm volatile("s_nop 0");
#else
#pragma unroll
                            for (int row = 0; row < 2; ++row)
                                if (jm_fwd_any_tie(tie[row]))
                                    jm_fwd_colour_fix(p[row][0], p[row][1], p[row][2], tie[row], uy[row], cb[row], cr[row]);
#endif
                        }
#endif
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
        if constexpr (kDump) {
            // the colour stage's planes as they stand in LDS: 8 Y rows now, the 8 chroma rows
            // once both halves are through
            if (it < 2) {
                if (x0 + lane * 8 < cols)
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        *reinterpret_cast<uint2 *>(counts.y + (size_t)(ty * 16 + it * 8 + r) * cols + x0 + lane * 8) =
                            *reinterpret_cast<const uint2 *>(stage + r * 512 + lane * 8);
                if (it == 1 && (x0 >> 1) + lane * 4 < (cols >> 1))
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const size_t o = (size_t)(ty * 8 + r) * (cols >> 1) + (x0 >> 1) + lane * 4;
                        *reinterpret_cast<uint32_t *>(counts.cb + o) = *reinterpret_cast<const uint32_t *>(cbuf + r * 256 + lane * 4);
                        *reinterpret_cast<uint32_t *>(counts.cr + o) =
                            *reinterpret_cast<const uint32_t *>(cbuf + 2048 + r * 256 + lane * 4);
                    }
                wave_lds_fence();
            }
            continue;
        }

        // ---- transform: lane = block ---------------------------------------
        const bool luma = it < 2;
        const uint8_t *src = luma ? stage + lane * 8 : cbuf + (lane >> 5) * 2048 + (lane & 31) * 8;
        const int pitch = luma ? 512 : 256;
        const int bcol = luma ? tx * 64 + lane : tx * 32 + (lane & 31);
        const bool valid = bcol < (luma ? bw_y : bw_c);
        float v[64];
        if constexpr (kPipe == 1)
            rows_fdct_matrix_pipe(src, pitch, rowpass, v); // every lane: a matrix instruction is the wave's
        if (kPipe == 1 || valid) { // matrix pipe: the whole wave, one basic block with the matrix instructions
            if constexpr (kPipe == 1) {
                // recenterAroundZero (:289-298) reaches horizontal frequency 0 only: each of the eight row
                // sums of that column is short of -128 * 8 a_0 * 2^-10 = -a_0, the column's sum of 8 a_0 = 2 sqrt 2
                jm_fdct8<102400, true, 282842712, 100000000>(v[0], v[8], v[16], v[24], v[32], v[40], v[48], v[56]);
#pragma unroll
                for (int c = 1; c < 8; ++c)
                    jm_fdct8<102400, false>(v[0 * 8 + c], v[1 * 8 + c], v[2 * 8 + c], v[3 * 8 + c],
                                            v[4 * 8 + c], v[5 * 8 + c], v[6 * 8 + c], v[7 * 8 + c]);
                jm_quantise_block(v, qtab + (luma ? 0 : 64));
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    uint2 d = *reinterpret_cast<const uint2 *>(src + r * pitch);
                    v[r * 8 + 0] = ubyte_f32<0>(d.x), v[r * 8 + 1] = ubyte_f32<1>(d.x);
                    v[r * 8 + 2] = ubyte_f32<2>(d.x), v[r * 8 + 3] = ubyte_f32<3>(d.x);
                    v[r * 8 + 4] = ubyte_f32<0>(d.y), v[r * 8 + 5] = ubyte_f32<1>(d.y);
                    v[r * 8 + 6] = ubyte_f32<2>(d.y), v[r * 8 + 7] = ubyte_f32<3>(d.y);
                }
                block_fdct_quant(v, qtab + (luma ? 0 : 64));
            }
            // The staging area aliases the Y rows.  LDS instructions of one wave
            // retire in program order, so every lane's reads above are done before
            // the writes below; the fence only pins the compiler's order.
            wave_lds_fence();
            stage_write_zigzag(v, stage, lane);
            if constexpr (kCounts) {
                // pairs of this block = run ends of its zigzag sequence (:357-368).  Counted on the
                // staged block read back (the 64 values are dead by now: counting them in registers
                // before the staging costs the kernel its third wave per SIMD; counting the packed
                // dwords while they are staged keeps 168 registers and is 5 us slower, round 3)
                wave_lds_fence();
                uint32_t w[32];
                const uint4 *own = reinterpret_cast<const uint4 *>(stage + lane * kStagePitch);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint4 d = own[k];
                    w[4 * k] = d.x, w[4 * k + 1] = d.y, w[4 * k + 2] = d.z, w[4 * k + 3] = d.w;
                }
                uint8_t *dst = luma ? counts.y + (size_t)(ty * 2 + it) * bw_y + bcol
                                    : (lane < 32 ? counts.cb : counts.cr) + (size_t)ty * bw_c + bcol;
                if (valid)
                    *dst = (uint8_t)count_pairs(w);
            }
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
    if (!has_next)
        break;
    tile = next_tile, tx = next_tx, ty = next_ty, ++round_no;
    lane_off0 = next_off0, lane_off1 = next_off1;
    } // tiles of this wave
    if constexpr (kTyped) { // (the last tasks' look-ahead is still in flight: nothing may reuse its registers before it lands)
        asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int i = 0; i < kTypedBufs; ++i)
            asm volatile("" : "+v"(tf[i].a[0]), "+v"(tf[i].a[1]), "+v"(tf[i].a[2]), "+v"(tf[i].b[0]), "+v"(tf[i].b[1]), "+v"(tf[i].b[2]));
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
    int tile = blockIdx.x * kWavesPerWg + wave;
    if (tile >= n_tiles)
        return;
    // persistent like the encoder: a wave takes tiles tile, tile + tile_step, ... and fetches the
    // chroma coefficients of its next tile during the current one's last luma round
    const int tile_step = gridDim.x * kWavesPerWg;
    const int skew = (tile_step % tiles_x) == 0 ? 1 : 0; // as in the encoder: rotate the tile column per round
    int round_no = 0;
    int tx = tile % tiles_x, ty = tile / tiles_x;
    uint8_t *stage = lds + wave * kWaveLds;
    uint8_t *cbuf = stage + kStageBytes;
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
    for (;;) {
    const int x0 = tx * kWaveTileW;
    const int next_tile = tile + tile_step;
    const bool has_next = next_tile < n_tiles;
    const int next_tx = has_next ? (next_tile % tiles_x + skew * (round_no + 1)) % tiles_x : tx;
    const int next_ty = has_next ? next_tile / tiles_x : ty;
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
        else if (has_next)
            fetch_coefs(0, coef_y, coef_cb, coef_cr, next_tx, next_ty, bw_y, bw_c, lane, pre);
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
    if (!has_next)
        break;
    tile = next_tile, tx = next_tx, ty = next_ty, ++round_no;
    } // tiles of this wave
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
                        plane[(size_t)(r0 + r) * stride + c0 + c] = (uint8_t)jm_clamp255(jm_magic_to_int32(v[r * 8 + c]));
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
        dst[k] = make_int4(jm_magic_to_int32(v[4 * k]), jm_magic_to_int32(v[4 * k + 1]),
                           jm_magic_to_int32(v[4 * k + 2]), jm_magic_to_int32(v[4 * k + 3]));
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
#ifdef JPEGAMD_DEFAULT_XDL
    return Transform::xdl_f16;
#else
    return Transform::valu;
#endif
}

// JPEGAMD_LOADS=typed: the colour stage's pixels arrive as floats through typed buffer loads (kTyped above);
// the buffer's 32-bit offsets cover images below 4 GiB, larger ones take the plain loads
bool encoder_loads_typed(int rows, size_t stride)
{
    const char *e = getenv("JPEGAMD_LOADS");
    const bool want = e && strcmp(e, "typed") == 0;
    return want && (size_t)rows * stride < ((size_t)1 << 32);
}

// Grid of the two fused kernels.  Persistent waves (at most three workgroups per CU, a wave takes tiles t,
// t + waves, ... with its next tile's first fetch in flight) or one wave per tile (the hardware hands out
// workgroups as earlier ones finish).  JPEGAMD_GRID=persistent / pertile forces one of them; otherwise the
// encoder is persistent (84 against 91-93 us at 8192^2) and the decoder is persistent only when the image is
// a whole number of 512-pixel tiles wide: with a partial last column the tiles differ in cost, the static
// assignment leaves waves unevenly loaded and the per-tile grid is 5-11 % faster (64 images of 3840x2160 as one
// launch: 615 against 659 us; 17280x3840: 64.5 against 72.7; full widths: 65.8 against 66.7 at 8192^2, 32.9
// against 35.1 at 4320x7680; tools/partial_tile_probe.py, profiles/r03_g_grid_by_geometry.txt).
enum class Grid { automatic, persistent, per_tile };
Grid grid_choice()
{
    const char *e = getenv("JPEGAMD_GRID");
    if (e && strcmp(e, "pertile") == 0)
        return Grid::per_tile;
    if (e && strcmp(e, "persistent") == 0)
        return Grid::persistent;
    return Grid::automatic;
}
bool encoder_grid_persistent()
{
    return grid_choice() != Grid::per_tile;
}
bool decoder_grid_persistent(int cols)
{
    const Grid g = grid_choice();
    return g == Grid::persistent || (g == Grid::automatic && cols % kWaveTileW == 0);
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

// compute units of the current device (cached per device)
int jpegamd_cu_count(int *out)
{
    static int cached[kJpegamdMaxDevices];
    int dev = -1;
    JPEGAMD_HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kJpegamdMaxDevices)
        return jpegamd_fail("device %d beyond the %d this library knows", dev, kJpegamdMaxDevices);
    int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (!n) {
        JPEGAMD_HIPCHK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        n = n > 0 ? n : 1;
        __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
    }
    *out = n;
    return 0;
}

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
    // what the first host-pointer call would otherwise pay for (15 ms at 8192^2): the copier's
    // pinned staging buffers and streams
    if (jpegamd_rle_noalloc_setup()) // this device's address of the no-alloc entry point's scratch
        return 1;
    int cus = 0;
    if (jpegamd_cu_count(&cus)) // cached per device: the persistent grids are sized by it
        return 1;
    return jpegamd_staged_copy_warm();
}

// the two fused launches; with events, the kernel's own start and stop time stamps (the
// dispatch packet's, what rocprofv3 reports too) are recorded into them
static int launch_encode_image(const uint8_t *d_bgr, int rows, int cols, size_t stride, const int32_t q_luma[64],
                               const int32_t q_chroma[64], int16_t *d_coef_y, int16_t *d_coef_cb,
                               int16_t *d_coef_cr, void *stream, hipEvent_t start, hipEvent_t stop,
                               const CountPlanes *counts = nullptr)
{
    QuantPair qp;
    if (check_image(d_bgr, rows, cols, stride, "encode_image") || make_consts(q_luma, qp.t[0], "encode_image") ||
        make_consts(q_chroma, qp.t[1], "encode_image"))
        return 1;
    if (!d_coef_y || !d_coef_cb || !d_coef_cr)
        return fail("encode_image: null coefficient pointer");
    const int tiles_x = (cols + kWaveTileW - 1) / kWaveTileW;
    const int n_tiles = tiles_x * (rows / 16);
    const int grid_per_tile = (n_tiles + kWavesPerWg - 1) / kWavesPerWg; // one wave per tile
    int grid = grid_per_tile;
    if (encoder_grid_persistent()) { // no more workgroups than are resident at once: 3 per CU (LDS)
        int cus = 0;
        if (jpegamd_cu_count(&cus))
            return 1;
        grid = grid < 3 * cus ? grid : 3 * cus;
    }
    const bool typed = encoder_loads_typed(rows, stride);
    if (counts) {
        const auto kern = typed ? &encode_image_kernel<true, false, 0, true> : &encode_image_kernel<true, false>;
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, start, stop, 0, d_bgr, rows, cols,
                              stride, qp, d_coef_y, d_coef_cb, d_coef_cr, tiles_x, n_tiles, *counts);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const Transform tr = transform_choice();
    if (tr == Transform::xdl_f16) { // the same kernel with its row pass on the 16-bit matrix pipe
        const auto kern = &encode_image_kernel<false, false, 1>;
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, start, stop, 0, d_bgr, rows, cols,
                              stride, qp, d_coef_y, d_coef_cb, d_coef_cr, tiles_x, n_tiles,
                              CountPlanes{nullptr, nullptr, nullptr});
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (tr != Transform::valu)
        // the f32 matrix-core variant takes one tile per wave (never the capped grid: it would leave the
        // tiles beyond 12 x CUs unwritten); the f16 row-pass variant loops over tiles like the kernel above
        return jpegamd_launch_encode_variant(tr == Transform::mfma_f32 ? 1 : 2,
                                             tr == Transform::mfma_f32 ? grid_per_tile : grid, stream, start, stop, d_bgr, rows,
                                             cols, stride, qp, d_coef_y, d_coef_cb, d_coef_cr, tiles_x, n_tiles);
    const auto kern = typed ? &encode_image_kernel<false, false, 0, true> : &encode_image_kernel<false, false>;
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, start, stop, 0, d_bgr, rows, cols,
                          stride, qp, d_coef_y, d_coef_cb, d_coef_cr, tiles_x, n_tiles,
                          CountPlanes{nullptr, nullptr, nullptr});
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
    int grid = (n_tiles + kWavesPerWg - 1) / kWavesPerWg;
    if (decoder_grid_persistent(cols)) { // no more workgroups than are resident at once: 3 per CU (LDS)
        int cus = 0;
        if (jpegamd_cu_count(&cus))
            return 1;
        grid = grid < 3 * cus ? grid : 3 * cus;
    }
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

int jpegamd_colour_planes_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride, uint8_t *d_y, uint8_t *d_cb,
                              uint8_t *d_cr, void *stream)
{
    if (check_image(d_bgr, rows, cols, stride, "colour_planes"))
        return 1;
    if (!d_y || !d_cb || !d_cr || (reinterpret_cast<uintptr_t>(d_y) & 7) || (reinterpret_cast<uintptr_t>(d_cb) & 3) ||
        (reinterpret_cast<uintptr_t>(d_cr) & 3))
        return fail("colour_planes: null plane pointer, or Y not 8-byte / Cb, Cr not 4-byte aligned");
    QuantPair qp{}; // (the transform is not reached)
    const int tiles_x = (cols + kWaveTileW - 1) / kWaveTileW;
    const int n_tiles = tiles_x * (rows / 16);
    const int grid = (n_tiles + kWavesPerWg - 1) / kWavesPerWg;
    const auto kern = encoder_loads_typed(rows, stride) ? &encode_image_kernel<false, true, 0, true> : &encode_image_kernel<false, true>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWavesPerWg), 0, (hipStream_t)stream, d_bgr, rows, cols, stride, qp,
                       (int16_t *)nullptr, (int16_t *)nullptr, (int16_t *)nullptr, tiles_x, n_tiles,
                       CountPlanes{d_y, d_cb, d_cr});
    HIPCHK(hipGetLastError());
    return 0;
}

int jpegamd_encode_image_rle_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride, const int32_t q_luma[64],
                                 const int32_t q_chroma[64], int16_t *const d_coef[3], int32_t *const d_pairs[3],
                                 const size_t cap_pairs[3], uint32_t *const d_block_offsets[3], void *d_workspace,
                                 size_t workspace_bytes, void *stream)
{
    if (!d_coef || !d_pairs || !cap_pairs || !d_block_offsets || !d_workspace)
        return fail("encode_image_rle: null pointer");
    if (rows <= 0 || cols <= 0 || (rows & 15) || (cols & 15))
        return fail("encode_image_rle: rows=%d cols=%d must be positive multiples of 16", rows, cols);
    const size_t nby = (size_t)(rows / 8) * (cols / 8), nbc = (size_t)(rows / 16) * (cols / 16);
    const size_t nb[3] = {nby, nbc, nbc};
    if (workspace_bytes < jpegamd_rle_workspace_bytes(nby + 2 * nbc) || (reinterpret_cast<uintptr_t>(d_workspace) & 255))
        return fail("encode_image_rle: workspace of %zu bytes (need %zu, 256-byte aligned)", workspace_bytes,
                    jpegamd_rle_workspace_bytes(nby + 2 * nbc));
    // workspace: the three count arrays (one byte per block, each 256-byte aligned), then the tile sums
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    uint8_t *w = static_cast<uint8_t *>(d_workspace);
    const CountPlanes cp{w, w + up(nby), w + up(nby) + up(nbc)};
    uint32_t *sums = reinterpret_cast<uint32_t *>(w + up(nby) + 2 * up(nbc));
    if (launch_encode_image(d_bgr, rows, cols, stride, q_luma, q_chroma, d_coef[0], d_coef[1], d_coef[2], stream, nullptr,
                            nullptr, &cp))
        return 1;
    const uint8_t *const counts[3] = {cp.y, cp.cb, cp.cr};
    return jpegamd_rle_encode_counted(d_coef, nb, counts, d_pairs, cap_pairs, d_block_offsets, sums, stream);
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
    // device memory per call, like kernel.cu:97-104 / 160-164 (no state survives the call); the
    // copies go through the staged copier (pinned buffers on worker threads: ~10x a pageable hipMemcpy)
    DevBuf in, res;
    HIPCHK(hipMalloc(&in.p, nb * 64 * sizeof(int32_t)));
    HIPCHK(hipMalloc(&res.p, nb * 64 * sizeof(float)));
    if (jpegamd_staged_copy_linear(in.p, const_cast<int32_t *>(blocks), nb * 64 * sizeof(int32_t), true))
        return 1;
    if (jpegamd_fdct_quant_blocks_dev((const int32_t *)in.p, nb, q, (float *)res.p, nullptr))
        return 1;
    HIPCHK(hipDeviceSynchronize()); // the kernel ran on the null stream, the copier uses its own
    return jpegamd_staged_copy_linear(res.p, out, nb * 64 * sizeof(float), false);
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
    if (jpegamd_staged_copy_linear(in.p, const_cast<float *>(coef), nb * 64 * sizeof(float), true))
        return 1;
    if (jpegamd_dequant_idct_blocks_dev((const float *)in.p, nb, q, (int32_t *)res.p, nullptr))
        return 1;
    HIPCHK(hipDeviceSynchronize());
    return jpegamd_staged_copy_linear(res.p, out, nb * 64 * sizeof(int32_t), false);
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
