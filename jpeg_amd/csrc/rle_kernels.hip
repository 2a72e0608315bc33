// rle_kernels.hip -- the reference's run-length stage on the device (gfx950).
//
//   run_length_encode   encode_decode.cpp:350-373   per block, over all 64 zigzag
//                       values including DC: one (value, run) pair per maximal run
//                       of equal values; runs never cross blocks; the plane's
//                       stream is the blocks' pairs in raster order (:643-645).
//   decodeRLE           encode_decode.cpp:484-518   pairs expanded into one long
//                       sequence that is cut every 64 values.
//
// Pairs are two int32 (value, run), i.e. exactly the memory of the reference's
// std::vector<int>, so the host entropy coder can consume the copied-back
// buffer as is.  All of it is integer work and bit-exact.
//
// Encoding is count -> scan -> emit.  In count and emit a wave stages 64
// consecutive blocks through LDS (coalesced 16 B per lane in, one block per lane
// out, the same 144-B pitch as the transform kernels) and a lane walks its own
// block in registers.  HBM-bound: 2 x 2 B/coefficient read + 8 B/pair written.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/jpeg_amd.h"
#include "host_common.h"

namespace {

constexpr int kPitch = 144;          // staged block pitch (128 + 16: conflict-free b128)
constexpr int kBlocksPerWg = 256;    // 4 waves x 64 blocks
constexpr int kPairsPerWg = 1024;    // decode: 256 threads x 4 pairs

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d)
            v += n;
    }
    return v;
}

// exclusive scan over a workgroup of `NW` waves; `total` = the workgroup's sum
template <int NW>
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t *wave_sums /* LDS, NW */, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63)
        wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0, t = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t s = wave_sums[w];
        base += w < wave ? s : 0u;
        t += s;
    }
    __syncthreads(); // wave_sums may be reused by the caller
    total = t;
    return base + inc - v;
}

// 64 consecutive blocks starting at blk0 -> lane's own block as 32 dwords (two
// int16 each, zigzag order).  Blocks at or beyond nb read as nothing.
__device__ __forceinline__ void stage_blocks(const int16_t *__restrict__ coef, size_t blk0, size_t nb,
                                             uint8_t *stage, int lane, uint32_t (&w)[32])
{
    const uint8_t *src = reinterpret_cast<const uint8_t *>(coef + blk0 * 64);
    const size_t n_ok = nb > blk0 ? nb - blk0 : 0;
    const int blk = lane >> 3;
    uint8_t *sp = stage + blk * kPitch + (lane & 7) * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((size_t)(k * 8 + blk) < n_ok) {
            uint4 d = *reinterpret_cast<const uint4 *>(src + k * 1024 + lane * 16);
            *reinterpret_cast<uint4 *>(sp + k * 8 * kPitch) = d;
        }
    }
    wave_fence();
    const uint4 *mine = reinterpret_cast<const uint4 *>(stage + lane * kPitch);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint4 d = mine[k];
        w[4 * k] = d.x, w[4 * k + 1] = d.y, w[4 * k + 2] = d.z, w[4 * k + 3] = d.w;
    }
    wave_fence();
}

__device__ __forceinline__ int coef_at(const uint32_t (&w)[32], int i)
{
    return (i & 1) ? (int)w[i >> 1] >> 16 : (int)(int16_t)(w[i >> 1] & 0xFFFFu);
}

// number of (value, run) pairs of one block = 1 + number of places where
// neighbours differ (encode_decode.cpp:357-368)
__device__ __forceinline__ uint32_t count_pairs(const uint32_t (&w)[32])
{
    uint32_t n = 1;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        n += ((w[k] ^ (w[k] >> 16)) & 0xFFFFu) != 0u;
        if (k < 31)
            n += (((w[k] >> 16) ^ w[k + 1]) & 0xFFFFu) != 0u;
    }
    return n;
}

// pass 1: pairs per workgroup of 256 blocks
__global__ __launch_bounds__(256) void rle_count_kernel(const int16_t *__restrict__ coef, size_t nb,
                                                        uint32_t *__restrict__ wg_sums)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 64 * kPitch];
    __shared__ uint32_t wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t blk0 = (size_t)blockIdx.x * kBlocksPerWg + wave * 64;
    uint32_t w[32];
    stage_blocks(coef, blk0, nb, lds + wave * 64 * kPitch, lane, w);
    const uint32_t n = blk0 + lane < nb ? count_pairs(w) : 0u;
    uint32_t total;
    (void)wg_exclusive_scan<4>(n, wave_sums, total);
    if (threadIdx.x == 0)
        wg_sums[blockIdx.x] = total;
}

// pass 2: exclusive scan of n values in place by ONE workgroup; the grand total
// goes to *total_out
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t *__restrict__ v, size_t n,
                                                         uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t wave_sums[16];
    uint32_t carry = 0;
    for (size_t base = 0; base < n; base += 1024) { // uniform trip count
        const size_t i = base + threadIdx.x;
        const uint32_t x = i < n ? v[i] : 0u;
        uint32_t total;
        const uint32_t ex = wg_exclusive_scan<16>(x, wave_sums, total);
        if (i < n)
            v[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0)
        *total_out = carry;
}

// pass 3: per-block offsets and the pairs themselves
__global__ __launch_bounds__(256) void rle_emit_kernel(const int16_t *__restrict__ coef, size_t nb,
                                                       const uint32_t *__restrict__ wg_base,
                                                       uint32_t *__restrict__ block_offsets,
                                                       int2 *__restrict__ pairs, size_t cap_pairs)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 64 * kPitch];
    __shared__ uint32_t wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t blk0 = (size_t)blockIdx.x * kBlocksPerWg + wave * 64;
    uint32_t w[32];
    stage_blocks(coef, blk0, nb, lds + wave * 64 * kPitch, lane, w);
    const bool valid = blk0 + lane < nb;
    const uint32_t n = valid ? count_pairs(w) : 0u;
    uint32_t total;
    const uint32_t off = wg_base[blockIdx.x] + wg_exclusive_scan<4>(n, wave_sums, total);
    if (!valid)
        return;
    block_offsets[blk0 + lane] = off;
    // walk the block: a pair is emitted where the value changes, and at the end
    size_t o = off;
    int run = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int cur = coef_at(w, i);
        ++run;
        const bool last = i == 63 || coef_at(w, i == 63 ? 63 : i + 1) != cur;
        if (last) {
            if (o < cap_pairs)
                pairs[o] = make_int2(cur, run);
            ++o;
            run = 0;
        }
    }
}

// ---- decodeRLE: sum of runs per workgroup of 1024 pairs, scan, fill ----------
__device__ __forceinline__ uint32_t run_of(const int2 &p)
{
    return p.y > 0 ? (uint32_t)p.y : 0u; // a non-positive run appends nothing (:489-497)
}

__global__ __launch_bounds__(256) void rld_count_kernel(const int2 *__restrict__ pairs, size_t n_pairs,
                                                        uint32_t *__restrict__ wg_sums)
{
    __shared__ uint32_t wave_sums[4];
    const size_t i0 = (size_t)blockIdx.x * kPairsPerWg + threadIdx.x * 4;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n_pairs)
            s += run_of(pairs[i0 + k]);
    uint32_t total;
    (void)wg_exclusive_scan<4>(s, wave_sums, total);
    if (threadIdx.x == 0)
        wg_sums[blockIdx.x] = total;
}

// out was zeroed beforehand: only non-zero values are written.  Positions at or
// beyond n_values are dropped (the loader's trailing pad symbols).
__global__ __launch_bounds__(256) void rld_fill_kernel(const int2 *__restrict__ pairs, size_t n_pairs,
                                                       const uint32_t *__restrict__ wg_base,
                                                       int16_t *__restrict__ out, size_t n_values)
{
    __shared__ uint32_t wave_sums[4];
    const size_t i0 = (size_t)blockIdx.x * kPairsPerWg + threadIdx.x * 4;
    int2 p[4];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p[k] = i0 + k < n_pairs ? pairs[i0 + k] : make_int2(0, 0);
        s += run_of(p[k]);
    }
    uint32_t total;
    size_t pos = (size_t)wg_base[blockIdx.x] + wg_exclusive_scan<4>(s, wave_sums, total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r = run_of(p[k]);
        if (p[k].x != 0) {
            for (uint32_t j = 0; j < r && pos + j < n_values; ++j)
                out[pos + j] = (int16_t)p[k].x;
        }
        pos += r;
    }
}

size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

} // namespace

extern "C" {

size_t jpegamd_rle_workspace_bytes(size_t nb)
{
    // encode: one sum per 256 blocks; decode: one per 1024 pairs, at most 64 pairs per block
    const size_t n = ceil_div(nb, kBlocksPerWg) > ceil_div(nb * 64, kPairsPerWg) ? ceil_div(nb, kBlocksPerWg)
                                                                                 : ceil_div(nb * 64, kPairsPerWg);
    return (n + 64) * sizeof(uint32_t);
}

int jpegamd_rle_encode_plane_dev(const int16_t *d_coef_zz, size_t nb, int32_t *d_pairs, size_t cap_pairs,
                                 uint32_t *d_block_offsets, void *d_workspace, size_t workspace_bytes,
                                 void *stream)
{
    if (!d_coef_zz || !d_pairs || !d_block_offsets || !d_workspace)
        return jpegamd_fail("rle_encode_plane: null pointer");
    if (nb == 0 || nb > ((size_t)1 << 26))
        return jpegamd_fail("rle_encode_plane: nb=%zu outside 1..2^26", nb);
    if (workspace_bytes < jpegamd_rle_workspace_bytes(nb))
        return jpegamd_fail("rle_encode_plane: workspace of %zu bytes, need %zu", workspace_bytes,
                            jpegamd_rle_workspace_bytes(nb));
    if ((reinterpret_cast<uintptr_t>(d_coef_zz) & 15) || (reinterpret_cast<uintptr_t>(d_pairs) & 7))
        return jpegamd_fail("rle_encode_plane: coefficient plane must be 16-byte and pairs 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t *sums = static_cast<uint32_t *>(d_workspace);
    const unsigned nwg = (unsigned)ceil_div(nb, kBlocksPerWg);
    hipLaunchKernelGGL(rle_count_kernel, dim3(nwg), dim3(256), 0, s, d_coef_zz, nb, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, (size_t)nwg, d_block_offsets + nb);
    hipLaunchKernelGGL(rle_emit_kernel, dim3(nwg), dim3(256), 0, s, d_coef_zz, nb, sums, d_block_offsets,
                       reinterpret_cast<int2 *>(d_pairs), cap_pairs);
    JPEGAMD_HIPCHK(hipGetLastError());
    return 0;
}

int jpegamd_rle_decode_plane_dev(const int32_t *d_pairs, size_t n_pairs, size_t nb, int16_t *d_coef_zz,
                                 void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!d_coef_zz || !d_workspace || (!d_pairs && n_pairs))
        return jpegamd_fail("rle_decode_plane: null pointer");
    if (nb == 0 || nb > ((size_t)1 << 26))
        return jpegamd_fail("rle_decode_plane: nb=%zu outside 1..2^26", nb);
    if (n_pairs > nb * 64 + 8)
        return jpegamd_fail("rle_decode_plane: %zu pairs cannot belong to %zu blocks", n_pairs, nb);
    if (workspace_bytes < jpegamd_rle_workspace_bytes(nb))
        return jpegamd_fail("rle_decode_plane: workspace of %zu bytes, need %zu", workspace_bytes,
                            jpegamd_rle_workspace_bytes(nb));
    if (reinterpret_cast<uintptr_t>(d_pairs) & 7)
        return jpegamd_fail("rle_decode_plane: pairs must be 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    JPEGAMD_HIPCHK(hipMemsetAsync(d_coef_zz, 0, nb * 64 * sizeof(int16_t), s));
    if (n_pairs == 0)
        return 0;
    uint32_t *sums = static_cast<uint32_t *>(d_workspace);
    const unsigned nwg = (unsigned)ceil_div(n_pairs, kPairsPerWg);
    const int2 *pairs = reinterpret_cast<const int2 *>(d_pairs);
    hipLaunchKernelGGL(rld_count_kernel, dim3(nwg), dim3(256), 0, s, pairs, n_pairs, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, (size_t)nwg, sums + nwg + 1);
    hipLaunchKernelGGL(rld_fill_kernel, dim3(nwg), dim3(256), 0, s, pairs, n_pairs, sums, d_coef_zz, nb * 64);
    JPEGAMD_HIPCHK(hipGetLastError());
    return 0;
}

} // extern "C"
