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
// Encoding is pairs per tile of 256 blocks -> emit (two launches for up to three planes; the emit
// kernel adds up the tiles before its own: the separate scan of round 2 is gone).  The per-block pair counts come from the fused encoder (one byte per block, summed by
// rle_tile_sums_kernel: the coefficients are then read ONCE, by the emit kernel) or, for planes from
// anywhere else, from rle_count_kernel (a second read).  In the emit kernel a wave stages 64
// consecutive blocks through LDS (coalesced 16 B per lane in, one block per lane out, the same
// 144-B pitch as the transform kernels), a lane walks the run ends of its own block, and the wave's
// pairs leave through LDS as contiguous 8-byte stores.  HBM-bound: 2 B/coefficient read (+ 1 B per
// block of counts) + 8 B/pair + 4 B/block written.  (A single pass with the counts chained between
// workgroups was measured and dropped: see below.)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "../../include/jpeg_amd.h"
#include "host_common.h"
#include "jpeg_device_common.h" // run_ends()

namespace {

constexpr int kPitch = 144;          // staged block pitch (128 + 16: conflict-free b128)
constexpr int kBlocksPerWg = 256;    // 4 waves x 64 blocks
constexpr int kPairsPerWg = 1024;    // decode: 256 threads x 4 pairs
constexpr int kMaxDevices = kJpegamdMaxDevices; // host_common.h

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

typedef unsigned long long u64;

// ---- run_length_encode ----------------------------------------------------------------
// A tile is 256 consecutive blocks of a plane (one workgroup, one block per lane).  To place a
// tile's pairs the number of pairs of all blocks before it is needed (the stream is the
// blocks' pairs in raster order, :643-645): pairs per tile -> emit (which sums the tiles before its own).
// Where the per-block pair counts come from:
//   * jpegamd_encode_image_rle_dev: from the fused ENCODER, whose lane holds the block's 64
//     quantised values in registers anyway and writes one count byte per block; the tile sums
//     are then 256-byte additions (rle_tile_sums_kernel) and the coefficients are read ONCE,
//     by the emit kernel;
//   * jpegamd_rle_encode_plane/_image_dev (coefficients from anywhere): rle_count_kernel reads
//     the plane for the tile sums, the emit kernel reads it again.
// (A single pass with the counts chained between workgroups, "decoupled look-back", was built
// and measured first: 129 us for the three planes of an 8192^2 image against 91 us with the
// chain cut, and 172 us with a persistent grid -- all resident workgroups publish together and
// every one of them then walks back a whole grid's worth of status words, at one agent-scope
// round trip per 64.  profiles/r02_a_rle_lookback_ablation.txt.)
// Up to three planes go through one launch of each kernel.
struct RlePlane {
    const int16_t *coef;
    uint32_t *block_offsets; // nb + 1
    int2 *pairs;
    size_t cap_pairs;
    const uint8_t *counts; // pairs per block from the encoder, or null
    uint32_t nb;
    uint32_t tile_begin; // first tile of this plane
};
struct RleJob {
    RlePlane p[3];
    uint32_t n_planes, n_tiles;
};
constexpr int kPairStage = 512; // pairs a wave stages in LDS for coalesced stores (4 KiB)

__device__ __forceinline__ const RlePlane &plane_of_tile(const RleJob &job, uint32_t tile)
{
    int pi = 0;
    if (job.n_planes > 1 && tile >= job.p[1].tile_begin)
        pi = 1;
    if (job.n_planes > 2 && tile >= job.p[2].tile_begin)
        pi = 2;
    return job.p[pi];
}

__device__ __forceinline__ uint32_t wave_sum32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    return v;
}

// pairs per tile from the coefficients themselves
__global__ __launch_bounds__(256) void rle_count_kernel(RleJob job, uint32_t *__restrict__ sums)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 64 * kPitch];
    __shared__ uint32_t wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x;
    const RlePlane &pl = plane_of_tile(job, tile);
    const size_t blk0 = (size_t)(tile - pl.tile_begin) * kBlocksPerWg + wave * 64;
    uint32_t w[32];
    stage_blocks(pl.coef, blk0, pl.nb, lds + wave * 64 * kPitch, lane, w);
    const uint32_t n = blk0 + lane < pl.nb ? count_pairs(w) : 0u;
    const uint32_t t = wave_sum32(n);
    if (lane == 0)
        wave_sums[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0)
        sums[tile] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
}

// pairs per tile from the encoder's count bytes: one thread per tile adds its 256 bytes
__global__ __launch_bounds__(256) void rle_tile_sums_kernel(RleJob job, uint32_t *__restrict__ sums)
{
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= job.n_tiles)
        return;
    const RlePlane &pl = plane_of_tile(job, tile);
    const size_t b0 = (size_t)(tile - pl.tile_begin) * kBlocksPerWg;
    const uint8_t *c = pl.counts + b0;
    uint32_t t = 0;
    if (b0 + kBlocksPerWg <= pl.nb) {
        const uint4 *c4 = reinterpret_cast<const uint4 *>(c); // (the count arrays are 256-byte aligned)
        uint32_t even = 0, odd = 0;                            // sums of bytes 0,2 / 1,3 as 16-bit pairs: 64 x 64 fits
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint4 d = c4[k];
            even += (d.x & 0x00FF00FFu) + (d.y & 0x00FF00FFu) + (d.z & 0x00FF00FFu) + (d.w & 0x00FF00FFu);
            odd += ((d.x >> 8) & 0x00FF00FFu) + ((d.y >> 8) & 0x00FF00FFu) + ((d.z >> 8) & 0x00FF00FFu) +
                   ((d.w >> 8) & 0x00FF00FFu);
        }
        t = (even & 0xFFFFu) + (even >> 16) + (odd & 0xFFFFu) + (odd >> 16);
    } else {
        for (size_t i = b0; i < pl.nb; ++i)
            t += pl.counts[i];
    }
    sums[tile] = t;
}

// per-block offsets and the pairs themselves; sums[tile] = pairs of tile `tile` (each workgroup adds up what lies before its tile).
// PERSISTENT since round 3: the host launches no more workgroups than are resident at once, a
// workgroup takes tiles blockIdx.x, + gridDim.x, ... and the 8 KiB of the NEXT tile's coefficients
// are already on their way (eight 16-byte loads per lane in registers) while the current tile's run
// ends are walked -- a wave of the one-tile-per-workgroup form lived 7.8 us for 0.9 us of instruction
// issue (profiles/r03_d_pmc_rle_entropy_kernels.txt: 12.7 M VALU, 24 576 waves, 114.5 M wave cycles x 4):
// launch, first fetch and the walk's LDS round trips one after the other.
__device__ __forceinline__ void fetch_blocks(const RleJob &job, uint32_t tile, int wave, int lane, uint4 (&d)[8])
{
    const RlePlane &pl = plane_of_tile(job, tile);
    const size_t blk0 = (size_t)(tile - pl.tile_begin) * kBlocksPerWg + wave * 64;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(pl.coef + blk0 * 64);
    const size_t n_ok = pl.nb > blk0 ? pl.nb - blk0 : 0;
    const int blk = lane >> 3;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        d[k] = make_uint4(0, 0, 0, 0);
        if ((size_t)(k * 8 + blk) < n_ok)
            d[k] = *reinterpret_cast<const uint4 *>(src + k * 1024 + lane * 16);
    }
}

__global__ __launch_bounds__(256) void rle_emit_kernel(RleJob job, const uint32_t *__restrict__ sums)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 64 * kPitch];
    __shared__ __attribute__((aligned(16))) int2 pair_stage[4][kPairStage];
    __shared__ uint32_t wave_sums[2][4], base_parts[2][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *stage = lds + wave * 64 * kPitch;
    uint32_t tile = blockIdx.x;
    if (tile >= job.n_tiles)
        return;
    uint4 pre[8];
    fetch_blocks(job, tile, wave, lane, pre);
    uint32_t prev_tile = 0, prev_base = 0;
    for (int round = 0;; ++round) {
        const RlePlane &pl = plane_of_tile(job, tile);
        const size_t blk0 = (size_t)(tile - pl.tile_begin) * kBlocksPerWg + wave * 64;
        // index of the tile's first pair = sum of the sums of the plane's tiles before it: every thread adds
        // a strided share (at most 16 words of a 16 KiB array that stays in L2), the four waves' shares meet
        // in LDS at the barrier below.  This replaced rle_scan_kernel (one workgroup, 8 us, and a launch).
        // (Only the tiles since this workgroup's previous tile are new when both lie in one plane.)
        const bool same_plane = round > 0 && prev_tile >= pl.tile_begin;
        const uint32_t from = same_plane ? prev_tile : pl.tile_begin;
        const uint32_t base_before = same_plane ? prev_base : 0u;
        uint32_t part = 0;
        for (uint32_t t = from + threadIdx.x; t < tile; t += 256)
            part += sums[t];
        // staged: 64 blocks x 128 B at the 144-B pitch, then the lane's own block into registers
        {
            uint8_t *sp = stage + (lane >> 3) * kPitch + (lane & 7) * 16;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<uint4 *>(sp + k * 8 * kPitch) = pre[k];
        }
        const uint32_t next_tile = tile + gridDim.x;
        const bool has_next = next_tile < job.n_tiles;
        if (has_next) // in flight during everything below
            fetch_blocks(job, next_tile, wave, lane, pre);
        wave_fence();
        uint32_t w[32];
        const uint4 *mine = reinterpret_cast<const uint4 *>(stage + lane * kPitch);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint4 d = mine[k];
            w[4 * k] = d.x, w[4 * k + 1] = d.y, w[4 * k + 2] = d.z, w[4 * k + 3] = d.w;
        }
        const bool valid = blk0 + lane < pl.nb;
        unsigned long long ends = valid ? run_ends(w) : 0u;
        const uint32_t n = (uint32_t)__builtin_popcountll(ends);
        const uint32_t inc = wave_inclusive_scan(n, lane);
        const uint32_t wave_total = __shfl(inc, 63, 64);
        part = wave_sum32(part);
        if (lane == 63) { // (two sets: a fast wave's next round must not overwrite this one's)
            wave_sums[round & 1][wave] = inc;
            base_parts[round & 1][wave] = part;
        }
        __syncthreads();
        uint32_t wave_base = base_before + base_parts[round & 1][0] + base_parts[round & 1][1] + base_parts[round & 1][2] +
                             base_parts[round & 1][3];
        prev_tile = tile, prev_base = wave_base; // (the tile's own base, before the waves' shares are added)
        if (threadIdx.x == 0 && blk0 + kBlocksPerWg >= pl.nb) // the plane's last tile: its end is the plane's total
            pl.block_offsets[pl.nb] = wave_base + wave_sums[round & 1][0] + wave_sums[round & 1][1] +
                                      wave_sums[round & 1][2] + wave_sums[round & 1][3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            wave_base += k < wave ? wave_sums[round & 1][k] : 0u;
        // one step per PAIR, not per value: the next set bit of `ends` is where the next run ends,
        // its value is still in the lane's staged block (LDS), its length the distance to the
        // previous end.  A wave takes as many steps as its busiest block has pairs; the value of
        // step i + 1 is fetched before the pair of step i is written, so one LDS round trip, not
        // two, separates the steps.
        const size_t wave_off = wave_base; // plane-wide index of the wave's first pair
        const uint32_t lane_off = inc - n;
        if (valid)
            pl.block_offsets[blk0 + lane] = (uint32_t)(wave_off + lane_off);
        const int16_t *own = reinterpret_cast<const int16_t *>(stage + lane * kPitch);
        int prev = -1;
        if (wave_total <= (uint32_t)kPairStage) {
            // the wave's pairs are contiguous in the stream: collect them in LDS and store them 64 at
            // a time, 512 contiguous bytes per instruction
            int2 *ps = pair_stage[wave] + lane_off;
            if (ends) {
                int pos = __builtin_ctzll(ends);
                ends &= ends - 1;
                int val = own[pos];
                while (ends) {
                    const int npos = __builtin_ctzll(ends);
                    ends &= ends - 1;
                    const int nval = own[npos];
                    *ps++ = make_int2(val, pos - prev);
                    prev = pos, pos = npos, val = nval;
                }
                *ps = make_int2(val, pos - prev);
            }
            wave_fence();
            for (uint32_t i = lane; i < wave_total; i += 64)
                if (wave_off + i < pl.cap_pairs)
                    pl.pairs[wave_off + i] = pair_stage[wave][i];
        } else {
            const size_t off = wave_off + lane_off;
            int2 *__restrict__ out = pl.pairs + off;
            const size_t room = off < pl.cap_pairs ? pl.cap_pairs - off : 0;
            uint32_t j = 0;
            while (ends) {
                const int pos = __builtin_ctzll(ends);
                ends &= ends - 1;
                if (j < room)
                    out[j] = make_int2((int)own[pos], pos - prev);
                ++j;
                prev = pos;
            }
        }
        if (!has_next)
            break;
        wave_fence(); // the staged blocks and pairs are consumed (in-order LDS) before the next round overwrites them
        tile = next_tile;
    }
}

// ---- decodeRLE: sum of runs per workgroup of 1024 pairs, scan, fill ----------
// The pairs come from a file: run lengths are untrusted.  A run is clamped to n_values + 1 (it
// cannot place more values than that in the plane; the reference's decodeRLE would build the
// overflow and drop it at :702) and all positions are 64-bit, so nothing wraps and a crafted
// stream cannot send values to the wrong blocks or keep the GPU busy beyond one plane fill.

__device__ __forceinline__ u64 run_of(const int2 &p, u64 n_values)
{
    const u64 r = p.y > 0 ? (u64)p.y : 0u; // a non-positive run appends nothing (:489-497)
    return r <= n_values ? r : n_values + 1;
}

__device__ __forceinline__ u64 wave_inclusive_scan64(u64 v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 n = __shfl_up(v, d, 64);
        if (lane >= d)
            v += n;
    }
    return v;
}

template <int NW>
__device__ __forceinline__ u64 wg_exclusive_scan64(u64 v, u64 *wave_sums /* LDS, NW */, u64 &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 inc = wave_inclusive_scan64(v, lane);
    if (lane == 63)
        wave_sums[wave] = inc;
    __syncthreads();
    u64 base = 0, t = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u64 s = wave_sums[w];
        base += w < wave ? s : 0u;
        t += s;
    }
    __syncthreads();
    total = t;
    return base + inc - v;
}

__global__ __launch_bounds__(256) void rld_count_kernel(const int2 *__restrict__ pairs, size_t n_pairs,
                                                        u64 *__restrict__ wg_sums, u64 n_values)
{
    __shared__ u64 wave_sums[4];
    const size_t i0 = (size_t)blockIdx.x * kPairsPerWg + threadIdx.x * 4;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n_pairs)
            s += run_of(pairs[i0 + k], n_values);
    u64 total;
    (void)wg_exclusive_scan64<4>(s, wave_sums, total);
    if (threadIdx.x == 0)
        wg_sums[blockIdx.x] = total;
}

// exclusive scan of n 64-bit values in place by ONE workgroup
__global__ __launch_bounds__(1024) void scan_sums64_kernel(u64 *__restrict__ v, size_t n)
{
    __shared__ u64 wave_sums[16];
    u64 carry = 0;
    for (size_t base = 0; base < n; base += 1024) { // uniform trip count
        const size_t i = base + threadIdx.x;
        const u64 x = i < n ? v[i] : 0u;
        u64 total;
        const u64 ex = wg_exclusive_scan64<16>(x, wave_sums, total);
        if (i < n)
            v[i] = carry + ex;
        carry += total;
    }
}

// out was zeroed beforehand: only non-zero values are written.  Positions at or
// beyond n_values are dropped (the loader's trailing pad symbols).
__global__ __launch_bounds__(256) void rld_fill_kernel(const int2 *__restrict__ pairs, size_t n_pairs,
                                                       const u64 *__restrict__ wg_base,
                                                       int16_t *__restrict__ out, u64 n_values)
{
    __shared__ u64 wave_sums[4];
    const int lane = threadIdx.x & 63;
    const size_t i0 = (size_t)blockIdx.x * kPairsPerWg + threadIdx.x * 4;
    int2 p[4];
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p[k] = i0 + k < n_pairs ? pairs[i0 + k] : make_int2(0, 0);
        s += run_of(p[k], n_values);
    }
    u64 total;
    u64 pos = wg_base[blockIdx.x] + wg_exclusive_scan64<4>(s, wave_sums, total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u64 r = run_of(p[k], n_values);
        const bool fill = p[k].x != 0 && pos < n_values;
        const u64 end = pos + r < n_values ? pos + r : n_values;
        // run_length_encode never emits a run above 64; a longer one (a corrupt file) is filled
        // by the whole wave, 64 values at a time, so that no single lane walks a plane
        if (fill && r <= 64) {
            for (u64 j = pos; j < end; ++j)
                out[j] = (int16_t)p[k].x;
        }
        u64 todo = __ballot(fill && r > 64);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const u64 b = __shfl(pos, src, 64), e = __shfl(end, src, 64);
            const int16_t val = (int16_t)__shfl(p[k].x, src, 64);
            for (u64 j = b + lane; j < e; j += 64)
                out[j] = val;
        }
        pos += r;
    }
}

size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

int check_rle_encode(const int16_t *d_coef_zz, size_t nb, const int32_t *d_pairs, const uint32_t *d_block_offsets,
                     const char *what)
{
    if (!d_coef_zz || !d_pairs || !d_block_offsets)
        return jpegamd_fail("%s: null pointer", what);
    if (nb == 0 || nb > ((size_t)1 << 26))
        return jpegamd_fail("%s: nb=%zu outside 1..2^26", what, nb);
    if ((reinterpret_cast<uintptr_t>(d_coef_zz) & 15) || (reinterpret_cast<uintptr_t>(d_pairs) & 7))
        return jpegamd_fail("%s: coefficient plane must be 16-byte and pairs 8-byte aligned", what);
    return 0;
}

// the launches: `sums` = one uint32 per tile of 256 blocks (device memory, nothing to clear)
int launch_rle_encode(const RlePlane *planes, int n_planes, uint32_t *sums, hipStream_t s)
{
    RleJob job{};
    uint32_t n_tiles = 0;
    bool counted = true;
    for (int k = 0; k < n_planes; ++k) {
        job.p[k] = planes[k];
        job.p[k].tile_begin = n_tiles;
        n_tiles += (uint32_t)ceil_div(planes[k].nb, kBlocksPerWg);
        counted = counted && planes[k].counts != nullptr;
    }
    job.n_planes = (uint32_t)n_planes, job.n_tiles = n_tiles;
    if (counted)
        hipLaunchKernelGGL(rle_tile_sums_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, s, job, sums);
    else
        hipLaunchKernelGGL(rle_count_kernel, dim3(n_tiles), dim3(256), 0, s, job, sums);
    // persistent emit: no more workgroups than are resident at once (three per CU: 53 KB of LDS each)
    int cus = 0;
    if (jpegamd_cu_count(&cus))
        return 1;
    const uint32_t emit_grid = n_tiles < (uint32_t)(3 * cus) ? n_tiles : (uint32_t)(3 * cus);
    hipLaunchKernelGGL(rle_emit_kernel, dim3(emit_grid), dim3(256), 0, s, job, sums);
    JPEGAMD_HIPCHK(hipGetLastError());
    return 0;
}

// ---- the entry point that allocates nothing (include/jpeg_amd.h) ------------------------
// Its scratch (the per-workgroup pair counts that the three kernels hand to each other) is a
// file-scope device array: every GPU the library runs on holds its own copy at its own
// address for the life of the process.  `scratch` below is only the HOST-side handle of that
// array -- never a device address: the address on the current device comes from
// hipGetSymbolAddress, once per device, and that pointer is what the kernels are given.
constexpr size_t kNoAllocMaxBlocks = (size_t)1 << 22; // 4 Mi blocks = a 16384 x 16384 luma plane
constexpr size_t kNoAllocScratchWords = kNoAllocMaxBlocks / kBlocksPerWg + 64; // one uint32 per tile

} // namespace

static __device__ __align__(256) uint32_t scratch[kNoAllocScratchWords]; // 64.25 KiB per GPU

namespace {

std::mutex g_scratch_mu;
uint32_t *g_scratch_ptr[kMaxDevices]; // device address of `scratch` on each device, once looked up

int noalloc_scratch_for_current_device(uint32_t **out, hipStream_t stream = nullptr, bool check_capture = false)
{
    int dev = -1;
    JPEGAMD_HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices)
        return jpegamd_fail("rle_encode_plane_noalloc: device %d beyond the %d this library keeps scratch for", dev,
                            kMaxDevices);
    // the common case takes no lock: the address never changes once it is there
    if (uint32_t *p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE)) {
        *out = p;
        return 0;
    }
    if (check_capture) { // a symbol lookup inside a stream capture would break or invalidate the capture
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
            return jpegamd_fail("rle_encode_plane_noalloc: device %d has not been set up and the stream is capturing; call "
                                "jpegamd_init(%d) (or one un-captured call) before capturing", dev, dev);
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        JPEGAMD_HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p)
            return jpegamd_fail("rle_encode_plane_noalloc: hipGetSymbolAddress returned a null address");
        __atomic_store_n(&g_scratch_ptr[dev], static_cast<uint32_t *>(p), __ATOMIC_RELEASE);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

} // namespace

int jpegamd_rle_encode_counted(int16_t *const coef[3], const size_t nb[3], const uint8_t *const counts[3],
                               int32_t *const pairs[3], const size_t cap_pairs[3], uint32_t *const block_offsets[3],
                               uint32_t *sums, void *stream)
{
    RlePlane pl[3];
    for (int k = 0; k < 3; ++k) {
        if (check_rle_encode(coef[k], nb[k], pairs[k], block_offsets[k], "encode_image_rle"))
            return 1;
        pl[k] = RlePlane{coef[k], block_offsets[k], reinterpret_cast<int2 *>(pairs[k]), cap_pairs[k], counts[k],
                         (uint32_t)nb[k], 0};
    }
    return launch_rle_encode(pl, 3, sums, static_cast<hipStream_t>(stream));
}

// jpegamd_init(): look the scratch address up while set-up is going on, with the device current
int jpegamd_rle_noalloc_setup()
{
    uint32_t *p = nullptr;
    return noalloc_scratch_for_current_device(&p);
}


extern "C" {

size_t jpegamd_rle_workspace_bytes(size_t nb)
{
    // encode: one sum per 256 blocks; decode: one per 1024 pairs, at most 64 pairs per block
    const size_t n = ceil_div(nb, kBlocksPerWg) > ceil_div(nb * 64, kPairsPerWg) ? ceil_div(nb, kBlocksPerWg)
                                                                                 : ceil_div(nb * 64, kPairsPerWg);
    // the decoder's sums are 64-bit (untrusted run lengths); jpegamd_encode_image_rle_dev keeps one count
    // byte per block (three arrays, 256-byte aligned) and one uint32 per 256 blocks there
    const size_t dec = (n + 64) * sizeof(uint64_t), enc = nb + 3 * 256 + (ceil_div(nb, kBlocksPerWg) + 4) * sizeof(uint32_t);
    return dec > enc ? dec : enc;
}

int jpegamd_rle_encode_plane_dev(const int16_t *d_coef_zz, size_t nb, int32_t *d_pairs, size_t cap_pairs,
                                 uint32_t *d_block_offsets, void *d_workspace, size_t workspace_bytes,
                                 void *stream)
{
    if (!d_workspace || (reinterpret_cast<uintptr_t>(d_workspace) & 7))
        return jpegamd_fail("rle_encode_plane: workspace is null or not 8-byte aligned");
    if (check_rle_encode(d_coef_zz, nb, d_pairs, d_block_offsets, "rle_encode_plane"))
        return 1;
    if (workspace_bytes < jpegamd_rle_workspace_bytes(nb))
        return jpegamd_fail("rle_encode_plane: workspace of %zu bytes, need %zu", workspace_bytes,
                            jpegamd_rle_workspace_bytes(nb));
    const RlePlane pl{d_coef_zz, d_block_offsets, reinterpret_cast<int2 *>(d_pairs), cap_pairs, nullptr, (uint32_t)nb, 0};
    return launch_rle_encode(&pl, 1, static_cast<uint32_t *>(d_workspace), static_cast<hipStream_t>(stream));
}

int jpegamd_rle_encode_image_dev(const int16_t *const d_coef_zz[3], const size_t nb[3], int32_t *const d_pairs[3],
                                 const size_t cap_pairs[3], uint32_t *const d_block_offsets[3], void *d_workspace,
                                 size_t workspace_bytes, void *stream)
{
    if (!d_coef_zz || !nb || !d_pairs || !cap_pairs || !d_block_offsets || !d_workspace ||
        (reinterpret_cast<uintptr_t>(d_workspace) & 7))
        return jpegamd_fail("rle_encode_image: null pointer or workspace not 8-byte aligned");
    RlePlane pl[3];
    size_t all = 0;
    for (int k = 0; k < 3; ++k) {
        if (check_rle_encode(d_coef_zz[k], nb[k], d_pairs[k], d_block_offsets[k], "rle_encode_image"))
            return 1;
        pl[k] = RlePlane{d_coef_zz[k], d_block_offsets[k], reinterpret_cast<int2 *>(d_pairs[k]), cap_pairs[k],
                         nullptr, (uint32_t)nb[k], 0};
        all += nb[k];
    }
    if (workspace_bytes < jpegamd_rle_workspace_bytes(all))
        return jpegamd_fail("rle_encode_image: workspace of %zu bytes, need %zu (jpegamd_rle_workspace_bytes of the "
                            "three planes' blocks together)", workspace_bytes, jpegamd_rle_workspace_bytes(all));
    return launch_rle_encode(pl, 3, static_cast<uint32_t *>(d_workspace), static_cast<hipStream_t>(stream));
}

size_t jpegamd_rle_noalloc_max_blocks(void) { return kNoAllocMaxBlocks; }

int jpegamd_rle_encode_plane_noalloc_dev(const int16_t *d_coef_zz, size_t nb, int32_t *d_pairs, size_t cap_pairs,
                                         uint32_t *d_block_offsets, void *stream)
{
    if (check_rle_encode(d_coef_zz, nb, d_pairs, d_block_offsets, "rle_encode_plane_noalloc"))
        return 1;
    if (nb > kNoAllocMaxBlocks)
        return jpegamd_fail("rle_encode_plane_noalloc: nb=%zu exceeds the built-in scratch (%zu blocks); use "
                            "jpegamd_rle_encode_plane_dev with a workspace", nb, kNoAllocMaxBlocks);
    uint32_t *work = nullptr;
    if (noalloc_scratch_for_current_device(&work, static_cast<hipStream_t>(stream), true))
        return 1;
    const RlePlane pl{d_coef_zz, d_block_offsets, reinterpret_cast<int2 *>(d_pairs), cap_pairs, nullptr, (uint32_t)nb, 0};
    return launch_rle_encode(&pl, 1, work, static_cast<hipStream_t>(stream));
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
    if ((reinterpret_cast<uintptr_t>(d_pairs) & 7) || (reinterpret_cast<uintptr_t>(d_workspace) & 7))
        return jpegamd_fail("rle_decode_plane: pairs and workspace must be 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    JPEGAMD_HIPCHK(hipMemsetAsync(d_coef_zz, 0, nb * 64 * sizeof(int16_t), s));
    if (n_pairs == 0)
        return 0;
    u64 *sums = static_cast<u64 *>(d_workspace);
    const unsigned nwg = (unsigned)ceil_div(n_pairs, kPairsPerWg);
    const int2 *pairs = reinterpret_cast<const int2 *>(d_pairs);
    const u64 n_values = (u64)nb * 64;
    hipLaunchKernelGGL(rld_count_kernel, dim3(nwg), dim3(256), 0, s, pairs, n_pairs, sums, n_values);
    hipLaunchKernelGGL(scan_sums64_kernel, dim3(1), dim3(1024), 0, s, sums, (size_t)nwg);
    hipLaunchKernelGGL(rld_fill_kernel, dim3(nwg), dim3(256), 0, s, pairs, n_pairs, sums, d_coef_zz, n_values);
    JPEGAMD_HIPCHK(hipGetLastError());
    return 0;
}

} // extern "C"
