// entropy_decode_kernels.hip -- the reference's huffman_decode (encode_decode.cpp:468-482) of one
// plane on the device (gfx950).
//
// A plane's bit stream has no restart points, but prefix codes resynchronise: a decoder started
// at a wrong bit falls into step with the true one after a few codes (the host decoder,
// PlaneDecoder in host/bin_container.cpp, proved the scheme on threads).  Here the stream is
// cut into subsequences of 256 bits, one per thread:
//   hd_sync_kernel   every thread decodes its subsequence from an assumed start -- at first its
//                    own first bit -- counting codes until it passes its end; then, round by
//                    round within the workgroup, a thread whose predecessor ended somewhere
//                    else than it assumed starts again from there, until nothing changes.  A
//                    second and third launch do the same across workgroup boundaries (each
//                    workgroup's first subsequence takes the previous workgroup's last end);
//                    the third must find nothing to change, otherwise the host decoder takes over.
//   hd_scan_kernel   exclusive scan of the subsequences' code counts = where each one's symbols go
//   hd_emit_kernel   every thread decodes its subsequence again from its true start, writing symbols
// Codes are looked up 12 bits at a time in a table held in LDS (the average code is 2-3 bits);
// longer ones continue bit by bit down the tree.  Exactly the bit-serial walk's symbols: every bit
// of every payload byte is decoded, the zero padding included (encode_decode.cpp:1261-1263), and a
// code the stream ends in is dropped.  All integer work.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "host/dev_alloc.h"
#include "host/entropy_internal.h"
#include "host_common.h"

namespace {

constexpr int kSubBits = 256; // bits per subsequence (one thread)
constexpr int kThreads = 256;
constexpr uint32_t kNoEnd = 0xFFFFFFFFu;

struct Stream {
    const uint32_t *words; // payload, zero padded to whole words plus two
    uint32_t n_words;      // words that hold payload bytes
    uint32_t n_bits;
    const int32_t *left, *right, *value;
};

// 32 bits of the stream from bit `pos` on, MSB first; bits beyond the payload read as zero
__device__ __forceinline__ uint32_t peek32(const Stream &s, uint32_t pos)
{
    const uint32_t i = pos >> 5, sh = pos & 31;
    const uint32_t hi = i < s.n_words ? __builtin_bswap32(s.words[i]) : 0u;
    const uint32_t lo = i + 1 < s.n_words ? __builtin_bswap32(s.words[i + 1]) : 0u;
    return (uint32_t)(((((unsigned long long)hi << 32) | lo) << sh) >> 32);
}

__device__ __forceinline__ uint32_t bit_at(const Stream &s, uint32_t pos)
{
    return (__builtin_bswap32(s.words[pos >> 5]) >> (31 - (pos & 31))) & 1u;
}

// Decodes codes from the boundary `pos` while pos < stop; returns the boundary reached (>= stop,
// or where the stream ends inside a code: then kNoEnd-like `stalled`) and counts the codes.
// kEmit: symbols go to out[n0 + k] while that index is below cap.
template <bool kEmit>
__device__ __forceinline__ uint32_t decode_range(const Stream &s, const uint32_t *win /* LDS */, uint32_t pos,
                                                 uint32_t stop, uint32_t &count, int32_t *__restrict__ out,
                                                 size_t n0, size_t cap)
{
    uint32_t n = 0;
    while (pos < stop) {
        const uint32_t avail = s.n_bits - pos;
        const uint32_t e = win[peek32(s, pos) >> (32 - kDecodeWindow)];
        uint32_t len = e >> 24;
        int sym;
        if (len) {
            if (len > avail)
                break; // the stream ends inside this code: the reference emits nothing for it either
            sym = (int)(e & 0xFFFFu) - 32768;
        } else {
            if (avail <= (uint32_t)kDecodeWindow)
                break;
            int node = (int)e; // longer than the window: bit by bit from the node reached
            uint32_t p = pos + kDecodeWindow;
            while (s.left[node] >= 0 && p < s.n_bits)
                node = bit_at(s, p++) ? s.right[node] : s.left[node];
            if (s.left[node] >= 0)
                break;
            len = p - pos;
            sym = s.value[node];
        }
        if (kEmit && n0 + n < cap)
            out[n0 + n] = sym;
        ++n;
        pos += len;
    }
    count = n;
    return pos;
}

// round 0: wg_start_in == nullptr, every workgroup assumes its own first bit.  Later rounds:
// wg_start_in[w] = the end the previous round found for workgroup w - 1's last subsequence; a
// workgroup whose assumption was right already does nothing.  *changed is raised by any workgroup
// that had to start again.
__global__ __launch_bounds__(kThreads) void hd_sync_kernel(Stream s, const uint32_t *__restrict__ window,
                                                           uint32_t n_sub, uint32_t *__restrict__ sub_start,
                                                           uint32_t *__restrict__ sub_count,
                                                           const uint32_t *__restrict__ wg_end_in,
                                                           uint32_t *__restrict__ wg_end_out,
                                                           uint32_t *__restrict__ changed)
{
    __shared__ uint32_t win[1 << kDecodeWindow];
    __shared__ uint32_t ends[kThreads];
    __shared__ int any;
    const uint32_t wg = blockIdx.x, t = threadIdx.x;
    const uint32_t sub = wg * kThreads + t;
    const uint32_t first_bit = wg * kThreads * kSubBits;
    uint32_t wg_start = first_bit;
    if (wg_end_in) {
        wg_start = wg ? wg_end_in[wg - 1] : 0u;
        const uint32_t assumed = sub_start[wg * kThreads]; // uniform
        if (wg_start == assumed) {
            if (t == 0)
                wg_end_out[wg] = wg_end_in[wg];
            return; // nothing to redo here
        }
        if (t == 0)
            atomicOr(changed, 1u);
    }
    for (int i = t; i < (1 << kDecodeWindow); i += kThreads)
        win[i] = window[i];
    __syncthreads();
    const bool live = sub < n_sub;
    const uint32_t lo = sub * kSubBits;
    const uint32_t hi = lo + kSubBits < s.n_bits ? lo + kSubBits : s.n_bits;
    // my start: the workgroup's for its first subsequence, else my own first bit (assumption)
    // (later rounds: the others keep the start they had settled on, most of them still hold)
    uint32_t start = t == 0 ? wg_start : (wg_end_in && live ? sub_start[sub] : lo), end = 0, count = 0;
    if (live)
        end = start < hi ? decode_range<false>(s, win, start, hi, count, nullptr, 0, 0) : start;
    ends[t] = live ? end : kNoEnd;
    for (int round = 0; round < kThreads; ++round) { // at most one subsequence settles per round, usually all at once
        __syncthreads();
        if (t == 0)
            any = 0;
        __syncthreads();
        if (live && t > 0) {
            const uint32_t want = ends[t - 1]; // where my predecessor stopped = where I begin
            if (want != start) {
                start = want;
                count = 0;
                end = start < hi ? decode_range<false>(s, win, start, hi, count, nullptr, 0, 0) : start;
                any = 1;
            }
        }
        __syncthreads();
        if (live)
            ends[t] = end;
        if (!any)
            break;
    }
    __syncthreads();
    if (live) {
        sub_start[sub] = start;
        sub_count[sub] = count;
    }
    // the end of the workgroup's last live subsequence opens the next workgroup
    const uint32_t last = n_sub - wg * kThreads < (uint32_t)kThreads ? n_sub - wg * kThreads - 1 : kThreads - 1;
    if (t == last)
        wg_end_out[wg] = end;
}

// exclusive scan of n counts in place by ONE workgroup; the total goes to *total (64-bit)
__global__ __launch_bounds__(1024) void hd_scan_kernel(uint32_t *__restrict__ v, uint32_t n,
                                                       unsigned long long *__restrict__ total)
{
    // one workgroup, eight consecutive counts per thread and pass: a plane of 8192^2 has about
    // 110 000 subsequences, i.e. 14 passes
    constexpr uint32_t kPer = 8;
    __shared__ unsigned long long wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < n; base += 1024 * kPer) { // uniform trip count
        const uint32_t i = base + threadIdx.x * kPer;
        uint32_t x[kPer];
        if (i + kPer <= n) {
            const uint4 a = *reinterpret_cast<const uint4 *>(v + i), b = *reinterpret_cast<const uint4 *>(v + i + 4);
            x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < kPer; ++j)
                x[j] = i + j < n ? v[i + j] : 0u;
        }
        unsigned long long own = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j)
            own += x[j];
        unsigned long long inc = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long y = __shfl_up(inc, d, 64);
            if (lane >= d)
                inc += y;
        }
        if (lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        unsigned long long before = 0, all = 0;
        for (int w = 0; w < 16; ++w) {
            before += w < wave ? wave_tot[w] : 0;
            all += wave_tot[w];
        }
        __syncthreads();
        // offsets beyond 32 bits cannot be reached: the caller's buffer (cap) is far smaller and
        // the emit pass drops what lies beyond it
        unsigned long long run = carry + before + inc - own;
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) {
            if (i + j < n)
                v[i + j] = (uint32_t)(run < 0xFFFFFFFFull ? run : 0xFFFFFFFFull);
            run += x[j];
        }
        carry += all;
    }
    if (threadIdx.x == 0)
        *total = carry;
}

__global__ __launch_bounds__(kThreads) void hd_emit_kernel(Stream s, const uint32_t *__restrict__ window,
                                                           uint32_t n_sub, const uint32_t *__restrict__ sub_start,
                                                           const uint32_t *__restrict__ sub_offset,
                                                           int32_t *__restrict__ out, size_t cap)
{
    __shared__ uint32_t win[1 << kDecodeWindow];
    for (int i = threadIdx.x; i < (1 << kDecodeWindow); i += kThreads)
        win[i] = window[i];
    __syncthreads();
    const uint32_t sub = blockIdx.x * kThreads + threadIdx.x;
    if (sub >= n_sub)
        return;
    const uint32_t lo = sub * kSubBits;
    const uint32_t hi = lo + kSubBits < s.n_bits ? lo + kSubBits : s.n_bits;
    const uint32_t start = sub_start[sub];
    uint32_t count;
    if (start < hi)
        (void)decode_range<true>(s, win, start, hi, count, out, (size_t)sub_offset[sub], cap);
}

typedef JpegamdDevMem DevMem;

} // namespace

int jpegamd_device_decode_plane(const uint8_t *payload, size_t len, const uint8_t *dict, int dsz, int32_t *d_sym,
                                size_t cap, size_t *n)
{
    *n = 0;
    if (len == 0)
        return 0;
    if (len >= ((size_t)1 << 28)) // bit positions are 32-bit here
        return 2;
    JpegamdDecodeTables tab;
    const int rc = jpegamd_decode_tables(dict, dsz, tab);
    if (rc)
        return rc;
    const uint32_t n_bits = (uint32_t)(len * 8);
    const uint32_t n_words = (uint32_t)((len + 3) / 4);
    const uint32_t n_sub = (n_bits + kSubBits - 1) / kSubBits;
    const uint32_t n_wg = (n_sub + kThreads - 1) / kThreads;
    const size_t nodes = tab.left.size();
    DevMem d_pay, d_tab, d_work;
    // work: total (8 B), changed flag (+ pad), sub_count (16-byte aligned: the scan reads it four at a
    // time), sub_start (n_sub each, rounded up to four), wg ends a/b (n_wg each)
    const size_t n_sub_pad = ((size_t)n_sub + 3) & ~(size_t)3;
    const size_t work_words = 4 + 2 * n_sub_pad + 2 * (size_t)n_wg;
    if (d_pay.alloc(((size_t)n_words + 2) * 4) || d_tab.alloc((tab.window.size() + 3 * nodes) * 4) ||
        d_work.alloc(work_words * 4))
        return jpegamd_fail("entropy decode: out of device memory");
    // payload: the last word and the two spare ones zeroed, then the bytes
    JPEGAMD_HIPCHK(hipMemsetAsync(d_pay.as<uint32_t>() + (n_words - 1), 0, 12, nullptr));
    if (len >= (1u << 20)) {
        JPEGAMD_HIPCHK(hipDeviceSynchronize());
        if (jpegamd_staged_copy_linear(d_pay.p, const_cast<uint8_t *>(payload), len, true))
            return 1;
    } else {
        JPEGAMD_HIPCHK(hipMemcpy(d_pay.p, payload, len, hipMemcpyHostToDevice));
    }
    uint32_t *d_window = d_tab.as<uint32_t>();
    int32_t *d_left = reinterpret_cast<int32_t *>(d_window + tab.window.size());
    int32_t *d_right = d_left + nodes, *d_value = d_right + nodes;
    JPEGAMD_HIPCHK(hipMemcpy(d_window, tab.window.data(), tab.window.size() * 4, hipMemcpyHostToDevice));
    JPEGAMD_HIPCHK(hipMemcpy(d_left, tab.left.data(), nodes * 4, hipMemcpyHostToDevice));
    JPEGAMD_HIPCHK(hipMemcpy(d_right, tab.right.data(), nodes * 4, hipMemcpyHostToDevice));
    JPEGAMD_HIPCHK(hipMemcpy(d_value, tab.value.data(), nodes * 4, hipMemcpyHostToDevice));
    unsigned long long *total = d_work.as<unsigned long long>();
    uint32_t *changed = d_work.as<uint32_t>() + 2;
    uint32_t *sub_count = changed + 2, *sub_start = sub_count + n_sub_pad;
    uint32_t *ends_a = sub_start + n_sub_pad, *ends_b = ends_a + n_wg;
    JPEGAMD_HIPCHK(hipMemsetAsync(d_work.p, 0, 16, nullptr));
    const Stream s{d_pay.as<uint32_t>(), n_words, n_bits, d_left, d_right, d_value};
    hipLaunchKernelGGL(hd_sync_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, s, d_window, n_sub, sub_start, sub_count,
                       (const uint32_t *)nullptr, ends_a, changed);
    // across workgroup boundaries: twice, the second time nothing may change any more
    hipLaunchKernelGGL(hd_sync_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, s, d_window, n_sub, sub_start, sub_count,
                       (const uint32_t *)ends_a, ends_b, changed);
    JPEGAMD_HIPCHK(hipMemsetAsync(changed, 0, 4, nullptr));
    hipLaunchKernelGGL(hd_sync_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, s, d_window, n_sub, sub_start, sub_count,
                       (const uint32_t *)ends_b, ends_a, changed);
    hipLaunchKernelGGL(hd_scan_kernel, dim3(1), dim3(1024), 0, nullptr, sub_count, n_sub, total);
    JPEGAMD_HIPCHK(hipGetLastError());
    struct {
        uint32_t changed, pad;
        unsigned long long total;
    } back = {0, 0, 0};
    JPEGAMD_HIPCHK(hipMemcpy(&back.changed, changed, 4, hipMemcpyDeviceToHost));
    JPEGAMD_HIPCHK(hipMemcpy(&back.total, total, 8, hipMemcpyDeviceToHost));
    if (back.changed)
        return 2; // a chunk that does not fall into step within a workgroup's reach: the host decoder knows how
    *n = (size_t)back.total;
    if (cap && d_sym) {
        hipLaunchKernelGGL(hd_emit_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, s, d_window, n_sub, sub_start,
                           sub_count, d_sym, cap);
        JPEGAMD_HIPCHK(hipGetLastError());
        JPEGAMD_HIPCHK(hipDeviceSynchronize()); // the tables and the payload go out of scope
    }
    return 0;
}
