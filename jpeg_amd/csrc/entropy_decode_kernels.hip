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
//   hd_emit_kernel   every thread decodes its subsequence again from its true start, writing symbols; where
//                    they go is the sum of the counts before it: the workgroups' totals before its own (left
//                    by hd_sync_kernel) added up by the workgroup itself + a scan of its 256 counts (round 2
//                    had a one-workgroup scan kernel and a wait between: 68 us + a launch at 8192^2)
// Codes are looked up 12 bits at a time in a table held in LDS (the average code is 2-3 bits);
// longer ones continue bit by bit down the tree.  Exactly the bit-serial walk's symbols: every bit
// of every payload byte is decoded, the zero padding included (encode_decode.cpp:1261-1263), and a
// code the stream ends in is dropped.  All integer work.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host/dev_alloc.h"
#include "host/entropy_internal.h"
#include "host_common.h"

namespace {

constexpr int kSubBits = 256; // bits per subsequence (one thread)
constexpr int kThreads = 256;
constexpr uint32_t kNoEnd = 0xFFFFFFFFu;

// A plane's payload sits where the file has it: the three payloads are uploaded as ONE range (they are
// contiguous in the .bin, encode_decode.cpp:1178-1186), so a plane begins at any byte of a word:
// `words` is the word holding its first byte and bit0 = 8 x (the byte's place in that word).  Stream
// positions are plane relative.  Bits read beyond the plane's end belong to whatever follows in the
// buffer; every use of them is cut off by the bit count (a code is accepted only if it ends at or
// before n_bits, and a prefix code's first `len` bits identify it whatever follows).
struct Stream {
    const uint32_t *words;
    uint32_t n_words; // words that hold bytes of this plane
    uint32_t n_bits;
    uint32_t bit0;
    const int32_t *left, *right, *value;
};

__device__ __forceinline__ uint32_t bit_at(const Stream &s, uint32_t pos)
{
    const uint32_t q = pos + s.bit0;
    return (__builtin_bswap32(s.words[q >> 5]) >> (31 - (q & 31))) & 1u;
}

// Up to three planes go through one launch of each kernel (round 3): a workgroup finds its plane from
// the first workgroup index of each.
struct HdPlane {
    Stream s;
    const uint32_t *window;
    uint32_t n_sub, wg_begin;
    uint32_t *sub_start, *sub_count, *ends_a, *ends_b;
    uint32_t *wg_count; // codes per workgroup of 256 subsequences
    uint32_t *changed;
    unsigned long long *total;
    int32_t *out;
    size_t cap;
};
struct HdJob {
    HdPlane p[3];
};
__device__ __forceinline__ const HdPlane &hd_plane_of(const HdJob &job, uint32_t wgid)
{
    return job.p[wgid >= job.p[2].wg_begin ? 2 : wgid >= job.p[1].wg_begin ? 1 : 0];
}

// A workgroup's share of the payload in LDS (round 3): its 256 subsequences are 8 KiB of consecutive words,
// byte-swapped once on the way in, + two words because a plane starts at any bit of a word (bit0) and a
// 32-bit peek at the last bit reaches into the next word.  A peek is then two LDS reads instead of two
// dependent trips to the L2 per code (1.9 us per symbol in round 2's emit kernel: profiles/r03_f).
constexpr int kPayWords = kThreads * kSubBits / 32 + 2;
struct PayLds {
    const uint32_t *w; // LDS, kPayWords
    uint32_t word0;    // index in s.words of w[0]
};
__device__ __forceinline__ void load_payload(const Stream &s, uint32_t first_bit, uint32_t *pay /* LDS */, PayLds &pl)
{
    pl.w = pay;
    pl.word0 = (first_bit + s.bit0) >> 5;
    for (int i = threadIdx.x; i < kPayWords; i += kThreads) {
        const uint32_t j = pl.word0 + (uint32_t)i;
        pay[i] = j < s.n_words ? __builtin_bswap32(s.words[j]) : 0u;
    }
}
// 32 bits from plane bit `pos` on, for first_bit <= pos < first_bit + kThreads * kSubBits
__device__ __forceinline__ uint32_t peek32(const Stream &s, const PayLds &pl, uint32_t pos)
{
    const uint32_t q = pos + s.bit0;
    const uint32_t i = (q >> 5) - pl.word0, sh = q & 31;
    return (uint32_t)(((((unsigned long long)pl.w[i] << 32) | pl.w[i + 1]) << sh) >> 32);
}

// Decodes codes from the boundary `pos` while pos < stop; returns the boundary reached (>= stop,
// or where the stream ends inside a code: then kNoEnd-like `stalled`) and counts the codes.
// kEmit: symbols go to out[n0 + k] while that index is below cap.
template <bool kEmit>
__device__ __forceinline__ uint32_t decode_range(const Stream &s, const PayLds &pay, const uint32_t *win /* LDS */,
                                                 uint32_t pos, uint32_t stop, uint32_t &count,
                                                 int32_t *__restrict__ out, size_t n0, size_t cap)
{
    uint32_t n = 0;
    while (pos < stop) {
        const uint32_t avail = s.n_bits - pos;
        const uint32_t e = win[peek32(s, pay, pos) >> (32 - kDecodeWindow)];
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
__global__ __launch_bounds__(kThreads) void hd_sync_kernel(HdJob job, int phase)
{
    __shared__ uint32_t win[1 << kDecodeWindow];
    __shared__ uint32_t pay_words[kPayWords];
    __shared__ uint32_t ends[kThreads];
    __shared__ int any;
    const HdPlane &pl = hd_plane_of(job, blockIdx.x);
    const Stream &s = pl.s;
    const uint32_t *__restrict__ window = pl.window;
    const uint32_t n_sub = pl.n_sub;
    uint32_t *__restrict__ sub_start = pl.sub_start, *__restrict__ sub_count = pl.sub_count;
    // phase 0: every workgroup assumes its own first bit, ends -> a; 1: a -> b; 2: b -> a
    const uint32_t *__restrict__ wg_end_in = phase == 0 ? nullptr : phase == 1 ? pl.ends_a : pl.ends_b;
    uint32_t *__restrict__ wg_end_out = phase == 1 ? pl.ends_b : pl.ends_a;
    uint32_t *__restrict__ changed = pl.changed;
    const uint32_t wg = blockIdx.x - pl.wg_begin, t = threadIdx.x;
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
    PayLds pay;
    load_payload(s, first_bit, pay_words, pay); // (every start lies at or after the workgroup's first bit: a
    __syncthreads();                            //  subsequence begins where its predecessor passed its own end)
    const bool live = sub < n_sub;
    const uint32_t lo = sub * kSubBits;
    const uint32_t hi = lo + kSubBits < s.n_bits ? lo + kSubBits : s.n_bits;
    // my start: the workgroup's for its first subsequence, else my own first bit (assumption)
    // (later rounds: the others keep the start they had settled on, most of them still hold)
    uint32_t start = t == 0 ? wg_start : (wg_end_in && live ? sub_start[sub] : lo), end = 0, count = 0;
    if (live)
        end = start < hi ? decode_range<false>(s, pay, win, start, hi, count, nullptr, 0, 0) : start;
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
                end = start < hi ? decode_range<false>(s, pay, win, start, hi, count, nullptr, 0, 0) : start;
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
    // the workgroup's codes (a workgroup that returned above has left its earlier total, which still holds)
    uint32_t c = live ? count : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        c += __shfl_down(c, d, 64);
    if ((t & 63) == 0)
        ends[t >> 6] = c; // (the rounds are over: ends[] is free; the barrier above separates the last read)
    __syncthreads();
    if (t == 0)
        pl.wg_count[wg] = ends[0] + ends[1] + ends[2] + ends[3];
}

__global__ __launch_bounds__(kThreads) void hd_emit_kernel(HdJob job)
{
    __shared__ uint32_t win[1 << kDecodeWindow];
    __shared__ uint32_t pay_words[kPayWords];
    __shared__ unsigned long long part[kThreads / 64];
    __shared__ uint32_t wave_codes[kThreads / 64];
    const HdPlane &pl = hd_plane_of(job, blockIdx.x);
    if (*pl.changed)
        return; // (uniform) the third sync pass still moved a boundary: the host decoder takes this plane
    const Stream &s = pl.s;
    const uint32_t wg = blockIdx.x - pl.wg_begin, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    for (int i = t; i < (1 << kDecodeWindow); i += kThreads)
        win[i] = pl.window[i];
    PayLds pay;
    load_payload(s, wg * kThreads * kSubBits, pay_words, pay);
    // codes before this workgroup: the totals of the workgroups before it (a few hundred words)
    unsigned long long before = 0;
    for (uint32_t i = t; i < wg; i += kThreads)
        before += pl.wg_count[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        before += __shfl_down(before, d, 64);
    const uint32_t sub = wg * kThreads + t;
    const bool live = sub < pl.n_sub;
    const uint32_t mine = live ? pl.sub_count[sub] : 0u;
    uint32_t inc = mine; // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 64);
        if (lane >= d)
            inc += y;
    }
    if (lane == 0)
        part[wave] = before;
    if (lane == 63)
        wave_codes[wave] = inc;
    __syncthreads();
    unsigned long long n0 = part[0] + part[1] + part[2] + part[3] + (inc - mine);
    for (int w = 0; w < wave; ++w)
        n0 += wave_codes[w];
    if (live && sub == pl.n_sub - 1)
        *pl.total = n0 + mine; // the plane's number of symbols
    if (!live || pl.cap == 0)
        return; // (a sizing call has nowhere to write)
    const uint32_t lo = sub * kSubBits;
    const uint32_t hi = lo + kSubBits < s.n_bits ? lo + kSubBits : s.n_bits;
    const uint32_t start = pl.sub_start[sub];
    uint32_t count;
    if (start < hi)
        (void)decode_range<true>(s, pay, win, start, hi, count, pl.out, (size_t)n0, pl.cap);
}

typedef JpegamdDevMem DevMem;

// Pinned host scratch of the calling thread (tables up, status words back): grows, never shrinks, lives as
// long as the thread.  Small asynchronous copies from and to it cost a queue entry each instead of the
// staging and the wait of a pageable hipMemcpy.
struct PinnedScratch {
    void *p = nullptr;
    size_t n = 0;
    ~PinnedScratch()
    {
        if (p)
            (void)hipHostFree(p);
    }
    void *ensure(size_t bytes)
    {
        if (n < bytes) {
            if (p)
                (void)hipHostFree(p);
            p = nullptr, n = 0;
            if (hipHostMalloc(&p, bytes + bytes / 4, hipHostMallocDefault) != hipSuccess)
                return nullptr;
            n = bytes + bytes / 4;
        }
        return p;
    }
};

} // namespace

void *jpegamd_pinned_scratch(size_t bytes)
{
    static thread_local PinnedScratch scratch;
    return scratch.ensure(bytes);
}

// huffman_decode of the three planes of a file in one go (round 3): ONE upload of the payload range, ONE of
// the three planes' tables, every kernel of every plane queued without a wait, ONE read-back of the three
// (changed, total) words -- the only synchronisation -- and then the emit kernels.  Round 2 ran plane
// after plane with five synchronous table copies, two read-backs and a device synchronisation each
// (1.7 ms for 0.8 ms of kernels at 8192^2).  rc[k]: 0 decoded, 2 the host decoder must take plane k.
int jpegamd_device_decode_planes(const JpegamdBinParts &parts, int32_t *const d_sym[3], const size_t cap[3], size_t n[3],
                                 int rc[3])
{
    JpegamdDecodeTables tab[3];
    const bool trace2 = getenv("JPEGAMD_TRACE") && getenv("JPEGAMD_TRACE")[0] == '2';
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    size_t off[3], span = 0, tab_words[3], tab_total = 0;
    bool live[3];
    for (int k = 0; k < 3; ++k) {
        n[k] = 0, rc[k] = 0;
        off[k] = (size_t)(parts.payload[k] - parts.payload[0]); // contiguous in the file (jpegamd_bin_parse)
        span = off[k] + parts.len[k];
        live[k] = false;
        tab_words[k] = 0;
        if (parts.len[k] == 0)
            continue;
        if (parts.len[k] >= ((size_t)1 << 28)) { // bit positions are 32-bit here
            rc[k] = 2;
            continue;
        }
        rc[k] = jpegamd_decode_tables(parts.dict[k], parts.dsz[k], tab[k]);
        if (rc[k] == 1)
            return 1;
        if (rc[k] == 0) {
            live[k] = true;
            tab_words[k] = tab[k].window.size() + 3 * tab[k].left.size();
            tab_total += tab_words[k];
        }
    }
    if (!live[0] && !live[1] && !live[2])
        return 0;
    // device memory: the payload range (+ two spare words), the tables, and per plane: sub_count (16-byte
    // aligned: the scan reads four at a time), sub_start, workgroup ends a/b; in front of them the three
    // status records {changed, pad, total(8)}
    uint32_t n_sub[3] = {0, 0, 0}, n_wg[3] = {0, 0, 0};
    size_t work_off[3], work_words = 16; // 3 x 4 words of status, padded to 64 bytes
    for (int k = 0; k < 3; ++k) {
        if (!live[k])
            continue;
        const uint32_t n_bits = (uint32_t)(parts.len[k] * 8);
        n_sub[k] = (n_bits + kSubBits - 1) / kSubBits;
        n_wg[k] = (n_sub[k] + kThreads - 1) / kThreads;
        work_off[k] = work_words;
        const size_t n_sub_pad = ((size_t)n_sub[k] + 3) & ~(size_t)3;
        work_words += 2 * n_sub_pad + 3 * (((size_t)n_wg[k] + 3) & ~(size_t)3);
    }
    const auto t1 = now();
    DevMem d_pay, d_tab, d_work;
    if (d_pay.alloc(((span + 3) / 4 + 2) * 4) || d_tab.alloc(tab_total * 4) || d_work.alloc(work_words * 4))
        return jpegamd_fail("entropy decode: out of device memory");
    uint32_t *pinned = static_cast<uint32_t *>(jpegamd_pinned_scratch(tab_total * 4 + 64));
    if (!pinned)
        return jpegamd_fail("entropy decode: no pinned host memory");
    const auto t2 = now();
    if (span >= (1u << 20)) {
        JPEGAMD_HIPCHK(hipDeviceSynchronize()); // (the copier runs on its own streams)
        if (jpegamd_staged_copy_linear(d_pay.p, const_cast<uint8_t *>(parts.payload[0]), span, true))
            return 1;
    } else {
        JPEGAMD_HIPCHK(hipMemcpy(d_pay.p, parts.payload[0], span, hipMemcpyHostToDevice));
    }
    const auto t3 = now();
    uint32_t *status_back = pinned; // 16 words
    uint32_t *tab_host = pinned + 16;
    {
        uint32_t *w = tab_host;
        for (int k = 0; k < 3; ++k) {
            if (!live[k])
                continue;
            const size_t nodes = tab[k].left.size();
            std::memcpy(w, tab[k].window.data(), tab[k].window.size() * 4), w += tab[k].window.size();
            std::memcpy(w, tab[k].left.data(), nodes * 4), w += nodes;
            std::memcpy(w, tab[k].right.data(), nodes * 4), w += nodes;
            std::memcpy(w, tab[k].value.data(), nodes * 4), w += nodes;
        }
    }
    JPEGAMD_HIPCHK(hipMemcpyAsync(d_tab.p, tab_host, tab_total * 4, hipMemcpyHostToDevice, nullptr));
    JPEGAMD_HIPCHK(hipMemsetAsync(d_work.p, 0, 64, nullptr));
    HdJob job{};
    uint32_t wg_total = 0;
    size_t tab_at = 0;
    for (int k = 0; k < 3; ++k) {
        HdPlane &pl = job.p[k];
        pl.wg_begin = wg_total; // (a plane that is not decoded here has no workgroups: the next one begins at the same index)
        if (!live[k])
            continue;
        const size_t nodes = tab[k].left.size();
        pl.window = d_tab.as<uint32_t>() + tab_at;
        const int32_t *d_left = reinterpret_cast<const int32_t *>(pl.window + tab[k].window.size());
        tab_at += tab_words[k];
        const uint32_t lead = (uint32_t)(off[k] & 3);
        pl.s = Stream{d_pay.as<uint32_t>() + (off[k] >> 2), (uint32_t)((lead + parts.len[k] + 3) / 4),
                      (uint32_t)(parts.len[k] * 8), 8 * lead, d_left, d_left + nodes, d_left + 2 * nodes};
        const size_t n_sub_pad = ((size_t)n_sub[k] + 3) & ~(size_t)3;
        pl.n_sub = n_sub[k];
        pl.sub_count = d_work.as<uint32_t>() + work_off[k];
        pl.sub_start = pl.sub_count + n_sub_pad;
        pl.ends_a = pl.sub_start + n_sub_pad;
        pl.ends_b = pl.ends_a + (((size_t)n_wg[k] + 3) & ~(size_t)3);
        pl.wg_count = pl.ends_b + (((size_t)n_wg[k] + 3) & ~(size_t)3);
        pl.changed = d_work.as<uint32_t>() + 4 * k;
        pl.total = reinterpret_cast<unsigned long long *>(pl.changed + 2);
        pl.out = cap[k] ? d_sym[k] : nullptr;
        pl.cap = d_sym[k] ? cap[k] : 0;
        wg_total += n_wg[k];
    }
    hipLaunchKernelGGL(hd_sync_kernel, dim3(wg_total), dim3(kThreads), 0, nullptr, job, 0);
    // across workgroup boundaries: twice, the second time nothing may change any more (the first of the two
    // raises `changed` too: the flags are cleared between them, in stream order)
    hipLaunchKernelGGL(hd_sync_kernel, dim3(wg_total), dim3(kThreads), 0, nullptr, job, 1);
    JPEGAMD_HIPCHK(hipMemsetAsync(d_work.p, 0, 64, nullptr));
    hipLaunchKernelGGL(hd_sync_kernel, dim3(wg_total), dim3(kThreads), 0, nullptr, job, 2);
    // a plane whose third pass still moved a boundary skips itself in the emit kernel (its `changed` flag)
    hipLaunchKernelGGL(hd_emit_kernel, dim3(wg_total), dim3(kThreads), 0, nullptr, job);
    JPEGAMD_HIPCHK(hipGetLastError());
    JPEGAMD_HIPCHK(hipMemcpyAsync(status_back, d_work.p, 48, hipMemcpyDeviceToHost, nullptr));
    JPEGAMD_HIPCHK(hipStreamSynchronize(nullptr)); // the one wait of the stage (per-call buffers may go after it)
    if (trace2)
        fprintf(stderr, "  entropy decode [ms]: tables %.3f alloc %.3f upload %.3f kernels+wait %.3f\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    for (int k = 0; k < 3; ++k) {
        if (!live[k])
            continue;
        if (status_back[4 * k]) {
            rc[k] = 2; // a chunk that does not fall into step within a workgroup's reach: the host decoder knows how
            continue;
        }
        unsigned long long total;
        std::memcpy(&total, status_back + 4 * k + 2, 8);
        n[k] = (size_t)total;
    }
    return 0;
}

// one plane (kept for callers that have a single payload)
int jpegamd_device_decode_plane(const uint8_t *payload, size_t len, const uint8_t *dict, int dsz, int32_t *d_sym,
                                size_t cap, size_t *n)
{
    JpegamdBinParts parts{};
    for (int k = 0; k < 3; ++k)
        parts.payload[k] = payload + (k ? len : 0), parts.len[k] = 0, parts.dict[k] = dict, parts.dsz[k] = 0;
    parts.len[0] = len, parts.dsz[0] = dsz;
    int32_t *const syms[3] = {d_sym, nullptr, nullptr};
    const size_t caps[3] = {cap, 0, 0};
    size_t ns[3];
    int rcs[3];
    if (jpegamd_device_decode_planes(parts, syms, caps, ns, rcs))
        return 1;
    *n = ns[0];
    return rcs[0];
}
