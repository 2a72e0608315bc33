// entropy_kernels.hip -- the reference's entropy stage for one plane with the two passes over
// the symbols on the device (gfx950):
//
//   frequency dictionary   encode_decode.cpp:375-383   counts per symbol, and for the
//                          dictionary ORDER the index of each symbol's first occurrence
//   huffman_encode         encode_decode.cpp:458-466 + stringToBitstream :1093-1127
//                          code lookup and MSB-first bit packing
//
// What fixes the bytes of the file -- the std::unordered_map iteration order and the
// std::priority_queue tie-breaks -- stays on the host (jpegamd_codes_from_order): the device
// hands it 2 x 64 Ki counters, the host hands back flat (code, length) tables.  The symbols
// (76 MB for an 8192^2 image) never leave the device; only the payload does.
//
//   sym_hist_kernel     counts + first occurrence, window of 2048 hot keys private in LDS
//   sym_bits_kernel     code-length sum of every workgroup's 2048 symbols
//   (scan_kernel)       exclusive scan of those sums (64-bit)
//   sym_pack_kernel     bit offset of every thread's 8 symbols, codes OR-ed into the payload
//
// All integer work, bit-exact by construction; tested through jpegamd_compress_image
// against the reference's files and the oracle's container writer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "host/dev_alloc.h"
#include "host/entropy_internal.h"
#include "host_common.h"

namespace {

constexpr int kKeys = 65536; // int16 coefficients and runs 1..64
constexpr int kWin = 2048;   // keys -1024 .. 1023 are counted in LDS
constexpr int kThreads = 256;
constexpr int kPerThread = 8;
constexpr int kPerWg = kThreads * kPerThread; // 2048 symbols

__global__ __launch_bounds__(kThreads) void sym_hist_kernel(const int32_t *__restrict__ sym, uint32_t n,
                                                            uint32_t *__restrict__ count,
                                                            uint32_t *__restrict__ first,
                                                            uint32_t *__restrict__ bad)
{
    __shared__ uint32_t l_count[kWin], l_first[kWin];
    for (int k = threadIdx.x; k < kWin; k += kThreads)
        l_count[k] = 0, l_first[k] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int s = sym[i];
        if (s < -32768 || s > 32767) {
            atomicOr(bad, 1u);
            continue;
        }
        const int w = s + kWin / 2;
        if ((unsigned)w < (unsigned)kWin) {
            atomicAdd(&l_count[w], 1u);
            atomicMin(&l_first[w], i);
        } else {
            atomicAdd(&count[s + 32768], 1u);
            atomicMin(&first[s + 32768], i);
            atomicOr(bad, 2u); // a key outside the window: the host then needs the whole tables
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kWin; k += kThreads)
        if (l_count[k]) {
            atomicAdd(&count[k - kWin / 2 + 32768], l_count[k]);
            atomicMin(&first[k - kWin / 2 + 32768], l_first[k]);
        }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, 64);
    return v;
}

// code bits of one workgroup's symbols
__global__ __launch_bounds__(kThreads) void sym_bits_kernel(const int32_t *__restrict__ sym, uint32_t n,
                                                            const uint8_t *__restrict__ clen,
                                                            uint32_t *__restrict__ wg_bits)
{
    __shared__ uint32_t part[kThreads / 64];
    const uint32_t i0 = blockIdx.x * kPerWg + threadIdx.x * kPerThread;
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k)
        if (i0 + k < n)
            bits += clen[sym[i0 + k] + 32768];
    bits = wave_sum(bits);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0)
        wg_bits[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the workgroups' bit counts: one workgroup walks them 1024 at a time
__global__ __launch_bounds__(1024) void wg_offsets_kernel(const uint32_t *__restrict__ wg_bits, uint32_t n_wg,
                                                          unsigned long long *__restrict__ wg_off)
{
    __shared__ unsigned long long wave_tot[16];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_wg; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_wg ? wg_bits[i] : 0;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(inc, d, 64);
            if (lane >= d)
                inc += t;
        }
        if (lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        unsigned long long before = carry_s, all = 0;
        for (int w = 0; w < 16; ++w) {
            before += w < wave ? wave_tot[w] : 0;
            all += wave_tot[w];
        }
        if (i < n_wg)
            wg_off[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0)
            carry_s += all;
        __syncthreads();
    }
}

// Every thread packs its 8 symbols: bit offset = workgroup offset + exclusive scan over the
// threads, then the codes go MSB-first into 32-bit big-endian words of the (zeroed) payload.
// Words can be shared with the neighbouring threads, hence atomicOr.
__global__ __launch_bounds__(kThreads) void sym_pack_kernel(const int32_t *__restrict__ sym, uint32_t n,
                                                            const uint32_t *__restrict__ cbits,
                                                            const uint8_t *__restrict__ clen,
                                                            const unsigned long long *__restrict__ wg_off,
                                                            uint32_t *__restrict__ payload, unsigned long long bit_base)
{
    __shared__ uint32_t wave_tot[kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * kPerWg + threadIdx.x * kPerThread;
    uint32_t code[kPerThread], len[kPerThread], mine = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        code[k] = 0, len[k] = 0;
        if (i0 + k < n) {
            const int key = sym[i0 + k] + 32768;
            code[k] = cbits[key], len[k] = clen[key];
        }
        mine += len[k];
    }
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d)
            inc += t;
    }
    if (lane == 63)
        wave_tot[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < kThreads / 64; ++w)
        before += w < wave ? wave_tot[w] : 0;
    unsigned long long pos = bit_base + wg_off[blockIdx.x] + before + inc - mine; // first bit of this thread (bit_base: where the plane begins in `payload`)
    // BitWriter (host/bin_container.cpp) with the partial first word as leading zero bits
    unsigned long long acc = 0;
    int have = (int)(pos & 31);
    uint32_t *out = payload + (pos >> 5);
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        acc = (acc << len[k]) | code[k];
        have += (int)len[k];
        if (have >= 32) {
            atomicOr(out++, __builtin_bswap32((uint32_t)(acc >> (have - 32))));
            have -= 32;
        }
    }
    if (have > 0)
        atomicOr(out, __builtin_bswap32((uint32_t)(acc << (32 - have))));
}

typedef JpegamdDevMem DevMem; // per call, or a slot of the running context call (host/dev_alloc.h)

} // namespace

int jpegamd_device_encode_plane(const int32_t *d_sym, size_t n, JpegamdPlaneCode &pl)
{
    pl.dict.clear();
    pl.payload.clear();
    if (n == 0)
        return 0;
    if (n >= 0xFFFFFFFFull)
        return 2;
    const uint32_t n32 = (uint32_t)n;
    // counters: count[64 Ki], first[64 Ki], bad
    DevMem ctr;
    if (ctr.alloc((2 * kKeys + 1) * sizeof(uint32_t)))
        return jpegamd_fail("entropy stage: out of device memory");
    uint32_t *d_count = ctr.as<uint32_t>(), *d_first = d_count + kKeys, *d_bad = d_first + kKeys;
    JPEGAMD_HIPCHK(hipMemsetAsync(d_count, 0, kKeys * sizeof(uint32_t), nullptr));
    JPEGAMD_HIPCHK(hipMemsetAsync(d_first, 0xFF, kKeys * sizeof(uint32_t), nullptr));
    JPEGAMD_HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), nullptr));
    const uint32_t hist_grid = std::min<uint32_t>((n32 + kThreads - 1) / kThreads, 2048);
    hipLaunchKernelGGL(sym_hist_kernel, dim3(hist_grid), dim3(kThreads), 0, nullptr, d_sym, n32, d_count, d_first, d_bad);
    JPEGAMD_HIPCHK(hipGetLastError());
    // Counters back to the host: the flags first; unless a key lay outside the LDS window (which
    // coefficients and runs of an image never do) only the window's 2 x 8 KB instead of 2 x 256 KB.
    std::vector<uint32_t> h(2 * kKeys + 1);
    JPEGAMD_HIPCHK(hipMemcpy(&h[2 * kKeys], d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (h[2 * kKeys] & 1u)
        return 2; // a symbol outside int16: the host path has the general tables
    const bool whole = (h[2 * kKeys] & 2u) != 0;
    const int w0 = whole ? 0 : 32768 - kWin / 2, wn = whole ? kKeys : kWin; // keys [w0, w0 + wn) may be in use
    JPEGAMD_HIPCHK(hipMemcpy(h.data() + w0, d_count + w0, (size_t)wn * sizeof(uint32_t), hipMemcpyDeviceToHost));
    JPEGAMD_HIPCHK(hipMemcpy(h.data() + kKeys + w0, d_first + w0, (size_t)wn * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // distinct symbols in first-occurrence order: the insertion sequence of the reference's map
    std::vector<int> order;
    for (int k = w0; k < w0 + wn; ++k)
        if (h[k])
            order.push_back(k - 32768);
    const uint32_t *first = h.data() + kKeys;
    std::sort(order.begin(), order.end(), [first](int a, int b) { return first[a + 32768] < first[b + 32768]; });
    std::vector<uint32_t> cbits;
    std::vector<uint8_t> clen;
    unsigned long long total_bits = 0;
    const int rc = jpegamd_codes_from_order(order, h.data(), pl, cbits, clen, total_bits);
    if (rc)
        return rc;
    const size_t bytes = (size_t)((total_bits + 7) / 8);
    if (bytes > 0x7FFFFFFFu)
        return jpegamd_fail("bin_encode: payload exceeds the container's int32 length field");
    pl.payload.resize(bytes);
    if (bytes == 0)
        return 0; // a single distinct symbol has the empty code (encode_decode.cpp:444-456)
    const uint32_t n_wg = (n32 + kPerWg - 1) / kPerWg;
    const size_t words = (bytes + 3) / 4 + 2;
    DevMem tab, sums, offs, out;
    if (tab.alloc(kKeys * 5) || sums.alloc(n_wg * sizeof(uint32_t)) || offs.alloc(n_wg * sizeof(unsigned long long)) ||
        out.alloc(words * 4))
        return jpegamd_fail("entropy stage: out of device memory");
    uint32_t *d_cbits = tab.as<uint32_t>();
    uint8_t *d_clen = reinterpret_cast<uint8_t *>(d_cbits + kKeys);
    // (only the entries of keys in use are ever read)
    JPEGAMD_HIPCHK(hipMemcpy(d_cbits + w0, cbits.data() + w0, (size_t)wn * 4, hipMemcpyHostToDevice));
    JPEGAMD_HIPCHK(hipMemcpy(d_clen + w0, clen.data() + w0, (size_t)wn, hipMemcpyHostToDevice));
    JPEGAMD_HIPCHK(hipMemsetAsync(out.p, 0, words * 4, nullptr));
    hipLaunchKernelGGL(sym_bits_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, d_sym, n32, d_clen, sums.as<uint32_t>());
    hipLaunchKernelGGL(wg_offsets_kernel, dim3(1), dim3(1024), 0, nullptr, sums.as<uint32_t>(), n_wg,
                       offs.as<unsigned long long>());
    hipLaunchKernelGGL(sym_pack_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, d_sym, n32, d_cbits, d_clen,
                       offs.as<unsigned long long>(), out.as<uint32_t>(), 0ull);
    JPEGAMD_HIPCHK(hipGetLastError());
    if (bytes >= (1u << 20)) { // through the pinned staging buffers: pageable hipMemcpy moves a fifth of that
        JPEGAMD_HIPCHK(hipDeviceSynchronize());
        if (jpegamd_staged_copy_linear(out.p, pl.payload.data(), bytes, false))
            return 1;
    } else {
        JPEGAMD_HIPCHK(hipMemcpy(pl.payload.data(), out.p, bytes, hipMemcpyDeviceToHost));
    }
    return 0;
}


// The entropy stage of the three planes of an image in one go (round 3).  Round 2 ran plane after plane:
// three memsets, the histogram, three synchronous read-backs, the host's code construction, two synchronous
// table uploads, three kernels and a synchronous payload copy EACH (1.2 ms for 0.26 ms of kernels at 8192^2).
// Here: the three histograms are queued together and their flags and windows come back in ONE wait (through
// pinned memory); the host builds the three codes; tables go up and the nine kernels are queued without a
// wait; the packed payloads land in ONE device buffer laid out like the file's payload section (planes back
// to back at byte granularity: encode_decode.cpp:1178-1186), which the caller copies straight into the
// file image (jpegamd_bin_assemble_from_device) -- no per-plane host vectors, no second copy.
// Returns 0 (code[k].dict filled, bytes[k] = payload bytes of plane k, d_out = the payload section, kernels
// still running on the null stream), 2 when some plane needs the host path (nothing is left behind), 1 on error.
int jpegamd_device_encode_image(const int32_t *const d_sym[3], const size_t n[3], JpegamdPlaneCode code[3], size_t bytes[3],
                                JpegamdDevMem &d_out)
{
    for (int k = 0; k < 3; ++k) {
        code[k].dict.clear(), code[k].payload.clear();
        bytes[k] = 0;
        if (n[k] >= 0xFFFFFFFFull)
            return 2;
    }
    // counters of the three planes: count[3][64 Ki], first[3][64 Ki], bad[4]
    DevMem ctr;
    if (ctr.alloc((6 * (size_t)kKeys + 4) * sizeof(uint32_t)))
        return jpegamd_fail("entropy stage: out of device memory");
    uint32_t *d_count = ctr.as<uint32_t>(), *d_first = d_count + 3 * kKeys, *d_bad = d_first + 3 * kKeys;
    JPEGAMD_HIPCHK(hipMemsetAsync(d_count, 0, 3 * (size_t)kKeys * sizeof(uint32_t), nullptr));
    JPEGAMD_HIPCHK(hipMemsetAsync(d_first, 0xFF, 3 * (size_t)kKeys * sizeof(uint32_t), nullptr));
    JPEGAMD_HIPCHK(hipMemsetAsync(d_bad, 0, 4 * sizeof(uint32_t), nullptr));
    for (int k = 0; k < 3; ++k) {
        if (!n[k])
            continue;
        const uint32_t n32 = (uint32_t)n[k];
        const uint32_t hist_grid = std::min<uint32_t>((n32 + kThreads - 1) / kThreads, 2048);
        hipLaunchKernelGGL(sym_hist_kernel, dim3(hist_grid), dim3(kThreads), 0, nullptr, d_sym[k], n32, d_count + k * kKeys,
                           d_first + k * kKeys, d_bad + k);
    }
    JPEGAMD_HIPCHK(hipGetLastError());
    // back through pinned memory, one wait: the flags and, optimistically, only the windows' 2 x 8 KB per plane
    // (a key outside the window -- which coefficients and runs of an image never are -- costs that plane two more copies)
    const int w0 = 32768 - kWin / 2;
    const size_t pin_words = 4 + 6 * (size_t)kWin;
    uint32_t *pin = static_cast<uint32_t *>(jpegamd_pinned_scratch((pin_words + 3 * ((size_t)kWin * 5 / 4 + 4)) * 4));
    if (!pin)
        return jpegamd_fail("entropy stage: no pinned host memory");
    JPEGAMD_HIPCHK(hipMemcpyAsync(pin, d_bad, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, nullptr));
    for (int k = 0; k < 3; ++k) {
        JPEGAMD_HIPCHK(hipMemcpyAsync(pin + 4 + (2 * k) * kWin, d_count + k * kKeys + w0, kWin * sizeof(uint32_t),
                                      hipMemcpyDeviceToHost, nullptr));
        JPEGAMD_HIPCHK(hipMemcpyAsync(pin + 4 + (2 * k + 1) * kWin, d_first + k * kKeys + w0, kWin * sizeof(uint32_t),
                                      hipMemcpyDeviceToHost, nullptr));
    }
    JPEGAMD_HIPCHK(hipStreamSynchronize(nullptr)); // wait 1 of 2
    std::vector<uint32_t> h(2 * (size_t)kKeys);
    std::vector<uint32_t> cbits[3];
    std::vector<uint8_t> clen[3];
    int win0[3], winn[3];
    size_t total_bytes = 0;
    for (int k = 0; k < 3; ++k) {
        win0[k] = w0, winn[k] = kWin;
        if (!n[k])
            continue;
        if (pin[k] & 1u)
            return 2; // a symbol outside int16: the host path has the general tables
        std::fill(h.begin(), h.end(), 0u);
        if (pin[k] & 2u) { // keys outside the window: the whole tables
            win0[k] = 0, winn[k] = kKeys;
            JPEGAMD_HIPCHK(hipMemcpy(h.data(), d_count + k * kKeys, kKeys * sizeof(uint32_t), hipMemcpyDeviceToHost));
            JPEGAMD_HIPCHK(hipMemcpy(h.data() + kKeys, d_first + k * kKeys, kKeys * sizeof(uint32_t), hipMemcpyDeviceToHost));
        } else {
            std::memcpy(h.data() + w0, pin + 4 + (2 * k) * kWin, kWin * sizeof(uint32_t));
            std::memcpy(h.data() + kKeys + w0, pin + 4 + (2 * k + 1) * kWin, kWin * sizeof(uint32_t));
        }
        // distinct symbols in first-occurrence order: the insertion sequence of the reference's map
        std::vector<int> order;
        for (int key = win0[k]; key < win0[k] + winn[k]; ++key)
            if (h[key])
                order.push_back(key - 32768);
        const uint32_t *first = h.data() + kKeys;
        std::sort(order.begin(), order.end(), [first](int a, int b) { return first[a + 32768] < first[b + 32768]; });
        unsigned long long total_bits = 0;
        const int rc = jpegamd_codes_from_order(order, h.data(), code[k], cbits[k], clen[k], total_bits);
        if (rc)
            return rc;
        bytes[k] = (size_t)((total_bits + 7) / 8);
        if (bytes[k] > 0x7FFFFFFFu)
            return jpegamd_fail("bin_encode: payload exceeds the container's int32 length field");
        total_bytes += bytes[k];
    }
    const size_t words = (total_bytes + 3) / 4 + 2;
    if (d_out.alloc(words * 4))
        return jpegamd_fail("entropy stage: out of device memory");
    if (total_bytes == 0)
        return 0; // (single distinct symbols have the empty code, encode_decode.cpp:444-456)
    JPEGAMD_HIPCHK(hipMemsetAsync(d_out.p, 0, words * 4, nullptr));
    // tables up from pinned memory (only the entries of keys in use are ever read), kernels queued behind them
    DevMem tab[3], sums[3], offs[3];
    uint8_t *pin_tab = reinterpret_cast<uint8_t *>(pin + pin_words);
    size_t bit_base = 0;
    for (int k = 0; k < 3; ++k) {
        if (!bytes[k])
            continue; // no symbols, or one distinct symbol: no bits
        const uint32_t n32 = (uint32_t)n[k];
        const uint32_t n_wg = (n32 + kPerWg - 1) / kPerWg;
        if (tab[k].alloc(kKeys * 5) || sums[k].alloc(n_wg * sizeof(uint32_t)) || offs[k].alloc(n_wg * sizeof(unsigned long long)))
            return jpegamd_fail("entropy stage: out of device memory");
        uint32_t *d_cbits = tab[k].as<uint32_t>();
        uint8_t *d_clen = reinterpret_cast<uint8_t *>(d_cbits + kKeys);
        if (winn[k] == kWin) {
            uint8_t *pt = pin_tab + (size_t)k * (kWin * 5 + 16);
            std::memcpy(pt, cbits[k].data() + w0, kWin * 4);
            std::memcpy(pt + kWin * 4, clen[k].data() + w0, kWin);
            JPEGAMD_HIPCHK(hipMemcpyAsync(d_cbits + w0, pt, kWin * 4, hipMemcpyHostToDevice, nullptr));
            JPEGAMD_HIPCHK(hipMemcpyAsync(d_clen + w0, pt + kWin * 4, kWin, hipMemcpyHostToDevice, nullptr));
        } else {
            JPEGAMD_HIPCHK(hipMemcpy(d_cbits, cbits[k].data(), (size_t)kKeys * 4, hipMemcpyHostToDevice));
            JPEGAMD_HIPCHK(hipMemcpy(d_clen, clen[k].data(), (size_t)kKeys, hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(sym_bits_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, d_sym[k], n32, d_clen, sums[k].as<uint32_t>());
        hipLaunchKernelGGL(wg_offsets_kernel, dim3(1), dim3(1024), 0, nullptr, sums[k].as<uint32_t>(), n_wg,
                           offs[k].as<unsigned long long>());
        hipLaunchKernelGGL(sym_pack_kernel, dim3(n_wg), dim3(kThreads), 0, nullptr, d_sym[k], n32, d_cbits, d_clen,
                           offs[k].as<unsigned long long>(), d_out.as<uint32_t>(), (unsigned long long)bit_base * 8);
        bit_base += bytes[k];
    }
    JPEGAMD_HIPCHK(hipGetLastError());
    // tab / sums / offs: arena slots inside a context call (they outlive this function); per-call buffers
    // are freed on return, so wait for the kernels that use them
    if (!jpegamd_thread_arena())
        JPEGAMD_HIPCHK(hipDeviceSynchronize());
    return 0;
}
