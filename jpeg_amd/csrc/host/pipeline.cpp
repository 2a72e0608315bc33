// pipeline.cpp -- whole-image convenience entry points on HOST buffers: what the
// reference's main.cpp does between cv::imread and the .bin file, and back
// (main.cpp:96-168 with platform "GPU"): fused encode -> device run-length stage ->
// host entropy coder -> .bin bytes, and the inverse.  Device memory is allocated
// and freed per call, like the reference's kernel.cu:97-104 / 160-164.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../../include/jpeg_amd.h"
#include "../host_common.h"
#include "dev_alloc.h"
#include "entropy_internal.h"

namespace {


typedef JpegamdDevMem DevMem; // per call, or a slot of the running context call (dev_alloc.h)

// malloc'ed, deliberately uninitialised host array (a std::vector would zero 76 MB first)
struct HostSymbols {
    int32_t *p = nullptr;
    size_t n = 0;
    ~HostSymbols() { std::free(p); }
    int32_t *data() { return p; }
    size_t size() const { return n; }
    bool resize(size_t count)
    {
        std::free(p);
        p = static_cast<int32_t *>(std::malloc((count ? count : 1) * sizeof(int32_t)));
        n = p ? count : 0;
        return p != nullptr;
    }
};

// JPEGAMD_TRACE=1: the host-pointer calls report where their time went, on stderr
class StageTrace {
  public:
    explicit StageTrace(const char *what) : on_(std::getenv("JPEGAMD_TRACE") != nullptr), what_(what), last_(now()) {}
    void mark(const char *stage) // time since the previous mark (device work is synchronised first)
    {
        if (!on_)
            return;
        (void)hipDeviceSynchronize();
        const double t = now();
        n_ += std::snprintf(line_ + n_, sizeof line_ - (size_t)n_, " %s %.2f", stage, t - last_);
        n_ = n_ < (int)sizeof line_ ? n_ : (int)sizeof line_ - 1;
        last_ = now();
    }
    ~StageTrace()
    {
        if (on_)
            std::fprintf(stderr, "jpeg_amd %s [ms]:%s\n", what_, line_);
    }

  private:
    static double now()
    {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    bool on_;
    const char *what_;
    double last_;
    char line_[256] = "";
    int n_ = 0;
};

// JPEGAMD_ENTROPY=host keeps both passes of the entropy stage on the host (the first
// implementation, still the path for inputs the device tables do not cover)
bool entropy_on_host()
{
    const char *e = std::getenv("JPEGAMD_ENTROPY");
    return e && std::strcmp(e, "host") == 0;
}

// fused encode + run-length stage (jpegamd_encode_image_rle_dev) and entropy stage
// The coded planes of an image: either whole on the host (code[k].payload), or -- the usual case -- the
// dictionaries on the host and the payload section still on the device, to be copied straight into the file
struct CodedImage {
    JpegamdPlaneCode code[3];
    bool payload_on_device = false;
    size_t bytes[3] = {0, 0, 0};
    DevMem d_payloads;
    int assemble(int y_rows, int y_cols, int c_rows, int c_cols, uint8_t *out, size_t cap, size_t *out_len) const
    {
        if (payload_on_device)
            return jpegamd_bin_assemble_from_device(code, bytes, d_payloads.p, y_rows, y_cols, c_rows, c_cols, out, cap, out_len);
        return jpegamd_bin_assemble(code, y_rows, y_cols, c_rows, c_cols, out, cap, out_len);
    }
    size_t file_bytes() const
    {
        size_t need = 9 * 4 + 3 * 4;
        for (int k = 0; k < 3; ++k)
            need += (payload_on_device ? bytes[k] : code[k].payload.size()) + 8 * code[k].dict.size();
        return need;
    }
};

// entropy stage of three symbol streams on the device: all three planes in one go, or plane by plane with
// the host coder for what the device tables do not cover
int code_streams(const int32_t *const d_sym[3], const size_t n[3], CodedImage &ci, const char *what)
{
    if (!entropy_on_host()) {
        const int rc = jpegamd_device_encode_image(d_sym, n, ci.code, ci.bytes, ci.d_payloads);
        if (rc == 0) {
            ci.payload_on_device = true;
            return 0;
        }
        if (rc != 2)
            return rc;
    }
    ci.payload_on_device = false;
    for (int k = 0; k < 3; ++k) {
        if (!entropy_on_host()) {
            const int rc = jpegamd_device_encode_plane(d_sym[k], n[k], ci.code[k]);
            if (rc == 0)
                continue;
            if (rc != 2)
                return rc;
        }
        HostSymbols sym; // the host path: the pairs are the memory of the reference's std::vector<int>
        if (!sym.resize(n[k]))
            return jpegamd_fail("%s: out of host memory", what);
        if (n[k] && jpegamd_staged_copy_linear(const_cast<int32_t *>(d_sym[k]), sym.data(), n[k] * 4, false))
            return 1;
        if (jpegamd_host_encode_plane(sym.data(), sym.size(), ci.code[k]))
            return 1;
    }
    return 0;
}

int image_code(const uint8_t *d_bgr, int rows, int cols, const int32_t q_luma[64], const int32_t q_chroma[64],
               int16_t *const d_coef[3], const size_t nb[3], CodedImage &ci, StageTrace &trace)
{
    DevMem work, offs[3], pairs[3];
    const size_t wbytes = jpegamd_rle_workspace_bytes(nb[0] + nb[1] + nb[2]);
    if (work.alloc(wbytes))
        return jpegamd_fail("compress_image: out of device memory");
    size_t cap[3];
    for (int k = 0; k < 3; ++k) {
        cap[k] = nb[k] * 16 < nb[k] * 64 ? nb[k] * 16 : nb[k] * 64; // a typical plane needs far fewer than 64 pairs per block
        if (offs[k].alloc((nb[k] + 1) * sizeof(uint32_t)))
            return jpegamd_fail("compress_image: out of device memory");
    }
    uint32_t total[3] = {0, 0, 0};
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (int k = 0; k < 3; ++k)
            if (pairs[k].alloc(cap[k] * 8)) // (a second attempt lets go of the first buffer)
                return jpegamd_fail("compress_image: out of device memory");
        int32_t *const pp[3] = {pairs[0].as<int32_t>(), pairs[1].as<int32_t>(), pairs[2].as<int32_t>()};
        uint32_t *const po[3] = {offs[0].as<uint32_t>(), offs[1].as<uint32_t>(), offs[2].as<uint32_t>()};
        if (jpegamd_encode_image_rle_dev(d_bgr, rows, cols, (size_t)cols * 3, q_luma, q_chroma, d_coef, pp, cap, po, work.p,
                                         wbytes, nullptr))
            return 1;
        bool fits = true;
        for (int k = 0; k < 3; ++k) {
            JPEGAMD_HIPCHK(hipMemcpy(&total[k], po[k] + nb[k], 4, hipMemcpyDeviceToHost));
            if (total[k] > cap[k])
                fits = false, cap[k] = total[k];
        }
        if (fits)
            break;
        if (attempt == 1)
            return jpegamd_fail("compress_image: run-length pass did not converge");
    }
    trace.mark("transform+run-length");
    const int32_t *const d_sym[3] = {pairs[0].as<int32_t>(), pairs[1].as<int32_t>(), pairs[2].as<int32_t>()};
    const size_t n_sym[3] = {(size_t)total[0] * 2, (size_t)total[1] * 2, (size_t)total[2] * 2};
    return code_streams(d_sym, n_sym, ci, "compress_image");
}

} // namespace

int jpegamd_bin_assemble_from_device(const JpegamdPlaneCode (&pl)[3], const size_t bytes[3], const void *d_payloads, int y_rows,
                                     int y_cols, int c_rows, int c_cols, uint8_t *out, size_t cap, size_t *out_len)
{
    size_t need = 9 * 4 + 3 * 4, payload = 0;
    for (int k = 0; k < 3; ++k) {
        if (bytes[k] > 0x7FFFFFFFu)
            return jpegamd_fail("bin_encode: payload exceeds the container's int32 length field");
        payload += bytes[k];
        need += bytes[k] + 8 * pl[k].dict.size();
    }
    *out_len = need;
    if (!out || cap < need)
        return jpegamd_fail("bin_encode: output buffer too small (the size needed is in *out_len)");
    auto put = [](uint8_t *&w, int v) { std::memcpy(w, &v, 4), w += 4; }; // (little endian like the reference's fwrite of ints)
    uint8_t *w = out;
    put(w, y_rows), put(w, y_cols), put(w, c_rows), put(w, c_cols), put(w, c_rows), put(w, c_cols); // saveEncodedData, :1161-1217
    for (int k = 0; k < 3; ++k)
        put(w, (int)bytes[k]);
    if (payload) {
        JPEGAMD_HIPCHK(hipDeviceSynchronize()); // the pack kernels ran on the null stream, the copier uses its own
        if (payload >= (1u << 20)) {
            if (jpegamd_staged_copy_linear(const_cast<void *>(d_payloads), w, payload, false))
                return 1;
        } else {
            JPEGAMD_HIPCHK(hipMemcpy(w, d_payloads, payload, hipMemcpyDeviceToHost));
        }
        w += payload;
    }
    for (int k = 0; k < 3; ++k)
        put(w, (int)pl[k].dict.size());
    for (int k = 0; k < 3; ++k)
        for (const auto &kv : pl[k].dict)
            put(w, kv.first), put(w, kv.second);
    return 0;
}

JpegamdArena *&jpegamd_thread_arena()
{
    static thread_local JpegamdArena *arena = nullptr;
    return arena;
}

// SURVEY 8b's context: a device and the device buffers of the host-pointer calls, kept from
// call to call.  One call at a time per context (a mutex serialises them).
struct jpegamd_context {
    int device = 0;
    std::mutex mu;
    JpegamdArena arena;
};

namespace {
// runs `body` on the context's device with the context's buffers
template <class Body> int in_context(jpegamd_context *ctx, const char *what, Body body)
{
    if (!ctx)
        return jpegamd_fail("%s: null context", what);
    std::lock_guard<std::mutex> lock(ctx->mu);
    int before = 0;
    JPEGAMD_HIPCHK(hipGetDevice(&before));
    JPEGAMD_HIPCHK(hipSetDevice(ctx->device));
    ctx->arena.rewind();
    jpegamd_thread_arena() = &ctx->arena;
    const int rc = jpegamd_guarded(what, body);
    jpegamd_thread_arena() = nullptr;
    (void)hipSetDevice(before);
    return rc;
}
} // namespace

extern "C" {

void jpegamd_free(void *p) { std::free(p); }

int jpegamd_quality_tables(int quality, int32_t q_luma[64], int32_t q_chroma[64])
{
    // main.cpp:3-20 (base tables, ITU-T T.81 Annex K) and :51-80 (scaling)
    static const uint8_t base_luma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                          14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                          18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                          49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const uint8_t base_chroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                            24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                            99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                            99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
    if (!q_luma || !q_chroma)
        return jpegamd_fail("quality_tables: null pointer");
    if (quality < 1 || quality > 100)
        return jpegamd_fail("Quality must be between 1 and 100");
    const int sf = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; ++i) {
        q_luma[i] = std::max(1, std::min(255, (base_luma[i] * sf + 50) / 100));
        q_chroma[i] = std::max(1, std::min(255, (base_chroma[i] * sf + 50) / 100));
    }
    return 0;
}

int jpegamd_analysis_tables(int quality, int32_t q_luma[64], int32_t q_chroma[64])
{
    // analysis.cpp:43-90, literally: no scaling, a fixed pair for quality >= 90 (whose luma table has
    // 12 at [5][5] where the scaled table would have 21, and whose chroma table is not the Annex-K one)
    // and the base pair otherwise
    static const uint8_t hi_luma[64] = {3,  2,  2,  3,  5,  8,  10, 12, 2,  2,  3,  4,  5,  12, 12, 11,
                                        3,  3,  3,  5,  8,  11, 14, 11, 3,  3,  4,  6,  10, 17, 16, 12,
                                        4,  4,  7,  11, 14, 22, 21, 15, 5,  7,  11, 13, 16, 12, 23, 18,
                                        10, 13, 16, 17, 21, 24, 24, 21, 14, 18, 19, 20, 22, 20, 20, 20};
    static const uint8_t hi_chroma[64] = {11, 12, 14, 19, 26, 58, 60, 55, 12, 18, 21, 28, 32, 57, 59, 56,
                                          14, 21, 25, 30, 59, 59, 59, 59, 19, 28, 30, 59, 59, 59, 59, 59,
                                          26, 32, 59, 59, 59, 59, 59, 59, 58, 57, 59, 59, 59, 59, 59, 59,
                                          60, 59, 59, 59, 59, 59, 59, 59, 55, 56, 59, 59, 59, 59, 59, 59};
    if (!q_luma || !q_chroma)
        return jpegamd_fail("analysis_tables: null pointer");
    if (quality < 1 || quality > 100)
        return jpegamd_fail("Quality must be between 1 and 100");
    if (quality >= 90) {
        for (int i = 0; i < 64; ++i)
            q_luma[i] = hi_luma[i], q_chroma[i] = hi_chroma[i];
        return 0;
    }
    return jpegamd_quality_tables(50, q_luma, q_chroma); // scale factor 100: the base tables themselves
}

static int bin_encode_dev_impl(const int32_t *d_sym_y, size_t n_y, const int32_t *d_sym_cb, size_t n_cb,
                           const int32_t *d_sym_cr, size_t n_cr, int y_rows, int y_cols, int c_rows, int c_cols,
                           uint8_t *out, size_t cap, size_t *out_len)
{
    if (!out_len || (!d_sym_y && n_y) || (!d_sym_cb && n_cb) || (!d_sym_cr && n_cr))
        return jpegamd_fail("bin_encode_dev: null pointer");
    const int32_t *const d_sym[3] = {d_sym_y, d_sym_cb, d_sym_cr};
    const size_t n[3] = {n_y, n_cb, n_cr};
    JPEGAMD_HIPCHK(hipDeviceSynchronize()); // the symbols may come from kernels on other streams
    CodedImage ci;
    if (code_streams(d_sym, n, ci, "bin_encode_dev"))
        return 1;
    return ci.assemble(y_rows, y_cols, c_rows, c_cols, out, cap, out_len);
}

static int compress_image_impl(const uint8_t *bgr, int rows, int cols, size_t stride, const int32_t q_luma[64],
                           const int32_t q_chroma[64], uint8_t **bin, size_t *bin_len)
{
    if (!bgr || !bin || !bin_len)
        return jpegamd_fail("compress_image: null pointer");
    if (rows <= 0 || cols <= 0 || (rows & 15) || (cols & 15) || stride < (size_t)cols * 3)
        return jpegamd_fail("compress_image: rows=%d cols=%d must be positive multiples of 16 and stride >= 3*cols "
                            "(the reference resizes first, encode_decode.cpp:29-37)", rows, cols);
    const size_t nby = (size_t)(rows / 8) * (cols / 8), nbc = (size_t)(rows / 16) * (cols / 16);
    StageTrace trace("compress_image");
    DevMem d_bgr, d_y, d_cb, d_cr;
    if (d_bgr.alloc((size_t)rows * cols * 3) || d_y.alloc(nby * 128) || d_cb.alloc(nbc * 128) || d_cr.alloc(nbc * 128))
        return jpegamd_fail("compress_image: out of device memory");
    trace.mark("alloc");
    if (jpegamd_staged_copy(d_bgr.p, const_cast<uint8_t *>(bgr), stride, (size_t)cols * 3, rows, true))
        return 1;
    trace.mark("upload");
    CodedImage ci;
    int16_t *const d_coef[3] = {d_y.as<int16_t>(), d_cb.as<int16_t>(), d_cr.as<int16_t>()};
    const size_t nbs[3] = {nby, nbc, nbc};
    if (image_code(d_bgr.as<uint8_t>(), rows, cols, q_luma, q_chroma, d_coef, nbs, ci, trace))
        return 1;
    trace.mark("entropy");
    size_t need = ci.file_bytes();
    uint8_t *buf = static_cast<uint8_t *>(std::malloc(need ? need : 1));
    if (!buf)
        return jpegamd_fail("compress_image: out of host memory");
    if (ci.assemble(rows, cols, rows / 2, cols / 2, buf, need, &need)) {
        std::free(buf);
        return 1;
    }
    *bin = buf;
    *bin_len = need;
    trace.mark("assemble");
    return 0;
}

static int decompress_image_impl(const uint8_t *bin, size_t len, const int32_t q_luma[64], const int32_t q_chroma[64],
                             uint8_t *bgr, size_t cap_bytes, size_t stride, int *rows, int *cols)
{
    if (!bin || !rows || !cols)
        return jpegamd_fail("decompress_image: null pointer");
    int32_t dims[6];
    if (len < 36)
        return jpegamd_fail("decompress_image: shorter than the 9-int header");
    std::memcpy(dims, bin, 24);
    const int r = dims[0], c = dims[1];
    *rows = r, *cols = c;
    if (r <= 0 || c <= 0 || (r & 15) || (c & 15) || dims[2] != r / 2 || dims[3] != c / 2 || dims[4] != r / 2 ||
        dims[5] != c / 2)
        return jpegamd_fail("decompress_image: header %dx%d / %dx%d / %dx%d is not a 4:2:0 image with sides that "
                            "are multiples of 16", dims[0], dims[1], dims[2], dims[3], dims[4], dims[5]);
    if (!bgr)
        return 0; // size query
    if (stride < (size_t)c * 3 || cap_bytes < (size_t)(r - 1) * stride + (size_t)c * 3)
        return jpegamd_fail("decompress_image: output buffer too small for %dx%d", r, c);
    const size_t nb[3] = {(size_t)(r / 8) * (c / 8), (size_t)(r / 16) * (c / 16), (size_t)(r / 16) * (c / 16)};
    StageTrace trace("decompress_image");
    JpegamdBinParts parts;
    if (jpegamd_bin_parse(bin, len, parts))
        return 1;
    // every buffer of the call up front: the planes then go through without a wait between them.
    // A plane of nb blocks has at most 64 pairs per block (plus what up to 7 pad bits decode to),
    // and no more symbols than payload bits.
    DevMem d_coef[3], d_sym[3], work[3], d_out;
    size_t cap[3], wbytes[3];
    for (int k = 0; k < 3; ++k) {
        cap[k] = nb[k] * 128 + 16;
        if (parts.len[k] * 8 < cap[k])
            cap[k] = parts.len[k] * 8;
        wbytes[k] = jpegamd_rle_workspace_bytes(nb[k]);
        if (d_coef[k].alloc(nb[k] * 128) || d_sym[k].alloc(cap[k] * 4) || work[k].alloc(wbytes[k]))
            return jpegamd_fail("decompress_image: out of device memory");
    }
    if (d_out.alloc((size_t)r * c * 3))
        return jpegamd_fail("decompress_image: out of device memory");
    // huffman_decode (encode_decode.cpp:468-482) on the device; a plane the device tables do not
    // cover (or JPEGAMD_ENTROPY=host) is decoded on the host and its symbols uploaded
    size_t n[3] = {0, 0, 0};
    int prc[3] = {2, 2, 2};
    int32_t *const d_sym_p[3] = {d_sym[0].as<int32_t>(), d_sym[1].as<int32_t>(), d_sym[2].as<int32_t>()};
    if (!entropy_on_host() && jpegamd_device_decode_planes(parts, d_sym_p, cap, n, prc))
        return 1;
    for (int k = 0; k < 3; ++k) {
        if (prc[k] == 2) { // the host decoder's cases (see jpegamd_decode_tables), or a stream that did not settle
            HostSymbols sym;
            if (!sym.resize(cap[k]))
                return jpegamd_fail("decompress_image: out of host memory");
            if (jpegamd_host_decode_plane(parts.payload[k], parts.len[k], parts.dict[k], parts.dsz[k], sym.data(), cap[k], &n[k],
                                          16))
                return 1;
            const size_t have = n[k] < cap[k] ? n[k] : cap[k];
            JPEGAMD_HIPCHK(hipDeviceSynchronize()); // (the copier runs on its own streams)
            if (have && jpegamd_staged_copy_linear(d_sym[k].p, sym.data(), have * 4, true))
                return 1;
        }
        if (n[k] > cap[k])
            return jpegamd_fail("decompress_image: plane %d decodes to %zu symbols, more than its %zu blocks can hold", k,
                                n[k], nb[k]);
    }
    trace.mark("entropy-decode");
    for (int k = 0; k < 3; ++k) {
        const size_t np = n[k] / 2; // an odd trailing symbol has no run: dropped
        const size_t use = np < nb[k] * 64 + 8 ? np : nb[k] * 64 + 8; // pairs beyond that cannot reach the plane
        if (jpegamd_rle_decode_plane_dev(d_sym[k].as<int32_t>(), use, nb[k], d_coef[k].as<int16_t>(), work[k].p,
                                         wbytes[k], nullptr))
            return 1;
    }
    trace.mark("run-length");
    if (jpegamd_decode_image_dev(d_coef[0].as<int16_t>(), d_coef[1].as<int16_t>(), d_coef[2].as<int16_t>(), r, c,
                                 q_luma, q_chroma, d_out.as<uint8_t>(), (size_t)c * 3, nullptr))
        return 1;
    JPEGAMD_HIPCHK(hipDeviceSynchronize()); // the decode kernel ran on the null stream, the copier uses its own
    trace.mark("transform");
    const int rc = jpegamd_staged_copy(d_out.p, bgr, stride, (size_t)c * 3, r, false);
    trace.mark("download");
    return rc;
}

int jpegamd_bin_encode_dev(const int32_t *d_sym_y, size_t n_y, const int32_t *d_sym_cb, size_t n_cb,
                           const int32_t *d_sym_cr, size_t n_cr, int y_rows, int y_cols, int c_rows, int c_cols,
                           uint8_t *out, size_t cap, size_t *out_len)
{
    return jpegamd_guarded("bin_encode_dev", [&] { return bin_encode_dev_impl(d_sym_y, n_y, d_sym_cb, n_cb, d_sym_cr, n_cr, y_rows, y_cols, c_rows, c_cols, out, cap, out_len); });
}

static int bin_decode_dev_impl(const uint8_t *bin, size_t len, int32_t dims[6], int32_t *const d_sym[3],
                               const size_t cap[3], size_t n[3])
{
    if (!bin || !dims || !d_sym || !cap || !n)
        return jpegamd_fail("bin_decode_dev: null pointer");
    n[0] = n[1] = n[2] = 0;
    JpegamdBinParts parts;
    if (jpegamd_bin_parse(bin, len, parts))
        return 1;
    std::memcpy(dims, parts.dims, sizeof parts.dims);
    bool small = false;
    int prc[3] = {2, 2, 2};
    size_t use_cap[3];
    for (int k = 0; k < 3; ++k)
        use_cap[k] = d_sym[k] ? cap[k] : 0;
    if (!entropy_on_host() && jpegamd_device_decode_planes(parts, d_sym, use_cap, n, prc))
        return 1;
    for (int k = 0; k < 3; ++k) {
        if (prc[k] == 2) { // the host decoder's cases: see jpegamd_decode_tables
            HostSymbols sym;
            const size_t c = use_cap[k];
            if (!sym.resize(c))
                return jpegamd_fail("bin_decode_dev: out of host memory");
            if (jpegamd_host_decode_plane(parts.payload[k], parts.len[k], parts.dict[k], parts.dsz[k], c ? sym.data() : nullptr, c,
                                          &n[k], 16))
                return 1;
            const size_t have = n[k] < c ? n[k] : c;
            JPEGAMD_HIPCHK(hipDeviceSynchronize());
            if (have && jpegamd_staged_copy_linear(d_sym[k], sym.data(), have * 4, true))
                return 1;
        }
        small = small || n[k] > cap[k] || !d_sym[k];
    }
    JPEGAMD_HIPCHK(hipDeviceSynchronize()); // the symbols are the caller's from here on
    if (small)
        return jpegamd_fail("bin_decode: symbol buffer too small (the counts needed are in n[0..2])");
    return 0;
}

int jpegamd_bin_decode_dev(const uint8_t *bin, size_t len, int32_t dims[6], int32_t *const d_sym[3], const size_t cap[3],
                           size_t n[3])
{
    return jpegamd_guarded("bin_decode_dev", [&] { return bin_decode_dev_impl(bin, len, dims, d_sym, cap, n); });
}

int jpegamd_compress_image(const uint8_t *bgr, int rows, int cols, size_t stride, const int32_t q_luma[64],
                           const int32_t q_chroma[64], uint8_t **bin, size_t *bin_len)
{
    return jpegamd_guarded("compress_image", [&] { return compress_image_impl(bgr, rows, cols, stride, q_luma, q_chroma, bin, bin_len); });
}

int jpegamd_decompress_image(const uint8_t *bin, size_t len, const int32_t q_luma[64], const int32_t q_chroma[64],
                             uint8_t *bgr, size_t cap_bytes, size_t stride, int *rows, int *cols)
{
    return jpegamd_guarded("decompress_image", [&] { return decompress_image_impl(bin, len, q_luma, q_chroma, bgr, cap_bytes, stride, rows, cols); });
}

int jpegamd_context_create(int device, jpegamd_context **ctx)
{
    if (!ctx)
        return jpegamd_fail("context_create: null pointer");
    *ctx = nullptr;
    int before = 0;
    JPEGAMD_HIPCHK(hipGetDevice(&before));
    if (jpegamd_init(device)) // checks the device, selects it, creates the copier's staging buffers
        return 1;
    (void)hipSetDevice(before);
    jpegamd_context *c = new (std::nothrow) jpegamd_context;
    if (!c)
        return jpegamd_fail("context_create: out of host memory");
    c->device = device;
    *ctx = c;
    return 0;
}

void jpegamd_context_destroy(jpegamd_context *ctx)
{
    if (!ctx)
        return;
    int before = 0;
    const bool have = hipGetDevice(&before) == hipSuccess;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->arena.release();
    if (have)
        (void)hipSetDevice(before);
    delete ctx;
}

int jpegamd_context_compress_image(jpegamd_context *ctx, const uint8_t *bgr, int rows, int cols, size_t stride,
                                   const int32_t q_luma[64], const int32_t q_chroma[64], uint8_t **bin, size_t *bin_len)
{
    return in_context(ctx, "compress_image",
                      [&] { return compress_image_impl(bgr, rows, cols, stride, q_luma, q_chroma, bin, bin_len); });
}

int jpegamd_context_decompress_image(jpegamd_context *ctx, const uint8_t *bin, size_t len, const int32_t q_luma[64],
                                     const int32_t q_chroma[64], uint8_t *bgr, size_t cap_bytes, size_t stride, int *rows,
                                     int *cols)
{
    return in_context(ctx, "decompress_image", [&] {
        return decompress_image_impl(bin, len, q_luma, q_chroma, bgr, cap_bytes, stride, rows, cols);
    });
}

} // extern "C"
