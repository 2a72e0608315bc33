// host_common.h -- error reporting and host<->device copies shared by the translation units of libjpeg_amd.so
#ifndef JPEGAMD_HOST_COMMON_H
#define JPEGAMD_HOST_COMMON_H
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

// device indices the library keeps per-device state for (copier, scratch address, CU count): one limit for all of it
constexpr int kJpegamdMaxDevices = 64;

// stores the message for jpegamd_last_error() (thread-local) and returns 1
__attribute__((visibility("hidden"), format(printf, 1, 2))) int jpegamd_fail(const char *fmt, ...);

// `rows` rows of `width` bytes between a dense device buffer and (pageable) host rows `hpitch`
// apart, through pinned staging buffers on worker threads (host/staged_copy.cpp); returns
// when the data has arrived.  The _linear form takes one contiguous range.
__attribute__((visibility("hidden"))) int jpegamd_staged_copy(void *dev, uint8_t *host, size_t hpitch, size_t width,
                                                              size_t rows, bool to_device);
__attribute__((visibility("hidden"))) int jpegamd_staged_copy_linear(void *dev, void *host, size_t bytes,
                                                                     bool to_device);

// C++ exceptions must not leave through the C ABI: run `body` and turn any of them into a
// status with a message (host containers can run out of memory on very large inputs)
#ifdef __cplusplus
#include <exception>
#include <new>
template <class Body> int jpegamd_guarded(const char *what, Body body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return jpegamd_fail("%s: out of host memory", what);
    } catch (const std::exception &e) {
        return jpegamd_fail("%s: %s", what, e.what());
    } catch (...) {
        return jpegamd_fail("%s: unknown C++ exception", what);
    }
}
#endif

// compute units of the current device (jpeg_kernels.hip)
__attribute__((visibility("hidden"))) int jpegamd_cu_count(int *out);

// the copier's pinned buffers and streams for the current device, ahead of the first copy (jpegamd_init)
__attribute__((visibility("hidden"))) int jpegamd_staged_copy_warm();

// address of the no-alloc entry point's device scratch on the current device (rle_kernels.hip)
__attribute__((visibility("hidden"))) int jpegamd_rle_noalloc_setup();

// run-length stage of three planes whose per-block pair counts the encoder has written
// (rle_kernels.hip); `sums` = one uint32 of device memory per 256 blocks of each plane
__attribute__((visibility("hidden"))) int jpegamd_rle_encode_counted(
    int16_t *const coef[3], const size_t nb[3], const uint8_t *const counts[3], int32_t *const pairs[3],
    const size_t cap_pairs[3], uint32_t *const block_offsets[3], uint32_t *sums, void *stream);

#define JPEGAMD_HIPCHK(expr)                                                                          \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return jpegamd_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#endif
