// host_common.h -- error reporting shared by the translation units of libjpeg_amd.so
#ifndef JPEGAMD_HOST_COMMON_H
#define JPEGAMD_HOST_COMMON_H
#include <hip/hip_runtime.h>

// stores the message for jpegamd_last_error() (thread-local) and returns 1
__attribute__((visibility("hidden"), format(printf, 1, 2))) int jpegamd_fail(const char *fmt, ...);

#define JPEGAMD_HIPCHK(expr)                                                                          \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return jpegamd_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#endif
