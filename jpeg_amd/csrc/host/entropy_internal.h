// entropy_internal.h -- pieces of the entropy coder shared between the host implementation
// (bin_container.cpp) and the device one (entropy_device.cpp).  Not part of the C ABI.
#ifndef JPEGAMD_ENTROPY_INTERNAL_H
#define JPEGAMD_ENTROPY_INTERNAL_H
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#define JPEGAMD_HIDDEN __attribute__((visibility("hidden")))

struct JpegamdPlaneCode {
    std::vector<std::pair<int, int>> dict; // (key, frequency) in unordered_map iteration order
    std::vector<uint8_t> payload;          // stringToBitstream's bytes
};

// the whole host path for one plane: dictionary, tree, codes, payload
JPEGAMD_HIDDEN int jpegamd_host_encode_plane(const int32_t *sym, size_t n, JpegamdPlaneCode &pl);

// Dictionary, tree and codes from the distinct symbols in first-occurrence order (all within
// int16) and their counts (`count[s + 32768]`).  Fills pl.dict, flat (code, length) tables
// indexed by s + 32768 and the payload's bit count.  Returns 0, or 2 when a code is longer
// than 32 bits (the caller then takes the host path).
JPEGAMD_HIDDEN int jpegamd_codes_from_order(const std::vector<int> &order, const uint32_t *count,
                                            JpegamdPlaneCode &pl, std::vector<uint32_t> &cbits,
                                            std::vector<uint8_t> &clen, unsigned long long &total_bits);

// saveEncodedData's layout (encode_decode.cpp:1161-1217) from three coded planes
JPEGAMD_HIDDEN int jpegamd_bin_assemble(const JpegamdPlaneCode (&pl)[3], int y_rows, int y_cols, int c_rows, int c_cols,
                                        uint8_t *out, size_t cap, size_t *out_len);

// The entropy stage of one plane on the device: d_sym = n int32 symbols (value, run, value,
// run, ...).  Returns 0 with pl filled, 2 when this input needs the host path (a symbol
// outside int16, a code longer than 32 bits, 2^32 symbols or more), 1 on error.
JPEGAMD_HIDDEN int jpegamd_device_encode_plane(const int32_t *d_sym, size_t n, JpegamdPlaneCode &pl);
#endif
