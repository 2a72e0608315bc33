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

// The same for the three planes of an image in one go (two waits in all): code[k].dict filled (payload left
// empty), bytes[k] = payload bytes of plane k, d_out = the file's payload section on the device (planes back to
// back), its kernels still running on the null stream.  0 / 2 (some plane needs the host path) / 1.
struct JpegamdDevMem;
JPEGAMD_HIDDEN int jpegamd_device_encode_image(const int32_t *const d_sym[3], const size_t n[3], JpegamdPlaneCode code[3],
                                               size_t bytes[3], JpegamdDevMem &d_out);
// saveEncodedData's layout with the payload section copied straight from the device into the file image
JPEGAMD_HIDDEN int jpegamd_bin_assemble_from_device(const JpegamdPlaneCode (&pl)[3], const size_t bytes[3], const void *d_payloads,
                                                    int y_rows, int y_cols, int c_rows, int c_cols, uint8_t *out, size_t cap,
                                                    size_t *out_len);

// Decoding tables of one plane from its dictionary as stored in the file (dsz (key, frequency)
// int32 pairs): the tree built exactly as decodeChannel does (encode_decode.cpp:664: the pairs go
// through a std::unordered_map first), flattened for the device decoder.
//   window[v], v = the next kDecodeWindow bits of the stream (MSB first):
//       length << 24 | (symbol + 32768)   the first whole code among those bits, or
//       node                              (length 0) the inner node reached after all of them
//   left / right / value: the tree, node by node (left < 0: a leaf carrying `value`)
// Returns 0, or 2 when the device decoder does not cover this dictionary (fewer than two
// symbols, a symbol outside int16): the caller then takes the host decoder.
constexpr int kDecodeWindow = 12;
struct JpegamdDecodeTables {
    std::vector<uint32_t> window;
    std::vector<int32_t> left, right, value;
    int root = -1;
};
JPEGAMD_HIDDEN int jpegamd_decode_tables(const uint8_t *dict, int dsz, JpegamdDecodeTables &t);

// The pieces of a .bin file (loadEncodedData's layout, encode_decode.cpp:1236-1296), bounds checked.
struct JpegamdBinParts {
    int32_t dims[6];
    const uint8_t *payload[3];
    size_t len[3];
    const uint8_t *dict[3]; // dsz (key, frequency) int32 pairs each
    int dsz[3];
};
JPEGAMD_HIDDEN int jpegamd_bin_parse(const uint8_t *bin, size_t len, JpegamdBinParts &parts);
// huffman_decode of one plane on the host (PlaneDecoder, `threads` chunks); as the device form below
JPEGAMD_HIDDEN int jpegamd_host_decode_plane(const uint8_t *payload, size_t len, const uint8_t *dict, int dsz,
                                             int32_t *sym, size_t cap, size_t *n, unsigned threads);

// huffman_decode of one plane on the device (entropy_decode_kernels.hip): payload = `len` bytes
// on the HOST, every bit of every byte is decoded (the reference's loader knows no bit count,
// encode_decode.cpp:1261-1263) into d_sym[0 .. cap) on the DEVICE; *n = the number of symbols
// found (more than cap: the buffer was too small, only cap were written).  Returns 0, 2 when
// the host decoder must take over, 1 on error.
JPEGAMD_HIDDEN int jpegamd_device_decode_plane(const uint8_t *payload, size_t len, const uint8_t *dict, int dsz,
                                               int32_t *d_sym, size_t cap, size_t *n);
// The three planes of a parsed file in one go (one upload of the payload range, one of the tables, one wait).
// rc[k]: 0 = decoded (n[k] symbols found, min(n[k], cap[k]) written to d_sym[k] if it is not null), 2 = the host
// decoder must take plane k.  Returns 1 on error.  The emit kernels are left running on the null stream when the
// call runs inside a context (the arena's buffers outlive it); otherwise the device is synchronised.
JPEGAMD_HIDDEN int jpegamd_device_decode_planes(const JpegamdBinParts &parts, int32_t *const d_sym[3], const size_t cap[3],
                                                size_t n[3], int rc[3]);
// pinned host scratch of the calling thread, at least `bytes` (grows, lives as long as the thread); null on failure
JPEGAMD_HIDDEN void *jpegamd_pinned_scratch(size_t bytes);
#endif
