/*
 * jpeg_amd.h -- C ABI of libjpeg_amd.so, the MI355X (gfx950) implementation of
 * the Fulim13/jpeg block-transform hot path.  Plain pointers and sizes only.
 * Citations name the reference interface each entry point stands in for
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure;
 *     jpegamd_last_error() then describes the failure (thread-local text).
 *   - "_dev" entry points take DEVICE pointers, enqueue on `stream`
 *     (a hipStream_t passed as void*, NULL = the default stream), allocate
 *     nothing and do not synchronise.  The others take HOST pointers, allocate
 *     and free device memory per call and return when the result is in place
 *     (the ownership model of kernel.cu:97-104 / 160-164).
 *   - quantisation tables are 64 host int32 in natural (row-major) order with
 *     entries 1..255, as main.cpp:64-80 builds them.
 *   - images are interleaved 3 x u8 in cv::imread's channel order (B,G,R), with
 *     `stride` bytes between rows; rows and cols must be multiples of 16
 *     (encode_decode.cpp:29-37 guarantees it before the path is entered).
 *   - coefficient planes: one int16 per coefficient, 64 per 8x8 block in the
 *     reference's zigzag order (encode_decode.cpp:312-348), blocks in raster
 *     order with ceil(cols/8) blocks per block-row (encode_decode.cpp:233-258).
 */
#ifndef JPEG_AMD_H
#define JPEG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Selects `device` for the calling thread and checks that it is a gfx950.
 * Fails loudly if there is no usable GPU: there is no CPU fallback.  Also creates the
 * pinned staging buffers (32 MiB) and copy streams the host-pointer entry points use,
 * which their first call would otherwise create. */
int jpegamd_init(int device);
const char *jpegamd_last_error(void);
/* Library/ABI version, bumped when a signature changes. */
int jpegamd_abi_version(void);

/* ---- legacy block boundary --------------------------------------------- */
/* kernel.cuh:6-7 / kernel.cu:91-165 dctQuantizationParallel, flattened:
 * blocks = nb x 64 int32 pixels 0..255 (NOT level shifted, row-major 8x8),
 * out = nb x 64 float quantised coefficients, natural order, integer valued,
 * never -0.  HOST pointers. */
int jpegamd_fdct_quant_blocks(const int32_t *blocks, size_t nb, const int32_t q[64], float *out);
/* kernel.cuh:9-10 / kernel.cu:217-289 dequantizeInverseDCTParallel, flattened:
 * coef = nb x 64 float quantised coefficients in natural order,
 * out = nb x 64 int32 = IDCT + 128, UNCLAMPED (the caller clamps,
 * encode_decode.cpp:808).  HOST pointers. */
int jpegamd_dequant_idct_blocks(const float *coef, size_t nb, const int32_t q[64], int32_t *out);
/* the same two on DEVICE pointers */
int jpegamd_fdct_quant_blocks_dev(const int32_t *d_blocks, size_t nb, const int32_t q[64],
                                  float *d_out, void *stream);
int jpegamd_dequant_idct_blocks_dev(const float *d_coef, size_t nb, const int32_t q[64],
                                    int32_t *d_out, void *stream);

/* ---- one plane ------------------------------------------------------------ */
/* encodeChannelGPU up to and including zigzag_scan, encode_decode.cpp:720-743:
 * extractBlocks (zero fill outside the plane, 233-258) -> level shift -> DCT ->
 * quantise -> zigzag.  Any rows/cols >= 1.  d_coef_zz holds
 * ceil(rows/8)*ceil(cols/8)*64 int16.  DEVICE pointers. */
int jpegamd_encode_plane_dev(const uint8_t *d_plane, int rows, int cols, size_t stride,
                             const int32_t q[64], int16_t *d_coef_zz, void *stream);
/* decodeChannelGPU from inverse_zigzag on, encode_decode.cpp:784-818:
 * dequantise -> IDCT -> +128 -> clamp -> scatter (samples outside the plane are
 * dropped).  DEVICE pointers. */
int jpegamd_decode_plane_dev(const int16_t *d_coef_zz, int rows, int cols, const int32_t q[64],
                             uint8_t *d_plane, size_t stride, void *stream);

/* ---- fused colour image ----------------------------------------------------- */
/* RGB2YCbCr (encode_decode.cpp:39-75) + chromaSubsampling (117-152) + the three
 * encodeChannelGPU transforms (720-743) of mainEncode (1327-1329) in ONE launch.
 * d_coef_y: (rows/8)*(cols/8)*64 int16; d_coef_cb/cr: (rows/16)*(cols/16)*64.
 * A batch: n equally sized images stored back to back are one image of n*rows rows
 * (MCU rows are independent); their coefficient planes come out back to back.  The
 * same holds for jpegamd_decode_image_dev.  One launch fills the GPU where a small
 * image alone cannot. */
int jpegamd_encode_image_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride,
                             const int32_t q_luma[64], const int32_t q_chroma[64],
                             int16_t *d_coef_y, int16_t *d_coef_cb, int16_t *d_coef_cr,
                             void *stream);
/* RGB2YCbCr (encode_decode.cpp:39-75) + chromaSubsampling (117-152) on their own: the colour
 * stage of the fused encoder (the same kernel code, stopped before the transform) writing the
 * planes main.cpp:99-104 holds at that point.  d_y: rows x cols bytes; d_cb, d_cr: rows/2 x
 * cols/2 bytes, dense.  Bit-exact with the reference for every input triple (GPU test over all
 * 2^24). */
int jpegamd_colour_planes_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride, uint8_t *d_y, uint8_t *d_cb,
                              uint8_t *d_cr, void *stream);
/* The three decodeChannelGPU transforms (encode_decode.cpp:784-818) of mainDecode
 * (1365-1367) + upsampleChroma (154-173) + YCbCr2RGB (77-114) in ONE launch. */
int jpegamd_decode_image_dev(const int16_t *d_coef_y, const int16_t *d_coef_cb,
                             const int16_t *d_coef_cr, int rows, int cols,
                             const int32_t q_luma[64], const int32_t q_chroma[64],
                             uint8_t *d_bgr, size_t stride, void *stream);
/* HOST-pointer forms of the two above (allocate, copy in, launch, copy out). */
int jpegamd_encode_image(const uint8_t *bgr, int rows, int cols, size_t stride,
                         const int32_t q_luma[64], const int32_t q_chroma[64],
                         int16_t *coef_y, int16_t *coef_cb, int16_t *coef_cr);
int jpegamd_decode_image(const int16_t *coef_y, const int16_t *coef_cb, const int16_t *coef_cr,
                         int rows, int cols, const int32_t q_luma[64],
                         const int32_t q_chroma[64], uint8_t *bgr, size_t stride);

/* ---- run-length stage ----------------------------------------------------- */
/* Bytes of device workspace the two calls below need for a plane of nb blocks. */
size_t jpegamd_rle_workspace_bytes(size_t nb);
/* run_length_encode over a whole plane, encode_decode.cpp:350-373 applied block by
 * block in raster order (:643-645): d_coef_zz = nb x 64 int16 in zigzag order (the
 * output of the encode calls above) -> d_pairs = (value, run) int32 pairs, the
 * memory layout of the reference's std::vector<int>; d_block_offsets[b] = index of
 * block b's first pair, d_block_offsets[nb] = total number of pairs (nb + 1 uint32).
 * Pairs beyond cap_pairs are not written (the total still is): size the buffer
 * for nb*64 pairs, or call again when the total exceeds the capacity.
 * DEVICE pointers, three launches on `stream` (pairs per 256 blocks, scan, emit), no allocation. */
int jpegamd_rle_encode_plane_dev(const int16_t *d_coef_zz, size_t nb, int32_t *d_pairs, size_t cap_pairs,
                                 uint32_t *d_block_offsets, void *d_workspace, size_t workspace_bytes,
                                 void *stream);
/* The three planes of an image (mainEncode's three encodeChannel calls, encode_decode.cpp:
 * 1327-1329) in ONE launch: arrays of three device pointers / sizes, one workspace of
 * jpegamd_rle_workspace_bytes(nb[0] + nb[1] + nb[2]) bytes. */
int jpegamd_rle_encode_image_dev(const int16_t *const d_coef_zz[3], const size_t nb[3], int32_t *const d_pairs[3],
                                 const size_t cap_pairs[3], uint32_t *const d_block_offsets[3], void *d_workspace,
                                 size_t workspace_bytes, void *stream);
/* jpegamd_encode_image_dev followed by the run-length stage of its three planes
 * (encodeChannelGPU up to and including run_length_encode, encode_decode.cpp:720-751, for Y, Cb
 * and Cr) with the per-block pair counts handed over from the encoder's registers, so the
 * run-length kernels read the coefficient planes once.  Outputs as the two calls made
 * separately: d_coef[3] (kept: the decoder or a caller may want them), d_pairs[3],
 * d_block_offsets[3].  Workspace: jpegamd_rle_workspace_bytes(blocks of the three planes
 * together) bytes, 256-byte aligned.  Four launches on `stream`, no allocation. */
int jpegamd_encode_image_rle_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride, const int32_t q_luma[64],
                                 const int32_t q_chroma[64], int16_t *const d_coef[3], int32_t *const d_pairs[3],
                                 const size_t cap_pairs[3], uint32_t *const d_block_offsets[3], void *d_workspace,
                                 size_t workspace_bytes, void *stream);
/* jpegamd_rle_encode_plane_dev for callers that cannot allocate or be handed a workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It
 * has NO counterpart in the reference.  Same inputs, outputs and kernels; the per-workgroup
 * counts live in a fixed device array inside the library (64.25 KiB per GPU, held until
 * the process ends) instead of a workspace argument.  Calls no hipMalloc/hipFree or any
 * allocating routine.  nb is limited to jpegamd_rle_noalloc_max_blocks() (4 Mi blocks, a
 * 16384 x 16384 plane); larger planes are rejected with an error.  All calls on one GPU share
 * that array: calls that overlap on one GPU (different streams or threads) race -- serialise
 * them.  Nothing in the array needs to start at zero (every word read was written by the same
 * call first).  UNDER STREAM CAPTURE call jpegamd_init(device) (or make one un-captured call on
 * that device) FIRST: the array's address on a device is looked up once per device with
 * hipGetSymbolAddress, which must not run inside a capture; a call that finds the address not yet
 * looked up while `stream` is capturing returns an error instead of looking it up. */
int jpegamd_rle_encode_plane_noalloc_dev(const int16_t *d_coef_zz, size_t nb, int32_t *d_pairs, size_t cap_pairs,
                                         uint32_t *d_block_offsets, void *stream);
size_t jpegamd_rle_noalloc_max_blocks(void);
/* decodeRLE, encode_decode.cpp:484-518: n_pairs (value, run) int32 pairs expanded
 * into one sequence cut every 64 values -> nb x 64 int16.  Values that would fall
 * beyond block nb-1 are dropped (the trailing pad symbols of loadEncodedData,
 * encode_decode.cpp:1261-1263, which the reference discards at :702); blocks the
 * stream does not reach are zero; non-positive runs expand to nothing.  Run lengths are
 * untrusted (they come from a file): any run is treated as min(run, nb*64 + 1) and positions
 * are 64-bit, so an oversized run fills to the end of the plane and everything after it is
 * dropped, as the reference's expand-then-cut would, in bounded time. */
int jpegamd_rle_decode_plane_dev(const int32_t *d_pairs, size_t n_pairs, size_t nb, int16_t *d_coef_zz,
                                 void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- entropy stage and .bin container (HOST, no GPU involved) --------------- */
/* encodeChannel's tail + saveEncodedData, encode_decode.cpp:375-466, 648-658,
 * 1093-1127, 1161-1217: three symbol streams (the (value, run) ints of the
 * run-length stage, flattened) -> the bytes of a .bin file, bit-compatible with
 * the reference built against libstdc++ (dictionary order and Huffman tie-breaks
 * included).  *out_len receives the size needed; if out is NULL or cap is too
 * small the call fails after setting it. */
int jpegamd_bin_encode(const int32_t *sym_y, size_t n_y, const int32_t *sym_cb, size_t n_cb,
                       const int32_t *sym_cr, size_t n_cr, int y_rows, int y_cols, int c_rows, int c_cols,
                       uint8_t *out, size_t cap, size_t *out_len);
/* loadEncodedData + decodeChannel's head, encode_decode.cpp:1225-1302, 660-668,
 * 468-482: .bin bytes -> dims[6] (y, cb, cr rows/cols) and the three symbol
 * streams.  Like the reference it decodes every bit of every payload byte, so
 * up to 7 pad bits may yield spurious trailing symbols (the run-length decoder
 * above drops them).  *n_* receive the symbol counts; if a buffer is too small
 * the call fails after setting them. */
int jpegamd_bin_decode(const uint8_t *bin, size_t len, int32_t dims[6],
                       int32_t *sym_y, size_t cap_y, size_t *n_y,
                       int32_t *sym_cb, size_t cap_cb, size_t *n_cb,
                       int32_t *sym_cr, size_t cap_cr, size_t *n_cr);

/* ---- whole image, HOST buffers ------------------------------------------------ */
/* The same container from symbol streams that are already on the device (e.g. the pair
 * streams of jpegamd_rle_encode_plane_dev): histogram, first-occurrence order, code lookup
 * and bit packing run on the GPU (entropy_kernels.hip), the order-defining containers
 * (std::unordered_map, std::priority_queue; encode_decode.cpp:375-456) on the host; `out`
 * is host memory.  Inputs outside the device tables (a symbol beyond int16, a code longer
 * than 32 bits) take the host coder.  Synchronises the device. */
int jpegamd_bin_encode_dev(const int32_t *d_sym_y, size_t n_y, const int32_t *d_sym_cb, size_t n_cb,
                           const int32_t *d_sym_cr, size_t n_cr, int y_rows, int y_cols, int c_rows, int c_cols,
                           uint8_t *out, size_t cap, size_t *out_len);

/* jpegamd_bin_decode with huffman_decode (encode_decode.cpp:468-482) on the device: .bin bytes
 * on the host -> dims[6] and the three symbol streams in DEVICE memory (d_sym[k], cap[k] int32
 * each; n[k] receives the count found -- if a buffer is missing or too small the call fails
 * after setting all three).  The bit stream has no restart points: threads decode 256-bit
 * subsequences from assumed starts and fall into step (prefix codes resynchronise); the same
 * symbols as the bit-serial walk, pad-bit symbols included.  Dictionaries the device tables do
 * not cover (a symbol outside int16) take the host decoder.  Synchronises the device. */
int jpegamd_bin_decode_dev(const uint8_t *bin, size_t len, int32_t dims[6], int32_t *const d_sym[3], const size_t cap[3],
                           size_t n[3]);

/* main.cpp:96-133 with platform "GPU" (RGB2YCbCr, chromaSubsampling, mainEncode,
 * saveEncodedData) : image -> malloc'ed .bin bytes (release with jpegamd_free). */
int jpegamd_compress_image(const uint8_t *bgr, int rows, int cols, size_t stride, const int32_t q_luma[64],
                           const int32_t q_chroma[64], uint8_t **bin, size_t *bin_len);
/* main.cpp:140-168 (loadEncodedData, mainDecode, upsampleChroma, YCbCr2RGB): .bin bytes
 * -> image.  The tables are not stored in the file (the reference passes them
 * again).  *rows, *cols are always set from the header; bgr == NULL only queries them. */
int jpegamd_decompress_image(const uint8_t *bin, size_t len, const int32_t q_luma[64], const int32_t q_chroma[64],
                             uint8_t *bgr, size_t cap_bytes, size_t stride, int *rows, int *cols);
void jpegamd_free(void *p);

/* A context keeps the device buffers of the two calls above from call to call, where the
 * plain calls allocate and free them each time like the reference's kernel.cu:97-104 and
 * 160-164 (no counterpart in the reference; SURVEY.md 8b proposes it).  `device` is selected
 * only for the duration of a call.  One call at a time per context (serialised inside);
 * buffers grow to the largest image seen and are released by jpegamd_context_destroy. */
typedef struct jpegamd_context jpegamd_context;
int jpegamd_context_create(int device, jpegamd_context **ctx);
void jpegamd_context_destroy(jpegamd_context *ctx);
int jpegamd_context_compress_image(jpegamd_context *ctx, const uint8_t *bgr, int rows, int cols, size_t stride,
                                   const int32_t q_luma[64], const int32_t q_chroma[64], uint8_t **bin,
                                   size_t *bin_len);
int jpegamd_context_decompress_image(jpegamd_context *ctx, const uint8_t *bin, size_t len, const int32_t q_luma[64],
                                     const int32_t q_chroma[64], uint8_t *bgr, size_t cap_bytes, size_t stride,
                                     int *rows, int *cols);
/* main.cpp:3-20, 51-80: the Annex-K base tables scaled for quality 1..100
 * (row-major 8x8, clamped to 1..255); non-zero for a quality outside that range. */
int jpegamd_quality_tables(int quality, int32_t q_luma[64], int32_t q_chroma[64]);

/* analysis.cpp:43-90: the two FIXED table pairs analysis uses instead of scaling (quality >= 90:
 * its own pair, the luma one with 12 at [5][5]; otherwise the base tables). */
int jpegamd_analysis_tables(int quality, int32_t q_luma[64], int32_t q_chroma[64]);

/* Times `iters` steps of encode followed by decode on `stream`, in that alternating
 * sequence, with HIP events recorded on that same stream between every two launches;
 * writes the mean milliseconds per launch of each kernel.  Used by bench.py for the
 * roofline figures. */
int jpegamd_time_image_dev(const uint8_t *d_bgr, int rows, int cols, size_t stride,
                           const int32_t q_luma[64], const int32_t q_chroma[64],
                           int16_t *d_coef_y, int16_t *d_coef_cb, int16_t *d_coef_cr,
                           uint8_t *d_bgr_out, int iters, void *stream,
                           float *enc_ms, float *dec_ms);

#ifdef __cplusplus
}
#endif
#endif
