// kernel_cuh_adapter.cpp -- the reference's own device boundary, kernel.cuh:6-10,
// re-exported on top of libjpeg_amd.so: the two C++ symbols a build of the
// reference links against instead of its kernel.o (README.md:79-85:
// `g++ main.o kernel.o ...` becomes `g++ main.o -ljpeg_amd_kernelcuh -ljpeg_amd`).
// The reference's header is not shipped here; its translation units keep
// including their own kernel.cuh, only the definitions change.
//
// Semantics follow kernel.cu:91-165 and 217-289: the callee sizes the output
// containers, blocks are row-major 8x8, pixels arrive 0..255 (the callee level
// shifts), coefficients are natural order and never -0, the inverse returns
// IDCT + 128 unclamped.  Unlike the reference, failures are not silent: they
// throw std::runtime_error (the convention of encode_decode.cpp:1156-1159).
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/jpeg_amd.h"

using std::vector;

namespace {
void flatten_table(const vector<vector<int>> &t, int32_t q[64])
{
    if (t.size() != 8)
        throw std::runtime_error("quantization_table must be 8x8");
    for (int i = 0; i < 8; ++i) {
        if (t[i].size() != 8)
            throw std::runtime_error("quantization_table must be 8x8");
        for (int j = 0; j < 8; ++j)
            q[i * 8 + j] = t[i][j];
    }
}
[[noreturn]] void fail(const char *where)
{
    throw std::runtime_error(std::string(where) + ": " + jpegamd_last_error());
}
} // namespace

// kernel.cuh:6-7
void dctQuantizationParallel(const vector<vector<vector<int>>> &input_blocks,
                             const vector<vector<int>> &quantization_table,
                             vector<vector<float>> &quantize_coefficients)
{
    const size_t nb = input_blocks.size();
    int32_t q[64];
    flatten_table(quantization_table, q);
    vector<int32_t> flat(nb * 64);
    for (size_t b = 0; b < nb; ++b)
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j)
                flat[b * 64 + i * 8 + j] = input_blocks[b][i][j]; // kernel.cu:120-130
    vector<float> out(nb * 64);
    if (jpegamd_fdct_quant_blocks(flat.data(), nb, q, out.data()))
        fail("dctQuantizationParallel");
    quantize_coefficients.resize(nb); // kernel.cu:154-158
    for (size_t b = 0; b < nb; ++b)
        quantize_coefficients[b].assign(out.begin() + b * 64, out.begin() + b * 64 + 64);
}

// kernel.cuh:9-10
void dequantizeInverseDCTParallel(const vector<vector<vector<float>>> &quantize_coefficients,
                                  const vector<vector<int>> &quantization_table,
                                  vector<vector<vector<int>>> &output_blocks)
{
    const size_t nb = quantize_coefficients.size();
    int32_t q[64];
    flatten_table(quantization_table, q);
    vector<float> flat(nb * 64);
    for (size_t b = 0; b < nb; ++b)
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j)
                flat[b * 64 + i * 8 + j] = quantize_coefficients[b][i][j]; // kernel.cu:247-257
    vector<int32_t> out(nb * 64);
    if (jpegamd_dequant_idct_blocks(flat.data(), nb, q, out.data()))
        fail("dequantizeInverseDCTParallel");
    output_blocks.resize(nb, vector<vector<int>>(8, vector<int>(8))); // kernel.cu:272
    for (size_t b = 0; b < nb; ++b)
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j)
                output_blocks[b][i][j] = out[b * 64 + i * 8 + j];
}
