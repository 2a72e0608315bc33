/*
 * ref_driver.cpp -- thin extern "C" skin over the REFERENCE'S OWN
 * jpeg_concept/jpeg_function.cpp, compiled from where it lies under
 * /root/reference (oracle/Makefile passes its path as REF_SRC).  Nothing of
 * the reference is copied into this repository; the output goes to
 * oracle/_ref/ (git-ignored) and is used only to validate oracle/jpeg_oracle.c
 * and to generate tests/golden vectors.  TEST INFRASTRUCTURE ONLY.
 *
 * jpeg_function.cpp is the OpenCV-free single-block walk-through whose
 * per-block functions (dct, quantize, zigzag_scan, run_length_encode, the
 * Huffman coder, decodeRLE, inverse_zigzag, inverse_quantize, idct,
 * add_back_128) are the same algorithms as encode_decode.cpp:260-605.
 * encode_decode.cpp itself needs OpenCV headers, which this image lacks, so
 * it is treated as unbuildable here ("Oracle").
 */
#ifndef REF_SRC
#error "pass -DREF_SRC='\"/root/reference/jpeg_concept/jpeg_function.cpp\"'"
#endif

#define main jpeg_concept_main
#include REF_SRC
#undef main

#include <cstdint>
#include <cstring>
#include <sstream>

namespace {
/* the walk-through prints from inside some helpers; silence std::cout */
struct Mute {
    std::streambuf *old;
    std::ostringstream sink;
    Mute() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~Mute() { std::cout.rdbuf(old); }
};

vector<vector<int>> to_vvi(const int32_t *p)
{
    vector<vector<int>> m(8, vector<int>(8));
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            m[i][j] = p[i * 8 + j];
    return m;
}
vector<vector<float>> to_vvf(const float *p)
{
    vector<vector<float>> m(8, vector<float>(8));
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            m[i][j] = p[i * 8 + j];
    return m;
}
} // namespace

extern "C" {

/* pixels (0..255) -> recenter -> dct -> (2-decimal) coefficients */
void ref_recenter_dct(const int32_t *pixels, float *out)
{
    Mute m;
    auto blk = to_vvi(pixels);
    recenterAroundZero(blk);
    vector<vector<float>> d(8, vector<float>(8, 0.0));
    dct(blk, d);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            out[i * 8 + j] = d[i][j];
}

void ref_quantize(float *blk, const int32_t *q)
{
    auto d = to_vvf(blk);
    quantize(d, to_vvi(q));
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            blk[i * 8 + j] = d[i][j];
}

void ref_zigzag(const float *blk, int32_t *zz)
{
    vector<int> z;
    zigzag_scan(to_vvf(blk), z);
    for (int i = 0; i < 64; ++i)
        zz[i] = z[i];
}

int ref_rle(const int32_t *zz, int32_t *out)
{
    vector<int> z(zz, zz + 64);
    vector<int> r = run_length_encode(z);
    for (size_t i = 0; i < r.size(); ++i)
        out[i] = r[i];
    return (int)r.size();
}

/* symbols -> '0'/'1' string; returns its length (or -1 if cap is too small) */
long ref_huffman_encode(const int32_t *sym, long n, char *bits, long cap)
{
    Mute m;
    vector<int> data(sym, sym + n);
    unordered_map<int, int> freq = build_frequency_dict(data);
    HuffmanNode *tree = build_huffman_tree(freq);
    unordered_map<int, string> codes;
    generate_huffman_codes(tree, "", codes);
    string s = huffman_encode(data, codes);
    if ((long)s.size() > cap)
        return -1;
    std::memcpy(bits, s.data(), s.size());
    return (long)s.size();
}

/* frequency dictionary in the container's iteration order: (key, freq) pairs */
long ref_freq_dict_order(const int32_t *sym, long n, int32_t *pairs, long cap_pairs)
{
    vector<int> data(sym, sym + n);
    unordered_map<int, int> freq = build_frequency_dict(data);
    long k = 0;
    for (const auto &p : freq) {
        if (k >= cap_pairs)
            return -1;
        pairs[2 * k] = p.first;
        pairs[2 * k + 1] = p.second;
        ++k;
    }
    return k;
}

/* '0'/'1' string + the symbols it was built from -> decoded symbols */
long ref_huffman_roundtrip(const int32_t *sym, long n, int32_t *out, long cap)
{
    Mute m;
    vector<int> data(sym, sym + n);
    unordered_map<int, int> freq = build_frequency_dict(data);
    HuffmanNode *tree = build_huffman_tree(freq);
    unordered_map<int, string> codes;
    generate_huffman_codes(tree, "", codes);
    string s = huffman_encode(data, codes);
    vector<int> dec = huffman_decode(s, tree);
    if ((long)dec.size() > cap)
        return -1;
    for (size_t i = 0; i < dec.size(); ++i)
        out[i] = dec[i];
    return (long)dec.size();
}

/* decodeRLE: returns number of blocks; block b has lens[b] values at out+64*b */
long ref_decode_rle(const int32_t *sym, long n, int32_t *out, int32_t *lens, long cap_blocks)
{
    vector<int> data(sym, sym + n);
    vector<vector<int>> blocks = decodeRLE(data);
    if ((long)blocks.size() > cap_blocks)
        return -1;
    for (size_t b = 0; b < blocks.size(); ++b) {
        lens[b] = (int32_t)blocks[b].size();
        for (size_t i = 0; i < blocks[b].size(); ++i)
            out[64 * b + i] = blocks[b][i];
    }
    return (long)blocks.size();
}

/* zigzag vector -> inverse_zigzag -> inverse_quantize -> idct -> add_back_128 */
void ref_decode_block(const int32_t *zz, const int32_t *q, int32_t *pixels)
{
    vector<int> z(zz, zz + 64);
    vector<vector<float>> blk;
    inverse_zigzag(z, blk);
    inverse_quantize(blk, to_vvi(q));
    vector<vector<int>> img(8, vector<int>(8, 0));
    idct(blk, img);
    add_back_128(img);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            pixels[i * 8 + j] = img[i][j];
}

/* the walk-through's table scaling (jpeg_function.cpp:465-484) */
void ref_luma_table(int quality, int32_t *q)
{
    int sf = (quality < 50) ? 5000 / quality : 200 - quality * 2;
    vector<vector<int>> t = baseline_quantization_table;
    for (auto &row : t)
        for (int &val : row)
            val = max(1, min(255, (val * sf + 50) / 100));
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j)
            q[i * 8 + j] = t[i][j];
}

} // extern "C"
