// png_io.h -- PNG in and out, and baseline TIFF in, for the command-line harness (jpegamd_main),
// on zlib alone.
// The reference reads its inputs with cv::imread(path, IMREAD_COLOR) and writes the
// reconstruction with cv::imwrite(".png") (main.cpp:86, :300); its own test images are PNG
// files.  OpenCV is not in this image, zlib is.  What imread's 8-bit colour mode makes of a PNG
// is reproduced for the non-interlaced files the reference ships: 8- and 16-bit samples (the
// high byte is kept), grey, grey + alpha, RGB, RGBA (alpha dropped) and palette images with 1,
// 2, 4 or 8 bits per index; the result is interleaved B,G,R.  Interlaced files are rejected.
#ifndef JPEGAMD_PNG_IO_H
#define JPEGAMD_PNG_IO_H
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pngio {

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline bool is_png(const std::vector<uint8_t> &f)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    return f.size() >= 8 && std::memcmp(f.data(), sig, 8) == 0;
}

// file bytes -> rows, cols, interleaved B,G,R
inline void decode(const std::vector<uint8_t> &f, int &rows, int &cols, std::vector<uint8_t> &bgr)
{
    if (!is_png(f))
        throw std::runtime_error("not a PNG file");
    size_t off = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool end = false;
    while (!end && off + 12 <= f.size()) {
        const uint32_t len = be32(&f[off]);
        const uint8_t *type = &f[off + 4], *data = &f[off + 8];
        if (len > f.size() - off - 12)
            throw std::runtime_error("PNG chunk runs past the end of the file");
        if (crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(data + len))
            throw std::runtime_error("PNG chunk checksum mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13)
                throw std::runtime_error("malformed PNG header");
            w = be32(data), h = be32(data + 4);
            depth = data[8], ctype = data[9], interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            end = true;
        }
        off += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0 || w > 65535 || h > 65535)
        throw std::runtime_error("PNG without a usable header");
    if (interlace)
        throw std::runtime_error("interlaced PNG files are not supported");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error("PNG colour type not supported");
    }
    const bool sub_byte = depth == 1 || depth == 2 || depth == 4;
    if (!(depth == 8 || depth == 16 || (sub_byte && (ctype == 0 || ctype == 3))))
        throw std::runtime_error("PNG bit depth not supported");
    if (ctype == 3 && (depth == 16 || plte.size() < 3))
        throw std::runtime_error("malformed palette PNG");
    const size_t bits_px = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bits_px + 7) / 8, bpp = bits_px < 8 ? 1 : bits_px / 8;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        throw std::runtime_error("PNG image data does not inflate to the size its header announces");
    // undo the scanline filters in place (PNG specification, section 9)
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *cur = &raw[(stride + 1) * y + 1];
        const uint8_t *up = y ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
        const int filter = cur[-1];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int pred;
            switch (filter) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: throw std::runtime_error("PNG scanline with an unknown filter type");
            }
            cur[i] = (uint8_t)(cur[i] + pred);
        }
    }
    rows = (int)h, cols = (int)w;
    bgr.resize((size_t)w * h * 3);
    const int step = depth == 16 ? 2 : 1; // 16-bit samples: big endian, the high byte is what an 8-bit reader keeps
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *row = &raw[(stride + 1) * y + 1];
        uint8_t *out = &bgr[(size_t)y * w * 3];
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t r, g, b;
            if (ctype == 2 || ctype == 6) {
                const uint8_t *p = row + (size_t)x * channels * step;
                r = p[0], g = p[step], b = p[2 * step];
            } else {
                unsigned v;
                if (sub_byte) {
                    const unsigned per = 8 / depth, shift = (per - 1 - x % per) * depth;
                    v = (row[x / per] >> shift) & ((1u << depth) - 1);
                    if (ctype == 0)
                        v = v * 255 / ((1u << depth) - 1);
                } else {
                    v = row[(size_t)x * channels * step];
                }
                if (ctype == 3) {
                    if ((size_t)v * 3 + 2 >= plte.size())
                        throw std::runtime_error("PNG palette index out of range");
                    r = plte[v * 3], g = plte[v * 3 + 1], b = plte[v * 3 + 2];
                } else {
                    r = g = b = (uint8_t)v;
                }
            }
            out[3 * x] = b, out[3 * x + 1] = g, out[3 * x + 2] = r;
        }
    }
}

// interleaved B,G,R -> an 8-bit RGB PNG (filter type 0 on every line, fastest deflate level)
inline void write(const std::string &path, int rows, int cols, const uint8_t *bgr)
{
    const size_t stride = (size_t)cols * 3;
    std::vector<uint8_t> raw((stride + 1) * rows);
    for (int y = 0; y < rows; ++y) {
        uint8_t *line = &raw[(stride + 1) * y];
        const uint8_t *src = bgr + stride * y;
        *line++ = 0;
        for (int x = 0; x < cols; ++x)
            line[3 * x] = src[3 * x + 2], line[3 * x + 1] = src[3 * x + 1], line[3 * x + 2] = src[3 * x];
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 1) != Z_OK)
        throw std::runtime_error("deflate failed while writing " + path);
    std::ofstream out(path, std::ios::binary);
    if (!out)
        throw std::runtime_error("Cannot open file for writing: " + path);
    auto chunk = [&out](const char type[4], const uint8_t *data, uint32_t len) {
        uint8_t head[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len,
                           (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
        uLong crc = crc32(crc32(0L, Z_NULL, 0), head + 4, 4);
        if (len)
            crc = crc32(crc, data, len);
        const uint8_t tail[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
        out.write((const char *)head, 8);
        if (len)
            out.write((const char *)data, len);
        out.write((const char *)tail, 4);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.write((const char *)sig, 8);
    const uint8_t ihdr[13] = {(uint8_t)(cols >> 24), (uint8_t)(cols >> 16), (uint8_t)(cols >> 8), (uint8_t)cols,
                              (uint8_t)(rows >> 24), (uint8_t)(rows >> 16), (uint8_t)(rows >> 8), (uint8_t)rows,
                              8, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    for (size_t off = 0; off < zlen; off += 1u << 20) // 1 MiB IDAT chunks
        chunk("IDAT", z.data() + off, (uint32_t)std::min<size_t>(1u << 20, zlen - off));
    chunk("IEND", nullptr, 0);
    if (!out)
        throw std::runtime_error("write failed: " + path);
}

} // namespace pngio

// Baseline TIFF, as far as the reference's own img/*.tif need it: uncompressed strips of 8-bit
// grey (either photometric sense) or RGB / RGBA samples, chunky layout, either byte order.
namespace tiffio {

inline bool is_tiff(const std::vector<uint8_t> &f)
{
    return f.size() >= 8 && ((f[0] == 'I' && f[1] == 'I' && f[2] == 42 && f[3] == 0) ||
                             (f[0] == 'M' && f[1] == 'M' && f[2] == 0 && f[3] == 42));
}

inline void decode(const std::vector<uint8_t> &f, int &rows, int &cols, std::vector<uint8_t> &bgr)
{
    if (!is_tiff(f))
        throw std::runtime_error("not a TIFF file");
    const bool le = f[0] == 'I';
    auto u16 = [&](size_t o) -> uint32_t {
        if (o + 2 > f.size())
            throw std::runtime_error("TIFF structure runs past the end of the file");
        return le ? (uint32_t)f[o] | ((uint32_t)f[o + 1] << 8) : ((uint32_t)f[o] << 8) | f[o + 1];
    };
    auto u32 = [&](size_t o) -> uint32_t {
        if (o + 4 > f.size())
            throw std::runtime_error("TIFF structure runs past the end of the file");
        return le ? u16(o) | (u16(o + 2) << 16) : (u16(o) << 16) | u16(o + 2);
    };
    const size_t ifd = u32(4);
    const uint32_t n = u16(ifd);
    uint32_t w = 0, h = 0, bits = 1, compression = 1, photometric = 1, spp = 1, rows_per_strip = 0xFFFFFFFFu, planar = 1;
    std::vector<uint32_t> offsets, counts;
    auto values = [&](size_t entry, std::vector<uint32_t> &out) { // SHORT or LONG arrays, inline or at an offset
        const uint32_t type = u16(entry + 2), count = u32(entry + 4);
        const size_t size = type == 3 ? 2 : type == 4 ? 4 : 0;
        if (!size || count > (1u << 24))
            throw std::runtime_error("TIFF field of an unsupported type");
        size_t at = size * count <= 4 ? entry + 8 : u32(entry + 8);
        out.clear();
        for (uint32_t i = 0; i < count; ++i, at += size)
            out.push_back(size == 2 ? u16(at) : u32(at));
    };
    for (uint32_t i = 0; i < n; ++i) {
        const size_t e = ifd + 2 + 12 * (size_t)i;
        std::vector<uint32_t> v;
        switch (u16(e)) {
        case 256: values(e, v); w = v.at(0); break;
        case 257: values(e, v); h = v.at(0); break;
        case 258:
            values(e, v);
            bits = v.at(0);
            for (uint32_t b : v)
                if (b != bits)
                    throw std::runtime_error("TIFF with different bit depths per sample is not supported");
            break;
        case 259: values(e, v); compression = v.at(0); break;
        case 262: values(e, v); photometric = v.at(0); break;
        case 273: values(e, offsets); break;
        case 277: values(e, v); spp = v.at(0); break;
        case 278: values(e, v); rows_per_strip = v.at(0); break;
        case 279: values(e, counts); break;
        case 284: values(e, v); planar = v.at(0); break;
        default: break;
        }
    }
    if (w == 0 || h == 0 || w > 65535 || h > 65535 || offsets.empty())
        throw std::runtime_error("TIFF without image dimensions or strips");
    if (compression != 1)
        throw std::runtime_error("compressed TIFF files are not supported (only uncompressed strips)");
    if (bits != 8 || planar != 1 || !((photometric <= 1 && spp >= 1) || (photometric == 2 && spp >= 3)))
        throw std::runtime_error("TIFF layout not supported (8-bit grey or RGB, chunky, is)");
    rows = (int)h, cols = (int)w;
    bgr.resize((size_t)w * h * 3);
    const size_t line = (size_t)w * spp;
    rows_per_strip = rows_per_strip < h ? rows_per_strip : h;
    for (uint32_t y = 0; y < h; ++y) {
        const size_t strip = y / rows_per_strip;
        if (strip >= offsets.size())
            throw std::runtime_error("TIFF strip table is too short");
        const size_t at = (size_t)offsets[strip] + (size_t)(y % rows_per_strip) * line;
        if (at + line > f.size())
            throw std::runtime_error("TIFF strip runs past the end of the file");
        const uint8_t *src = &f[at];
        uint8_t *out = &bgr[(size_t)y * w * 3];
        for (uint32_t x = 0; x < w; ++x) {
            if (photometric == 2) {
                out[3 * x] = src[x * spp + 2], out[3 * x + 1] = src[x * spp + 1], out[3 * x + 2] = src[x * spp];
            } else {
                const uint8_t v = photometric == 0 ? (uint8_t)(255 - src[x * spp]) : src[x * spp];
                out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = v;
            }
        }
    }
}

} // namespace tiffio
#endif
