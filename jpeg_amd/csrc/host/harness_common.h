// harness_common.h -- what the two command-line harnesses (jpegamd_main, jpegamd_analysis) share:
// image files in and out without OpenCV (png_io.h: PNG and uncompressed TIFF on zlib; binary
// PPM), calculateMSE, timing.
#ifndef JPEGAMD_HARNESS_COMMON_H
#define JPEGAMD_HARNESS_COMMON_H
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" {
#include "../../../include/jpeg_amd.h"
}
#include "png_io.h"

namespace {

struct Image {
    int rows = 0, cols = 0;
    std::vector<uint8_t> bgr; // interleaved B,G,R like cv::Mat CV_8UC3
};

int ppm_token(std::istream &in)
{
    // header integers, skipping whitespace and '#' comments
    for (;;) {
        int c = in.peek();
        if (c == '#') {
            std::string skip;
            std::getline(in, skip);
        } else if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
            in.get();
        } else
            break;
    }
    int v = -1;
    in >> v;
    if (!in || v < 0)
        throw std::runtime_error("malformed PPM header");
    return v;
}

Image read_ppm(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("Could not open or find the image"); // readImage, encode_decode.cpp:23
    char magic[2] = {0, 0};
    in.read(magic, 2);
    if (magic[0] != 'P' || magic[1] != '6')
        throw std::runtime_error("not a binary PPM (P6): " + path);
    Image img;
    img.cols = ppm_token(in);
    img.rows = ppm_token(in);
    const int maxval = ppm_token(in);
    in.get(); // the single whitespace byte after maxval
    if (maxval != 255 || img.rows <= 0 || img.cols <= 0)
        throw std::runtime_error("PPM must have maxval 255 and positive size: " + path);
    const size_t n = (size_t)img.rows * img.cols;
    std::vector<uint8_t> rgb(n * 3);
    in.read((char *)rgb.data(), (std::streamsize)rgb.size());
    if ((size_t)in.gcount() != rgb.size())
        throw std::runtime_error("PPM pixel data is truncated: " + path);
    img.bgr.resize(n * 3);
    for (size_t i = 0; i < n; ++i) {
        img.bgr[3 * i + 0] = rgb[3 * i + 2];
        img.bgr[3 * i + 1] = rgb[3 * i + 1];
        img.bgr[3 * i + 2] = rgb[3 * i + 0];
    }
    return img;
}

// PNG or TIFF by their signatures, otherwise PPM; *png: write the reconstruction as PNG
Image read_image(const std::string &path, bool *png)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("Could not open or find the image"); // readImage, encode_decode.cpp:23
    uint8_t sig[8] = {0};
    in.read((char *)sig, 8);
    const std::vector<uint8_t> head(sig, sig + (in.gcount() == 8 ? 8 : 0));
    const bool tiff = tiffio::is_tiff(head);
    *png = pngio::is_png(head) || tiff; // either way the reconstruction is written as PNG, like main.cpp:300
    if (!*png)
        return read_ppm(path);
    in.clear();
    in.seekg(0, std::ios::end);
    std::vector<uint8_t> file((size_t)in.tellg());
    in.seekg(0);
    in.read((char *)file.data(), (std::streamsize)file.size());
    Image img;
    if (tiff)
        tiffio::decode(file, img.rows, img.cols, img.bgr);
    else
        pngio::decode(file, img.rows, img.cols, img.bgr);
    return img;
}

void write_ppm(const std::string &path, const Image &img)
{
    std::ofstream out(path, std::ios::binary);
    if (!out)
        throw std::runtime_error("Cannot open file for writing: " + path);
    out << "P6\n" << img.cols << " " << img.rows << "\n255\n";
    const size_t n = (size_t)img.rows * img.cols;
    std::vector<uint8_t> rgb(n * 3);
    for (size_t i = 0; i < n; ++i) {
        rgb[3 * i + 0] = img.bgr[3 * i + 2];
        rgb[3 * i + 1] = img.bgr[3 * i + 1];
        rgb[3 * i + 2] = img.bgr[3 * i + 0];
    }
    out.write((const char *)rgb.data(), (std::streamsize)rgb.size());
}

// calculateMSE (encode_decode.cpp:1386-1401): squared differences as float, summed
// per channel in double, over total()*channels().
double mean_squared_error(const Image &a, const Image &b)
{
    double sum[3] = {0, 0, 0};
    const size_t n = (size_t)a.rows * a.cols;
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            const float d = (float)std::abs((int)a.bgr[3 * i + c] - (int)b.bgr[3 * i + c]);
            sum[c] += (double)(d * d);
        }
    return (sum[0] + sum[1] + sum[2]) / ((double)n * 3.0);
}

void check(int rc)
{
    if (rc)
        throw std::runtime_error(jpegamd_last_error());
}

double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace
#endif
