// Test driver for jpeg_amd/csrc/host/png_io.h (the command-line harness's PNG reader and
// writer): `png_io_driver decode in.png out.raw` writes rows, cols (two little-endian int32)
// and the B,G,R bytes; `png_io_driver encode in.raw out.png` does the opposite.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <iterator>

#include "../../jpeg_amd/csrc/host/png_io.h"

int main(int argc, char **argv)
{
    if (argc != 4)
        return 2;
    try {
        std::ifstream in(argv[2], std::ios::binary);
        std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (std::string(argv[1]) == "decode") {
            int rows, cols;
            std::vector<uint8_t> bgr;
            if (tiffio::is_tiff(file))
                tiffio::decode(file, rows, cols, bgr);
            else
                pngio::decode(file, rows, cols, bgr);
            std::ofstream out(argv[3], std::ios::binary);
            const int32_t dims[2] = {rows, cols};
            out.write((const char *)dims, 8);
            out.write((const char *)bgr.data(), (std::streamsize)bgr.size());
        } else {
            int32_t dims[2];
            std::memcpy(dims, file.data(), 8);
            pngio::write(argv[3], dims[0], dims[1], file.data() + 8);
        }
    } catch (std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
