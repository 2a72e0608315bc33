// jpegamd_analysis: the GPU leg of the reference's analysis.cpp on top of libjpeg_amd.so.
//
//   jpegamd_analysis <image_path> [quality] [--cpu-encode-ms X --cpu-decode-ms Y] [-o DIR]
//
// Same protocol as analysis.cpp: the image path and an optional quality on the command line
// (:10-37), the two FIXED table pairs instead of quality scaling (:43-90, through
// jpegamd_analysis_tables), output/compressed_image_gpu.bin and output/decompress_image_gpu.png
// without a quality suffix (:156, :259), and THIRTEEN bare numbers on stdout in the order of
// analysis.cpp:281-293, which metric_image_file.py:49-65 and metric_image_folder.py parse by
// position:
//    1 encode gain GPU   2 decode gain GPU   3 encode gain OMP   4 decode gain OMP
//    5 ratio CPU         6 ratio GPU         7 ratio OMP
//    8 MSE CPU  9 PSNR CPU   10 MSE GPU  11 PSNR GPU   12 MSE OMP  13 PSNR OMP
// This library has ONE back-end and no CPU path, so only the GPU members are measured here:
// 6, 10 and 11 always; 1 and 2 (reference CPU time / this GPU time) when the reference's CPU
// times for the same image are passed with --cpu-encode-ms / --cpu-decode-ms.  Every other
// slot prints "nan" (float() in the reference's scripts accepts it; the plots show a gap).
#include "harness_common.h"

int main(int argc, char *argv[])
{
    using std::cout;
    using std::endl;
    int quality = 50;
    std::string image_path, out_dir = "output";
    double cpu_encode_ms = NAN, cpu_decode_ms = NAN;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "--cpu-encode-ms" && i + 1 < argc)
            cpu_encode_ms = std::atof(argv[++i]);
        else if (arg == "--cpu-decode-ms" && i + 1 < argc)
            cpu_decode_ms = std::atof(argv[++i]);
        else if (arg == "-o" && i + 1 < argc)
            out_dir = argv[++i];
        else
            pos.push_back(arg);
    }
    if (pos.size() < 1 || pos.size() > 2) {
        std::cerr << "Usage: " << argv[0] << " <image_path> [quality]" << endl; // analysis.cpp:35
        return 1;
    }
    image_path = pos[0];
    if (pos.size() == 2) {
        try {
            quality = std::stoi(pos[1]);
            if (quality < 1 || quality > 100)
                throw std::out_of_range("Quality must be between 1 and 100");
        } catch (std::exception &e) {
            std::cerr << "Invalid quality argument: " << e.what() << endl;
            return 1;
        }
    }
    try {
        int32_t q_luma[64], q_chroma[64];
        check(jpegamd_analysis_tables(quality, q_luma, q_chroma));
        bool png = false;
        const Image image = read_image(image_path, &png);
        if (image.rows % 16 || image.cols % 16)
            throw std::runtime_error("image sides must be multiples of 16 (the reference resizes with OpenCV; "
                                     "this harness does not)");
        check(jpegamd_init(0));
        jpegamd_context *ctx = nullptr;
        check(jpegamd_context_create(0, &ctx));
        {
            // The timed regions are the reference's: mainEncode and mainDecode only (encode_decode.cpp:1310-1375).
            // One untimed call on the same image grows the context's device buffers to size and keeps the
            // first use of the device out of the timings, as the reference's CUDA context is after its CPU leg.
            uint8_t *warm = nullptr;
            size_t warm_len = 0;
            check(jpegamd_context_compress_image(ctx, image.bgr.data(), image.rows, image.cols, (size_t)image.cols * 3,
                                                 q_luma, q_chroma, &warm, &warm_len));
            jpegamd_free(warm);
        }
        uint8_t *bin = nullptr;
        size_t bin_len = 0;
        auto t0 = std::chrono::steady_clock::now();
        check(jpegamd_context_compress_image(ctx, image.bgr.data(), image.rows, image.cols, (size_t)image.cols * 3, q_luma,
                                             q_chroma, &bin, &bin_len));
        const double encode_ms = ms_since(t0);
        const std::string bin_name = out_dir + "/compressed_image_gpu.bin";
        {
            std::ofstream f(bin_name, std::ios::binary);
            if (!f)
                throw std::runtime_error("Cannot open file for writing: " + bin_name);
            f.write((const char *)bin, (std::streamsize)bin_len);
        }
        std::vector<uint8_t> file(bin, bin + bin_len);
        jpegamd_free(bin);
        const double ratio = (double)image.rows * image.cols * 3 / (double)file.size(); // calculateCompressionStats

        Image rec;
        // output sized from the header and its pages touched before the clock starts (the reference's Mats
        // exist before mainDecode), one untimed decode for the context's decode buffers; the second is timed
        check(jpegamd_context_decompress_image(ctx, file.data(), file.size(), q_luma, q_chroma, nullptr, 0, 0, &rec.rows,
                                               &rec.cols));
        rec.bgr.resize((size_t)rec.rows * rec.cols * 3);
        check(jpegamd_context_decompress_image(ctx, file.data(), file.size(), q_luma, q_chroma, rec.bgr.data(),
                                               rec.bgr.size(), (size_t)rec.cols * 3, &rec.rows, &rec.cols));
        t0 = std::chrono::steady_clock::now();
        check(jpegamd_context_decompress_image(ctx, file.data(), file.size(), q_luma, q_chroma, rec.bgr.data(),
                                               rec.bgr.size(), (size_t)rec.cols * 3, &rec.rows, &rec.cols));
        const double decode_ms = ms_since(t0);
        jpegamd_context_destroy(ctx);
        if (png)
            pngio::write(out_dir + "/decompress_image_gpu.png", rec.rows, rec.cols, rec.bgr.data());
        else
            write_ppm(out_dir + "/decompress_image_gpu.ppm", rec);
        const double mse = mean_squared_error(image, rec);
        const double psnr = mse == 0 ? INFINITY : 10.0 * std::log10((255 * 255) / mse);

        const double nan = NAN;
        cout << cpu_encode_ms / encode_ms << endl; // 1 (nan unless the reference's CPU time was given)
        cout << cpu_decode_ms / decode_ms << endl; // 2
        cout << nan << endl;                        // 3 OMP
        cout << nan << endl;                        // 4 OMP
        cout << nan << endl;                        // 5 CPU
        cout << ratio << endl;                      // 6
        cout << nan << endl;                        // 7 OMP
        cout << nan << endl;                        // 8 CPU
        cout << nan << endl;                        // 9 CPU
        cout << mse << endl;                        // 10
        cout << psnr << endl;                       // 11
        cout << nan << endl;                        // 12 OMP
        cout << nan << endl;                        // 13 OMP
        std::cerr << "jpegamd_analysis: encode " << encode_ms << " ms, decode " << decode_ms << " ms (host to host, second call through a warm context)" << endl;
    } catch (std::exception &e) {
        std::cerr << "Error: " << e.what() << endl;
        return 1;
    }
    return 0;
}
