// jpegamd_main: the GPU branch of the reference's command-line harness
// (main.cpp:22-337) on top of libjpeg_amd.so.
//
//   echo image.png | jpegamd_main [-q N] [-o DIR]
//
// Same protocol as main.cpp: `-q N` (1..100, default 50, :24-47), the image
// name is read from stdin after the "Enter the image name: " prompt (:84-85),
// the compressed file goes to DIR/compressed_image_gpu_<q>.bin (:166) and the
// report lines carry the reference's wording for its GPU leg (:49, :93, :123,
// :150, :171, :191-194, :244, :264, :328-330).
//
// Differences, on purpose:
//  * no OpenCV in this image: the input is a PNG or an uncompressed 8-bit TIFF (png_io.h:
//    zlib only, what imread's 8-bit colour mode makes of the kinds of file the reference
//    ships) or a binary PPM (P6, maxval 255); the reconstruction is written as
//    DIR/decompress_image_gpu_<q>.png (main.cpp:300), for a PPM input as .ppm;
//  * ensureMultipleOf16 (encode_decode.cpp:29-37) resizes with cv::INTER_AREA;
//    here an image whose sides are not multiples of 16 is rejected;
//  * only the GPU leg runs, so there are no CPU/OMP lines and no speed-up lines;
//  * "Original size" is rows*cols*3 as in calculateCompressionStats
//    (encode_decode.cpp:185-231), taken from the image in memory.
#include "harness_common.h"

int main(int argc, char *argv[])
{
    using std::cout;
    using std::endl;
    int quality = 50;
    std::string out_dir = "output";
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "-q" && i + 1 < argc) {
            try {
                quality = std::stoi(argv[++i]);
                if (quality < 1 || quality > 100)
                    throw std::out_of_range("Quality must be between 1 and 100");
            } catch (std::exception &e) {
                std::cerr << "Invalid quality argument: " << e.what() << endl;
                return 1;
            }
        } else if (arg == "-o" && i + 1 < argc) {
            out_dir = argv[++i];
        }
    }
    cout << "Using quality: " << quality << "%" << endl;

    try {
        int32_t q_luma[64], q_chroma[64];
        check(jpegamd_quality_tables(quality, q_luma, q_chroma));

        std::string image_name;
        cout << "Enter the image name: ";
        std::cin >> image_name;
        bool png = false;
        const Image image = read_image(image_name, &png);
        if (image.rows % 16 || image.cols % 16)
            throw std::runtime_error("image sides must be multiples of 16 (the reference resizes with OpenCV; "
                                     "this harness does not)");
        cout << "Image Dimensions: " << image.rows << " x " << image.cols << endl;
        cout << "====================================================\n";

        check(jpegamd_init(0));
        // The timed regions are the reference's: mainEncode and mainDecode only (encode_decode.cpp:1310-1375).
        // Device buffers live in a context that one untimed call on the same image has grown to size
        // (the first use of the device -- code object load, the context's arena -- stays out of the
        // timings, as the reference's CUDA context does after its CPU leg has run).
        jpegamd_context *ctx = nullptr;
        check(jpegamd_context_create(0, &ctx));
        {
            uint8_t *warm = nullptr;
            size_t warm_len = 0;
            check(jpegamd_context_compress_image(ctx, image.bgr.data(), image.rows, image.cols, (size_t)image.cols * 3,
                                                 q_luma, q_chroma, &warm, &warm_len));
            jpegamd_free(warm);
        }

        cout << "Compressed with GPU" << endl;
        cout << "======================================\n";
        uint8_t *bin = nullptr;
        size_t bin_len = 0;
        auto t0 = std::chrono::steady_clock::now();
        check(jpegamd_context_compress_image(ctx, image.bgr.data(), image.rows, image.cols, (size_t)image.cols * 3, q_luma,
                                             q_chroma, &bin, &bin_len));
        const double encode_ms = ms_since(t0);
        cout << "Encoding time (GPU): " << encode_ms << " ms" << endl;
        cout << "======================================\n\n";

        const std::string bin_name = out_dir + "/compressed_image_gpu_" + std::to_string(quality) + ".bin";
        {
            std::ofstream f(bin_name, std::ios::binary);
            if (!f)
                throw std::runtime_error("Cannot open file for writing: " + bin_name);
            f.write((const char *)bin, (std::streamsize)bin_len);
        }
        jpegamd_free(bin);
        cout << "Compressed file (GPU) saved in: " << bin_name << endl;

        // calculateCompressionStats: sizes from the image and from the file on disk
        std::ifstream f(bin_name, std::ios::binary | std::ios::ate);
        if (!f)
            throw std::runtime_error("Cannot open the compressed file: " + bin_name);
        const size_t compressed_size = (size_t)f.tellg();
        const size_t original_size = (size_t)image.rows * image.cols * 3;
        cout << "Original size (GPU): " << original_size << " bytes (" << original_size / 1024.0 << " KB)" << endl;
        cout << "Compressed size (GPU): " << compressed_size << " bytes (" << compressed_size / 1024.0 << " KB)"
             << endl;
        cout << "Compression ratio (GPU): " << (compressed_size ? (double)original_size / compressed_size : 0.0)
             << endl;
        cout << "Percentage reduction (GPU): "
             << (compressed_size ? (1.0 - (double)compressed_size / original_size) * 100 : 0.0) << "%" << endl;
        cout << "======================================\n\n";

        // back from the file, like loadEncodedData + mainDecode + upsampleChroma + YCbCr2RGB
        std::vector<uint8_t> file(compressed_size);
        f.seekg(0);
        f.read((char *)file.data(), (std::streamsize)file.size());
        cout << "Decompressed with GPU" << endl;
        cout << "======================================\n";
        Image rec;
        // the output image is sized from the file's header and its pages are touched before the clock
        // starts (the reference's Mats exist before mainDecode too), then one untimed decode warms the
        // context's decode buffers; the timed call is the second
        check(jpegamd_context_decompress_image(ctx, file.data(), file.size(), q_luma, q_chroma, nullptr, 0, 0, &rec.rows,
                                               &rec.cols));
        rec.bgr.resize((size_t)rec.rows * rec.cols * 3);
        check(jpegamd_context_decompress_image(ctx, file.data(), file.size(), q_luma, q_chroma, rec.bgr.data(), rec.bgr.size(),
                                               (size_t)rec.cols * 3, &rec.rows, &rec.cols));
        t0 = std::chrono::steady_clock::now();
        check(jpegamd_context_decompress_image(ctx, file.data(), file.size(), q_luma, q_chroma, rec.bgr.data(), rec.bgr.size(),
                                               (size_t)rec.cols * 3, &rec.rows, &rec.cols));
        const double decode_ms = ms_since(t0);
        jpegamd_context_destroy(ctx);
        cout << "Decoding time (GPU): " << decode_ms << " ms" << endl;
        cout << "======================================\n\n";

        const std::string rec_name = out_dir + "/decompress_image_gpu_" + std::to_string(quality);
        if (png)
            pngio::write(rec_name + ".png", rec.rows, rec.cols, rec.bgr.data());
        else
            write_ppm(rec_name + ".ppm", rec);

        if (rec.rows != image.rows || rec.cols != image.cols)
            throw std::runtime_error("Images must have the same size and type for MSE calculation.");
        const double mse = mean_squared_error(image, rec);
        const double psnr = mse == 0 ? INFINITY : 10.0 * std::log10((255 * 255) / mse);
        cout << "Image Metrics (GPU):" << endl;
        cout << "  MSE: " << mse << endl;
        cout << "  PSNR: " << psnr << " dB" << endl;
        cout << "======================================\n\n";
    } catch (std::exception &e) {
        std::cerr << "Error: " << e.what() << endl;
        return 1;
    }
    return 0;
}
