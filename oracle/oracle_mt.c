/*
 * oracle_mt.c -- whole-image drivers over jpeg_oracle.c, used by the tests as
 * the end-to-end checker and by bench.py as the timed CPU baseline
 * ("cpu_baseline", kind "port").  TEST INFRASTRUCTURE ONLY.
 *
 * The per-block arithmetic is untouched; `threads` only spreads independent
 * blocks over OpenMP threads (as encode_decode.cpp:911 / 1019 do for the OMP
 * flavour), so results do not depend on the thread count.
 */
#include "jpeg_oracle.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

static void encode_plane_mt(const uint8_t *plane, int rows, int cols, const int32_t q[64],
                            int flavour, int threads, int16_t *coef_zz)
{
    int bw = (cols + 7) / 8, bh = (rows + 7) / 8;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int br = 0; br < bh; ++br) {
        for (int bc = 0; bc < bw; ++bc) {
            int32_t blk[64], zz[64];
            float c[64];
            for (int x = 0; x < 8; ++x)
                for (int y = 0; y < 8; ++y) {
                    int r = br * 8 + x, cidx = bc * 8 + y;
                    int v = (r < rows && cidx < cols) ? plane[(size_t)r * cols + cidx] : 0;
                    blk[x * 8 + y] = v - 128;
                }
            if (flavour == 0) {
                orc_dct_block(blk, c);
                orc_quantize_block(c, q);
            } else {
                orc_dct_quant_block_omp(blk, q, c);
            }
            orc_zigzag_block(c, zz);
            int16_t *o = coef_zz + ((size_t)br * bw + bc) * 64;
            for (int i = 0; i < 64; ++i)
                o[i] = (int16_t)zz[i];
        }
    }
}

static void decode_plane_mt(const int16_t *coef_zz, int rows, int cols, const int32_t q[64],
                            int flavour, int threads, uint8_t *plane)
{
    int bw = (cols + 7) / 8, bh = (rows + 7) / 8;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int br = 0; br < bh; ++br) {
        for (int bc = 0; bc < bw; ++bc) {
            const int16_t *in = coef_zz + ((size_t)br * bw + bc) * 64;
            int32_t zz[64], px[64];
            float c[64];
            for (int i = 0; i < 64; ++i)
                zz[i] = in[i];
            orc_inverse_zigzag_block(zz, c);
            if (flavour == 0) {
                orc_inverse_quantize_block(c, q);
                orc_idct_block(c, px);
                for (int i = 0; i < 64; ++i)
                    px[i] += 128;
            } else {
                orc_dequant_idct_block_omp(c, q, px);
            }
            for (int x = 0; x < 8; ++x)
                for (int y = 0; y < 8; ++y) {
                    int r = br * 8 + x, cidx = bc * 8 + y;
                    if (r < rows && cidx < cols) {
                        int v = px[x * 8 + y];
                        v = v < 0 ? 0 : (v > 255 ? 255 : v);
                        plane[(size_t)r * cols + cidx] = (uint8_t)v;
                    }
                }
        }
    }
}

/* BGR image (rows, cols multiples of 16) -> zigzag int16 coefficient planes.
 * RGB2YCbCr -> chromaSubsampling -> encodeChannel x3 (without the entropy coder).
 * If ycc_planes != NULL it receives Y | Cb | Cr (rows*cols + 2*(rows/2*cols/2) bytes). */
int orc_encode_image(const uint8_t *bgr, int rows, int cols,
                     const int32_t qY[64], const int32_t qC[64], int flavour, int threads,
                     int16_t *coefY, int16_t *coefCb, int16_t *coefCr, uint8_t *ycc_planes)
{
    if (rows % 2 || cols % 2)
        return 1;
    orc_init();
    size_t n = (size_t)rows * cols, nc = (size_t)(rows / 2) * (cols / 2);
    uint8_t *ycc = (uint8_t *)malloc(3 * n);
    uint8_t *planes = ycc_planes ? ycc_planes : (uint8_t *)malloc(n + 2 * nc);
    if (!ycc || !planes)
        return 2;
    orc_rgb2ycbcr(bgr, rows, cols, ycc);
    orc_chroma_subsample(ycc, rows, cols, planes, planes + n, planes + n + nc);
    free(ycc);
    encode_plane_mt(planes, rows, cols, qY, flavour, threads, coefY);
    encode_plane_mt(planes + n, rows / 2, cols / 2, qC, flavour, threads, coefCb);
    encode_plane_mt(planes + n + nc, rows / 2, cols / 2, qC, flavour, threads, coefCr);
    if (!ycc_planes)
        free(planes);
    return 0;
}

/* inverse: decodeChannel x3 -> upsampleChroma -> merge -> YCbCr2RGB */
int orc_decode_image(const int16_t *coefY, const int16_t *coefCb, const int16_t *coefCr,
                     int rows, int cols, const int32_t qY[64], const int32_t qC[64],
                     int flavour, int threads, uint8_t *bgr)
{
    if (rows % 2 || cols % 2)
        return 1;
    orc_init();
    size_t n = (size_t)rows * cols, nc = (size_t)(rows / 2) * (cols / 2);
    uint8_t *Y = (uint8_t *)malloc(n), *Cb = (uint8_t *)malloc(nc), *Cr = (uint8_t *)malloc(nc);
    uint8_t *up = (uint8_t *)malloc(2 * n), *ycc = (uint8_t *)malloc(3 * n);
    if (!Y || !Cb || !Cr || !up || !ycc)
        return 2;
    decode_plane_mt(coefY, rows, cols, qY, flavour, threads, Y);
    decode_plane_mt(coefCb, rows / 2, cols / 2, qC, flavour, threads, Cb);
    decode_plane_mt(coefCr, rows / 2, cols / 2, qC, flavour, threads, Cr);
    orc_upsample_chroma(Cb, rows / 2, cols / 2, up, cols, rows);
    orc_upsample_chroma(Cr, rows / 2, cols / 2, up + n, cols, rows);
    for (size_t p = 0; p < n; ++p) { /* cv::merge main.cpp:283-289 */
        ycc[3 * p] = Y[p];
        ycc[3 * p + 1] = up[p];
        ycc[3 * p + 2] = up[n + p];
    }
    orc_ycbcr2rgb(ycc, rows, cols, bgr);
    free(Y), free(Cb), free(Cr), free(up), free(ycc);
    return 0;
}

int orc_max_threads(void) { return omp_get_max_threads(); }
