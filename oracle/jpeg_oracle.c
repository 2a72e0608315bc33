/*
 * jpeg_oracle.c -- CPU restatement (plain C) of the Fulim13/jpeg block path.
 * TEST INFRASTRUCTURE ONLY -- see jpeg_oracle.h for the rules and the pins.
 *
 * Every function follows the arithmetic of the cited reference lines
 * operation for operation (operand types, evaluation order, where a float
 * accumulator is rounded).  Build WITHOUT -ffast-math / -march=native /
 * FMA contraction (oracle/Makefile passes -ffp-contract=off).
 *
 * The one liberty taken: cos((2x+1)*u*M_PI/16) is evaluated once per (x,u)
 * with the reference's exact argument expression and memoised; libm's cos is
 * a pure function, so the values are bit-identical to calling it per term.
 */
#include "jpeg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ tables */

/* main.cpp:2-10 */
static const int32_t base_luma[64] = {
    16, 11, 10, 16, 24, 40, 51, 61,
    12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77,
    24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101,
    72, 92, 95, 98, 112, 100, 103, 99};

/* main.cpp:12-20 */
static const int32_t base_chroma[64] = {
    17, 18, 24, 47, 99, 99, 99, 99,
    18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99};

/* analysis.cpp:45-54 (note the 12 at [5][5]) */
static const int32_t analysis_hi_luma[64] = {
    3, 2, 2, 3, 5, 8, 10, 12,
    2, 2, 3, 4, 5, 12, 12, 11,
    3, 3, 3, 5, 8, 11, 14, 11,
    3, 3, 4, 6, 10, 17, 16, 12,
    4, 4, 7, 11, 14, 22, 21, 15,
    5, 7, 11, 13, 16, 12, 23, 18,
    10, 13, 16, 17, 21, 24, 24, 21,
    14, 18, 19, 20, 22, 20, 20, 20};

/* analysis.cpp:56-65 */
static const int32_t analysis_hi_chroma[64] = {
    11, 12, 14, 19, 26, 58, 60, 55,
    12, 18, 21, 28, 32, 57, 59, 56,
    14, 21, 25, 30, 59, 59, 59, 59,
    19, 28, 30, 59, 59, 59, 59, 59,
    26, 32, 59, 59, 59, 59, 59, 59,
    58, 57, 59, 59, 59, 59, 59, 59,
    60, 59, 59, 59, 59, 59, 59, 59,
    55, 56, 59, 59, 59, 59, 59, 59};

void orc_quality_tables(int quality, int32_t q_luma[64], int32_t q_chroma[64])
{
    /* main.cpp:51-61 */
    int scaling_factor = (quality < 50) ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; ++i) {
        /* main.cpp:69, 78: max(1, min(255, (val*sf + 50) / 100)) */
        int a = (base_luma[i] * scaling_factor + 50) / 100;
        int b = (base_chroma[i] * scaling_factor + 50) / 100;
        a = a > 255 ? 255 : a;
        b = b > 255 ? 255 : b;
        q_luma[i] = a < 1 ? 1 : a;
        q_chroma[i] = b < 1 ? 1 : b;
    }
}

void orc_analysis_tables(int quality, int32_t q_luma[64], int32_t q_chroma[64])
{
    /* analysis.cpp:43-90: no scaling at all */
    memcpy(q_luma, quality >= 90 ? analysis_hi_luma : base_luma, 64 * sizeof(int32_t));
    memcpy(q_chroma, quality >= 90 ? analysis_hi_chroma : base_chroma, 64 * sizeof(int32_t));
}

/* ----------------------------------------------------------------- colour */

void orc_rgb2ycbcr(const uint8_t *src, int rows, int cols, uint8_t *dst)
{
    /* encode_decode.cpp:41-45 */
    const double offset[3] = {0.0, 128.0, 128.0};
    const double m[3][3] = {
        {0.299, 0.587, 0.114},
        {-0.1687, -0.3313, 0.5},
        {0.5, -0.4187, -0.0813}};
    size_t n = (size_t)rows * (size_t)cols;
    for (size_t p = 0; p < n; ++p) {
        const uint8_t *px = src + 3 * p;
        for (int k = 0; k < 3; ++k) {
            double t = offset[k];              /* :58 */
            for (int l = 0; l < 3; ++l)
                t += m[k][l] * px[l];          /* :61  uchar -> int -> double */
            t = round(t);                      /* :63 */
            /* :67-70 static_cast<uchar>(double): x86-64 truncates through int,
             * so 256.0 wraps to 0. */
            dst[3 * p + k] = (uint8_t)(int)t;
        }
    }
}

void orc_ycbcr2rgb(const uint8_t *src, int rows, int cols, uint8_t *dst)
{
    /* encode_decode.cpp:79-83 */
    const double offset[3] = {0.0, 128.0, 128.0};
    const double m[3][3] = {
        {1.0, 0.0, 1.402},
        {1.0, -0.344136, -0.714136},
        {1.0, 1.772, 0.0}};
    size_t n = (size_t)rows * (size_t)cols;
    for (size_t p = 0; p < n; ++p) {
        const uint8_t *px = src + 3 * p;
        for (int k = 0; k < 3; ++k) {
            double t = 0.0;                                   /* :96 */
            for (int l = 0; l < 3; ++l)
                t += m[k][l] * (px[l] - offset[l]);           /* :99 */
            t = round(t);                                     /* :102 */
            t = t > 255.0 ? 255.0 : t;
            t = t < 0.0 ? 0.0 : t;
            dst[3 * p + k] = (uint8_t)t;
        }
    }
}

void orc_chroma_subsample(const uint8_t *ycc, int rows, int cols,
                          uint8_t *Y, uint8_t *Cb, uint8_t *Cr)
{
    /* encode_decode.cpp:128-151 */
    int cw = cols / 2;
    for (int i = 0; i < rows; i += 2) {
        for (int j = 0; j < cols; j += 2) {
            const uint8_t *p00 = ycc + 3 * ((size_t)i * cols + j);
            const uint8_t *p10 = ycc + 3 * ((size_t)(i + 1) * cols + j);
            const uint8_t *p01 = p00 + 3;
            const uint8_t *p11 = p10 + 3;
            Y[(size_t)i * cols + j] = p00[0];
            Y[(size_t)(i + 1) * cols + j] = p10[0];
            Y[(size_t)i * cols + j + 1] = p01[0];
            Y[(size_t)(i + 1) * cols + j + 1] = p11[0];
            Cb[(size_t)(i / 2) * cw + j / 2] = (uint8_t)((p00[1] + p10[1] + p01[1] + p11[1]) / 4);
            Cr[(size_t)(i / 2) * cw + j / 2] = (uint8_t)((p00[2] + p10[2] + p01[2] + p11[2]) / 4);
        }
    }
}

void orc_upsample_chroma(const uint8_t *C, int crows, int ccols,
                         uint8_t *out, int width, int height)
{
    (void)crows;
    /* encode_decode.cpp:160-172 */
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j)
            out[(size_t)i * width + j] = C[(size_t)(i / 2) * ccols + j / 2];
}

/* ----------------------------------------------------------------- blocks */

size_t orc_num_blocks(int rows, int cols)
{
    return (size_t)((rows + 7) / 8) * (size_t)((cols + 7) / 8);
}

void orc_extract_blocks(const uint8_t *plane, int rows, int cols, int32_t *blocks)
{
    /* encode_decode.cpp:238-257 */
    size_t b = 0;
    for (int i = 0; i < rows; i += 8) {
        for (int j = 0; j < cols; j += 8) {
            int32_t *blk = blocks + 64 * b++;
            for (int x = 0; x < 8; ++x)
                for (int y = 0; y < 8; ++y)
                    blk[x * 8 + y] = (i + x < rows && j + y < cols)
                                         ? plane[(size_t)(i + x) * cols + j + y]
                                         : 0;
        }
    }
}

/* cos((2*x+1)*u*M_PI/16), argument formed exactly as encode_decode.cpp:278:
 * int product, times M_PI (double), divided by 16. */
static double cos_tab[8][8]; /* [x][u] */
static int cos_tab_ready = 0;

static void init_cos_tab(void)
{
    if (cos_tab_ready)
        return;
    for (int x = 0; x < 8; ++x)
        for (int u = 0; u < 8; ++u)
            cos_tab[x][u] = cos((2 * x + 1) * u * M_PI / 16);
    cos_tab_ready = 1;
}

void orc_dct_block(const int32_t in[64], float out[64])
{
    init_cos_tab();
    /* encode_decode.cpp:263-286 */
    for (int u = 0; u < 8; ++u) {
        for (int v = 0; v < 8; ++v) {
            float cu = (u == 0) ? (float)(1 / sqrt(2)) : (float)1.0; /* :267 */
            float cv = (v == 0) ? (float)(1 / sqrt(2)) : (float)1.0; /* :268 */
            float sum_val = 0;
            for (int x = 0; x < 8; ++x)
                for (int y = 0; y < 8; ++y) {
                    /* :277-279  int*double*double, float accumulator */
                    double term = in[x * 8 + y] * cos_tab[x][u] * cos_tab[y][v];
                    sum_val = (float)((double)sum_val + term);
                }
            /* :284  double expression, round(), /100 (int), narrowed to float */
            out[u * 8 + v] = (float)(round(0.25 * cu * cv * sum_val * 100) / 100);
        }
    }
}

void orc_quantize_block(float blk[64], const int32_t q[64])
{
    /* encode_decode.cpp:302-309: float / int -> float division, std::round(float) */
    for (int i = 0; i < 64; ++i) {
        float quantized_value = roundf(blk[i] / (float)q[i]);
        blk[i] = (quantized_value == -0.0f) ? 0.0f : quantized_value;
    }
}

/* Generated by walking encode_decode.cpp:317-347 once. */
const uint8_t orc_zigzag_order[64] = {
    0, 1, 8, 16, 9, 2, 3, 10,
    17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34,
    27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36,
    29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46,
    53, 60, 61, 54, 47, 55, 62, 63};

/* The same walk, executed (used by the self-check in the tests). */
static void zigzag_walk(uint8_t order[64])
{
    int n = 0;
    for (int i = 0; i < 15; ++i) {
        if (i < 8) {
            for (int j = 0; j <= i; ++j)
                order[n++] = (i % 2 == 0) ? (uint8_t)((i - j) * 8 + j) : (uint8_t)(j * 8 + (i - j));
        } else {
            for (int j = 0; j < 15 - i; ++j)
                order[n++] = (i % 2 == 0) ? (uint8_t)((7 - j) * 8 + (j + (i - 7)))
                                          : (uint8_t)((j + (i - 7)) * 8 + (7 - j));
        }
    }
}

int orc_zigzag_selfcheck(void)
{
    uint8_t o[64];
    zigzag_walk(o);
    return memcmp(o, orc_zigzag_order, 64) == 0;
}

void orc_zigzag_block(const float blk[64], int32_t zz[64])
{
    for (int i = 0; i < 64; ++i)
        zz[i] = (int)blk[orc_zigzag_order[i]]; /* :325 static_cast<int> */
}

int orc_rle_block(const int32_t zz[64], int32_t *out)
{
    /* encode_decode.cpp:352-372 */
    int n = 0;
    int run_length = 1;
    int prev = zz[0];
    for (int i = 1; i < 64; ++i) {
        if (zz[i] == prev) {
            run_length++;
        } else {
            out[n++] = prev;
            out[n++] = run_length;
            prev = zz[i];
            run_length = 1;
        }
    }
    out[n++] = prev;
    out[n++] = run_length;
    return n;
}

void orc_inverse_zigzag_block(const int32_t zz[64], float blk[64])
{
    for (int i = 0; i < 64; ++i)
        blk[orc_zigzag_order[i]] = (float)zz[i]; /* :533 */
}

void orc_inverse_quantize_block(float blk[64], const int32_t q[64])
{
    for (int i = 0; i < 64; ++i)
        blk[i] *= (float)q[i]; /* :564 float *= int */
}

void orc_idct_block(const float in[64], int32_t out[64])
{
    init_cos_tab();
    /* encode_decode.cpp:571-593 */
    const float inv_sqrt2 = (float)(1 / sqrt(2));
    for (int x = 0; x < 8; ++x) {
        for (int y = 0; y < 8; ++y) {
            float sum_val = 0.0f;
            for (int u = 0; u < 8; ++u)
                for (int v = 0; v < 8; ++v) {
                    float cu = (u == 0) ? inv_sqrt2 : 1.0f; /* :581 */
                    float cv = (v == 0) ? inv_sqrt2 : 1.0f; /* :582 */
                    /* :584-586  (cu*cv*c) in float, then * cos * cos in double */
                    float f = cu * cv * in[u * 8 + v];
                    double term = f * cos_tab[x][u] * cos_tab[y][v];
                    sum_val = (float)((double)sum_val + term);
                }
            out[x * 8 + y] = (int)round(0.25 * sum_val); /* :591 */
        }
    }
}

/* ---- OMP flavour (separable, float) ------------------------------------- */
static void init_omp_tab(void);

static float omp_tab[8][8]; /* initDCTTableOMP, encode_decode.cpp:823-834 */
static int omp_tab_ready = 0;

static void init_omp_tab(void)
{
    if (omp_tab_ready)
        return;
    for (int i = 0; i < 8; ++i) {
        float ci = (i == 0) ? 1.0f / sqrtf(2.0f) : 1.0f; /* :828 sqrt(float) overload */
        for (int j = 0; j < 8; ++j)
            omp_tab[i][j] = (float)(ci * cos((2.0 * j + 1.0) * i * M_PI / 16.0)); /* :831 */
    }
    omp_tab_ready = 1;
}

void orc_dct_quant_block_omp(const int32_t in[64], const int32_t q[64], float out[64])
{
    init_omp_tab();
    float tmp[64];
    /* performDCTOMP, encode_decode.cpp:855-880 */
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            float sum = 0.0f;
            for (int x = 0; x < 8; ++x)
                sum += in[u * 8 + x] * omp_tab[v][x]; /* :862 int*float -> float */
            tmp[u * 8 + v] = sum;
        }
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            float sum = 0.0f;
            for (int x = 0; x < 8; ++x)
                sum += tmp[x * 8 + v] * omp_tab[u][x]; /* :876 */
            out[u * 8 + v] = roundf(sum / 4.0f * 100) / 100; /* :878 */
        }
    /* quantizeBlockOMP :884-893 (no -0 normalisation) */
    for (int i = 0; i < 64; ++i)
        out[i] = roundf(out[i] / (float)q[i]);
}

void orc_dequant_idct_block_omp(const float coef[64], const int32_t q[64], int32_t out[64])
{
    float deq[64], tmp[64];
    for (int i = 0; i < 64; ++i)
        deq[i] = coef[i] * (float)q[i]; /* :1028 */
    /* inverseDCTOMP :981-1009.  (2.0f*j+1.0f)*k -> float, * M_PI -> double. */
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            float sum = 0.0f;
            for (int k = 0; k < 8; ++k) {
                float ck = (k == 0) ? 1.0f / sqrtf(2.0f) : 1.0f;
                sum = (float)((double)sum +
                              (double)(ck * deq[i * 8 + k]) *
                                  cos((2.0f * j + 1.0f) * k * M_PI / 16.0f)); /* :989 */
            }
            tmp[i * 8 + j] = sum;
        }
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            float sum = 0.0f;
            for (int k = 0; k < 8; ++k) {
                float ck = (k == 0) ? 1.0f / sqrtf(2.0f) : 1.0f;
                sum = (float)((double)sum +
                              (double)(ck * tmp[k * 8 + j]) *
                                  cos((2.0f * i + 1.0f) * k * M_PI / 16.0f)); /* :1005 */
            }
            out[i * 8 + j] = (int)roundf(sum / 4.0f) + 128; /* :1007 */
        }
}

/* Fill the memo tables up front (call before spreading blocks over threads). */
void orc_init(void)
{
    init_cos_tab();
    init_omp_tab();
}

/* ------------------------------------------------------------ plane level */

size_t orc_encode_plane(const uint8_t *plane, int rows, int cols, const int32_t q[64],
                        int flavour, int16_t *coef_zz, int32_t *symbols)
{
    size_t nb = orc_num_blocks(rows, cols);
    int bw = (cols + 7) / 8;
    size_t nsym = 0;
    for (size_t b = 0; b < nb; ++b) {
        int bi = (int)(b / bw) * 8, bj = (int)(b % bw) * 8;
        int32_t blk[64];
        float c[64];
        int32_t zz[64];
        for (int x = 0; x < 8; ++x)
            for (int y = 0; y < 8; ++y) {
                int v = (bi + x < rows && bj + y < cols) ? plane[(size_t)(bi + x) * cols + bj + y] : 0;
                blk[x * 8 + y] = v - 128; /* recenterAroundZero :289-298 */
            }
        if (flavour == 0) {
            orc_dct_block(blk, c);
            orc_quantize_block(c, q);
        } else {
            orc_dct_quant_block_omp(blk, q, c);
        }
        orc_zigzag_block(c, zz);
        if (coef_zz)
            for (int i = 0; i < 64; ++i)
                coef_zz[b * 64 + i] = (int16_t)zz[i];
        if (symbols)
            nsym += (size_t)orc_rle_block(zz, symbols + nsym);
    }
    return nsym;
}

void orc_decode_plane(const int16_t *coef_zz, int rows, int cols, const int32_t q[64],
                      int flavour, uint8_t *plane)
{
    size_t nb = orc_num_blocks(rows, cols);
    int bw = (cols + 7) / 8;
    memset(plane, 0, (size_t)rows * cols); /* Mat::zeros :690 */
    for (size_t b = 0; b < nb; ++b) {
        int bi = (int)(b / bw) * 8, bj = (int)(b % bw) * 8;
        int32_t zz[64], px[64];
        float c[64];
        for (int i = 0; i < 64; ++i)
            zz[i] = coef_zz[b * 64 + i];
        orc_inverse_zigzag_block(zz, c);
        if (flavour == 0) {
            orc_inverse_quantize_block(c, q);
            orc_idct_block(c, px);
            for (int i = 0; i < 64; ++i)
                px[i] += 128; /* add_back_128 :596-605 */
        } else {
            orc_dequant_idct_block_omp(c, q, px);
        }
        for (int x = 0; x < 8; ++x)
            for (int y = 0; y < 8; ++y)
                if (bi + x < rows && bj + y < cols) {
                    int v = px[x * 8 + y];
                    v = v < 0 ? 0 : (v > 255 ? 255 : v); /* std::clamp :705 */
                    plane[(size_t)(bi + x) * cols + bj + y] = (uint8_t)v;
                }
    }
}

size_t orc_decode_rle(const int32_t *symbols, size_t n_symbols, size_t nb, int16_t *coef_zz)
{
    /* encode_decode.cpp:490-509; stops once nb blocks are complete */
    size_t blocks = 0;
    int len = 0;
    for (size_t i = 0; i + 1 < n_symbols && blocks < nb; i += 2) {
        int value = symbols[i];
        int frequency = symbols[i + 1];
        for (int j = 0; j < frequency && blocks < nb; ++j) {
            coef_zz[blocks * 64 + len] = (int16_t)value;
            if (++len == 64) {
                len = 0;
                ++blocks;
            }
        }
    }
    return blocks;
}

double orc_mse(const uint8_t *a, const uint8_t *b, size_t n)
{
    /* encode_decode.cpp:1392-1398: |a-b| -> float, squared (float), summed per
     * channel in double by cv::sum; integer squares <= 65025 are exact in
     * float, so one double accumulator gives the same value. */
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) {
        int d = (int)a[i] - (int)b[i];
        s += (double)(d * d);
    }
    return s / (double)n;
}

double orc_psnr(double mse)
{
    if (mse == 0)
        return INFINITY;
    return 10.0 * log10((255 * 255) / mse); /* :1411 */
}
