// kernel_math_model.cpp -- host build of jpeg_amd/csrc/jpeg_math.h, the exact
// per-sample arithmetic the gfx950 kernels execute (same source, g++ instead
// of hipcc).  TEST INFRASTRUCTURE: lets the CPU-only suite check that
// arithmetic against the oracle (exhaustively for colour, the divisions and
// the rounding tricks) before anything runs on a GPU.  It is not a product
// path and nothing in jpeg_amd/ loads it.
#include "../../jpeg_amd/csrc/jpeg_math.h"

#include <cstddef>
#include <cstdint>
#include <cstring>

static const uint8_t kZigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

extern "C" {

// n interleaved 3-byte pixels -> interleaved Y,Cb,Cr
void jmm_fwd_colour(const uint8_t *src, size_t n, uint8_t *dst)
{
    for (size_t i = 0; i < n; ++i) {
        int y, cb, cr;
        jm_fwd_colour(src[3 * i], src[3 * i + 1], src[3 * i + 2], y, cb, cr);
        dst[3 * i] = (uint8_t)y;
        dst[3 * i + 1] = (uint8_t)cb;
        dst[3 * i + 2] = (uint8_t)cr;
    }
}

// the kernels' fast forward path: float chains, tie candidates redone exactly
void jmm_fwd_colour_fast(const uint8_t *src, size_t n, uint8_t *dst, long *n_candidates)
{
    long cand = 0;
    for (size_t i = 0; i < n; ++i) {
        float uy, cb, cr;
        JmFwdTies tie;
        const float p0 = (float)src[3 * i], p1 = (float)src[3 * i + 1], p2 = (float)src[3 * i + 2];
        jm_fwd_colour_fast(p0, p1, p2, uy, cb, cr, tie);
        if (jm_fwd_any_tie(tie)) {
            ++cand;
            jm_fwd_colour_fix(p0, p1, p2, tie, uy, cb, cr);
        }
        dst[3 * i] = (uint8_t)(jm_bits(uy) & 0xFFu);
        dst[3 * i + 1] = (uint8_t)((int)cb + 128); // centred; -128 is the wrapped 256
        dst[3 * i + 2] = (uint8_t)((int)cr + 128);
    }
    *n_candidates = cand;
}

void jmm_inv_colour(const uint8_t *src, size_t n, uint8_t *dst)
{
    for (size_t i = 0; i < n; ++i) {
        JmChromaTerm c = jm_inv_chroma(src[3 * i + 1], src[3 * i + 2]);
        int o0, o1, o2;
        jm_inv_colour(src[3 * i], src[3 * i + 1], src[3 * i + 2], c, o0, o1, o2);
        dst[3 * i] = (uint8_t)o0;
        dst[3 * i + 1] = (uint8_t)o1;
        dst[3 * i + 2] = (uint8_t)o2;
    }
}

// the kernels' row path: n groups of 4 pixels, each group = 4 Y bytes + two chroma
// samples (cb,cr for pixels 0-1 and for pixels 2-3) -> 12 bytes
void jmm_inv_rows(const uint8_t *y, const uint8_t *cbcr, size_t n, uint8_t *dst)
{
    for (size_t i = 0; i < n; ++i) {
        uint32_t y4, w[3];
        std::memcpy(&y4, y + 4 * i, 4);
        const int cba = cbcr[4 * i], cra = cbcr[4 * i + 1], cbb = cbcr[4 * i + 2], crb = cbcr[4 * i + 3];
        // the kernels' float chroma terms, bytes packed as the kernel reads them
        const uint32_t cbw = (uint32_t)cba | ((uint32_t)cbb << 8), crw = (uint32_t)cra | ((uint32_t)crb << 8);
        JmChromaPk a = jm_inv_chroma_pk_fast<0>(cbw, crw), b = jm_inv_chroma_pk_fast<1>(cbw, crw);
        jm_inv_row4(y4, a, b, cba, cra, cbb, crb, w[0], w[1], w[2]);
        std::memcpy(dst + 12 * i, w, 12);
    }
}

// nb blocks of 64 pixel values (0..255, not level shifted) -> quantised
// coefficients, natural order, exactly as the kernels' block stage does.
void jmm_fdct_quant_blocks(const int32_t *px, size_t nb, const int32_t *q, int32_t *out)
{
    JmQuantTab tab;
    jm_make_quant_tab(q, tab);
    for (size_t b = 0; b < nb; ++b) {
        float v[64];
        for (int i = 0; i < 64; ++i)
            v[i] = (float)px[b * 64 + i];
        for (int r = 0; r < 8; ++r) // rows: out index = horizontal frequency
            jm_fdct8<1, true>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3],
                              v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
        for (int c = 0; c < 8; ++c) // columns, scaled by 100
            jm_fdct8<100, false>(v[0 * 8 + c], v[1 * 8 + c], v[2 * 8 + c], v[3 * 8 + c],
                                 v[4 * 8 + c], v[5 * 8 + c], v[6 * 8 + c], v[7 * 8 + c]);
        jm_quantise_block(v, tab);
        for (int i = 0; i < 64; ++i)
            out[b * 64 + i] = jm_magic_to_int(v[i]);
    }
}

// nb blocks of quantised coefficients (natural order) -> pixels + 128, unclamped
void jmm_dequant_idct_blocks(const int32_t *coef, size_t nb, const int32_t *q, int32_t *out)
{
    for (size_t b = 0; b < nb; ++b) {
        float v[64];
        for (int i = 0; i < 64; ++i)
            v[i] = (float)coef[b * 64 + i] * (float)q[i];
        for (int c = 0; c < 8; ++c)
            jm_idct8<false>(v[0 * 8 + c], v[1 * 8 + c], v[2 * 8 + c], v[3 * 8 + c],
                     v[4 * 8 + c], v[5 * 8 + c], v[6 * 8 + c], v[7 * 8 + c]);
        for (int r = 0; r < 8; ++r)
            jm_idct8<true>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3],
                     v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
        for (int i = 0; i < 64; ++i)
            out[b * 64 + i] = jm_magic_to_int(jm_pixel_magic_preshrunk(v[i]));
    }
}

// all 65536 (cb, cr): the float chroma terms equal the exact ones wherever they are
// not flagged, and every exact tie is flagged.  Returns #failures; *n_flagged counts
// the candidates.
long jmm_check_inv_chroma_fast(long *n_flagged)
{
    long bad = 0, fl = 0;
    for (int cb = 0; cb < 256; ++cb)
        for (int cr = 0; cr < 256; ++cr) {
            const JmChromaPk e = jm_inv_chroma_pk(cb, cr);
            const JmChromaPk f = jm_inv_chroma_pk_fast<0>((uint32_t)cb, (uint32_t)cr);
            const JmChromaPk g = jm_inv_chroma_pk_fast<1>((uint32_t)cb << 8, (uint32_t)cr << 8);
            if (f.tie < 0)
                ++fl;
            else
                bad += f.d01 != e.d01 || f.d20 != e.d20 || f.d12 != e.d12;
            bad += e.tie < 0 && f.tie >= 0;
            bad += f.d01 != g.d01 || f.d20 != g.d20 || f.d12 != g.d12 || (f.tie < 0) != (g.tie < 0);
        }
    *n_flagged = fl;
    return bad;
}

// the matrix-core transform order (jm_mfma_*): same contracts as the two above
void jmm_mfma_fdct_quant_blocks(const int32_t *px, size_t nb, const int32_t *q, int32_t *out)
{
    JmQuantTab tab;
    jm_make_quant_tab(q, tab);
    for (size_t b = 0; b < nb; ++b) {
        float v[64], c[64];
        for (int i = 0; i < 64; ++i)
            v[i] = (float)px[b * 64 + i];
        jm_mfma_fdct_block(v, c);
        jm_quantise_block(c, tab);
        for (int i = 0; i < 64; ++i)
            out[b * 64 + i] = jm_magic_to_int(c[i]);
    }
}

void jmm_mfma_dequant_idct_blocks(const int32_t *coef, size_t nb, const int32_t *q, int32_t *out)
{
    for (size_t b = 0; b < nb; ++b) {
        float w[64], x[64];
        for (int i = 0; i < 64; ++i)
            w[i] = (float)coef[b * 64 + i] * (float)q[i];
        jm_mfma_idct_block(w, x);
        for (int i = 0; i < 64; ++i)
            out[b * 64 + i] = jm_magic_to_int(jm_pixel_magic(x[i]));
    }
}

float jmm_dct_basis(int u, int n, double scale) { return jm_dct_basis(u, n, scale); }

const uint8_t *jmm_zigzag(void) { return kZigzag; }

// every possible quad sum 0..1020 through the kernels' truncating-mean helper
long jmm_check_quad_mean(void)
{
    long bad = 0;
    for (int s = 0; s <= 1020; ++s) // the kernels sum centred values: s - 512
        bad += (int)(jm_bits(jm_quad_mean_magic((float)(s - 512))) & 0xFFu) != s / 4;
    return bad;
}

// ---- exhaustive self-checks of the arithmetic shortcuts; return #failures --

// jm_div1000 / jm_div10000 over their whole input ranges
long jmm_check_const_div(void)
{
    long bad = 0;
    for (uint32_t n = 0; n < (1u << 18); ++n)
        bad += jm_div1000(n) != n / 1000u;
    for (uint32_t n = 0; n < (1u << 22); ++n)
        bad += jm_div10000(n) != n / 10000u;
    return bad;
}

// jm_round_half_away == roundf on every float with |v| < 2^22 (+ both zeros)
long jmm_check_round(void)
{
    long bad = 0;
    for (uint32_t bits = 0; bits < 0x4A800000u; ++bits) { // 0 .. 2^22
        float v, w;
        std::memcpy(&v, &bits, 4);
        w = -v;
        float a = jm_round_half_away(v), b = roundf(v);
        float c = jm_round_half_away(w), d = roundf(w);
        bad += std::memcmp(&a, &b, 4) != 0;
        bad += std::memcmp(&c, &d, 4) != 0;
    }
    return bad;
}

// the snap + quantise chain against real divisions, every r in [-rmax, rmax]
// and every Q in 1..255: c = (float)(r/100.0) (double division narrowed, as
// encode_decode.cpp:284), v = c / (float)Q, q = roundf(v).
long jmm_check_quant(int rmax)
{
    long bad = 0;
    for (int Q = 1; Q <= 255; ++Q) {
        float qf = (float)Q, rq = 1.0f / qf;
        int32_t qq[64];
        for (int i = 0; i < 64; ++i)
            qq[i] = Q;
        JmQuantTab tab;
        jm_make_quant_tab(qq, tab);
        for (int r = -rmax; r <= rmax; ++r) {
            float c = (float)((double)r / 100);
            volatile float v = c / qf;
            float ref = roundf(v);
            int want = (int)((ref == -0.0f) ? 0.0f : ref);
            // feed the integer r itself and the edges of its rounding interval
            int got = jm_snap_quant((float)r, qf, rq);
            bad += got != want;
            // the kernels' quantiser: one fma with the biased multiplier
            float u = fmaf((float)r, tab.inv100q[0], JM_MAGIC);
            bad += jm_magic_to_int(u) != want;
        }
    }
    return bad;
}

} // extern "C"
