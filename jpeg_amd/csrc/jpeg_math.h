// jpeg_math.h -- the per-sample arithmetic of the gfx950 kernels, written so
// that the same source compiles as device code (hipcc) and as plain host C++
// (g++, for tests/kernel_math_model.cpp, which checks it against the oracle
// exhaustively on the CPU).  Compile with -ffp-contract=off: every fused
// multiply-add below is an explicit jm_fma().
//
// What each piece stands in for (reference = /root/reference):
//   jm_fwd_colour   RGB2YCbCr            encode_decode.cpp:39-75
//   jm_inv_*        YCbCr2RGB            encode_decode.cpp:77-114
//   jm_fdct8 x2     dct                  encode_decode.cpp:260-287
//   jm_snap_quant   dct's 2-decimal snap (284) + quantize (300-310)
//   jm_idct8 x2     idct                 encode_decode.cpp:569-594
//   jm_round_pixel  round(0.25*sum) (591), tie handling see below
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define JM_HD __host__ __device__ __forceinline__
#else
#define JM_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define jm_fma(a, b, c) __builtin_fmaf((a), (b), (c))
#define jm_trunc(a) __builtin_truncf(a)
#define jm_rint(a) __builtin_rintf(a)
#define jm_copysign(a, b) __builtin_copysignf((a), (b))
#else
#include <math.h>
#define jm_fma(a, b, c) fmaf((a), (b), (c))
#define jm_trunc(a) truncf(a)
#define jm_rint(a) rintf(a)
#define jm_copysign(a, b) copysignf((a), (b))
#endif

// ---------------------------------------------------------------------------
// Exact roundf() (half away from zero) for |v| < 2^22: adding the largest
// float below 0.5 never carries a value that is below k+0.5 up to k+1, and
// always carries k+0.5 itself.  Checked exhaustively in the model test.
JM_HD float jm_round_half_away(float v)
{
    return jm_trunc(v + jm_copysign(0x1.fffffep-2f, v));
}

// ------------------------------------------------------------ forward colour
// The reference evaluates  t = off + m0*p0 + m1*p1 + m2*p2  in double, one
// rounding per operation, then round() half away.  With p integers the exact
// value is a rational with denominator 1000 (Y) or 10000 (Cb, Cr), so unless
// it is an exact x.5 tie the nearest integer is decided 1e-4 away from any
// double rounding error (~1e-13): integer arithmetic gives the same answer.
// Exact ties (about 1 pixel in 1000 for Y) depend on the double rounding
// errors and are re-evaluated with the reference's own operation sequence.

JM_HD double jm_round_half_away_f64(double t)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_trunc(t);
    double d = t - r; // exact
    if (__builtin_fabs(d) >= 0.5)
        r += __builtin_copysign(1.0, t);
    return r;
#else
    return round(t);
#endif
}

// one channel, the reference's operation order (no contraction!)
JM_HD int jm_fwd_colour_f64(int k, int p0, int p1, int p2)
{
    const double off = (k == 0) ? 0.0 : 128.0;
    const double m0 = (k == 0) ? 0.299 : (k == 1) ? -0.1687 : 0.5;
    const double m1 = (k == 0) ? 0.587 : (k == 1) ? -0.3313 : -0.4187;
    const double m2 = (k == 0) ? 0.114 : (k == 1) ? 0.5 : -0.0813;
    double t = off;
    t = t + m0 * (double)p0;
    t = t + m1 * (double)p1;
    t = t + m2 * (double)p2;
    t = jm_round_half_away_f64(t);
    return (int)t & 0xFF; // static_cast<uchar>: 256 wraps to 0
}

// floor(n / 1000) for 0 <= n < 2^18 and floor(n / 10000) for 0 <= n < 2^22,
// through one float fma (n is exact in float; the bias keeps exact multiples
// on the right side of the truncation).  Checked over the full ranges.
JM_HD uint32_t jm_div1000(uint32_t n)
{
    return (uint32_t)jm_fma((float)n, 1.0e-3f, 1.0f / 4096.0f);
}
JM_HD uint32_t jm_div10000(uint32_t n)
{
    return (uint32_t)jm_fma((float)n, 1.0e-4f, 3.0f / 65536.0f);
}

// p0,p1,p2 = the stored channel order (cv::imread's B,G,R; the reference
// applies its "R,G,B" matrix to them as they come).
JM_HD void jm_fwd_colour(int p0, int p1, int p2, int &y, int &cb, int &cr)
{
    uint32_t ny = (uint32_t)(299 * p0 + 587 * p1 + 114 * p2 + 500);
    uint32_t nb = (uint32_t)(1285000 - 1687 * p0 - 3313 * p1 + 5000 * p2);
    uint32_t nr = (uint32_t)(1285000 + 5000 * p0 - 4187 * p1 - 813 * p2);
    uint32_t qy = jm_div1000(ny), qb = jm_div10000(nb), qr = jm_div10000(nr);
    uint32_t ry = ny - qy * 1000u, rb = nb - qb * 10000u, rr = nr - qr * 10000u;
    y = (int)qy;
    cb = (int)(qb & 0xFFu);
    cr = (int)(qr & 0xFFu);
    uint32_t rmin = ry < rb ? ry : rb;
    rmin = rmin < rr ? rmin : rr;
    if (__builtin_expect(rmin == 0u, 0)) { // an exact .5 somewhere: ask the doubles
        if (ry == 0u)
            y = jm_fwd_colour_f64(0, p0, p1, p2);
        if (rb == 0u)
            cb = jm_fwd_colour_f64(1, p0, p1, p2);
        if (rr == 0u)
            cr = jm_fwd_colour_f64(2, p0, p1, p2);
    }
}

// ------------------------------------------------------------ inverse colour
// out_k = clamp(round(Y + m_k1*(Cb-128) + m_k2*(Cr-128))): Y is an integer, so
// the rounding only concerns the chroma term, shared by the 2x2 pixels of one
// chroma sample.  Same tie argument as above (denominators 1000 and 10^6).
struct JmChromaTerm {
    int d0, d1, d2; // rounded chroma contributions to channels 0,1,2
    int tie;        // bit k set: channel k is an exact tie, use the doubles
};

JM_HD JmChromaTerm jm_inv_chroma(int cb, int cr)
{
    int b = cb - 128, r = cr - 128;
    uint32_t n0 = (uint32_t)(1402 * r + 500 + 256000);
    uint32_t n2 = (uint32_t)(1772 * b + 500 + 256000);
    uint32_t n1 = (uint32_t)(-344136 * b - 714136 * r + 500000 + 256000000);
    // plain integer divisions: once per chroma sample, not per pixel
    uint32_t q0 = n0 / 1000u, q2 = n2 / 1000u, q1 = n1 / 1000000u;
    JmChromaTerm t;
    t.d0 = (int)q0 - 256;
    t.d1 = (int)q1 - 256;
    t.d2 = (int)q2 - 256;
    t.tie = (n0 - q0 * 1000u == 0u ? 1 : 0) | (n1 - q1 * 1000000u == 0u ? 2 : 0) |
            (n2 - q2 * 1000u == 0u ? 4 : 0);
    return t;
}

JM_HD int jm_clamp255(int v)
{
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one channel, the reference's operation order
JM_HD int jm_inv_colour_f64(int k, int y, int cb, int cr)
{
    const double m1 = (k == 0) ? 0.0 : (k == 1) ? -0.344136 : 1.772;
    const double m2 = (k == 0) ? 1.402 : (k == 1) ? -0.714136 : 0.0;
    double t = 0.0;
    t = t + 1.0 * ((double)y - 0.0);
    t = t + m1 * ((double)cb - 128.0);
    t = t + m2 * ((double)cr - 128.0);
    t = jm_round_half_away_f64(t);
    t = t > 255.0 ? 255.0 : t;
    t = t < 0.0 ? 0.0 : t;
    return (int)t;
}

JM_HD void jm_inv_colour(int y, int cb, int cr, const JmChromaTerm &c, int &o0, int &o1, int &o2)
{
    o0 = jm_clamp255(y + c.d0);
    o1 = jm_clamp255(y + c.d1);
    o2 = jm_clamp255(y + c.d2);
    if (__builtin_expect(c.tie != 0, 0)) {
        if (c.tie & 1)
            o0 = jm_inv_colour_f64(0, y, cb, cr);
        if (c.tie & 2)
            o1 = jm_inv_colour_f64(1, y, cb, cr);
        if (c.tie & 4)
            o2 = jm_inv_colour_f64(2, y, cb, cr);
    }
}

// --------------------------------------------------------------- 8-point DCT
// X_u = (c_u/2) sum_n x_n cos((2n+1)u*pi/16), even/odd split, all in f32.
// Two passes give 0.25*cu*cv*sum of encode_decode.cpp:284.  S scales the
// outputs (the second forward pass folds the *100/4... factor: see kernels).
#define JM_A0 0.35355339059327376220 /* 1/(2*sqrt(2)) = cos(pi/4)/2 */
#define JM_C1 0.49039264020161522456
#define JM_C2 0.46193976625564337806
#define JM_C3 0.41573480615127261854
#define JM_C5 0.27778511650980111237
#define JM_C6 0.19134171618254488586
#define JM_C7 0.09754516100806413392

// LEVEL: subtract the 128 level shift (recenterAroundZero, :289-298) -- it only
// reaches the u=0 output, every other output is built from differences.
// LEVEL_SUM: what the level shift amounts to in (t0 + t1), the sum of the eight inputs: 8 * 128 for
// pixels; the matrix-pipe row pass hands over row sums scaled by 2^-10 * a_0 and shifts them in the
// column pass instead (jpeg_kernels.hip), where the sum of eight of them carries 8 * 128 * 8 a_0 / 1024.
template <int SNUM, bool LEVEL, int LEVEL_SUM_NUM = 1024, int LEVEL_SUM_DEN = 1>
JM_HD void jm_fdct8(float &x0, float &x1, float &x2, float &x3,
                    float &x4, float &x5, float &x6, float &x7)
{
    constexpr float LEVEL_SUM = (float)((double)LEVEL_SUM_NUM / (double)LEVEL_SUM_DEN);
    constexpr double S = (double)SNUM;
    constexpr float A0 = (float)(S * JM_A0), C1 = (float)(S * JM_C1), C2 = (float)(S * JM_C2),
                    C3 = (float)(S * JM_C3), C5 = (float)(S * JM_C5), C6 = (float)(S * JM_C6),
                    C7 = (float)(S * JM_C7);
    float s0 = x0 + x7, s1 = x1 + x6, s2 = x2 + x5, s3 = x3 + x4;
    float d0 = x0 - x7, d1 = x1 - x6, d2 = x2 - x5, d3 = x3 - x4;
    float t0 = s0 + s3, t1 = s1 + s2, t2 = s0 - s3, t3 = s1 - s2;
    x0 = LEVEL ? A0 * ((t0 + t1) - LEVEL_SUM) : A0 * (t0 + t1);
    x4 = A0 * (t0 - t1);
    x2 = jm_fma(C2, t2, C6 * t3);
    x6 = jm_fma(C6, t2, -(C2 * t3));
    x1 = jm_fma(C7, d3, jm_fma(C5, d2, jm_fma(C3, d1, C1 * d0)));
    x3 = jm_fma(-C5, d3, jm_fma(-C1, d2, jm_fma(-C7, d1, C3 * d0)));
    x5 = jm_fma(C3, d3, jm_fma(C7, d2, jm_fma(-C1, d1, C5 * d0)));
    x7 = jm_fma(-C1, d3, jm_fma(C3, d2, jm_fma(-C5, d1, C7 * d0)));
}

// x_n = sum_u (c_u/2) X_u cos((2n+1)u*pi/16)
// SHRINK: every constant carries a factor (1 - 1.5e-7), i.e. the outputs come out
// pulled toward zero by about two ulps -- the second inverse pass uses it so that the
// pixel rounding needs no separate shrink step (see jm_pixel_magic).
template <bool SHRINK>
JM_HD void jm_idct8(float &x0, float &x1, float &x2, float &x3,
                    float &x4, float &x5, float &x6, float &x7)
{
    constexpr double K = SHRINK ? 1.0 - 1.5e-7 : 1.0;
    constexpr float A0 = (float)(K * JM_A0), C1 = (float)(K * JM_C1), C2 = (float)(K * JM_C2),
                    C3 = (float)(K * JM_C3), C5 = (float)(K * JM_C5), C6 = (float)(K * JM_C6),
                    C7 = (float)(K * JM_C7);
    float p0 = A0 * (x0 + x4), p1 = A0 * (x0 - x4);
    float r0 = jm_fma(C2, x2, C6 * x6), r1 = jm_fma(C6, x2, -(C2 * x6));
    float e0 = p0 + r0, e3 = p0 - r0, e1 = p1 + r1, e2 = p1 - r1;
    float o0 = jm_fma(C7, x7, jm_fma(C5, x5, jm_fma(C3, x3, C1 * x1)));
    float o1 = jm_fma(-C5, x7, jm_fma(-C1, x5, jm_fma(-C7, x3, C3 * x1)));
    float o2 = jm_fma(C3, x7, jm_fma(C7, x5, jm_fma(-C1, x3, C5 * x1)));
    float o3 = jm_fma(-C1, x7, jm_fma(C3, x5, jm_fma(-C5, x3, C7 * x1)));
    x0 = e0 + o0;
    x7 = e0 - o0;
    x1 = e1 + o1;
    x6 = e1 - o1;
    x2 = e2 + o2;
    x5 = e2 - o2;
    x3 = e3 + o3;
    x4 = e3 - o3;
}

// ------------------------------------------------- 8x8 DCT on the matrix cores
// v_mfma_f32_16x16x4_f32 accumulates D = fma(a_k, b_k, ...fma(a_0, b_0, C)) in k order
// with one f32 rounding per product (cdna_hip_programming.md, FP32-input MFMA), so the
// kernels' matrix-core transform is a plain fma chain the host can replay bit for bit.
// The basis  m(u, n) = (c_u / 2) cos((2n+1) u pi / 16)  is rounded to float once.
// the same basis as a double (compile-time tables)
constexpr double jm_dct_basis_f64(int u, int n)
{
    int a = ((2 * n + 1) * u) & 31;
    a = a > 16 ? 32 - a : a;
    const bool neg = a > 8;
    a = neg ? 16 - a : a;
    const double m = (a == 0 || a == 4) ? JM_A0 : a == 1 ? JM_C1 : a == 2 ? JM_C2 : a == 3 ? JM_C3
                     : a == 5 ? JM_C5 : a == 6 ? JM_C6 : JM_C7;
    return u == 0 ? JM_A0 : neg ? -m : m;
}
JM_HD float jm_dct_basis(int u, int n, double scale)
{
    // (2n+1) u pi/16 folded into the first octant: only JM_A0 and JM_C1..JM_C7 occur
    int a = ((2 * n + 1) * u) & 31;
    a = a > 16 ? 32 - a : a;
    const bool neg = a > 8;
    a = neg ? 16 - a : a;
    const double m = (a == 0 || a == 4) ? JM_A0 : a == 1 ? JM_C1 : a == 2 ? JM_C2 : a == 3 ? JM_C3
                     : a == 5 ? JM_C5 : a == 6 ? JM_C6 : JM_C7;
    const float f = (float)(scale * m);
    return neg ? -f : f;
}

// The order in which the second MFMA pass visits the 8 entries of a block's
// contraction index: pass one leaves index i = 4*(lane>>4) + reg in accumulator
// register `reg`, and register r feeds K-step r of pass two, where lane group g
// supplies i = 4g + r.
JM_HD int jm_mfma_k_order(int step) // step 0..7 -> index 0,4,1,5,2,6,3,7
{
    return (step & 1) * 4 + (step >> 1);
}

// Level shift folded into the accumulator of the first forward pass (recenterAroundZero,
// :289-298): -128 * sum_n m(0, n) for u = 0, nothing for the other rows.
#define JM_LEVEL_DC (-362.03867196751236f) /* -128 * 8 / (2 sqrt 2) */

// px: 64 pixel values 0..255 (row-major) -> 100 * DCT coefficients (natural order)
JM_HD void jm_mfma_fdct_block(const float (&px)[64], float (&out)[64])
{
    float z[64]; // z[u*8 + x]: vertical pass, contraction over the row index y, ascending
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) {
            float acc = u == 0 ? JM_LEVEL_DC : 0.0f;
            for (int y = 0; y < 8; ++y)
                acc = jm_fma(px[y * 8 + x], jm_dct_basis(u, y, 1.0), acc);
            z[u * 8 + x] = acc;
        }
    for (int u = 0; u < 8; ++u) // horizontal pass, scaled by 100, contraction over x
        for (int v = 0; v < 8; ++v) {
            float acc = 0.0f;
            for (int st = 0; st < 8; ++st) {
                const int x = jm_mfma_k_order(st);
                acc = jm_fma(z[u * 8 + x], jm_dct_basis(v, x, 100.0), acc);
            }
            out[u * 8 + v] = acc;
        }
}

// w: 64 dequantised coefficients (natural order) -> 0.25*sum of :591 (before rounding)
JM_HD void jm_mfma_idct_block(const float (&w)[64], float (&out)[64])
{
    float t[64]; // t[u*8 + x]: contraction over v, ascending
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) {
            float acc = 0.0f;
            for (int v = 0; v < 8; ++v)
                acc = jm_fma(w[u * 8 + v], jm_dct_basis(v, x, 1.0), acc);
            t[u * 8 + x] = acc;
        }
    for (int y = 0; y < 8; ++y) // contraction over u in the register-fed order
        for (int x = 0; x < 8; ++x) {
            float acc = 0.0f;
            for (int st = 0; st < 8; ++st) {
                const int u = jm_mfma_k_order(st);
                acc = jm_fma(jm_dct_basis(u, y, 1.0), t[u * 8 + x], acc);
            }
            out[y * 8 + x] = acc;
        }
}

// ------------------------------------------------------------------ quantise
// x100 = 100 * (0.25*cu*cv*sum) as produced by the two forward passes (the
// second one scaled by 100).  Follows the reference bit for bit from there:
//   r = round(x100)                       encode_decode.cpp:284 round(...*100)
//   c = (float)(r / 100)                  :284  (/100, narrowed to float)
//   v = c / (float)Q                      :306  IEEE float division
//   q = roundf(v); -0 -> 0                :306-307
// Both divisions are done as multiply + one Newton correction (q0 = a*y,
// q = fma(fma(-b, q0, a), y, q0) with y = RN(1/b)), which is the correctly
// rounded quotient for these operand ranges; the model test checks every
// (r, Q) pair against real divisions.
JM_HD float jm_div_by(float a, float b, float rcp_b)
{
    float q0 = a * rcp_b;
    float rem = jm_fma(-b, q0, a);
    return jm_fma(rem, rcp_b, q0);
}

JM_HD int jm_snap_quant(float x100, float q, float rcp_q)
{
    float r = jm_round_half_away(x100);
    float c = jm_div_by(r, 100.0f, 0.01f);
    float v = jm_div_by(c, q, rcp_q);
    return (int)jm_round_half_away(v); // (int)(-0.0f) == 0
}

// ===========================================================================
// Block-level fast paths.  gfx950 issues v_fma/v_mul/v_add/v_sub (f32),
// v_add/v_sub (int), and/or/xor and right shifts at full rate, and nearly
// everything else (conversions, trunc/rndne, bfi, min/max/med3, 24-bit mads,
// compares, perm) at about 0.6x that rate (tools/valu_ubench2.hip), so the
// per-sample chains below are built from the first group: rounding is done by
// adding and subtracting 1.5*2^23 ("magic"), which leaves the nearest integer
// (ties to even) in the low mantissa bits of a float.
// ===========================================================================
#define JM_MAGIC 12582912.0f /* 1.5 * 2^23: ulp is 1 for values in [2^23, 2^24) */

JM_HD uint32_t jm_bits(float f)
{
    return __builtin_bit_cast(uint32_t, f);
}

// low 16 bits of a, low 16 bits of b -> one dword (a in the low half)
JM_HD uint32_t jm_pack_lo16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(b, a, 0x05040100u);
#else
    return (a & 0xFFFFu) | (b << 16);
#endif
}

// Quantiser constants for one table, natural order.
struct JmQuantTab {
    float inv100q[64]; // RN((1 + 2^-21) / (100 Q)): see jm_quantise_block
    float q[64];       // Q (dequantisation)
};

JM_HD void jm_make_quant_tab(const int32_t q[64], JmQuantTab &t)
{
    for (int i = 0; i < 64; ++i) {
        t.q[i] = (float)q[i];
        t.inv100q[i] = (float)((1.0 + 0x1p-21) / (100.0 * (double)q[i]));
    }
}

// x[i] = 100 * coefficient (natural order) on entry; on exit x[i] = MAGIC + q_i,
// i.e. a float whose low 16 bits are the quantised coefficient as int16.
//
//   r = nearest integer of x: the reference's round(.. * 100) (:284).  Ties to
//       even instead of away from zero only matters when the float x is exactly
//       k + 0.5, which the reference's own accumulation noise makes immaterial.
//   q = the reference's roundf((float)(r / 100) / Q) (:284, :306).  r/(100 Q) is
//       a rational that is either an exact tie j + 0.5 or at least 1/(100 Q) away
//       from one -- a relative distance of at least 1/|r| > 3.8e-6 for
//       |r| < 2^18, far beyond float error -- so away from ties the reference's
//       two float divisions cannot change the nearest integer.  AT an exact tie
//       r = 50 Q (2j+1): r/100 = Q (2j+1) / 2 has at most 21 significant bits, so
//       both divisions are exact, the quotient is exactly j + 0.5 and roundf takes
//       it away from zero.  Hence q = round-half-away(r / (100 Q)) as a rational,
//       which one fma delivers: the multiplier carries a 2^-21 relative push away
//       from zero (more than the float error, less than the 3.8e-6 gap), and the
//       magic add rounds to nearest.  tests/model checks every (r, Q) pair against
//       real float divisions.
JM_HD void jm_quantise_block(float (&x)[64], const float *inv100q)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 64; ++i) {
        float r = jm_rint(x[i]); // round to nearest even, what (x + MAGIC) - MAGIC does in two operations
        x[i] = jm_fma(r, inv100q[i], JM_MAGIC);
    }
}
JM_HD void jm_quantise_block(float (&x)[64], const JmQuantTab &t) { jm_quantise_block(x, t.inv100q); }

// IDCT output x = 0.25*sum (:591) -> float whose low 16 bits hold
// round(x) + 128 as int16, with exact x.5 ties resolved toward zero like the
// reference: x is first shrunk by 1.5e-7 of itself (a few
// ulps: more than this kernel's own rounding error at a DC-only tie), then the
// magic add rounds to nearest.
JM_HD float jm_pixel_magic(float x)
{
    float z = jm_fma(x, -1.5e-7f, x);
    return z + (JM_MAGIC + 128.0f);
}
// the same for an x that already carries the shrink (jm_idct8<true> as the last pass)
JM_HD float jm_pixel_magic_preshrunk(float x)
{
    return x + (JM_MAGIC + 128.0f);
}

JM_HD int jm_magic_to_int(float u)
{
    return (int)(int16_t)(jm_bits(u) & 0xFFFFu);
}
// MAGIC + 128 + x -> x + 128 as a full int (the legacy block boundary returns UNCLAMPED int32):
// exact while |x + 128| < 2^22
JM_HD int jm_magic_to_int32(float u)
{
    return (int)(u - JM_MAGIC);
}

// four jm_pixel_magic values -> four bytes saturated to 0..255 (std::clamp, :808).
// u - MAGIC is x + 128 as an exact float integer (while |x + 128| < 2^22) and v_cvt_pk_u8_f32
// converts with SATURATION into the byte it is told (tools/cvt_pk_u8_probe.hip), so a sum far
// outside int16 -- a corrupt file, a caller's own coefficients -- still clamps to 0 or 255
// like the reference's int conversion + std::clamp (:705, :808) instead of wrapping to the
// opposite extreme.  Two instructions per sample, no packing permutes.
JM_HD uint32_t jm_pack4_u8(float a, float b, float c, float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(a - JM_MAGIC));
    asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(b - JM_MAGIC));
    asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(r) : "v"(c - JM_MAGIC));
    asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(r) : "v"(d - JM_MAGIC));
    return r;
#else
    return (uint32_t)jm_clamp255(jm_magic_to_int32(a)) | ((uint32_t)jm_clamp255(jm_magic_to_int32(b)) << 8) |
           ((uint32_t)jm_clamp255(jm_magic_to_int32(c)) << 16) | ((uint32_t)jm_clamp255(jm_magic_to_int32(d)) << 24);
#endif
}

// ---------------------------------------------------------------- colour, fast
// Forward: the same decision as jm_fwd_colour (exact rational rounding, with
// exact x.5 ties handed to the reference's double sequence), but the common
// case runs on float fma/add only.
//   Y      float chain of value + 6e-5 (error < 4.6e-5; the rational values are 1e-3 apart).
//   Cb, Cr centred chains  s = value - 128 + 5e-5  (error < 2e-5; the rational
//          values are 1e-4 apart).  The bias makes the magic add round exact ties
//          UP, which is what round() does unless the double sequence lands below
//          the tie; c = rounded - 128 comes back as an exact float for the quad
//          sums, and the byte (with the 256 -> 0 wrap of static_cast<uchar>,
//          :67-70) sits in the low 8 bits of s + (MAGIC + 128).
// Only triples whose double sequence lands BELOW an exact tie need anything else than
// the biased chains (which round ties up): 3464 Y, 34 Cb and 4186 Cr triples of 2^24,
// plus the two wraps Cb(0,0,255) = Cr(255,0,0) = 256 -> 0.  Candidates are found from
// the rounding residuals r = s - round(s) (a tie sits at -0.5 + bias):
//   Y, Cr  candidate iff r < JM_TIE_A (every exact tie).
//   Cb     all 34 triples have Cb <= 64 and the wrap has Cb = 256, so the test value
//          r - 1.5e-4 |cb| is compared with JM_TIE_B: exact ties with |Cb - 128| >= 64
//          (plus, harmlessly, near-ties of more saturated pixels).  Every Cb tie has
//          R == G, which photographs have in ~0.5 % of their pixels: without this filter
//          they were 40 % of all candidates.
// Smaller = more suspicious, so a wave takes the MINIMUM over a few pixels and compares once.
JM_HD float jm_abs(float x) { return __builtin_fabsf(x); }
// min of three / two floats, none of them NaN (v_min3_f32 without the canonicalising
// extras fminf would get)
JM_HD float jm_min3(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    float m = a < b ? a : b;
    return m < c ? m : c;
#endif
}
JM_HD float jm_min2(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a < b ? a : b;
#endif
}
struct JmFwdTies {
    float sy, sb, sr; // the unrounded float values (Y; Cb, Cr centred and biased)
    float ry, rb, rr; // test values: candidate iff ry, rr < JM_TIE_A or rb < JM_TIE_B
};
#define JM_CHROMA_BIAS 5.0e-5f
#define JM_LUMA_BIAS 6.0e-5f
#define JM_TIE_A (-0.49988f)
#define JM_CB_SLOPE 1.5e-4f
#define JM_TIE_B (-0.5095f) // -0.49995 - 64 * JM_CB_SLOPE = -0.50955 fires, 63 -> -0.5094 does not

JM_HD void jm_fwd_colour_fast(float p0, float p1, float p2, float &uy, float &cb, float &cr, JmFwdTies &t)
{
    // Y: value + 6e-5 (the chain's error is below 4.6e-5, the rational values are 1e-3 apart)
    t.sy = jm_fma(p2, 0.114f, jm_fma(p1, 0.587f, jm_fma(p0, 0.299f, JM_LUMA_BIAS)));
    uy = t.sy + JM_MAGIC;
    t.ry = t.sy - (uy - JM_MAGIC);
    t.sb = jm_fma(p2, 0.5f, jm_fma(p1, -0.3313f, jm_fma(p0, -0.1687f, JM_CHROMA_BIAS)));
    cb = jm_rint(t.sb); // nearest even like a magic add, one operation
    t.rb = jm_fma(jm_abs(cb), -JM_CB_SLOPE, t.sb - cb);
    t.sr = jm_fma(p2, -0.0813f, jm_fma(p1, -0.4187f, jm_fma(p0, 0.5f, JM_CHROMA_BIAS)));
    cr = jm_rint(t.sr);
    t.rr = t.sr - cr;
}

JM_HD bool jm_fwd_any_tie(const JmFwdTies &t)
{
    return t.ry < JM_TIE_A || t.rr < JM_TIE_A || t.rb < JM_TIE_B;
}

// Exact ties: the value is k + 0.5 as a rational; the reference's double sequence
// (encode_decode.cpp:58-63, one rounding per operation, no contraction) lands on,
// above or below it, and round() goes up unless it is below.  The fast path has rounded
// the tie UP, so its own value names the tie: "up - 0.5".  (A Cb candidate that is no
// exact tie compares at least 1e-4 above "its rounded value - 0.5" and stays as it is.)
// Exact products (0.5 x, and 128 + 0.5 x) are folded into the neighbouring addition: one
// rounding either way.  cb/cr stay centred (value - 128); 256 wraps to 0, i.e. -128 centred.
JM_HD double jm_fma64(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return fma(a, b, c);
#endif
}
JM_HD void jm_fwd_colour_fix(float p0, float p1, float p2, const JmFwdTies &t, float &uy, float &cb, float &cr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // keep the conversions on the floats at hand (left alone, the compiler goes back to
    // the packed bytes: eight more instructions per visit)
    asm("" : "+v"(p0), "+v"(p1), "+v"(p2));
#endif
    const double d0 = (double)p0, d1 = (double)p1, d2 = (double)p2;
    if (t.ry < JM_TIE_A) {
        double v = 0.299 * d0; // 0.0 + x is exact
        v = v + 0.587 * d1;
        v = v + 0.114 * d2;
        const float tie = (uy - JM_MAGIC) - 0.5f; // exact in float: an integer below 256 less one half
        uy = v < (double)tie ? uy - 1.0f : uy;
    }
    if (t.rb < JM_TIE_B) {
        double v = 128.0 + -0.1687 * d0;
        v = v + -0.3313 * d1;
        v = jm_fma64(0.5, d2, v);
        cb = v < (double)(cb + 127.5f) ? cb - 1.0f : cb; // (the sum is exact in float)
        cb = cb == 128.0f ? -128.0f : cb;
    }
    if (t.rr < JM_TIE_A) {
        double v = (double)jm_fma(0.5f, p0, 128.0f); // exact in float as in double: one conversion instead of two operations
        v = v + -0.4187 * d1;
        v = v + -0.0813 * d2;
        cr = v < (double)(cr + 127.5f) ? cr - 1.0f : cr;
        cr = cr == 128.0f ? -128.0f : cr;
    }
}

// truncating mean of four chroma values (chromaSubsampling, :139-149) as MAGIC + mean;
// `sum` is the sum of the four CENTRED values (each value - 128)
JM_HD float jm_quad_mean_magic(float sum)
{
    // sum/4 + 128 is a multiple of 0.25: -0.375 makes nearest == floor
    return jm_fma(sum, 0.25f, 127.625f) + JM_MAGIC;
}

// low bytes of four magic floats -> one dword (a in byte 0)
JM_HD uint32_t jm_pack_lo8x4(float a, float b, float c, float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t ab = __builtin_amdgcn_perm(jm_bits(b), jm_bits(a), 0x0C0C0400u);
    uint32_t cd = __builtin_amdgcn_perm(jm_bits(d), jm_bits(c), 0x0C0C0400u);
    return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
#else
    return (jm_bits(a) & 0xFFu) | ((jm_bits(b) & 0xFFu) << 8) | ((jm_bits(c) & 0xFFu) << 16) |
           ((jm_bits(d) & 0xFFu) << 24);
#endif
}

// ------------------------------------------------------------- inverse colour, fast
// One chroma sample serves a 2x2 pixel quad (upsampleChroma, :165-170), so its
// rounded contributions d0,d1,d2 (jm_inv_chroma) are packed once as int16 pairs
// and each output row then costs one packed add + one saturating pack per two
// output bytes: B0 G0 | R0 B1 | G1 R1 for a pixel pair.
struct JmChromaPk {
    uint32_t d01, d20, d12; // {d0,d1}, {d2,d0}, {d1,d2} as int16 pairs (first in the low half)
    int tie;                // < 0: some pixel under this sample needs the exact path
};

JM_HD JmChromaPk jm_inv_chroma_pk(int cb, int cr)
{
    const JmChromaTerm t = jm_inv_chroma(cb, cr);
    JmChromaPk p;
    p.d01 = jm_pack_lo16((uint32_t)t.d0, (uint32_t)t.d1);
    p.d20 = jm_pack_lo16((uint32_t)t.d2, (uint32_t)t.d0);
    p.d12 = jm_pack_lo16((uint32_t)t.d1, (uint32_t)t.d2);
    p.tie = t.tie ? -1 : 0;
    return p;
}

// The same three contributions from float chains.  With b = Cb-128, r = Cr-128:
//   d0 = round(1.402 r):  the float product is within 2e-5 of the rational value,
//        whose distance to any x.5 is at least 1e-3 (1402 r = 500 mod 1000 has no
//        solution), so the magic add lands on the right integer;
//   d2 = round(1.772 b):  likewise except for the exact ties at |b| = 125;
//   d1 = round(-0.344136 b - 0.714136 r): denominators of 1e6, so anything
//        within 1e-4 of x.5 is a candidate.
// `tie` < 0 (sign bit) flags a candidate: jm_inv_row4 then redoes its pixels with the
// exact integer/double path, so d1/d2 need not be right for flagged samples.
template <int BI>
JM_HD JmChromaPk jm_inv_chroma_pk_fast(uint32_t cbw, uint32_t crw) // byte BI of each word
{
    const float b = (float)((cbw >> (8 * BI)) & 0xFFu) - 128.0f;
    const float r = (float)((crw >> (8 * BI)) & 0xFFu) - 128.0f;
    const float u0 = jm_fma(r, 1.402f, JM_MAGIC);
    const float u2 = jm_fma(b, 1.772f, JM_MAGIC);
    const float f1 = jm_fma(r, -0.714136f, b * -0.344136f);
    const float u1 = f1 + JM_MAGIC;
    const float t1 = 0.4999f - __builtin_fabsf(f1 - (u1 - JM_MAGIC));
    const float t2 = __builtin_fabsf(__builtin_fabsf(b) - 125.0f) - 0.5f;
    JmChromaPk p;
    p.d01 = jm_pack_lo16(jm_bits(u0), jm_bits(u1));
    p.d20 = jm_pack_lo16(jm_bits(u2), jm_bits(u0));
    p.d12 = jm_pack_lo16(jm_bits(u1), jm_bits(u2));
    p.tie = (int)(jm_bits(t1) | jm_bits(t2));
    return p;
}

// {a.lo + b.lo, a.hi + b.hi} as int16, each saturated to 0..255 -> two bytes in the low half
JM_HD uint32_t jm_add_sat_u8x2(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short jm_s2 __attribute__((ext_vector_type(2)));
    jm_s2 s = __builtin_bit_cast(jm_s2, a) + __builtin_bit_cast(jm_s2, b);
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, s)));
    return r;
#else
    int lo = (int16_t)(a & 0xFFFFu) + (int16_t)(b & 0xFFFFu);
    int hi = (int16_t)(a >> 16) + (int16_t)(b >> 16);
    return (uint32_t)jm_clamp255(lo) | ((uint32_t)jm_clamp255(hi) << 8);
#endif
}

template <int I, int J>
JM_HD uint32_t jm_ypair(uint32_t y4) // {Y_I, Y_J} as uint16 pair from four packed Y bytes
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, y4, 0x0C000C00u | (uint32_t)I | ((uint32_t)J << 16));
#else
    return ((y4 >> (8 * I)) & 0xFFu) | (((y4 >> (8 * J)) & 0xFFu) << 16);
#endif
}

// four pixels of one row (Y bytes y4), pixels 0,1 under chroma sample a and 2,3
// under b -> 12 output bytes: one packed add + one saturating pack per two bytes.
// Valid unless a.tie or b.tie is flagged (< 0); then use jm_inv_row4_exact.
JM_HD void jm_inv_row4_fast(uint32_t y4, const JmChromaPk &a, const JmChromaPk &b,
                            uint32_t &w0, uint32_t &w1, uint32_t &w2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // One register {Y0, Y1} serves the three adds of a pixel pair: the operand-half selectors of
    // the packed add pick {Y0,Y0}, {Y0,Y1} or {Y1,Y1}; the second saturating pack of every output
    // dword writes the upper half of the first one's register (SDWA), so no permute assembles it.
    const uint32_t y01 = jm_ypair<0, 1>(y4), y23 = jm_ypair<2, 3>(y4);
    uint32_t s0, s1, s2, s3, s4, s5;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(s0) : "v"(y01), "v"(a.d01));
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(s1) : "v"(y01), "v"(a.d20));
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(s2) : "v"(y01), "v"(a.d12));
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(s3) : "v"(y23), "v"(b.d01));
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(s4) : "v"(y23), "v"(b.d20));
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(s5) : "v"(y23), "v"(b.d12));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(w0) : "v"(s0));
    asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w0) : "v"(s1));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(w1) : "v"(s2));
    asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w1) : "v"(s3));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(w2) : "v"(s4));
    asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w2) : "v"(s5));
#else
    uint32_t h0 = jm_add_sat_u8x2(jm_ypair<0, 0>(y4), a.d01);
    uint32_t h1 = jm_add_sat_u8x2(jm_ypair<0, 1>(y4), a.d20);
    uint32_t h2 = jm_add_sat_u8x2(jm_ypair<1, 1>(y4), a.d12);
    uint32_t h3 = jm_add_sat_u8x2(jm_ypair<2, 2>(y4), b.d01);
    uint32_t h4 = jm_add_sat_u8x2(jm_ypair<2, 3>(y4), b.d20);
    uint32_t h5 = jm_add_sat_u8x2(jm_ypair<3, 3>(y4), b.d12);
    w0 = jm_pack_lo16(h0, h1);
    w1 = jm_pack_lo16(h2, h3);
    w2 = jm_pack_lo16(h4, h5);
#endif
}

// the same four pixels by the exact path: integer chroma terms, and the reference's
// double sequence (jm_inv_colour_f64) for the channels that are exact ties
JM_HD void jm_inv_row4_exact(uint32_t y4, int cba, int cra, int cbb, int crb,
                             uint32_t &w0, uint32_t &w1, uint32_t &w2)
{
    uint8_t o[12];
    for (int j = 0; j < 4; ++j) {
        const int y = (int)((y4 >> (8 * j)) & 0xFFu);
        const int cb = j < 2 ? cba : cbb, cr = j < 2 ? cra : crb;
        const JmChromaTerm t = jm_inv_chroma(cb, cr);
        int o0, o1, o2;
        jm_inv_colour(y, cb, cr, t, o0, o1, o2);
        o[3 * j] = (uint8_t)o0, o[3 * j + 1] = (uint8_t)o1, o[3 * j + 2] = (uint8_t)o2;
    }
    w0 = o[0] | (o[1] << 8) | (o[2] << 16) | ((uint32_t)o[3] << 24);
    w1 = o[4] | (o[5] << 8) | (o[6] << 16) | ((uint32_t)o[7] << 24);
    w2 = o[8] | (o[9] << 8) | (o[10] << 16) | ((uint32_t)o[11] << 24);
}

JM_HD void jm_inv_row4(uint32_t y4, const JmChromaPk &a, const JmChromaPk &b,
                       int cba, int cra, int cbb, int crb, uint32_t &w0, uint32_t &w1, uint32_t &w2)
{
    jm_inv_row4_fast(y4, a, b, w0, w1, w2);
    if (__builtin_expect((a.tie | b.tie) < 0, 0))
        jm_inv_row4_exact(y4, cba, cra, cbb, crb, w0, w1, w2);
}
