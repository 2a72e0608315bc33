// pk_fdct_experiment.h -- the forward DCT + quantiser stage on packed f32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  MEASURED AND NOT USED by the library: bit-identical
// to block_fdct_quant (the GPU parity suite passed with it in the encoder), 18 % fewer VALU
// instructions in the fused encoder (53.1 M -> 43.6 M at 8192^2) and 16 % faster on its own
// (tools/fdct_ubench.hip), but the fused encoder did not get faster (88.0-90.5 us against
// 86.3-89.6 us): with the transform removed altogether the kernel only drops to 78-80 us, so most
// of the transform's VALU time already hides behind the colour stage's waits
// (profiles/r02_d_encoder_ablation.txt).  Kept here so that the measurement can be repeated.
#ifndef JPEGAMD_PK_FDCT_EXPERIMENT_H
#define JPEGAMD_PK_FDCT_EXPERIMENT_H
#include "../jpeg_amd/csrc/jpeg_device_common.h"

// ------------------------------------------------- the same stage on packed f32 instructions
// On gfx950 a SIMD issues about one VALU instruction per 4 cycles whatever the number of waves
// (tools/pk_ubench.hip: 3.8-4.1 cycles per v_fma_f32 from 3 to 8 waves per SIMD) and the fused
// kernels are bound by exactly that; v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 take ~5 cycles
// and do the work of two.  The arithmetic below is jm_fdct8's and jm_quantise_block's operation
// for operation (each lane of a packed instruction is one IEEE operation of the scalar code, in
// the same order), so the results are bit-identical to the scalar stage and to the host model.
//   row pass:    the even/odd structure of ONE 8-point transform packs two outputs per
//                instruction with the operand-half selectors (op_sel) and per-half negation of
//                VOP3P: 19 packed instructions instead of 37.  A row comes out as the pairs of
//                frequencies (0,4) (2,6) (1,3) (5,7).
//   column pass: those pairs stay together; each packed instruction does the same step of the
//                8-point transform for two columns: 4 x 36 instead of 8 x 36.
//   quantiser:   32 x 3 packed instead of 64 x 3; the table is read in the paired order
//                (kPkOrder within every row of 8).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long pkc_t; // two float bit patterns (lo, hi) for a scalar register pair

__device__ constexpr uint8_t kPkOrder[8] = {0, 4, 2, 6, 1, 7, 3, 5};

__device__ __forceinline__ constexpr pkc_t pkc(float lo, float hi)
{
    return (pkc_t)__builtin_bit_cast(uint32_t, lo) | ((pkc_t)__builtin_bit_cast(uint32_t, hi) << 32);
}

// Every constant of the stage comes out of eight scalar register pairs through the operand-half
// selectors and the per-half negation of VOP3P; the pairs are made opaque once per kernel so
// that they stay in registers instead of being rebuilt (two s_mov each) in front of every use.
struct PkBank {
    pkc_t r17, r35, r26; // (C1,C7) (C3,C5) (C2,C6) of the row pass
    pkc_t c17, c35, c26; // the same times 100 for the column pass
    pkc_t a0;            // (A0, 100 A0)
    pkc_t magic;         // (MAGIC, MAGIC)
};
__device__ __forceinline__ PkBank pk_bank()
{
    PkBank b;
    b.r17 = pkc((float)JM_C1, (float)JM_C7), b.r35 = pkc((float)JM_C3, (float)JM_C5), b.r26 = pkc((float)JM_C2, (float)JM_C6);
    b.c17 = pkc((float)(100.0 * JM_C1), (float)(100.0 * JM_C7)), b.c35 = pkc((float)(100.0 * JM_C3), (float)(100.0 * JM_C5));
    b.c26 = pkc((float)(100.0 * JM_C2), (float)(100.0 * JM_C6));
    b.a0 = pkc((float)JM_A0, (float)(100.0 * JM_A0)), b.magic = pkc(JM_MAGIC, JM_MAGIC);
    asm volatile("" : "+s"(b.r17), "+s"(b.r35), "+s"(b.r26), "+s"(b.c17), "+s"(b.c35), "+s"(b.c26), "+s"(b.a0), "+s"(b.magic));
    return b;
}

// A packed f32 instruction whose operand was written by the instruction just before it costs a
// wait state (an s_nop, one issue slot).  The statements below are therefore `asm volatile` --
// kept in the order written -- and every sequence is written so that no instruction reads the
// result of its predecessor.
#define PK_ASM asm volatile
// operand selectors: which half of the data (src0) and of the constant pair (src1) feeds the
// low result (ds, ks) and the high result (dh, kh); nl / nh negate the constant for that half
#define PK_SEL2(ds, dh, ks, kh, nl, nh) \
    " op_sel:[" #ds "," #ks "] op_sel_hi:[" #dh "," #kh "] neg_lo:[0," #nl "] neg_hi:[0," #nh "]"
#define PK_SEL3(ds, dh, ks, kh, nl, nh) \
    " op_sel:[" #ds "," #ks ",0] op_sel_hi:[" #dh "," #kh ",1] neg_lo:[0," #nl ",0] neg_hi:[0," #nh ",0]"
#define PK_MULK(d, a, k, ...) PK_ASM("v_pk_mul_f32 %0, %1, %2" PK_SEL2(__VA_ARGS__) : "=v"(d) : "v"(a), "s"(k))
#define PK_ADDK(d, a, k, ...) PK_ASM("v_pk_add_f32 %0, %1, %2" PK_SEL2(__VA_ARGS__) : "=v"(d) : "v"(a), "s"(k))
#define PK_FMAK(d, a, k, c, ...) PK_ASM("v_pk_fma_f32 %0, %1, %2, %3" PK_SEL3(__VA_ARGS__) : "=v"(d) : "v"(a), "s"(k), "v"(c))
// d = a + b, a - b, lane by lane
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { f2 d; PK_ASM("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) { f2 d; PK_ASM("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// (a.lo + b.hi, a.hi + b.lo) and the same with minus: the second operand's halves swapped
__device__ __forceinline__ f2 pk_add_swap(f2 a, f2 b) { f2 d; PK_ASM("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f2 pk_sub_swap(f2 a, f2 b) { f2 d; PK_ASM("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// (a.lo + a.hi, a.lo - a.hi)
__device__ __forceinline__ f2 pk_sum_diff(f2 a) { f2 d; PK_ASM("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }

// jm_fdct8<1, true> on TWO rows (a, b), each given as the pairs (x0,x1) (x2,x3) (x4,x5) (x6,x7); a
// row comes back as the frequency pairs (X0,X4) (X2,X6) (X1,X7) (X3,X5).  The two rows' steps
// alternate, and within a row no step follows the one it depends on: a result is read three or
// more instructions after it was written (the hazard needs two in between).
__device__ __forceinline__ void pk_fdct8_rows(f2 (&a)[4], f2 (&b)[4], const PkBank &k)
{
    f2 s01[2], d01[2], s23[2], d23[2], t01[2], t23[2], o17[2], o35[2], e[2], m[2];
    f2 (*const x[2])[4] = {&a, &b};
#define PK_BOTH(stmt)                  \
    _Pragma("unroll") for (int w = 0; w < 2; ++w) { stmt; }
    PK_BOTH(s01[w] = pk_add_swap((*x[w])[0], (*x[w])[3]))   // (s0,s1)
    PK_BOTH(d01[w] = pk_sub_swap((*x[w])[0], (*x[w])[3]))   // (d0,d1)
    PK_BOTH(s23[w] = pk_add_swap((*x[w])[1], (*x[w])[2]))   // (s2,s3)
    PK_BOTH(d23[w] = pk_sub_swap((*x[w])[1], (*x[w])[2]))   // (d2,d3)
    PK_BOTH(t01[w] = pk_add_swap(s01[w], s23[w]))           // (t0,t1)
    PK_BOTH(t23[w] = pk_sub_swap(s01[w], s23[w]))           // (t2,t3)
    PK_BOTH(PK_MULK(o17[w], d01[w], k.r17, 0, 0, 0, 1, 0, 0))          // d0 ( C1,  C7)
    PK_BOTH(e[w] = pk_sum_diff(t01[w]))                                // (t0+t1, t0-t1)
    PK_BOTH(PK_MULK(o35[w], d01[w], k.r35, 0, 0, 0, 1, 0, 0))          // d0 ( C3,  C5)
    PK_BOTH(PK_MULK(m[w], t23[w], k.r26, 1, 1, 1, 0, 0, 1))            // t3 ( C6, -C2)
    PK_BOTH(PK_FMAK(o17[w], d01[w], k.r35, o17[w], 1, 1, 0, 1, 0, 1))  // d1 ( C3, -C5)
    PK_BOTH(e[w].x = e[w].x + -1024.0f)                                // the level shift reaches u = 0 only
    PK_BOTH(PK_FMAK(o35[w], d01[w], k.r17, o35[w], 1, 1, 1, 0, 1, 1))  // d1 (-C7, -C1)
    PK_BOTH(PK_FMAK((*x[w])[1], t23[w], k.r26, m[w], 0, 0, 0, 1, 0, 0)) // t2 ( C2,  C6): (X2,X6)
    PK_BOTH(PK_FMAK(o17[w], d23[w], k.r35, o17[w], 0, 0, 1, 0, 0, 0))  // d2 ( C5,  C3)
    PK_BOTH(PK_MULK((*x[w])[0], e[w], k.a0, 0, 1, 0, 0, 0, 0))         // (X0,X4)
    PK_BOTH(PK_FMAK(o35[w], d23[w], k.r17, o35[w], 0, 0, 0, 1, 1, 0))  // d2 (-C1,  C7)
    PK_BOTH(PK_FMAK((*x[w])[2], d23[w], k.r17, o17[w], 1, 1, 1, 0, 0, 1)) // d3 ( C7, -C1): (X1,X7)
    PK_BOTH(PK_FMAK((*x[w])[3], d23[w], k.r35, o35[w], 1, 1, 1, 0, 1, 0)) // d3 (-C5,  C3): (X3,X5)
#undef PK_BOTH
}

// jm_fdct8<100, false> down two columns at once (x_r = the pair of the two columns' values in row r)
__device__ __forceinline__ void pk_fdct8_cols(f2 &x0, f2 &x1, f2 &x2, f2 &x3, f2 &x4, f2 &x5, f2 &x6, f2 &x7,
                                              const PkBank &k)
{
    // the constant's low half (LO) or high half (HI) for both results, plain or negated
#define LO 0, 1, 0, 0, 0, 0
#define HI 0, 1, 1, 1, 0, 0
#define NLO 0, 1, 0, 0, 1, 1
#define NHI 0, 1, 1, 1, 1, 1
    const f2 s0 = pk_add(x0, x7), s1 = pk_add(x1, x6), s2 = pk_add(x2, x5), s3 = pk_add(x3, x4);
    const f2 d0 = pk_sub(x0, x7), d1 = pk_sub(x1, x6), d2 = pk_sub(x2, x5), d3 = pk_sub(x3, x4);
    const f2 t0 = pk_add(s0, s3), t1 = pk_add(s1, s2), t2 = pk_sub(s0, s3), t3 = pk_sub(s1, s2);
    f2 o1, o3, o5, o7, m2, m6;
    PK_MULK(o1, d0, k.c17, LO); PK_MULK(o3, d0, k.c35, LO); PK_MULK(o5, d0, k.c35, HI); PK_MULK(o7, d0, k.c17, HI);
    const f2 a0 = pk_add(t0, t1), a4 = pk_sub(t0, t1);
    PK_MULK(m2, t3, k.c26, HI); PK_MULK(m6, t3, k.c26, NLO);
    PK_FMAK(o1, d1, k.c35, o1, LO); PK_FMAK(o3, d1, k.c17, o3, NHI);
    PK_FMAK(o5, d1, k.c17, o5, NLO); PK_FMAK(o7, d1, k.c35, o7, NHI);
    PK_MULK(x0, a0, k.a0, HI); PK_MULK(x4, a4, k.a0, HI);
    PK_FMAK(x2, t2, k.c26, m2, LO); PK_FMAK(x6, t2, k.c26, m6, HI);
    PK_FMAK(o1, d2, k.c35, o1, HI); PK_FMAK(o3, d2, k.c17, o3, NLO);
    PK_FMAK(o5, d2, k.c17, o5, HI); PK_FMAK(o7, d2, k.c35, o7, LO);
    PK_FMAK(x1, d3, k.c17, o1, HI); PK_FMAK(x3, d3, k.c35, o3, NHI);
    PK_FMAK(x5, d3, k.c35, o5, LO); PK_FMAK(x7, d3, k.c17, o7, NLO);
}

// block_fdct_quant on packed instructions.  v: 64 pixel values (row-major) on entry, MAGIC +
// quantised coefficient (natural order) on exit; inv100q_pk: the block's multipliers with every
// row of 8 in kPkOrder.
__device__ __forceinline__ void block_fdct_quant_pk(float (&v)[64], const float *inv100q_pk, const PkBank &k)
{
    f2 p[8][4]; // row r: pairs of columns (0,1) (2,3) (4,5) (6,7), then of frequencies (0,4) (2,6) (1,7) (3,5)
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            p[r][j] = f2{v[r * 8 + 2 * j], v[r * 8 + 2 * j + 1]};
#pragma unroll
    for (int r = 0; r < 8; r += 2)
        pk_fdct8_rows(p[r], p[r + 1], k);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        pk_fdct8_cols(p[0][j], p[1][j], p[2][j], p[3][j], p[4][j], p[5][j], p[6][j], p[7][j], k);
    const f2 *tab = reinterpret_cast<const f2 *>(inv100q_pk);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        // jm_quantise_block: r = (x + M) - M; q = fma(r, inv100q, M), a row of four pairs at a time
        f2 t[4], m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            m[j] = tab[r * 4 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            PK_ADDK(t[j], p[r][j], k.magic, LO);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            PK_ADDK(t[j], t[j], k.magic, NLO);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f2 q;
            PK_ASM("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(q) : "v"(t[j]), "v"(m[j]), "s"(k.magic));
            v[r * 8 + kPkOrder[2 * j]] = q.x;
            v[r * 8 + kPkOrder[2 * j + 1]] = q.y;
        }
    }
#undef LO
#undef HI
#undef NLO
#undef NHI
}

#endif
