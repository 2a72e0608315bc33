// Developer microbenchmark: issue cost of individual gfx950 VALU instructions (inline asm).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters, unsigned s)   \
    {                                                                                    \
        unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        unsigned long long w = ((unsigned long long)x0 << 32) | x1;                     \
        for (int it = 0; it < iters; ++it) {                                            \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                             \
                asm volatile(ASM(0) : "+v"(x0) : "v"(s), "v"(x7));                      \
                asm volatile(ASM(1) : "+v"(x1) : "v"(s), "v"(x6));                      \
                asm volatile(ASM(2) : "+v"(x2) : "v"(s), "v"(x5));                      \
                asm volatile(ASM(3) : "+v"(x3) : "v"(s), "v"(x4));                      \
                asm volatile(ASM(4) : "+v"(x4) : "v"(s), "v"(x3));                      \
                asm volatile(ASM(5) : "+v"(x5) : "v"(s), "v"(x2));                      \
                asm volatile(ASM(6) : "+v"(x6) : "v"(s), "v"(x1));                      \
                asm volatile(ASM(7) : "+v"(x7) : "v"(s), "v"(x0));                      \
            }                                                                            \
        }                                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (unsigned)w; \
    }
#define A_FMA(i) "v_fma_f32 %0, %0, %1, %2"
#define A_MADI24(i) "v_mad_i32_i24 %0, %0, %1, %2"
#define A_MADU24(i) "v_mad_u32_u24 %0, %0, %1, %2"
#define A_MULLO(i) "v_mul_lo_u32 %0, %0, %1"
#define A_MULHI(i) "v_mul_hi_u32 %0, %0, %1"
#define A_CVTF(i) "v_cvt_f32_u32 %0, %0"
#define A_CVTU(i) "v_cvt_u32_f32 %0, %0"
#define A_CVTUB(i) "v_cvt_f32_ubyte1 %0, %0"
#define A_BFE(i) "v_bfe_u32 %0, %0, 8, 8"
#define A_MIN3(i) "v_min3_u32 %0, %0, %1, %2"
#define A_RNDNE(i) "v_rndne_f32 %0, %0"
#define A_FRACT(i) "v_fract_f32 %0, %0"
#define A_BFI(i) "v_bfi_b32 %0, %1, %0, %2"
#define A_MED3(i) "v_med3_i32 %0, %0, %1, %2"
#define A_PERM(i) "v_perm_b32 %0, %0, %1, %2"
#define A_ADD3(i) "v_add3_u32 %0, %0, %1, %2"
#define A_LSHLOR(i) "v_lshl_or_b32 %0, %0, 8, %2"
#define A_ADDF(i) "v_add_f32 %0, %0, %1"
#define A_ADDU(i) "v_add_u32 %0, %0, %1"
#define A_CVTI(i) "v_cvt_i32_f32 %0, %0"
#define A_TRUNC(i) "v_trunc_f32 %0, %0"
#define A_SATPK(i) "v_sat_pk_u8_i16 %0, %0"
#define A_PKADDI16(i) "v_pk_add_i16 %0, %0, %1"
#define A_CVTPKU8(i) "v_cvt_pk_u8_f32 %0, %0, 1, %2"
#define A_CVTSDWA(i) "v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define A_CNDMASK(i) "v_cndmask_b32 %0, %0, %1, vcc"
#define A_CMP(i) "v_cmp_eq_u32 vcc, %0, %1"
KERNEL(k_fma, A_FMA) KERNEL(k_madi24, A_MADI24) KERNEL(k_madu24, A_MADU24) KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI) KERNEL(k_cvtf, A_CVTF) KERNEL(k_cvtu, A_CVTU) KERNEL(k_cvtub, A_CVTUB)
KERNEL(k_bfe, A_BFE) KERNEL(k_min3, A_MIN3) KERNEL(k_rndne, A_RNDNE) KERNEL(k_fract, A_FRACT)
KERNEL(k_bfi, A_BFI) KERNEL(k_med3, A_MED3) KERNEL(k_perm, A_PERM) KERNEL(k_add3, A_ADD3)
KERNEL(k_lshlor, A_LSHLOR) KERNEL(k_addf, A_ADDF) KERNEL(k_addu, A_ADDU) KERNEL(k_cvti, A_CVTI)
KERNEL(k_trunc, A_TRUNC) KERNEL(k_satpk, A_SATPK) KERNEL(k_pkaddi16, A_PKADDI16) KERNEL(k_cvtpku8, A_CVTPKU8)
KERNEL(k_cvtsdwa, A_CVTSDWA) KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cmp, A_CMP)

#define B_AND(i) "v_and_b32 %0, %0, %1"
#define B_OR(i) "v_or_b32 %0, %0, %1"
#define B_LSHL(i) "v_lshlrev_b32 %0, 3, %0"
#define B_LSHR(i) "v_lshrrev_b32 %0, 3, %0"
#define B_MULF(i) "v_mul_f32 %0, %0, %1"
#define B_MAXF(i) "v_max_f32 %0, %0, %1"
#define B_MOV(i) "v_mov_b32 %0, %2"
#define B_SUBF(i) "v_sub_f32 %0, %0, %1"
#define B_MULU24(i) "v_mul_u32_u24 %0, %0, %1"
#define B_MULI24(i) "v_mul_i32_i24 %0, %0, %1"
#define B_XOR(i) "v_xor_b32 %0, %0, %1"
#define B_ASHR(i) "v_ashrrev_i32 %0, 3, %0"
#define B_CVTPKI16(i) "v_cvt_pk_i16_i32 %0, %0, %1"
#define B_ANDOR(i) "v_and_or_b32 %0, %0, %1, %2"
#define B_ALIGNB(i) "v_alignbyte_b32 %0, %0, %1, 1"
#define B_FMAC(i) "v_fmac_f32 %0, %1, %2"
#define B_FMAK(i) "v_fmaak_f32 %0, %0, %1, 0x3f000000"
#define B_MAXI(i) "v_max_i32 %0, %0, %1"
#define B_MINI(i) "v_min_u32 %0, %0, %1"
#define B_SUBU(i) "v_sub_u32 %0, %0, %1"
#define B_LSHLADD(i) "v_lshl_add_u32 %0, %0, 2, %1"
#define B_FLOOR(i) "v_floor_f32 %0, %0"
KERNEL(k_and, B_AND) KERNEL(k_or, B_OR) KERNEL(k_lshl, B_LSHL) KERNEL(k_lshr, B_LSHR) KERNEL(k_mulf, B_MULF) KERNEL(k_maxf, B_MAXF) KERNEL(k_mov, B_MOV) KERNEL(k_subf, B_SUBF) KERNEL(k_mulu24, B_MULU24) KERNEL(k_muli24, B_MULI24) KERNEL(k_xor, B_XOR) KERNEL(k_ashr, B_ASHR) KERNEL(k_cvtpki16, B_CVTPKI16) KERNEL(k_andor, B_ANDOR) KERNEL(k_alignb, B_ALIGNB) KERNEL(k_fmac, B_FMAC) KERNEL(k_fmak, B_FMAK) KERNEL(k_maxi, B_MAXI) KERNEL(k_mini, B_MINI) KERNEL(k_subu, B_SUBU) KERNEL(k_lshladd, B_LSHLADD) KERNEL(k_floor, B_FLOOR) 
typedef void (*kern_t)(unsigned *, int, unsigned);
static double base_ms = 0;
void run(const char *name, kern_t k)
{
    unsigned *d; (void)hipMalloc(&d, 4096 * 256 * 4);
    int iters = 1000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, 10, 3u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, iters, 3u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (base_ms == 0) base_ms = ms;
    printf("%-18s %.3f ms  relative to v_fma_f32: %.2fx\n", name, ms, ms / base_ms);
    (void)hipFree(d);
}
int main()
{
#define R(n) run(#n, n);
    R(k_fma) R(k_madi24) R(k_madu24) R(k_mullo) R(k_mulhi) R(k_cvtf) R(k_cvtu) R(k_cvtub) R(k_bfe) R(k_min3)
    R(k_rndne) R(k_fract) R(k_bfi) R(k_med3) R(k_perm) R(k_add3) R(k_lshlor) R(k_addf) R(k_addu) R(k_cvti)
    R(k_trunc) R(k_satpk) R(k_pkaddi16) R(k_cvtpku8) R(k_cvtsdwa) R(k_cndmask) R(k_cmp)
    R(k_and) R(k_or) R(k_lshl) R(k_lshr) R(k_mulf) R(k_maxf) R(k_mov) R(k_subf) R(k_mulu24) R(k_muli24) R(k_xor) R(k_ashr) R(k_cvtpki16) R(k_andor) R(k_alignb) R(k_fmac) R(k_fmak) R(k_maxi) R(k_mini) R(k_subu) R(k_lshladd) R(k_floor) 
    return 0;
}
