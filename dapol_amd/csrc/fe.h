// GF(2^255-19) arithmetic for gfx950: nine signed limbs, radix 2^29 (limb i has weight 2^(29 i); 9 x 29 = 261 bits, so
// a reduced value keeps 23 bits in limb 8 and 2^261 = 2^6 * 2^255 wraps to 1216).
//
// Why this form (measured, tools/ubench_valu.hip on MI355X): v_mad_{u,i}64_{u,i}32 issues at ~4.2-5 cycles per wave, every
// other VOP3 op at ~3.6-4.3, plain VOP2 logic at ~2: a 255-bit product costs what its MULTIPLY-ADDs cost, so the cheapest
// representation is the one with the fewest of them whose column sums still ride the MAD's 64-bit addend.  Round 1 used
// the classic 10 x 25.5-bit limbs: 100 MADs + 9 multiplies by 19 + 10 carries.  Nine 29-bit limbs need 81 MADs; the high
// columns 9..16 cannot be pre-multiplied by the wrap factor (1216 * 2^29 does not fit 32 bits), so they are carried into
// limbs first and re-enter through 9 more MADs by 1216: 90 MADs + 17 carries, ~8 % fewer issue cycles per product
// (profiles/r02_fe29_ab.txt), and a point is 36 registers instead of 40.  fe_sq: 45 + 9 MADs.
//
// Bounds (floor carries, so "reduced" limbs are non-negative: limbs 0..7 in [0, 2^29 + 2^17), limb 8 in [0, 2^23]):
//   TIGHT  = |limb| <= 2^29 + 2^17  (a reduced value, its negation, or the DIFFERENCE of two reduced values);
//   LOOSE  = up to 3x a reduced value in magnitude (2Z - C, B + A, ...).
//   fe_mul(h, f, g): f may be LOOSE, g must be TIGHT: a column is at most 9 products of 3*2^29 * 2^29 = 27 * 2^58 < 2^63.
//   fe_sq(h, f):     f must be TIGHT (the doubled limbs 2 f_i must fit int32; columns <= 9 * 2^59).
//   Outputs of fe_mul / fe_sq / fe_carry are reduced.  A SUM of two reduced values is loose: carry it before it is squared
//   or used as g.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DAPOL_HD __host__ __device__ __forceinline__
#define DAPOL_HD_NOINLINE __host__ __device__ __noinline__
#else
#define DAPOL_HD inline
#define DAPOL_HD_NOINLINE inline
#endif

namespace dapol {

enum { FE_NL = 9, FE_M29 = 0x1fffffff, FE_M23 = 0x7fffff };

struct fe {
    int32_t v[FE_NL];
};

DAPOL_HD void fe_0(fe& h) {
    for (int i = 0; i < FE_NL; i++) h.v[i] = 0;
}
DAPOL_HD void fe_1(fe& h) {
    fe_0(h);
    h.v[0] = 1;
}
DAPOL_HD void fe_add(fe& h, const fe& f, const fe& g) {
    for (int i = 0; i < FE_NL; i++) h.v[i] = f.v[i] + g.v[i];
}
DAPOL_HD void fe_sub(fe& h, const fe& f, const fe& g) {
    for (int i = 0; i < FE_NL; i++) h.v[i] = f.v[i] - g.v[i];
}
DAPOL_HD void fe_neg(fe& h, const fe& f) {
    for (int i = 0; i < FE_NL; i++) h.v[i] = -f.v[i];
}
// h = c ? g : h  (branch-free select)
DAPOL_HD void fe_cmov(fe& h, const fe& g, bool c) {
    for (int i = 0; i < FE_NL; i++) h.v[i] = c ? g.v[i] : h.v[i];
}
DAPOL_HD void fe_cswap(fe& f, fe& g, bool c) {
    for (int i = 0; i < FE_NL; i++) {
        int32_t a = f.v[i], b = g.v[i];
        f.v[i] = c ? b : a;
        g.v[i] = c ? a : b;
    }
}

// Limbs from the nine low columns whose carries are already chained (c_k includes c_(k-1) >> 29): limbs 0..7 are the low 29
// bits; column 8 keeps 23 bits and everything above bit 255 re-enters limb 0 times 19.
DAPOL_HD void fe_limbs_from_cols(fe& h, int64_t c0, int64_t c1, int64_t c2, int64_t c3, int64_t c4, int64_t c5, int64_t c6, int64_t c7,
                                 int64_t c8) {
    const int64_t t = (c8 >> 23) * 19 + (int64_t)((int32_t)c0 & FE_M29);
    h.v[0] = (int32_t)t & FE_M29;
    h.v[1] = ((int32_t)c1 & FE_M29) + (int32_t)(t >> 29);
    h.v[2] = (int32_t)c2 & FE_M29; h.v[3] = (int32_t)c3 & FE_M29; h.v[4] = (int32_t)c4 & FE_M29; h.v[5] = (int32_t)c5 & FE_M29;
    h.v[6] = (int32_t)c6 & FE_M29; h.v[7] = (int32_t)c7 & FE_M29;
    h.v[8] = (int32_t)c8 & FE_M23;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(DAPOL_NO_MAD_CHAIN)
// Device form of the column sums.  Every column is ONE chain of v_mad_i64_i32 whose first addend is the carry out of the
// column below, so no 64-bit carry addition is ever issued on its own.  Inline assembly, one statement per column, because
// LLVM re-associates a C sum so that the carry is added last, as a separate instruction; one statement per column (not per
// MAD) keeps the hazard recogniser from padding the dependent MADs with s_nop.  madNc: N products added to a carry;
// madNz: N products from zero; suffix l / h: plus the low (unsigned) / high (signed) half of a wrapped high column times its
// constant.  (Generated once; the second destination is VOP3b's unused scalar carry-out.)
#define DAPOL_MAD_CHAIN 1
__device__ __forceinline__ int64_t mad1z(int32_t a0, int32_t b0) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0));
    return d;
}

__device__ __forceinline__ int64_t mad1zl(int32_t a0, int32_t b0, uint32_t lo, uint32_t kl) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %5, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(lo), "v"(kl));
    return d;
}

__device__ __forceinline__ int64_t mad1clh(int64_t d, int32_t a0, int32_t b0, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad2z(int32_t a0, int32_t b0, int32_t a1, int32_t b1) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
    return d;
}

__device__ __forceinline__ int64_t mad2clh(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_u64_u32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad3z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2));
    return d;
}

__device__ __forceinline__ int64_t mad3clh(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_u64_u32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad4z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
    return d;
}

__device__ __forceinline__ int64_t mad4clh(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_u64_u32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad5z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4));
    return d;
}

__device__ __forceinline__ int64_t mad5ch(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad5clh(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_u64_u32 %0, %1, %12, %13, %0\n\t"
        "v_mad_i64_i32 %0, %1, %14, %15, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad6z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5));
    return d;
}

__device__ __forceinline__ int64_t mad6clh(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0\n\t"
        "v_mad_u64_u32 %0, %1, %14, %15, %0\n\t"
        "v_mad_i64_i32 %0, %1, %16, %17, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad7z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, int32_t a6, int32_t b6) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0\n\t"
        "v_mad_i64_i32 %0, %1, %14, %15, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6));
    return d;
}

__device__ __forceinline__ int64_t mad7clh(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, int32_t a6, int32_t b6, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0\n\t"
        "v_mad_i64_i32 %0, %1, %14, %15, %0\n\t"
        "v_mad_u64_u32 %0, %1, %16, %17, %0\n\t"
        "v_mad_i64_i32 %0, %1, %18, %19, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad8z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, int32_t a6, int32_t b6, int32_t a7, int32_t b7) {
    int64_t d;
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0\n\t"
        "v_mad_i64_i32 %0, %1, %14, %15, %0\n\t"
        "v_mad_i64_i32 %0, %1, %16, %17, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7));
    return d;
}

__device__ __forceinline__ int64_t mad8clh(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, int32_t a6, int32_t b6, int32_t a7, int32_t b7, uint32_t lo, uint32_t kl, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0\n\t"
        "v_mad_i64_i32 %0, %1, %14, %15, %0\n\t"
        "v_mad_i64_i32 %0, %1, %16, %17, %0\n\t"
        "v_mad_u64_u32 %0, %1, %18, %19, %0\n\t"
        "v_mad_i64_i32 %0, %1, %20, %21, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7), "v"(lo), "v"(kl), "v"(hi), "v"(kh));
    return d;
}

__device__ __forceinline__ int64_t mad9ch(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, int32_t a6, int32_t b6, int32_t a7, int32_t b7, int32_t a8, int32_t b8, int32_t hi, int32_t kh) {
    uint64_t sdst;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0\n\t"
        "v_mad_i64_i32 %0, %1, %14, %15, %0\n\t"
        "v_mad_i64_i32 %0, %1, %16, %17, %0\n\t"
        "v_mad_i64_i32 %0, %1, %18, %19, %0\n\t"
        "v_mad_i64_i32 %0, %1, %20, %21, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7), "v"(a8), "v"(b8), "v"(hi), "v"(kh));
    return d;
}
#endif

#define M64(a, b) ((int64_t)(a) * (int64_t)(b))

DAPOL_HD void fe_mul(fe& h, const fe& f, const fe& g) {
    const int32_t f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8];
    const int32_t g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4], g5 = g.v[5], g6 = g.v[6], g7 = g.v[7], g8 = g.v[8];
#if defined(DAPOL_MAD_CHAIN)
    const int32_t k1216 = 1216, k9728 = 9728;
    const int64_t c9 = mad8z(f1, g8, f2, g7, f3, g6, f4, g5, f5, g4, f6, g3, f7, g2, f8, g1);
    const int64_t c10 = mad7z(f2, g8, f3, g7, f4, g6, f5, g5, f6, g4, f7, g3, f8, g2);
    const int64_t c11 = mad6z(f3, g8, f4, g7, f5, g6, f6, g5, f7, g4, f8, g3);
    const int64_t c12 = mad5z(f4, g8, f5, g7, f6, g6, f7, g5, f8, g4);
    const int64_t c13 = mad4z(f5, g8, f6, g7, f7, g6, f8, g5);
    const int64_t c14 = mad3z(f6, g8, f7, g7, f8, g6);
    const int64_t c15 = mad2z(f7, g8, f8, g7);
    const int64_t c16 = mad1z(f8, g8);
    const int64_t c0 = mad1zl(f0, g0, (uint32_t)c9, k1216);
    const int64_t c1 = mad2clh(c0 >> 29, f0, g1, f1, g0, (uint32_t)c10, k1216, (int32_t)(c9 >> 32), k9728);
    const int64_t c2 = mad3clh(c1 >> 29, f0, g2, f1, g1, f2, g0, (uint32_t)c11, k1216, (int32_t)(c10 >> 32), k9728);
    const int64_t c3 = mad4clh(c2 >> 29, f0, g3, f1, g2, f2, g1, f3, g0, (uint32_t)c12, k1216, (int32_t)(c11 >> 32), k9728);
    const int64_t c4 = mad5clh(c3 >> 29, f0, g4, f1, g3, f2, g2, f3, g1, f4, g0, (uint32_t)c13, k1216, (int32_t)(c12 >> 32), k9728);
    const int64_t c5 = mad6clh(c4 >> 29, f0, g5, f1, g4, f2, g3, f3, g2, f4, g1, f5, g0, (uint32_t)c14, k1216, (int32_t)(c13 >> 32), k9728);
    const int64_t c6 = mad7clh(c5 >> 29, f0, g6, f1, g5, f2, g4, f3, g3, f4, g2, f5, g1, f6, g0, (uint32_t)c15, k1216, (int32_t)(c14 >> 32), k9728);
    const int64_t c7 = mad8clh(c6 >> 29, f0, g7, f1, g6, f2, g5, f3, g4, f4, g3, f5, g2, f6, g1, f7, g0, (uint32_t)c16, k1216, (int32_t)(c15 >> 32), k9728);
    const int64_t c8 = mad9ch(c7 >> 29, f0, g8, f1, g7, f2, g6, f3, g5, f4, g4, f5, g3, f6, g2, f7, g1, f8, g0, (int32_t)(c16 >> 32), k9728);
#else
    const int64_t c9 = M64(f1, g8) + M64(f2, g7) + M64(f3, g6) + M64(f4, g5) + M64(f5, g4) + M64(f6, g3) + M64(f7, g2) + M64(f8, g1);
    const int64_t c10 = M64(f2, g8) + M64(f3, g7) + M64(f4, g6) + M64(f5, g5) + M64(f6, g4) + M64(f7, g3) + M64(f8, g2);
    const int64_t c11 = M64(f3, g8) + M64(f4, g7) + M64(f5, g6) + M64(f6, g5) + M64(f7, g4) + M64(f8, g3);
    const int64_t c12 = M64(f4, g8) + M64(f5, g7) + M64(f6, g6) + M64(f7, g5) + M64(f8, g4);
    const int64_t c13 = M64(f5, g8) + M64(f6, g7) + M64(f7, g6) + M64(f8, g5);
    const int64_t c14 = M64(f6, g8) + M64(f7, g7) + M64(f8, g6);
    const int64_t c15 = M64(f7, g8) + M64(f8, g7);
    const int64_t c16 = M64(f8, g8);
    const int64_t c0 = M64(f0, g0) + (int64_t)((uint64_t)(uint32_t)c9 * 1216u);
    const int64_t c1 = (c0 >> 29) + M64(f0, g1) + M64(f1, g0) + (int64_t)((uint64_t)(uint32_t)c10 * 1216u) + M64((int32_t)(c9 >> 32), 9728);
    const int64_t c2 = (c1 >> 29) + M64(f0, g2) + M64(f1, g1) + M64(f2, g0) + (int64_t)((uint64_t)(uint32_t)c11 * 1216u) + M64((int32_t)(c10 >> 32), 9728);
    const int64_t c3 = (c2 >> 29) + M64(f0, g3) + M64(f1, g2) + M64(f2, g1) + M64(f3, g0) + (int64_t)((uint64_t)(uint32_t)c12 * 1216u) + M64((int32_t)(c11 >> 32), 9728);
    const int64_t c4 = (c3 >> 29) + M64(f0, g4) + M64(f1, g3) + M64(f2, g2) + M64(f3, g1) + M64(f4, g0) + (int64_t)((uint64_t)(uint32_t)c13 * 1216u) + M64((int32_t)(c12 >> 32), 9728);
    const int64_t c5 = (c4 >> 29) + M64(f0, g5) + M64(f1, g4) + M64(f2, g3) + M64(f3, g2) + M64(f4, g1) + M64(f5, g0) + (int64_t)((uint64_t)(uint32_t)c14 * 1216u) + M64((int32_t)(c13 >> 32), 9728);
    const int64_t c6 = (c5 >> 29) + M64(f0, g6) + M64(f1, g5) + M64(f2, g4) + M64(f3, g3) + M64(f4, g2) + M64(f5, g1) + M64(f6, g0) + (int64_t)((uint64_t)(uint32_t)c15 * 1216u) + M64((int32_t)(c14 >> 32), 9728);
    const int64_t c7 = (c6 >> 29) + M64(f0, g7) + M64(f1, g6) + M64(f2, g5) + M64(f3, g4) + M64(f4, g3) + M64(f5, g2) + M64(f6, g1) + M64(f7, g0) + (int64_t)((uint64_t)(uint32_t)c16 * 1216u) + M64((int32_t)(c15 >> 32), 9728);
    const int64_t c8 = (c7 >> 29) + M64(f0, g8) + M64(f1, g7) + M64(f2, g6) + M64(f3, g5) + M64(f4, g4) + M64(f5, g3) + M64(f6, g2) + M64(f7, g1) + M64(f8, g0) + M64((int32_t)(c16 >> 32), 9728);
#endif
    fe_limbs_from_cols(h, c0, c1, c2, c3, c4, c5, c6, c7, c8);
}

DAPOL_HD void fe_sq(fe& h, const fe& f) {
    const int32_t f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8];
    const int32_t f0_2 = 2 * f0, f1_2 = 2 * f1, f2_2 = 2 * f2, f3_2 = 2 * f3, f4_2 = 2 * f4, f5_2 = 2 * f5, f6_2 = 2 * f6, f7_2 = 2 * f7;
#if defined(DAPOL_MAD_CHAIN)
    const int32_t k1216 = 1216, k9728 = 9728;
    const int64_t c9 = mad4z(f1_2, f8, f2_2, f7, f3_2, f6, f4_2, f5);
    const int64_t c10 = mad4z(f2_2, f8, f3_2, f7, f4_2, f6, f5, f5);
    const int64_t c11 = mad3z(f3_2, f8, f4_2, f7, f5_2, f6);
    const int64_t c12 = mad3z(f4_2, f8, f5_2, f7, f6, f6);
    const int64_t c13 = mad2z(f5_2, f8, f6_2, f7);
    const int64_t c14 = mad2z(f6_2, f8, f7, f7);
    const int64_t c15 = mad1z(f7_2, f8);
    const int64_t c16 = mad1z(f8, f8);
    const int64_t c0 = mad1zl(f0, f0, (uint32_t)c9, k1216);
    const int64_t c1 = mad1clh(c0 >> 29, f0_2, f1, (uint32_t)c10, k1216, (int32_t)(c9 >> 32), k9728);
    const int64_t c2 = mad2clh(c1 >> 29, f0_2, f2, f1, f1, (uint32_t)c11, k1216, (int32_t)(c10 >> 32), k9728);
    const int64_t c3 = mad2clh(c2 >> 29, f0_2, f3, f1_2, f2, (uint32_t)c12, k1216, (int32_t)(c11 >> 32), k9728);
    const int64_t c4 = mad3clh(c3 >> 29, f0_2, f4, f1_2, f3, f2, f2, (uint32_t)c13, k1216, (int32_t)(c12 >> 32), k9728);
    const int64_t c5 = mad3clh(c4 >> 29, f0_2, f5, f1_2, f4, f2_2, f3, (uint32_t)c14, k1216, (int32_t)(c13 >> 32), k9728);
    const int64_t c6 = mad4clh(c5 >> 29, f0_2, f6, f1_2, f5, f2_2, f4, f3, f3, (uint32_t)c15, k1216, (int32_t)(c14 >> 32), k9728);
    const int64_t c7 = mad4clh(c6 >> 29, f0_2, f7, f1_2, f6, f2_2, f5, f3_2, f4, (uint32_t)c16, k1216, (int32_t)(c15 >> 32), k9728);
    const int64_t c8 = mad5ch(c7 >> 29, f0_2, f8, f1_2, f7, f2_2, f6, f3_2, f5, f4, f4, (int32_t)(c16 >> 32), k9728);
#else
    const int64_t c9 = M64(f1_2, f8) + M64(f2_2, f7) + M64(f3_2, f6) + M64(f4_2, f5);
    const int64_t c10 = M64(f2_2, f8) + M64(f3_2, f7) + M64(f4_2, f6) + M64(f5, f5);
    const int64_t c11 = M64(f3_2, f8) + M64(f4_2, f7) + M64(f5_2, f6);
    const int64_t c12 = M64(f4_2, f8) + M64(f5_2, f7) + M64(f6, f6);
    const int64_t c13 = M64(f5_2, f8) + M64(f6_2, f7);
    const int64_t c14 = M64(f6_2, f8) + M64(f7, f7);
    const int64_t c15 = M64(f7_2, f8);
    const int64_t c16 = M64(f8, f8);
    const int64_t c0 = M64(f0, f0) + (int64_t)((uint64_t)(uint32_t)c9 * 1216u);
    const int64_t c1 = (c0 >> 29) + M64(f0_2, f1) + (int64_t)((uint64_t)(uint32_t)c10 * 1216u) + M64((int32_t)(c9 >> 32), 9728);
    const int64_t c2 = (c1 >> 29) + M64(f0_2, f2) + M64(f1, f1) + (int64_t)((uint64_t)(uint32_t)c11 * 1216u) + M64((int32_t)(c10 >> 32), 9728);
    const int64_t c3 = (c2 >> 29) + M64(f0_2, f3) + M64(f1_2, f2) + (int64_t)((uint64_t)(uint32_t)c12 * 1216u) + M64((int32_t)(c11 >> 32), 9728);
    const int64_t c4 = (c3 >> 29) + M64(f0_2, f4) + M64(f1_2, f3) + M64(f2, f2) + (int64_t)((uint64_t)(uint32_t)c13 * 1216u) + M64((int32_t)(c12 >> 32), 9728);
    const int64_t c5 = (c4 >> 29) + M64(f0_2, f5) + M64(f1_2, f4) + M64(f2_2, f3) + (int64_t)((uint64_t)(uint32_t)c14 * 1216u) + M64((int32_t)(c13 >> 32), 9728);
    const int64_t c6 = (c5 >> 29) + M64(f0_2, f6) + M64(f1_2, f5) + M64(f2_2, f4) + M64(f3, f3) + (int64_t)((uint64_t)(uint32_t)c15 * 1216u) + M64((int32_t)(c14 >> 32), 9728);
    const int64_t c7 = (c6 >> 29) + M64(f0_2, f7) + M64(f1_2, f6) + M64(f2_2, f5) + M64(f3_2, f4) + (int64_t)((uint64_t)(uint32_t)c16 * 1216u) + M64((int32_t)(c15 >> 32), 9728);
    const int64_t c8 = (c7 >> 29) + M64(f0_2, f8) + M64(f1_2, f7) + M64(f2_2, f6) + M64(f3_2, f5) + M64(f4, f4) + M64((int32_t)(c16 >> 32), 9728);
#endif
    fe_limbs_from_cols(h, c0, c1, c2, c3, c4, c5, c6, c7, c8);
}
#undef M64

// Weak carry of a LOOSE value (|limb| < 2^31 / 20) back to reduced form, all in 32-bit ops.
DAPOL_HD void fe_carry(fe& h, const fe& f) {
    int32_t h0 = f.v[0], h1 = f.v[1], h2 = f.v[2], h3 = f.v[3], h4 = f.v[4], h5 = f.v[5], h6 = f.v[6], h7 = f.v[7], h8 = f.v[8];
    h1 += h0 >> 29; h0 &= FE_M29;
    h2 += h1 >> 29; h1 &= FE_M29;
    h3 += h2 >> 29; h2 &= FE_M29;
    h4 += h3 >> 29; h3 &= FE_M29;
    h5 += h4 >> 29; h4 &= FE_M29;
    h6 += h5 >> 29; h5 &= FE_M29;
    h7 += h6 >> 29; h6 &= FE_M29;
    h8 += h7 >> 29; h7 &= FE_M29;
    h0 += 19 * (h8 >> 23); h8 &= FE_M23;
    h1 += h0 >> 29; h0 &= FE_M29;
    h.v[0] = h0; h.v[1] = h1; h.v[2] = h2; h.v[3] = h3; h.v[4] = h4; h.v[5] = h5; h.v[6] = h6; h.v[7] = h7; h.v[8] = h8;
}
// carried sum: reduced output from two reduced (or tight) inputs
DAPOL_HD void fe_addc(fe& h, const fe& f, const fe& g) {
    fe t;
    fe_add(t, f, g);
    fe_carry(h, t);
}

DAPOL_HD void fe_sqn(fe& h, const fe& f, int n) {
    fe_sq(h, f);
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

// z^(2^252 - 3) = z^((p-5)/8)
DAPOL_HD_NOINLINE void fe_pow22523(fe& out, const fe& z) {
    fe t0, t1, t2;
    fe_sq(t0, z);            // 2
    fe_sqn(t1, t0, 2);       // 8
    fe_mul(t1, z, t1);       // 9
    fe_mul(t0, t0, t1);      // 11
    fe_sq(t0, t0);           // 22
    fe_mul(t0, t1, t0);      // 31 = 2^5-1
    fe_sqn(t1, t0, 5);
    fe_mul(t0, t1, t0);      // 2^10-1
    fe_sqn(t1, t0, 10);
    fe_mul(t1, t1, t0);      // 2^20-1
    fe_sqn(t2, t1, 20);
    fe_mul(t1, t2, t1);      // 2^40-1
    fe_sqn(t1, t1, 10);
    fe_mul(t0, t1, t0);      // 2^50-1
    fe_sqn(t1, t0, 50);
    fe_mul(t1, t1, t0);      // 2^100-1
    fe_sqn(t2, t1, 100);
    fe_mul(t1, t2, t1);      // 2^200-1
    fe_sqn(t1, t1, 50);
    fe_mul(t0, t1, t0);      // 2^250-1
    fe_sqn(t0, t0, 2);       // 2^252-4
    fe_mul(out, t0, z);      // 2^252-3
}

// z^(p-2)
DAPOL_HD_NOINLINE void fe_invert(fe& out, const fe& z) {
    fe t, z3;
    fe_pow22523(t, z);       // z^(2^252-3)
    fe_sqn(t, t, 3);         // z^(2^255-24)
    fe_sq(z3, z);
    fe_mul(z3, z3, z);       // z^3
    fe_mul(out, t, z3);      // z^(2^255-21)
}

// Eight little-endian words of the canonical (fully reduced) value.  Accepts any tight / loose input, negative limbs
// included: two signed carry passes bring the value into [0, 2^255 + 19], the last step subtracts p when it is >= p.
DAPOL_HD void fe_towords(uint32_t* w, const fe& f) {
    int32_t l[FE_NL];
    for (int i = 0; i < FE_NL; i++) l[i] = f.v[i];
    for (int pass = 0; pass < 3; pass++) {
        for (int i = 0; i < 8; i++) { l[i + 1] += l[i] >> 29; l[i] &= FE_M29; }
        l[0] += 19 * (l[8] >> 23);
        l[8] &= FE_M23;
    }
    // now limbs 1..8 are canonical-width, l0 in [0, 2^29 + 19]; value in [0, 2^255 + 19).  q = 1 iff value >= p.
    int32_t q = (l[0] + 19) >> 29;
    for (int i = 1; i < 8; i++) q = (l[i] + q) >> 29;
    q = (l[8] + q) >> 23;
    l[0] += 19 * q;
    for (int i = 0; i < 8; i++) { l[i + 1] += l[i] >> 29; l[i] &= FE_M29; }
    l[8] &= FE_M23;                                             // drops q * 2^255
    const uint32_t u0 = (uint32_t)l[0], u1 = (uint32_t)l[1], u2 = (uint32_t)l[2], u3 = (uint32_t)l[3], u4 = (uint32_t)l[4],
                   u5 = (uint32_t)l[5], u6 = (uint32_t)l[6], u7 = (uint32_t)l[7], u8 = (uint32_t)l[8];
    w[0] = u0 | (u1 << 29);                  // bits   0.. 31: limb 0 (29) + 3 of limb 1
    w[1] = (u1 >> 3) | (u2 << 26);           // bits  32.. 63: 26 of limb 1 + 6 of limb 2
    w[2] = (u2 >> 6) | (u3 << 23);           // bits  64.. 95: 23 + 9
    w[3] = (u3 >> 9) | (u4 << 20);           // bits  96..127: 20 + 12
    w[4] = (u4 >> 12) | (u5 << 17);          // bits 128..159: 17 + 15
    w[5] = (u5 >> 15) | (u6 << 14);          // bits 160..191: 14 + 18
    w[6] = (u6 >> 18) | (u7 << 11);          // bits 192..223: 11 + 21
    w[7] = (u7 >> 21) | (u8 << 8);           // bits 224..255:  8 + 23 (+ bit 255 = 0)
}

// Canonical little-endian 32-byte encoding (fully reduced).
DAPOL_HD void fe_tobytes(uint8_t* s, const fe& f) {
    uint32_t w[8];
    fe_towords(w, f);
    for (int i = 0; i < 8; i++) {
        s[4 * i] = (uint8_t)w[i];
        s[4 * i + 1] = (uint8_t)(w[i] >> 8);
        s[4 * i + 2] = (uint8_t)(w[i] >> 16);
        s[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}

// Load from eight little-endian words; bit 255 is ignored (dalek FieldElement::from_bytes semantics).
DAPOL_HD void fe_fromwords(fe& h, const uint32_t* w) {
    h.v[0] = (int32_t)(w[0] & FE_M29);
    h.v[1] = (int32_t)(((w[0] >> 29) | (w[1] << 3)) & FE_M29);
    h.v[2] = (int32_t)(((w[1] >> 26) | (w[2] << 6)) & FE_M29);
    h.v[3] = (int32_t)(((w[2] >> 23) | (w[3] << 9)) & FE_M29);
    h.v[4] = (int32_t)(((w[3] >> 20) | (w[4] << 12)) & FE_M29);
    h.v[5] = (int32_t)(((w[4] >> 17) | (w[5] << 15)) & FE_M29);
    h.v[6] = (int32_t)(((w[5] >> 14) | (w[6] << 18)) & FE_M29);
    h.v[7] = (int32_t)(((w[6] >> 11) | (w[7] << 21)) & FE_M29);
    h.v[8] = (int32_t)((w[7] >> 8) & FE_M23);
}

DAPOL_HD void fe_frombytes(fe& h, const uint8_t* s) {
    uint32_t w[8];
    for (int i = 0; i < 8; i++)
        w[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
    fe_fromwords(h, w);
}

// "Negative" = least significant bit of the canonical encoding (RFC 9496 IS_NEGATIVE).
DAPOL_HD bool fe_isnegative(const fe& f) {
    uint32_t w[8];
    fe_towords(w, f);
    return w[0] & 1;
}
DAPOL_HD bool fe_iszero(const fe& f) {
    uint32_t w[8];
    fe_towords(w, f);
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r |= w[i];
    return r == 0;
}
DAPOL_HD bool fe_equal(const fe& f, const fe& g) {
    fe d;
    fe_sub(d, f, g);
    return fe_iszero(d);
}
// |f| as a REDUCED value (non-negative limbs): a bare negation would leave negative limbs, and a later Y - X with
// such an X is no longer tight (this bit ge_add on decompressed points before the carry was added).
DAPOL_HD void fe_abs(fe& h, const fe& f) {
    bool n = fe_isnegative(f);
    fe m;
    fe_neg(m, f);
    fe_carry(m, m);
    h = f;
    fe_cmov(h, m, n);
}

}  // namespace dapol
