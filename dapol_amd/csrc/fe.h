// GF(2^255-19) arithmetic for gfx950: ten signed limbs, radix 2^25.5 (limb i holds ceil(25.5 i) .. bits).
//
// Why this form (measured, tools/ubench_valu.hip on MI355X): v_mad_{u,i}64_{u,i}32 issues at ~4.2-5 cycles
// per wave, the same as v_fma_f64 and every other VOP3 op, while carry-producing adds (v_add_co/v_addc) cost
// as much as a multiply.  So the cheapest 255-bit multiply is the one with the fewest *instructions*: an
// unsaturated radix whose column sums fit the MAD's 64-bit addend (no carry instructions inside a column),
// with the 2^255 = 19 wrap folded into pre-multiplied operands that still fit 32 bits.  That is the classic
// 10 x 25.5-bit signed representation: 100 MADs + ~35 shift/mask/pre-multiply instructions per product.
//
// Bounds (floor carries, so "reduced" limbs are non-negative: even limbs in [0,2^26), odd limbs in [0,2^25+2^16)):
//   TIGHT  = |even limb| <= 1.68*2^26 and |odd limb| <= 1.68*2^25  (a reduced value, its negation, or the
//            DIFFERENCE of two reduced values);   LOOSE = up to 8x a reduced value (sums of a few terms).
//   fe_mul(h, f, g): f may be LOOSE, g must be TIGHT (19*g_i must fit int32; column sums < 2^63 need A*B < 16).
//   fe_sq(h, f):     f must be TIGHT (38*f_odd, 19*f_even must fit int32).
//   Outputs of fe_mul / fe_sq / fe_carry are reduced.  A SUM of two reduced values is loose: carry it before
//   it is squared or used as g.
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

struct fe {
    int32_t v[10];
};

DAPOL_HD void fe_0(fe& h) {
    for (int i = 0; i < 10; i++) h.v[i] = 0;
}
DAPOL_HD void fe_1(fe& h) {
    fe_0(h);
    h.v[0] = 1;
}
DAPOL_HD void fe_add(fe& h, const fe& f, const fe& g) {
    for (int i = 0; i < 10; i++) h.v[i] = f.v[i] + g.v[i];
}
DAPOL_HD void fe_sub(fe& h, const fe& f, const fe& g) {
    for (int i = 0; i < 10; i++) h.v[i] = f.v[i] - g.v[i];
}
DAPOL_HD void fe_neg(fe& h, const fe& f) {
    for (int i = 0; i < 10; i++) h.v[i] = -f.v[i];
}
// h = c ? g : h  (branch-free select)
DAPOL_HD void fe_cmov(fe& h, const fe& g, bool c) {
    for (int i = 0; i < 10; i++) h.v[i] = c ? g.v[i] : h.v[i];
}
DAPOL_HD void fe_cswap(fe& f, fe& g, bool c) {
    for (int i = 0; i < 10; i++) {
        int32_t a = f.v[i], b = g.v[i];
        f.v[i] = c ? b : a;
        g.v[i] = c ? a : b;
    }
}

// Sequential floor carry of ten 64-bit column sums into limbs; the 2^255 wrap re-enters limb 0 times 19.
DAPOL_HD void fe_reduce_cols(fe& h, int64_t c0, int64_t c1, int64_t c2, int64_t c3, int64_t c4, int64_t c5, int64_t c6,
                             int64_t c7, int64_t c8, int64_t c9) {
    c1 += c0 >> 26; int32_t h0 = (int32_t)c0 & 0x3ffffff;
    c2 += c1 >> 25; int32_t h1 = (int32_t)c1 & 0x1ffffff;
    c3 += c2 >> 26; int32_t h2 = (int32_t)c2 & 0x3ffffff;
    c4 += c3 >> 25; int32_t h3 = (int32_t)c3 & 0x1ffffff;
    c5 += c4 >> 26; int32_t h4 = (int32_t)c4 & 0x3ffffff;
    c6 += c5 >> 25; int32_t h5 = (int32_t)c5 & 0x1ffffff;
    c7 += c6 >> 26; int32_t h6 = (int32_t)c6 & 0x3ffffff;
    c8 += c7 >> 25; int32_t h7 = (int32_t)c7 & 0x1ffffff;
    c9 += c8 >> 26; int32_t h8 = (int32_t)c8 & 0x3ffffff;
    int64_t t = (c9 >> 25) * 19 + h0; int32_t h9 = (int32_t)c9 & 0x1ffffff;
    h.v[0] = (int32_t)t & 0x3ffffff;
    h.v[1] = h1 + (int32_t)(t >> 26);
    h.v[2] = h2; h.v[3] = h3; h.v[4] = h4; h.v[5] = h5; h.v[6] = h6; h.v[7] = h7; h.v[8] = h8; h.v[9] = h9;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(DAPOL_NO_MAD_CHAIN)
// Device form of the column sums.  Every column is ONE chain of v_mad_i64_i32 whose first addend is the carry out of
// the column below, so the 64-bit carry additions of fe_reduce_cols disappear into MADs that are issued anyway
// (per product: 100 MAD + 10 shift + 10 mask, without the 9 v_lshl_add_u64; same values, same limbs).  Written as
// inline assembly, one statement per column, because LLVM re-associates a C sum so that the carry is added last, as
// a separate instruction; one statement per column (not per MAD) keeps the hazard recogniser from padding the
// dependent MADs with s_nop.
#define DAPOL_MAD_CHAIN 1
__device__ __forceinline__ int64_t mad_col10z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2,
        int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, int32_t a6, int32_t b6, int32_t a7,
        int32_t b7, int32_t a8, int32_t b8, int32_t a9, int32_t b9) {
    int64_t d;
    uint64_t sdst;               // VOP3b scalar destination (carry out), unused
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0\n\t"
        "v_mad_i64_i32 %0, %1, %14, %15, %0\n\t"
        "v_mad_i64_i32 %0, %1, %16, %17, %0\n\t"
        "v_mad_i64_i32 %0, %1, %18, %19, %0\n\t"
        "v_mad_i64_i32 %0, %1, %20, %21, %0"
        : "=&v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5),
          "v"(a6), "v"(b6), "v"(a7), "v"(b7), "v"(a8), "v"(b8), "v"(a9), "v"(b9));
    return d;
}
__device__ __forceinline__ int64_t mad_col10c(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2,
        int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5, int32_t a6, int32_t b6,
        int32_t a7, int32_t b7, int32_t a8, int32_t b8, int32_t a9, int32_t b9) {
    uint64_t sdst;               // VOP3b scalar destination (carry out), unused
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
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5),
          "v"(a6), "v"(b6), "v"(a7), "v"(b7), "v"(a8), "v"(b8), "v"(a9), "v"(b9));
    return d;
}
__device__ __forceinline__ int64_t mad_col6z(int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2, int32_t b2,
        int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5) {
    int64_t d;
    uint64_t sdst;               // VOP3b scalar destination (carry out), unused
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
__device__ __forceinline__ int64_t mad_col6c(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2,
        int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4, int32_t a5, int32_t b5) {
    uint64_t sdst;               // VOP3b scalar destination (carry out), unused
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0\n\t"
        "v_mad_i64_i32 %0, %1, %12, %13, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5));
    return d;
}
__device__ __forceinline__ int64_t mad_col5c(int64_t d, int32_t a0, int32_t b0, int32_t a1, int32_t b1, int32_t a2,
        int32_t b2, int32_t a3, int32_t b3, int32_t a4, int32_t b4) {
    uint64_t sdst;               // VOP3b scalar destination (carry out), unused
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %1, %6, %7, %0\n\t"
        "v_mad_i64_i32 %0, %1, %8, %9, %0\n\t"
        "v_mad_i64_i32 %0, %1, %10, %11, %0"
        : "+v"(d), "=s"(sdst)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4));
    return d;
}
// Limbs from columns whose carries are already chained (c_k includes c_{k-1} >> shift).
__device__ __forceinline__ void fe_reduce_chained(fe& h, int64_t c0, int64_t c1, int64_t c2, int64_t c3, int64_t c4, int64_t c5,
                                                  int64_t c6, int64_t c7, int64_t c8, int64_t c9) {
    int32_t h0 = (int32_t)c0 & 0x3ffffff, h1 = (int32_t)c1 & 0x1ffffff;
    int64_t t = (c9 >> 25) * 19 + h0;
    h.v[0] = (int32_t)t & 0x3ffffff;
    h.v[1] = h1 + (int32_t)(t >> 26);
    h.v[2] = (int32_t)c2 & 0x3ffffff; h.v[3] = (int32_t)c3 & 0x1ffffff; h.v[4] = (int32_t)c4 & 0x3ffffff;
    h.v[5] = (int32_t)c5 & 0x1ffffff; h.v[6] = (int32_t)c6 & 0x3ffffff; h.v[7] = (int32_t)c7 & 0x1ffffff;
    h.v[8] = (int32_t)c8 & 0x3ffffff; h.v[9] = (int32_t)c9 & 0x1ffffff;
}
#endif

#define M64(a, b) ((int64_t)(a) * (int64_t)(b))

DAPOL_HD void fe_mul(fe& h, const fe& f, const fe& g) {
    const int32_t f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7],
                  f8 = f.v[8], f9 = f.v[9];
    const int32_t g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4], g5 = g.v[5], g6 = g.v[6], g7 = g.v[7],
                  g8 = g.v[8], g9 = g.v[9];
    const int32_t g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4, g5_19 = 19 * g5, g6_19 = 19 * g6,
                  g7_19 = 19 * g7, g8_19 = 19 * g8, g9_19 = 19 * g9;
    const int32_t f1_2 = 2 * f1, f3_2 = 2 * f3, f5_2 = 2 * f5, f7_2 = 2 * f7, f9_2 = 2 * f9;
#if defined(DAPOL_MAD_CHAIN)
    const int64_t c0 = mad_col10z(f0, g0, f1_2, g9_19, f2, g8_19, f3_2, g7_19, f4, g6_19, f5_2, g5_19, f6, g4_19, f7_2,
        g3_19, f8, g2_19, f9_2, g1_19);
    const int64_t c1 = mad_col10c(c0 >> 26, f0, g1, f1, g0, f2, g9_19, f3, g8_19, f4, g7_19, f5, g6_19, f6, g5_19, f7,
        g4_19, f8, g3_19, f9, g2_19);
    const int64_t c2 = mad_col10c(c1 >> 25, f0, g2, f1_2, g1, f2, g0, f3_2, g9_19, f4, g8_19, f5_2, g7_19, f6, g6_19, f7_2,
        g5_19, f8, g4_19, f9_2, g3_19);
    const int64_t c3 = mad_col10c(c2 >> 26, f0, g3, f1, g2, f2, g1, f3, g0, f4, g9_19, f5, g8_19, f6, g7_19, f7, g6_19, f8,
        g5_19, f9, g4_19);
    const int64_t c4 = mad_col10c(c3 >> 25, f0, g4, f1_2, g3, f2, g2, f3_2, g1, f4, g0, f5_2, g9_19, f6, g8_19, f7_2,
        g7_19, f8, g6_19, f9_2, g5_19);
    const int64_t c5 = mad_col10c(c4 >> 26, f0, g5, f1, g4, f2, g3, f3, g2, f4, g1, f5, g0, f6, g9_19, f7, g8_19, f8,
        g7_19, f9, g6_19);
    const int64_t c6 = mad_col10c(c5 >> 25, f0, g6, f1_2, g5, f2, g4, f3_2, g3, f4, g2, f5_2, g1, f6, g0, f7_2, g9_19, f8,
        g8_19, f9_2, g7_19);
    const int64_t c7 = mad_col10c(c6 >> 26, f0, g7, f1, g6, f2, g5, f3, g4, f4, g3, f5, g2, f6, g1, f7, g0, f8, g9_19, f9,
        g8_19);
    const int64_t c8 = mad_col10c(c7 >> 25, f0, g8, f1_2, g7, f2, g6, f3_2, g5, f4, g4, f5_2, g3, f6, g2, f7_2, g1, f8, g0,
        f9_2, g9_19);
    const int64_t c9 = mad_col10c(c8 >> 26, f0, g9, f1, g8, f2, g7, f3, g6, f4, g5, f5, g4, f6, g3, f7, g2, f8, g1, f9, g0);
    fe_reduce_chained(h, c0, c1, c2, c3, c4, c5, c6, c7, c8, c9);
#else
    int64_t c0 = M64(f0, g0) + M64(f1_2, g9_19) + M64(f2, g8_19) + M64(f3_2, g7_19) + M64(f4, g6_19) + M64(f5_2, g5_19) +
                 M64(f6, g4_19) + M64(f7_2, g3_19) + M64(f8, g2_19) + M64(f9_2, g1_19);
    int64_t c1 = M64(f0, g1) + M64(f1, g0) + M64(f2, g9_19) + M64(f3, g8_19) + M64(f4, g7_19) + M64(f5, g6_19) +
                 M64(f6, g5_19) + M64(f7, g4_19) + M64(f8, g3_19) + M64(f9, g2_19);
    int64_t c2 = M64(f0, g2) + M64(f1_2, g1) + M64(f2, g0) + M64(f3_2, g9_19) + M64(f4, g8_19) + M64(f5_2, g7_19) +
                 M64(f6, g6_19) + M64(f7_2, g5_19) + M64(f8, g4_19) + M64(f9_2, g3_19);
    int64_t c3 = M64(f0, g3) + M64(f1, g2) + M64(f2, g1) + M64(f3, g0) + M64(f4, g9_19) + M64(f5, g8_19) + M64(f6, g7_19) +
                 M64(f7, g6_19) + M64(f8, g5_19) + M64(f9, g4_19);
    int64_t c4 = M64(f0, g4) + M64(f1_2, g3) + M64(f2, g2) + M64(f3_2, g1) + M64(f4, g0) + M64(f5_2, g9_19) + M64(f6, g8_19) +
                 M64(f7_2, g7_19) + M64(f8, g6_19) + M64(f9_2, g5_19);
    int64_t c5 = M64(f0, g5) + M64(f1, g4) + M64(f2, g3) + M64(f3, g2) + M64(f4, g1) + M64(f5, g0) + M64(f6, g9_19) +
                 M64(f7, g8_19) + M64(f8, g7_19) + M64(f9, g6_19);
    int64_t c6 = M64(f0, g6) + M64(f1_2, g5) + M64(f2, g4) + M64(f3_2, g3) + M64(f4, g2) + M64(f5_2, g1) + M64(f6, g0) +
                 M64(f7_2, g9_19) + M64(f8, g8_19) + M64(f9_2, g7_19);
    int64_t c7 = M64(f0, g7) + M64(f1, g6) + M64(f2, g5) + M64(f3, g4) + M64(f4, g3) + M64(f5, g2) + M64(f6, g1) + M64(f7, g0) +
                 M64(f8, g9_19) + M64(f9, g8_19);
    int64_t c8 = M64(f0, g8) + M64(f1_2, g7) + M64(f2, g6) + M64(f3_2, g5) + M64(f4, g4) + M64(f5_2, g3) + M64(f6, g2) +
                 M64(f7_2, g1) + M64(f8, g0) + M64(f9_2, g9_19);
    int64_t c9 = M64(f0, g9) + M64(f1, g8) + M64(f2, g7) + M64(f3, g6) + M64(f4, g5) + M64(f5, g4) + M64(f6, g3) + M64(f7, g2) +
                 M64(f8, g1) + M64(f9, g0);
    fe_reduce_cols(h, c0, c1, c2, c3, c4, c5, c6, c7, c8, c9);
#endif
}

DAPOL_HD void fe_sq(fe& h, const fe& f) {
    const int32_t f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7],
                  f8 = f.v[8], f9 = f.v[9];
    const int32_t f0_2 = 2 * f0, f1_2 = 2 * f1, f2_2 = 2 * f2, f3_2 = 2 * f3, f4_2 = 2 * f4, f5_2 = 2 * f5, f6_2 = 2 * f6,
                  f7_2 = 2 * f7;
    const int32_t f5_38 = 38 * f5, f6_19 = 19 * f6, f7_38 = 38 * f7, f8_19 = 19 * f8, f9_38 = 38 * f9;
#if defined(DAPOL_MAD_CHAIN)
    const int64_t c0 = mad_col6z(f0, f0, f1_2, f9_38, f2_2, f8_19, f3_2, f7_38, f4_2, f6_19, f5, f5_38);
    const int64_t c1 = mad_col5c(c0 >> 26, f0_2, f1, f2, f9_38, f3_2, f8_19, f4, f7_38, f5_2, f6_19);
    const int64_t c2 = mad_col6c(c1 >> 25, f0_2, f2, f1_2, f1, f3_2, f9_38, f4_2, f8_19, f5_2, f7_38, f6, f6_19);
    const int64_t c3 = mad_col5c(c2 >> 26, f0_2, f3, f1_2, f2, f4, f9_38, f5_2, f8_19, f6, f7_38);
    const int64_t c4 = mad_col6c(c3 >> 25, f0_2, f4, f1_2, f3_2, f2, f2, f5_2, f9_38, f6_2, f8_19, f7, f7_38);
    const int64_t c5 = mad_col5c(c4 >> 26, f0_2, f5, f1_2, f4, f2_2, f3, f6, f9_38, f7_2, f8_19);
    const int64_t c6 = mad_col6c(c5 >> 25, f0_2, f6, f1_2, f5_2, f2_2, f4, f3_2, f3, f7_2, f9_38, f8, f8_19);
    const int64_t c7 = mad_col5c(c6 >> 26, f0_2, f7, f1_2, f6, f2_2, f5, f3_2, f4, f8, f9_38);
    const int64_t c8 = mad_col6c(c7 >> 25, f0_2, f8, f1_2, f7_2, f2_2, f6, f3_2, f5_2, f4, f4, f9, f9_38);
    const int64_t c9 = mad_col5c(c8 >> 26, f0_2, f9, f1_2, f8, f2_2, f7, f3_2, f6, f4_2, f5);
    fe_reduce_chained(h, c0, c1, c2, c3, c4, c5, c6, c7, c8, c9);
#else
    int64_t c0 = M64(f0, f0) + M64(f1_2, f9_38) + M64(f2_2, f8_19) + M64(f3_2, f7_38) + M64(f4_2, f6_19) + M64(f5, f5_38);
    int64_t c1 = M64(f0_2, f1) + M64(f2, f9_38) + M64(f3_2, f8_19) + M64(f4, f7_38) + M64(f5_2, f6_19);
    int64_t c2 = M64(f0_2, f2) + M64(f1_2, f1) + M64(f3_2, f9_38) + M64(f4_2, f8_19) + M64(f5_2, f7_38) + M64(f6, f6_19);
    int64_t c3 = M64(f0_2, f3) + M64(f1_2, f2) + M64(f4, f9_38) + M64(f5_2, f8_19) + M64(f6, f7_38);
    int64_t c4 = M64(f0_2, f4) + M64(f1_2, f3_2) + M64(f2, f2) + M64(f5_2, f9_38) + M64(f6_2, f8_19) + M64(f7, f7_38);
    int64_t c5 = M64(f0_2, f5) + M64(f1_2, f4) + M64(f2_2, f3) + M64(f6, f9_38) + M64(f7_2, f8_19);
    int64_t c6 = M64(f0_2, f6) + M64(f1_2, f5_2) + M64(f2_2, f4) + M64(f3_2, f3) + M64(f7_2, f9_38) + M64(f8, f8_19);
    int64_t c7 = M64(f0_2, f7) + M64(f1_2, f6) + M64(f2_2, f5) + M64(f3_2, f4) + M64(f8, f9_38);
    int64_t c8 = M64(f0_2, f8) + M64(f1_2, f7_2) + M64(f2_2, f6) + M64(f3_2, f5_2) + M64(f4, f4) + M64(f9, f9_38);
    int64_t c9 = M64(f0_2, f9) + M64(f1_2, f8) + M64(f2_2, f7) + M64(f3_2, f6) + M64(f4_2, f5);
    fe_reduce_cols(h, c0, c1, c2, c3, c4, c5, c6, c7, c8, c9);
#endif
}
#undef M64

// h = f * small constant (|c| < 2^20), carried
DAPOL_HD void fe_mul_small(fe& h, const fe& f, int32_t c) {
    fe_reduce_cols(h, (int64_t)f.v[0] * c, (int64_t)f.v[1] * c, (int64_t)f.v[2] * c, (int64_t)f.v[3] * c, (int64_t)f.v[4] * c,
                   (int64_t)f.v[5] * c, (int64_t)f.v[6] * c, (int64_t)f.v[7] * c, (int64_t)f.v[8] * c, (int64_t)f.v[9] * c);
}

// Weak carry of a LOOSE value (|limb| < 2^30) back to reduced form, all in 32-bit VOP2 ops.
DAPOL_HD void fe_carry(fe& h, const fe& f) {
    int32_t h0 = f.v[0], h1 = f.v[1], h2 = f.v[2], h3 = f.v[3], h4 = f.v[4], h5 = f.v[5], h6 = f.v[6], h7 = f.v[7], h8 = f.v[8],
            h9 = f.v[9];
    h1 += h0 >> 26; h0 &= 0x3ffffff;
    h2 += h1 >> 25; h1 &= 0x1ffffff;
    h3 += h2 >> 26; h2 &= 0x3ffffff;
    h4 += h3 >> 25; h3 &= 0x1ffffff;
    h5 += h4 >> 26; h4 &= 0x3ffffff;
    h6 += h5 >> 25; h5 &= 0x1ffffff;
    h7 += h6 >> 26; h6 &= 0x3ffffff;
    h8 += h7 >> 25; h7 &= 0x1ffffff;
    h9 += h8 >> 26; h8 &= 0x3ffffff;
    h0 += 19 * (h9 >> 25); h9 &= 0x1ffffff;
    h1 += h0 >> 26; h0 &= 0x3ffffff;
    h.v[0] = h0; h.v[1] = h1; h.v[2] = h2; h.v[3] = h3; h.v[4] = h4; h.v[5] = h5; h.v[6] = h6; h.v[7] = h7; h.v[8] = h8; h.v[9] = h9;
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

// Canonical little-endian 32-byte encoding (fully reduced).
DAPOL_HD void fe_tobytes(uint8_t* s, const fe& f) {
    int32_t h0 = f.v[0], h1 = f.v[1], h2 = f.v[2], h3 = f.v[3], h4 = f.v[4], h5 = f.v[5], h6 = f.v[6], h7 = f.v[7], h8 = f.v[8],
            h9 = f.v[9];
    int32_t q = (19 * h9 + (1 << 24)) >> 25;
    q = (h0 + q) >> 26; q = (h1 + q) >> 25; q = (h2 + q) >> 26; q = (h3 + q) >> 25; q = (h4 + q) >> 26;
    q = (h5 + q) >> 25; q = (h6 + q) >> 26; q = (h7 + q) >> 25; q = (h8 + q) >> 26; q = (h9 + q) >> 25;
    h0 += 19 * q;
    int32_t c;
    c = h0 >> 26; h1 += c; h0 -= c * (1 << 26);
    c = h1 >> 25; h2 += c; h1 -= c * (1 << 25);
    c = h2 >> 26; h3 += c; h2 -= c * (1 << 26);
    c = h3 >> 25; h4 += c; h3 -= c * (1 << 25);
    c = h4 >> 26; h5 += c; h4 -= c * (1 << 26);
    c = h5 >> 25; h6 += c; h5 -= c * (1 << 25);
    c = h6 >> 26; h7 += c; h6 -= c * (1 << 26);
    c = h7 >> 25; h8 += c; h7 -= c * (1 << 25);
    c = h8 >> 26; h9 += c; h8 -= c * (1 << 26);
    c = h9 >> 25; h9 -= c * (1 << 25);
    uint32_t w[8];
    w[0] = (uint32_t)h0 | ((uint32_t)h1 << 26);
    w[1] = ((uint32_t)h1 >> 6) | ((uint32_t)h2 << 19);
    w[2] = ((uint32_t)h2 >> 13) | ((uint32_t)h3 << 13);
    w[3] = ((uint32_t)h3 >> 19) | ((uint32_t)h4 << 6);
    w[4] = (uint32_t)h5 | ((uint32_t)h6 << 25);
    w[5] = ((uint32_t)h6 >> 7) | ((uint32_t)h7 << 19);
    w[6] = ((uint32_t)h7 >> 13) | ((uint32_t)h8 << 12);
    w[7] = ((uint32_t)h8 >> 20) | ((uint32_t)h9 << 6);
    for (int i = 0; i < 8; i++) {
        s[4 * i] = (uint8_t)w[i];
        s[4 * i + 1] = (uint8_t)(w[i] >> 8);
        s[4 * i + 2] = (uint8_t)(w[i] >> 16);
        s[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}

// Same as fe_tobytes but into eight little-endian words (device-friendly: no byte stores).
DAPOL_HD void fe_towords(uint32_t* w, const fe& f) {
    uint8_t s[32];
    fe_tobytes(s, f);
    for (int i = 0; i < 8; i++)
        w[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
}

// Load from eight little-endian words; bit 255 is ignored (dalek FieldElement::from_bytes semantics).
DAPOL_HD void fe_fromwords(fe& h, const uint32_t* w) {
    h.v[0] = (int32_t)(w[0] & 0x3ffffff);
    h.v[1] = (int32_t)(((w[0] >> 26) | (w[1] << 6)) & 0x1ffffff);
    h.v[2] = (int32_t)(((w[1] >> 19) | (w[2] << 13)) & 0x3ffffff);
    h.v[3] = (int32_t)(((w[2] >> 13) | (w[3] << 19)) & 0x1ffffff);
    h.v[4] = (int32_t)((w[3] >> 6) & 0x3ffffff);
    h.v[5] = (int32_t)(w[4] & 0x1ffffff);
    h.v[6] = (int32_t)(((w[4] >> 25) | (w[5] << 7)) & 0x3ffffff);
    h.v[7] = (int32_t)(((w[5] >> 19) | (w[6] << 13)) & 0x1ffffff);
    h.v[8] = (int32_t)(((w[6] >> 12) | (w[7] << 20)) & 0x3ffffff);
    h.v[9] = (int32_t)((w[7] >> 6) & 0x1ffffff);
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
