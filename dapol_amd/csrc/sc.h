// Scalars mod L = 2^252 + 27742317777372353535851937790883648493 in Montgomery form (R = 2^256), eight 32-bit words.
// Replaces curve25519-dalek-ng Scalar arithmetic used by the reference (src/dapol/node.rs:75, bulletproofs
// prover).  All values that live in HBM between kernels (l, r, s-vectors) stay in Montgomery form; bytes leave
// through sc_from_mont (canonical, < L), and enter through sc_from_wide / sc_to_mont.
#pragma once
#include "consts.h"

namespace dapol {

struct sc {
    uint32_t v[8];
};

DAPOL_HD void sc_zero(sc& r) {
    for (int i = 0; i < 8; i++) r.v[i] = 0;
}
DAPOL_HD void sc_one_mont(sc& r) {
    for (int i = 0; i < 8; i++) r.v[i] = SC_R1[i];
}
DAPOL_HD bool sc_is_zero(const sc& a) {
    uint32_t o = 0;
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

// 32-bit add / subtract with carry: clang's builtins become v_addc_co_u32 / v_subb_co_u32 chains on the device (one
// instruction per word; a uint64_t emulation costs a 64-bit add plus register moves per word); plain C elsewhere.
DAPOL_HD uint32_t sc_addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t& cout) {
#if defined(__clang__) && !defined(DAPOL_SC_NO_BUILTIN_CARRY)
    return __builtin_addc(a, b, cin, &cout);
#else
    uint32_t s = a + b, c1 = s < a ? 1u : 0u, s2 = s + cin;
    cout = c1 | (s2 < s ? 1u : 0u);
    return s2;
#endif
}
DAPOL_HD uint32_t sc_subb(uint32_t a, uint32_t b, uint32_t bin, uint32_t& bout) {
#if defined(__clang__) && !defined(DAPOL_SC_NO_BUILTIN_CARRY)
    return __builtin_subc(a, b, bin, &bout);
#else
    uint32_t d = a - b, b1 = a < b ? 1u : 0u, d2 = d - bin;
    bout = b1 | (d < bin ? 1u : 0u);
    return d2;
#endif
}

// r = (t >= L) ? t - L : t, where t = (hi : t[0..8)) < 2L
DAPOL_HD void sc_cond_sub(sc& r, const uint32_t* t, uint32_t hi) {
    uint32_t d[8], borrow = 0;
    for (int i = 0; i < 8; i++) d[i] = sc_subb(t[i], SC_L[i], borrow, borrow);
    bool ge = (hi != 0) | (borrow == 0);
    for (int i = 0; i < 8; i++) r.v[i] = ge ? d[i] : t[i];
}

// a * b + c as ONE v_mad_u64_u32 on the device.  Explicit because with the reduction's constant 2^20 (the top limb of L)
// written as a C product, hipcc -O1 and above dropped the term in two of the nine reduction steps
// (tools/sc_selftest.hip: device != host at -O3, equal at -O0); an opaque multiply-add leaves nothing to re-associate.
DAPOL_HD uint64_t sc_mad(uint32_t a, uint32_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t d, carry_out;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry_out) : "v"(a), "s"(b), "v"(c));   // b: a limb of L, kept in an SGPR
    return d;
#else
    return (uint64_t)a * b + c;
#endif
}

// Montgomery product a*b/R mod L (R = 2^256).  a < 2^256 (any), b < L  ->  result < L.
// Computed on nine 29-bit limbs so that every column of the product and of the reduction is a plain sum of 64-bit
// multiply-adds (at most 15 terms of 58 bits): ~250 VALU instructions, against ~600 for the word-serial form with its
// 64-bit carry additions -- the same lesson as fe.h.  L = 2^252 + delta has only six non-zero 29-bit limbs, so a
// reduction step is six MADs.  Eight steps retire 29 bits each and a ninth the remaining 24 (8 * 29 + 24 = 256), which
// keeps R = 2^256 and with it every constant and every stored Montgomery value.
// The three parts of the product, also used on their own where sums of products are reduced once (k_rvb_gh_lazy):
// nine 29-bit limbs of a 256-bit word array; columns += A x B; Montgomery reduction of the columns.
DAPOL_HD void sc_split29(uint32_t* A, const uint32_t* v) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int o = 29 * k, w = o >> 5, sh = o & 31;
        uint32_t x = v[w] >> sh;
        if (sh > 3 && w + 1 < 8) x |= v[w + 1] << (32 - sh);
        A[k] = x & 0x1fffffffu;
    }
}
DAPOL_HD void sc_mac29(uint64_t* c, const uint32_t* A, const uint32_t* B) {
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)A[i] * B[j];
}
// r = (sum of the columns) / R mod L for a column array holding at most 6 products of values < L (columns < 2^64,
// result < 2L before the final subtraction).  Destroys c.
DAPOL_HD void sc_redc29(sc& r, uint64_t* c) {
    const uint32_t M29 = 0x1fffffffu;
    const uint32_t L0 = 0x1cf5d3edu, L1 = 0x009318d2u, L2 = 0x1de73596u, L3 = 0x1df3bd45u, L4 = 0x0000014du, L8 = 0x00100000u;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t mask = i < 8 ? M29 : 0x00ffffffu;                     // the ninth step retires 24 bits
        const uint32_t m = ((uint32_t)c[i] * SC_LFACTOR) & mask;
        c[i] = sc_mad(m, L0, c[i]);
        c[i + 1] = sc_mad(m, L1, c[i + 1]);
        c[i + 2] = sc_mad(m, L2, c[i + 2]);
        c[i + 3] = sc_mad(m, L3, c[i + 3]);
        c[i + 4] = sc_mad(m, L4, c[i + 4]);
        c[i + 8] = sc_mad(m, L8, c[i + 8]);
        if (i < 8) c[i + 1] += c[i] >> 29;                                   // c[i] is now a multiple of 2^29
    }
    // normalise columns 8..17 (column 8 keeps its bits 24..28) and read the 256 bits that start at bit 24 of column 8
    uint32_t lim[11];
#pragma unroll
    for (int k = 8; k < 17; k++) {
        lim[k - 8] = (uint32_t)c[k] & M29;
        c[k + 1] += c[k] >> 29;
    }
    lim[9] = (uint32_t)c[17];
    lim[10] = 0;
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int o = 24 + 32 * j, q = o / 29, sh = o % 29;
        uint32_t x = lim[q] >> sh;
        x |= lim[q + 1] << (29 - sh);
        if (58 - sh < 32) x |= lim[q + 2] << (58 - sh);
        t[j] = x;
    }
    sc_cond_sub(r, t, 0);
}
DAPOL_HD void sc_montmul(sc& r, const sc& a, const sc& b) {
    uint32_t A[9], B[9];
    sc_split29(A, a.v);
    sc_split29(B, b.v);
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
    sc_mac29(c, A, B);
    sc_redc29(r, c);
}
DAPOL_HD void sc_montsq(sc& r, const sc& a) {
    sc_montmul(r, a, a);
}

// a + b mod L (both < L)
DAPOL_HD void sc_add(sc& r, const sc& a, const sc& b) {
    uint32_t t[8], c = 0;
    for (int i = 0; i < 8; i++) t[i] = sc_addc(a.v[i], b.v[i], c, c);
    sc_cond_sub(r, t, c);
}
// a - b mod L (both < L)
DAPOL_HD void sc_sub(sc& r, const sc& a, const sc& b) {
    uint32_t t[8], borrow = 0, c = 0;
    for (int i = 0; i < 8; i++) t[i] = sc_subb(a.v[i], b.v[i], borrow, borrow);
    for (int i = 0; i < 8; i++) r.v[i] = sc_addc(t[i], borrow ? SC_L[i] : 0u, c, c);
}
DAPOL_HD void sc_neg(sc& r, const sc& a) {
    sc z;
    sc_zero(z);
    sc_sub(r, z, a);
}

// any 256-bit integer x (eight words)  ->  Montgomery form of x mod L
DAPOL_HD void sc_to_mont(sc& r, const uint32_t* x) {
    sc a, r2;
    for (int i = 0; i < 8; i++) {
        a.v[i] = x[i];
        r2.v[i] = SC_R2[i];
    }
    sc_montmul(r, a, r2);
}
// 512-bit little-endian integer (sixteen words) -> Montgomery form of it mod L  (Scalar::from_bytes_mod_order_wide)
DAPOL_HD void sc_from_wide(sc& r, const uint32_t* w16) {
    sc lo, hi, k;
    for (int i = 0; i < 8; i++) {
        lo.v[i] = w16[i];
        hi.v[i] = w16[8 + i];
        k.v[i] = SC_R2[i];
    }
    sc_montmul(lo, lo, k);          // lo * R
    for (int i = 0; i < 8; i++) k.v[i] = SC_R3[i];
    sc_montmul(hi, hi, k);          // hi * 2^256 * R
    sc_add(r, lo, hi);
}
// Montgomery form -> canonical integer (< L) as eight words
DAPOL_HD void sc_from_mont(uint32_t* out, const sc& a) {
    sc one, r;
    sc_zero(one);
    one.v[0] = 1;
    sc_montmul(r, a, one);
    for (int i = 0; i < 8; i++) out[i] = r.v[i];
}
DAPOL_HD void sc_from_u64_mont(sc& r, uint64_t x) {
    uint32_t w[8] = {(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0};
    sc_to_mont(r, w);
}

// a^(L-2) in Montgomery form (Fermat inversion; a != 0)
DAPOL_HD_NOINLINE void sc_invert_mont(sc& r, const sc& a) {
    sc acc;
    sc_one_mont(acc);
    for (int i = 252; i >= 0; i--) {
        sc_montsq(acc, acc);
        if ((SC_LM2[i >> 5] >> (i & 31)) & 1) sc_montmul(acc, acc, a);
    }
    r = acc;
}
// Inversion of a PUBLIC scalar (a Fiat-Shamir challenge: y, u_k) in variable time, Montgomery form in and out: 0 -> 0 like the
// Fermat ladder above.  The ladder is 253 dependent squarings -- 70,000 instructions, 0.14 ms on a lone lane, once per round of
// a proof somebody is waiting for.  This is the Bernstein-Yang division-step recurrence ("Fast constant-time gcd computation and
// modular inversion", 2019) in the batched shape that suits a 32-bit machine: thirty division steps at a time are run on the low
// words of (f, g) alone and collected in a 2x2 integer matrix t, which is then applied once to the full-width f, g (exactly
// divisible by 2^30) and to the Bezout pair d, e (made divisible by adding a multiple of L).  Values are nine signed 30-bit limbs;
// about 20 batches of ~450 instructions.  NOT for secrets: the step count and the branches depend on the operand.
DAPOL_HD void sc_divsteps30_var(int32_t& eta, uint32_t f, uint32_t g, int32_t* t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    int i = 30;
    for (;;) {
        const int zeros = __builtin_ctz(g | (0xffffffffu << i));      // g even: halve (the f row doubles instead, keeping t integral)
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {                                                // delta > 0 and g odd: (f, g) <- (g, -f)
            eta = -eta;
            uint32_t tmp = f; f = g; g = 0u - tmp;
            tmp = u; u = q; q = 0u - tmp;
            tmp = v; v = r; r = 0u - tmp;
        }
        g += f; q += u; r += v;                                       // g odd: g <- g + f (even now; halved at the top)
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
}
DAPOL_HD_NOINLINE void sc_invert_vartime_mont(sc& out, const sc& a) {
    const int32_t M30 = (int32_t)0x3fffffff;
    int32_t d[9], e[9], f[9], g[9];
    for (int i = 0; i < 9; i++) { d[i] = 0; e[i] = 0; f[i] = SC_L30[i]; }
    e[0] = 1;
    for (int i = 0; i < 9; i++) {                                     // g = a (the integer a_mont, < L) in 30-bit limbs
        const int o = 30 * i, w = o >> 5, sh = o & 31;
        uint64_t x = (uint64_t)a.v[w] | (w + 1 < 8 ? (uint64_t)a.v[w + 1] << 32 : 0ull);
        g[i] = (int32_t)((uint32_t)(x >> sh) & (uint32_t)M30);
    }
    int32_t eta = -1;
    for (int it = 0; it < 40; it++) {                                 // 741 steps bound the recurrence for 256-bit operands
        int32_t t[4];
        sc_divsteps30_var(eta, (uint32_t)f[0] | ((uint32_t)f[1] << 30), (uint32_t)g[0] | ((uint32_t)g[1] << 30), t);
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        {   // (d, e) <- t (d, e) / 2^30 mod L, kept in (-2L, L)
            const int32_t sd = d[8] >> 31, se = e[8] >> 31;
            int32_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);
            int64_t cd = u * d[0] + v * e[0], ce = q * d[0] + r * e[0];
            md -= (int32_t)((SC_LINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
            me -= (int32_t)((SC_LINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd += (int64_t)SC_L30[0] * md; ce += (int64_t)SC_L30[0] * me;
            cd >>= 30; ce >>= 30;
            for (int i = 1; i < 9; i++) {
                cd += u * d[i] + v * e[i] + (int64_t)SC_L30[i] * md;
                ce += q * d[i] + r * e[i] + (int64_t)SC_L30[i] * me;
                d[i - 1] = (int32_t)cd & M30; e[i - 1] = (int32_t)ce & M30;
                cd >>= 30; ce >>= 30;
            }
            d[8] = (int32_t)cd; e[8] = (int32_t)ce;
        }
        {   // (f, g) <- t (f, g) / 2^30, exactly
            int64_t cf = u * f[0] + v * g[0], cg = q * f[0] + r * g[0];
            cf >>= 30; cg >>= 30;
            for (int i = 1; i < 9; i++) {
                cf += u * f[i] + v * g[i];
                cg += q * f[i] + r * g[i];
                f[i - 1] = (int32_t)cf & M30; g[i - 1] = (int32_t)cg & M30;
                cf >>= 30; cg >>= 30;
            }
            f[8] = (int32_t)cf; g[8] = (int32_t)cg;
        }
        int32_t nz = 0;
        for (int i = 0; i < 9; i++) nz |= g[i];
        if (nz == 0) break;
    }
    // gcd = |f| = 1 (f = +-L for a = 0, where d = 0): a^-1 = sign(f) d.  Pack d into nine two's-complement words, fix the sign,
    // bring it into [0, L).
    uint32_t w[9];
    {
        uint64_t acc = 0;
        int bits = 0, j = 0;
        for (int i = 0; i < 9; i++) {
            acc |= (uint64_t)(i < 8 ? (uint32_t)(d[i] & M30) : (uint32_t)d[8]) << bits;
            bits += i < 8 ? 30 : 32;
            while (bits >= 32) { w[j++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
        }
        w[8] = (uint32_t)(((int32_t)((uint32_t)acc << 16)) >> 16);   // 8 * 30 + 32 = 272 bits: the last 16, sign-extended
    }
    if (f[8] < 0) {                                                   // negate
        uint32_t c = 1;
        for (int i = 0; i < 9; i++) w[i] = sc_addc(~w[i], 0u, c, c);
    }
    for (int k = 0; k < 4 && (w[8] >> 31); k++) {                     // negative: + L
        uint32_t c = 0;
        for (int i = 0; i < 9; i++) w[i] = sc_addc(w[i], i < 8 ? SC_L[i] : 0u, c, c);
    }
    for (int k = 0; k < 4; k++) {                                     // >= L: - L
        uint32_t t9[9], bw = 0;
        for (int i = 0; i < 9; i++) t9[i] = sc_subb(w[i], i < 8 ? SC_L[i] : 0u, bw, bw);
        if (t9[8] >> 31) break;
        for (int i = 0; i < 9; i++) w[i] = t9[i];
    }
    // w = (a R)^-1 as an integer; the Montgomery form of a^-1 is a^-1 R = w R^2 = montmul(w, R^3)
    sc x, k3;
    for (int i = 0; i < 8; i++) { x.v[i] = w[i]; k3.v[i] = SC_R3[i]; }
    sc_montmul(out, x, k3);
}

// a^e for a small public exponent (variable time in e, like bulletproofs util::scalar_exp_vartime)
DAPOL_HD void sc_pow_mont(sc& r, const sc& a, uint32_t e) {
    sc acc, base = a;
    sc_one_mont(acc);
    while (e) {
        if (e & 1) sc_montmul(acc, acc, base);
        sc_montsq(base, base);
        e >>= 1;
    }
    r = acc;
}

// Signed radix-256 recoding of a 255-bit integer (eight words, bit 255 clear): digits d[0..32) in [-128, 128],
// sum d[i] 256^i = x.  The top digit reaches 128 only for unreduced inputs >= 2^255 - 2^247.
DAPOL_HD void sc_recode_s8(int16_t* d, const uint32_t* x) {
    int carry = 0;
    for (int i = 0; i < 32; i++) {
        int b = (int)((x[i >> 2] >> (8 * (i & 3))) & 0xff) + carry;
        carry = (b > 127 && i < 31) ? 1 : 0;
        d[i] = (int16_t)(b - (carry << 8));
    }
}

// Signed radix-2^W recoding into NW digits in [-2^(W-1), 2^(W-1)] (the top digit absorbs the last carry, so the
// input must be < 2^(W(NW-1) + W-1): NW = 255/W + 1 for any 255-bit integer, 253/W + 1 for canonical scalars).
// out(i, digit) is called for i = 0..NW-1.
template <typename F>
DAPOL_HD void sc_recode_w(int W, int NW, const uint32_t* x, F out) {
    int carry = 0;
    for (int i = 0; i < NW; i++) {
        int o = i * W, wd = o >> 5, sh = o & 31;
        uint32_t lo = wd < 8 ? x[wd] >> sh : 0u;
        uint32_t hi = (sh && wd + 1 < 8) ? (x[wd + 1] << (32 - sh)) : 0u;
        int b = (int)((lo | hi) & ((1u << W) - 1)) + carry;
        carry = (b >= (1 << (W - 1)) && i < NW - 1) ? 1 : 0;
        out(i, b - (carry << W));
    }
}

}  // namespace dapol
