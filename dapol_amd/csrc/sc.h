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

// r = (t >= L) ? t - L : t, where t = (hi : t[0..8)) < 2L
DAPOL_HD void sc_cond_sub(sc& r, const uint32_t* t, uint32_t hi) {
    uint32_t d[8];
    uint64_t borrow = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)t[i] - SC_L[i] - borrow;
        d[i] = (uint32_t)x;
        borrow = (x >> 32) & 1;
    }
    bool ge = (hi != 0) | (borrow == 0);
    for (int i = 0; i < 8; i++) r.v[i] = ge ? d[i] : t[i];
}

// Montgomery product a*b/R mod L.  a < 2^256 (any), b < L  ->  result < L.
DAPOL_HD void sc_montmul(sc& r, const sc& a, const sc& b) {
    uint32_t t[10];
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        uint64_t s = (uint64_t)t[8] + c;
        t[8] = (uint32_t)s;
        t[9] = (uint32_t)(s >> 32);
        uint32_t m = t[0] * SC_LFACTOR;
        c = ((uint64_t)m * SC_L[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            uint64_t s2 = (uint64_t)m * SC_L[j] + t[j] + c;
            t[j - 1] = (uint32_t)s2;
            c = s2 >> 32;
        }
        s = (uint64_t)t[8] + c;
        t[7] = (uint32_t)s;
        t[8] = t[9] + (uint32_t)(s >> 32);
    }
    sc_cond_sub(r, t, t[8]);
}
DAPOL_HD void sc_montsq(sc& r, const sc& a) {
    sc_montmul(r, a, a);
}

// a + b mod L (both < L)
DAPOL_HD void sc_add(sc& r, const sc& a, const sc& b) {
    uint32_t t[8];
    uint64_t c = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)a.v[i] + b.v[i] + c;
        t[i] = (uint32_t)s;
        c = s >> 32;
    }
    sc_cond_sub(r, t, (uint32_t)c);
}
// a - b mod L (both < L)
DAPOL_HD void sc_sub(sc& r, const sc& a, const sc& b) {
    uint32_t t[8];
    uint64_t borrow = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - borrow;
        t[i] = (uint32_t)x;
        borrow = (x >> 32) & 1;
    }
    uint64_t c = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)t[i] + (borrow ? SC_L[i] : 0u) + c;
        r.v[i] = (uint32_t)s;
        c = s >> 32;
    }
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
