/* oracle/ref_math.c -- TEST INFRASTRUCTURE ONLY.  ristretto255 (RFC 9496) and scalar arithmetic for the CPU oracle.
 * Restates what the reference reaches through curve25519-dalek-ng: RistrettoPoint add / compress / decompress /
 * from_uniform_bytes / (vartime_)multiscalar_mul and Scalar arithmetic (src/dapol/node.rs:31,35,67-76,
 * src/proof/node.rs:58-63,88). */
#include "ref_math.h"
#include <stdlib.h>

fe51 REF_D, REF_D2, REF_SQRT_M1, REF_SQRT_AD_MINUS_ONE, REF_INVSQRT_A_MINUS_D, REF_ONE_MINUS_D_SQ, REF_D_MINUS_ONE_SQ;
pt REF_B, REF_BB;
scl SC_ONE, SC_ZERO;

static void fe_pow_bytes(fe51* r, const fe51* x, const uint8_t e[32]) {
    fe51 acc;
    fe_1(&acc);
    for (int i = 255; i >= 0; i--) {
        fe_sq(&acc, &acc);
        if ((e[i >> 3] >> (i & 7)) & 1) fe_mul(&acc, &acc, x);
    }
    *r = acc;
}
static void fe_invert(fe51* r, const fe51* x) {
    fe51 t, x3;
    fe_pow22523(&t, x);
    fe_sqn(&t, &t, 3);
    fe_sq(&x3, x);
    fe_mul(&x3, &x3, x);
    fe_mul(r, &t, &x3);
}

int fe_sqrt_ratio_m1(fe51* r, const fe51* u, const fe51* v) {
    fe51 v3, v7, t, check, nu, nui, rr;
    fe_sq(&v3, v); fe_mul(&v3, &v3, v);
    fe_sq(&v7, &v3); fe_mul(&v7, &v7, v);
    fe_mul(&t, u, &v7);
    fe_pow22523(&t, &t);
    fe_mul(&t, &t, &v3);
    fe_mul(&rr, &t, u);
    fe_sq(&check, &rr); fe_mul(&check, &check, v);
    fe_neg(&nu, u);
    fe_mul(&nui, &nu, &REF_SQRT_M1);
    int correct = fe_eq(&check, u), flipped = fe_eq(&check, &nu), flipped_i = fe_eq(&check, &nui);
    if (flipped || flipped_i) fe_mul(&rr, &rr, &REF_SQRT_M1);
    fe_abs(r, &rr);
    return correct || flipped;
}

void pt_identity(pt* r) { fe_0(&r->X); fe_1(&r->Y); fe_1(&r->Z); fe_0(&r->T); }
void pt_neg(pt* r, const pt* p) { fe_neg(&r->X, &p->X); r->Y = p->Y; r->Z = p->Z; fe_neg(&r->T, &p->T); }
void pt_add(pt* r, const pt* p, const pt* q) {
    fe51 a, b, c, d, e, f, g, h, t0, t1;
    fe_sub(&t0, &p->Y, &p->X); fe_sub(&t1, &q->Y, &q->X); fe_mul(&a, &t0, &t1);
    fe_add(&t0, &p->Y, &p->X); fe_add(&t1, &q->Y, &q->X); fe_mul(&b, &t0, &t1);
    fe_mul(&c, &p->T, &q->T); fe_mul(&c, &c, &REF_D2);
    fe_mul(&d, &p->Z, &q->Z); fe_add(&d, &d, &d);
    fe_sub(&e, &b, &a); fe_sub(&f, &d, &c); fe_add(&g, &d, &c); fe_add(&h, &b, &a);
    fe_mul(&r->X, &e, &f); fe_mul(&r->Y, &g, &h); fe_mul(&r->Z, &f, &g); fe_mul(&r->T, &e, &h);
}
void pt_dbl(pt* r, const pt* p) {
    fe51 a, b, c, e, f, g, h, t;
    fe_sq(&a, &p->X); fe_sq(&b, &p->Y); fe_sq(&c, &p->Z); fe_add(&c, &c, &c);
    fe_add(&t, &p->X, &p->Y); fe_sq(&t, &t);
    fe_add(&h, &a, &b);                    /* A + B */
    fe_sub(&e, &h, &t);                    /* E = A + B - (X+Y)^2 ; with D = -A: signs below follow a = -1 */
    fe_sub(&g, &a, &b);                    /* G = A - B  (= -(B - A)) */
    fe_add(&f, &c, &g);                    /* F = C + G */
    fe_mul(&r->X, &e, &f); fe_mul(&r->Y, &g, &h); fe_mul(&r->Z, &f, &g); fe_mul(&r->T, &e, &h);
}
void pt_compress(uint8_t out[32], const pt* p) {
    fe51 u1, u2, t, invsqrt, den1, den2, z_inv, ix, iy, ench, x, y, den_inv, one;
    fe_add(&t, &p->Z, &p->Y); fe_sub(&u1, &p->Z, &p->Y); fe_mul(&u1, &t, &u1);
    fe_mul(&u2, &p->X, &p->Y);
    fe_sq(&t, &u2); fe_mul(&t, &t, &u1);
    fe_1(&one);
    fe_sqrt_ratio_m1(&invsqrt, &one, &t);
    fe_mul(&den1, &invsqrt, &u1); fe_mul(&den2, &invsqrt, &u2);
    fe_mul(&z_inv, &den1, &den2); fe_mul(&z_inv, &z_inv, &p->T);
    fe_mul(&ix, &p->X, &REF_SQRT_M1); fe_mul(&iy, &p->Y, &REF_SQRT_M1);
    fe_mul(&ench, &den1, &REF_INVSQRT_A_MINUS_D);
    fe_mul(&t, &p->T, &z_inv);
    if (fe_isneg(&t)) { x = iy; y = ix; den_inv = ench; } else { x = p->X; y = p->Y; den_inv = den2; }
    fe_mul(&t, &x, &z_inv);
    if (fe_isneg(&t)) fe_neg(&y, &y);
    fe_sub(&t, &p->Z, &y); fe_mul(&t, &t, &den_inv); fe_abs(&t, &t);
    fe_tobytes(out, &t);
}
int pt_decompress(pt* r, const uint8_t in[32]) {
    fe51 s, ss, u1, u2, u2s, v, t, invsqrt, den_x, den_y, one;
    uint8_t chk[32];
    fe_frombytes(&s, in);
    fe_tobytes(chk, &s);
    if (memcmp(chk, in, 32) != 0 || (in[0] & 1)) return 0;
    fe_1(&one);
    fe_sq(&ss, &s); fe_sub(&u1, &one, &ss); fe_add(&u2, &one, &ss); fe_sq(&u2s, &u2);
    fe_sq(&t, &u1); fe_mul(&t, &t, &REF_D); fe_neg(&t, &t); fe_sub(&v, &t, &u2s);
    fe_mul(&t, &v, &u2s);
    int was_square = fe_sqrt_ratio_m1(&invsqrt, &one, &t);
    fe_mul(&den_x, &invsqrt, &u2); fe_mul(&den_y, &invsqrt, &den_x); fe_mul(&den_y, &den_y, &v);
    fe_mul(&t, &s, &den_x); fe_add(&t, &t, &t); fe_abs(&r->X, &t);
    fe_mul(&r->Y, &u1, &den_y); fe_1(&r->Z); fe_mul(&r->T, &r->X, &r->Y);
    return was_square && !fe_isneg(&r->T) && !fe_iszero(&r->Y);
}
static void elligator(pt* out, const fe51* t0) {
    fe51 r, u, v, a, b, s, sp, c, N, w0, w1, w2, w3, one, tmp;
    fe_1(&one);
    fe_sq(&r, t0); fe_mul(&r, &r, &REF_SQRT_M1);
    fe_add(&u, &r, &one); fe_mul(&u, &u, &REF_ONE_MINUS_D_SQ);
    fe_mul(&a, &r, &REF_D); fe_add(&a, &a, &one); fe_neg(&a, &a);
    fe_add(&b, &r, &REF_D);
    fe_mul(&v, &a, &b);
    int was_square = fe_sqrt_ratio_m1(&s, &u, &v);
    fe_mul(&sp, &s, t0); fe_abs(&sp, &sp); fe_neg(&sp, &sp);
    if (!was_square) s = sp;
    if (was_square) fe_neg(&c, &one); else c = r;
    fe_sub(&tmp, &r, &one); fe_mul(&N, &c, &tmp); fe_mul(&N, &N, &REF_D_MINUS_ONE_SQ); fe_sub(&N, &N, &v);
    fe_mul(&w0, &s, &v); fe_add(&w0, &w0, &w0);
    fe_mul(&w1, &N, &REF_SQRT_AD_MINUS_ONE);
    fe_sq(&tmp, &s); fe_sub(&w2, &one, &tmp); fe_add(&w3, &one, &tmp);
    fe_mul(&out->X, &w0, &w3); fe_mul(&out->Y, &w2, &w1); fe_mul(&out->Z, &w1, &w3); fe_mul(&out->T, &w0, &w2);
}
void pt_from_uniform(pt* r, const uint8_t in[64]) {
    fe51 t1, t2;
    pt p1, p2;
    fe_frombytes(&t1, in); fe_frombytes(&t2, in + 32);
    elligator(&p1, &t1); elligator(&p2, &t2);
    pt_add(r, &p1, &p2);
}
void pt_mul(pt* r, const pt* p, const uint8_t k[32]) {
    pt tbl[16], acc;
    pt_identity(&tbl[0]);
    tbl[1] = *p;
    for (int i = 2; i < 16; i++) pt_add(&tbl[i], &tbl[i - 1], p);
    pt_identity(&acc);
    for (int i = 63; i >= 0; i--) {
        for (int d = 0; d < 4; d++) pt_dbl(&acc, &acc);
        int nib = (k[i >> 1] >> (4 * (i & 1))) & 15;
        if (nib) pt_add(&acc, &acc, &tbl[nib]);
    }
    *r = acc;
}
/* RistrettoPoint::vartime_multiscalar_mul: Straus (4-bit) for small sizes, bucket method (Pippenger) above. */
void pt_msm_vartime(pt* r, const scl* scalars, const pt* points, size_t n) {
    pt acc;
    pt_identity(&acc);
    if (n == 0) { *r = acc; return; }
    if (n <= 4) {                      /* the 2-term generator folds of the inner-product argument: no heap traffic */
        uint8_t k4[4 * 32];
        pt t4[4 * 16];
        for (size_t i = 0; i < n; i++) {
            sc_to_bytes(k4 + 32 * i, &scalars[i]);
            pt_identity(&t4[16 * i]);
            t4[16 * i + 1] = points[i];
            for (int k = 2; k < 16; k++) pt_add(&t4[16 * i + k], &t4[16 * i + k - 1], &points[i]);
        }
        for (int w = 63; w >= 0; w--) {
            for (int d = 0; d < 4; d++) pt_dbl(&acc, &acc);
            for (size_t i = 0; i < n; i++) {
                int nib = (k4[32 * i + (w >> 1)] >> (4 * (w & 1))) & 15;
                if (nib) pt_add(&acc, &acc, &t4[16 * i + nib]);
            }
        }
        *r = acc;
        return;
    }
    uint8_t* kb = (uint8_t*)malloc(32 * n);
    for (size_t i = 0; i < n; i++) sc_to_bytes(kb + 32 * i, &scalars[i]);
    if (n < 96) {
        pt* tbl = (pt*)malloc(sizeof(pt) * 16 * n);
        for (size_t i = 0; i < n; i++) {
            pt_identity(&tbl[16 * i]);
            tbl[16 * i + 1] = points[i];
            for (int k = 2; k < 16; k++) pt_add(&tbl[16 * i + k], &tbl[16 * i + k - 1], &points[i]);
        }
        for (int w = 63; w >= 0; w--) {
            for (int d = 0; d < 4; d++) pt_dbl(&acc, &acc);
            for (size_t i = 0; i < n; i++) {
                int nib = (kb[32 * i + (w >> 1)] >> (4 * (w & 1))) & 15;
                if (nib) pt_add(&acc, &acc, &tbl[16 * i + nib]);
            }
        }
        free(tbl);
    } else {
        int c = 6;
        while ((1u << (c + 3)) < n && c < 12) c++;
        int nb = 1 << c, windows = (253 + c - 1) / c;
        pt* bucket = (pt*)malloc(sizeof(pt) * nb);
        uint8_t* used = (uint8_t*)malloc(nb);
        for (int w = windows - 1; w >= 0; w--) {
            for (int d = 0; d < c; d++) pt_dbl(&acc, &acc);
            memset(used, 0, nb);
            for (size_t i = 0; i < n; i++) {
                int bitpos = w * c, idx = 0;
                for (int t = 0; t < c; t++) {
                    int bp = bitpos + t;
                    if (bp < 256) idx |= ((kb[32 * i + (bp >> 3)] >> (bp & 7)) & 1) << t;
                }
                if (!idx) continue;
                if (used[idx]) pt_add(&bucket[idx], &bucket[idx], &points[i]);
                else { bucket[idx] = points[i]; used[idx] = 1; }
            }
            pt run, sum;
            pt_identity(&run);
            pt_identity(&sum);
            int any = 0;
            for (int k = nb - 1; k >= 1; k--) {
                if (used[k]) { pt_add(&run, &run, &bucket[k]); any = 1; }
                if (any) pt_add(&sum, &sum, &run);
            }
            if (any) pt_add(&acc, &acc, &sum);
        }
        free(bucket);
        free(used);
    }
    free(kb);
    *r = acc;
}

/* ------------------------------------------------------------------------------------------------ scalars */
static const uint64_t L64[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};
static const uint64_t LFACTOR64 = 0xd2b51da312547e1bULL;
static const uint64_t R2_64[4] = {0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL};
static const uint64_t R3_64[4] = {0x2a9e49687b83a2dbULL, 0x278324e6aef7f3ecULL, 0x8065dc6c04ec5b65ULL, 0x0e530b773599cec7ULL};
static const uint64_t LM2_64[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};

static void cond_sub_l(uint64_t out[4], const uint64_t t[4], uint64_t hi) {
    uint64_t d[4];
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 x = (u128)t[i] - L64[i] - borrow;
        d[i] = (uint64_t)x;
        borrow = (x >> 64) & 1;
    }
    int ge = hi || !borrow;
    for (int i = 0; i < 4; i++) out[i] = ge ? d[i] : t[i];
}
static void montmul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            u128 s = (u128)a[j] * b[i] + t[j] + c;
            t[j] = (uint64_t)s;
            c = s >> 64;
        }
        u128 s = (u128)t[4] + c;
        t[4] = (uint64_t)s;
        t[5] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * LFACTOR64;
        c = ((u128)m * L64[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            u128 s2 = (u128)m * L64[j] + t[j] + c;
            t[j - 1] = (uint64_t)s2;
            c = s2 >> 64;
        }
        s = (u128)t[4] + c;
        t[3] = (uint64_t)s;
        t[4] = t[5] + (uint64_t)(s >> 64);
    }
    cond_sub_l(r, t, t[4]);
}
void sc_mul(scl* r, const scl* a, const scl* b) { montmul(r->v, a->v, b->v); }
void sc_add(scl* r, const scl* a, const scl* b) {
    uint64_t t[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) { u128 s = (u128)a->v[i] + b->v[i] + c; t[i] = (uint64_t)s; c = s >> 64; }
    cond_sub_l(r->v, t, (uint64_t)c);
}
void sc_sub(scl* r, const scl* a, const scl* b) {
    uint64_t t[4];
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) { u128 x = (u128)a->v[i] - b->v[i] - borrow; t[i] = (uint64_t)x; borrow = (x >> 64) & 1; }
    u128 c = 0;
    for (int i = 0; i < 4; i++) { u128 s = (u128)t[i] + (borrow ? L64[i] : 0) + c; r->v[i] = (uint64_t)s; c = s >> 64; }
}
void sc_from_bytes(scl* r, const uint8_t b[32]) {
    uint64_t x[4];
    memcpy(x, b, 32);
    montmul(r->v, x, R2_64);
}
void sc_from_wide(scl* r, const uint8_t b[64]) {
    uint64_t lo[4], hi[4];
    scl a, c;
    memcpy(lo, b, 32); memcpy(hi, b + 32, 32);
    montmul(a.v, lo, R2_64);
    montmul(c.v, hi, R3_64);
    sc_add(r, &a, &c);
}
void sc_to_bytes(uint8_t out[32], const scl* a) {
    uint64_t one[4] = {1, 0, 0, 0}, r[4];
    montmul(r, a->v, one);
    memcpy(out, r, 32);
}
void sc_from_u64(scl* r, uint64_t x) { uint8_t b[32] = {0}; memcpy(b, &x, 8); sc_from_bytes(r, b); }
void sc_inv(scl* r, const scl* a) {
    scl acc = SC_ONE;
    for (int i = 252; i >= 0; i--) {
        sc_mul(&acc, &acc, &acc);
        if ((LM2_64[i >> 6] >> (i & 63)) & 1) sc_mul(&acc, &acc, a);
    }
    *r = acc;
}
void sc_pow(scl* r, const scl* a, uint64_t e) {
    scl acc = SC_ONE, base = *a;
    while (e) { if (e & 1) sc_mul(&acc, &acc, &base); sc_mul(&base, &base, &base); e >>= 1; }
    *r = acc;
}
int sc_is_canonical(const uint8_t b[32]) {
    uint64_t x[4], d[4];
    memcpy(x, b, 32);
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)x[i] - L64[i] - borrow; d[i] = (uint64_t)t; borrow = (t >> 64) & 1; }
    (void)d;
    return borrow != 0;   /* x < l */
}
int sc_is_zero(const scl* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }

/* ------------------------------------------------------------------------------------------------ init */
static int g_inited = 0;
void ref_init(void) {
    if (g_inited) return;
    fe51 one, t, num, den;
    fe_1(&one);
    /* sqrt(-1) = 2^((p-1)/4),  (p-1)/4 = 2^253 - 5 */
    uint8_t e[32];
    memset(e, 0xff, 32);
    e[0] = 0xfb; e[31] = 0x1f;
    fe51 two; fe_from_u64(&two, 2);
    fe_pow_bytes(&REF_SQRT_M1, &two, e);
    /* d = -121665/121666 */
    fe_from_u64(&num, 121665); fe_from_u64(&den, 121666);
    fe_invert(&t, &den); fe_mul(&REF_D, &num, &t); fe_neg(&REF_D, &REF_D);
    fe_add(&REF_D2, &REF_D, &REF_D);
    static const uint8_t sqrt_ad_m1[32] = {0x1b,0x2e,0x7b,0x49,0xa0,0xf6,0x97,0x7e,0xbd,0x54,0x78,0x1b,0x0c,0x8e,0x9d,0xaf,0xfd,0xd1,0xf5,0x31,0xc9,0xfc,0x3c,0x0f,0xac,0x48,0x83,0x2b,0xbf,0x31,0x69,0x37};
    static const uint8_t invsqrt_amd[32] = {0xea,0x40,0x5d,0x80,0xaa,0xfd,0xc8,0x99,0xbe,0x72,0x41,0x5a,0x17,0x16,0x2f,0x9d,0x40,0xd8,0x01,0xfe,0x91,0x7b,0xc2,0x16,0xa2,0xfc,0xaf,0xcf,0x05,0x89,0x6c,0x78};
    fe_frombytes(&REF_SQRT_AD_MINUS_ONE, sqrt_ad_m1);
    fe_frombytes(&REF_INVSQRT_A_MINUS_D, invsqrt_amd);
    fe_sq(&t, &REF_D); fe_sub(&REF_ONE_MINUS_D_SQ, &one, &t);
    fe_sub(&t, &REF_D, &one); fe_sq(&REF_D_MINUS_ONE_SQ, &t);
    /* basepoint: y = 4/5, x = the non-negative root of (y^2-1)/(d y^2+1) */
    fe51 four, five, y, yy, u, v, x;
    fe_from_u64(&four, 4); fe_from_u64(&five, 5);
    fe_invert(&t, &five); fe_mul(&y, &four, &t);
    fe_sq(&yy, &y); fe_sub(&u, &yy, &one); fe_mul(&v, &REF_D, &yy); fe_add(&v, &v, &one);
    fe_sqrt_ratio_m1(&x, &u, &v);
    REF_B.X = x; REF_B.Y = y; fe_1(&REF_B.Z); fe_mul(&REF_B.T, &x, &y);
    memset(&SC_ZERO, 0, sizeof SC_ZERO);
    uint8_t ob[32] = {1};
    sc_from_bytes(&SC_ONE, ob);
    g_inited = 1;
}
