/* oracle/ref_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle / CPU baseline).  Never linked into libdapol_hip.so.
 *
 * Plain-C restatement of the arithmetic the reference reaches through curve25519-dalek-ng 4.1.1 (serial u64
 * backend, Cargo.toml:21): field 2^255-19 in five 51-bit limbs with unsigned __int128 products, ristretto255
 * per RFC 9496, scalars mod l in four 64-bit Montgomery limbs.  Deliberately a different limb design from the
 * GPU code (dapol_amd/csrc: ten 25.5-bit signed limbs, eight 32-bit scalar words) so the two check each other;
 * both are checked against the big-integer Python restatement (oracle/pyref.py).
 */
#ifndef REF_MATH_H
#define REF_MATH_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } fe51;
typedef struct { fe51 X, Y, Z, T; } pt;           /* extended twisted Edwards, a = -1 */
typedef struct { uint64_t v[4]; } scl;            /* Montgomery form mod l, R = 2^256 */

#define MASK51 ((1ULL << 51) - 1)

static inline void fe_0(fe51* h) { memset(h, 0, sizeof *h); }
static inline void fe_1(fe51* h) { fe_0(h); h->v[0] = 1; }
static inline void fe_carry(fe51* h) {
    uint64_t c;
    c = h->v[0] >> 51; h->v[0] &= MASK51; h->v[1] += c;
    c = h->v[1] >> 51; h->v[1] &= MASK51; h->v[2] += c;
    c = h->v[2] >> 51; h->v[2] &= MASK51; h->v[3] += c;
    c = h->v[3] >> 51; h->v[3] &= MASK51; h->v[4] += c;
    c = h->v[4] >> 51; h->v[4] &= MASK51; h->v[0] += 19 * c;
    c = h->v[0] >> 51; h->v[0] &= MASK51; h->v[1] += c;
}
static inline void fe_add(fe51* h, const fe51* f, const fe51* g) {
    for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i];
    fe_carry(h);
}
/* h = f - g, computed as f + 4p - g so limbs stay non-negative (inputs carried: limbs < 2^52) */
static inline void fe_sub(fe51* h, const fe51* f, const fe51* g) {
    h->v[0] = f->v[0] + 0x1FFFFFFFFFFFB4ULL - g->v[0];
    for (int i = 1; i < 5; i++) h->v[i] = f->v[i] + 0x1FFFFFFFFFFFFCULL - g->v[i];
    fe_carry(h);
}
static inline void fe_neg(fe51* h, const fe51* f) { fe51 z; fe_0(&z); fe_sub(h, &z, f); }
static inline void fe_mul(fe51* h, const fe51* f, const fe51* g) {
    const uint64_t *a = f->v, *b = g->v;
    uint64_t b1 = 19 * b[1], b2 = 19 * b[2], b3 = 19 * b[3], b4 = 19 * b[4];
    u128 c0 = (u128)a[0] * b[0] + (u128)a[1] * b4 + (u128)a[2] * b3 + (u128)a[3] * b2 + (u128)a[4] * b1;
    u128 c1 = (u128)a[0] * b[1] + (u128)a[1] * b[0] + (u128)a[2] * b4 + (u128)a[3] * b3 + (u128)a[4] * b2;
    u128 c2 = (u128)a[0] * b[2] + (u128)a[1] * b[1] + (u128)a[2] * b[0] + (u128)a[3] * b4 + (u128)a[4] * b3;
    u128 c3 = (u128)a[0] * b[3] + (u128)a[1] * b[2] + (u128)a[2] * b[1] + (u128)a[3] * b[0] + (u128)a[4] * b4;
    u128 c4 = (u128)a[0] * b[4] + (u128)a[1] * b[3] + (u128)a[2] * b[2] + (u128)a[3] * b[1] + (u128)a[4] * b[0];
    c1 += (uint64_t)(c0 >> 51); uint64_t r0 = (uint64_t)c0 & MASK51;
    c2 += (uint64_t)(c1 >> 51); uint64_t r1 = (uint64_t)c1 & MASK51;
    c3 += (uint64_t)(c2 >> 51); uint64_t r2 = (uint64_t)c2 & MASK51;
    c4 += (uint64_t)(c3 >> 51); uint64_t r3 = (uint64_t)c3 & MASK51;
    uint64_t carry = (uint64_t)(c4 >> 51), r4 = (uint64_t)c4 & MASK51;
    r0 += 19 * carry;
    r1 += r0 >> 51; r0 &= MASK51;
    h->v[0] = r0; h->v[1] = r1; h->v[2] = r2; h->v[3] = r3; h->v[4] = r4;
}
static inline void fe_sq(fe51* h, const fe51* f) { fe_mul(h, f, f); }
static inline void fe_sqn(fe51* h, const fe51* f, int n) { fe_sq(h, f); for (int i = 1; i < n; i++) fe_sq(h, h); }
static void fe_pow22523(fe51* out, const fe51* z) {
    fe51 t0, t1, t2;
    fe_sq(&t0, z); fe_sqn(&t1, &t0, 2); fe_mul(&t1, z, &t1); fe_mul(&t0, &t0, &t1); fe_sq(&t0, &t0); fe_mul(&t0, &t1, &t0);
    fe_sqn(&t1, &t0, 5); fe_mul(&t0, &t1, &t0); fe_sqn(&t1, &t0, 10); fe_mul(&t1, &t1, &t0); fe_sqn(&t2, &t1, 20);
    fe_mul(&t1, &t2, &t1); fe_sqn(&t1, &t1, 10); fe_mul(&t0, &t1, &t0); fe_sqn(&t1, &t0, 50); fe_mul(&t1, &t1, &t0);
    fe_sqn(&t2, &t1, 100); fe_mul(&t1, &t2, &t1); fe_sqn(&t1, &t1, 50); fe_mul(&t0, &t1, &t0); fe_sqn(&t0, &t0, 2);
    fe_mul(out, &t0, z);
}
static void fe_tobytes(uint8_t* s, const fe51* f) {
    fe51 t = *f;
    fe_carry(&t);
    fe_carry(&t);
    /* conditional subtract p: add 19, see if it overflows 2^255 */
    uint64_t q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51; q = (t.v[2] + q) >> 51; q = (t.v[3] + q) >> 51; q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= MASK51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= MASK51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= MASK51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= MASK51; t.v[4] += c;
    t.v[4] &= MASK51;
    uint64_t w0 = t.v[0] | (t.v[1] << 51), w1 = (t.v[1] >> 13) | (t.v[2] << 38), w2 = (t.v[2] >> 26) | (t.v[3] << 25),
             w3 = (t.v[3] >> 39) | (t.v[4] << 12);
    memcpy(s, &w0, 8); memcpy(s + 8, &w1, 8); memcpy(s + 16, &w2, 8); memcpy(s + 24, &w3, 8);
}
static void fe_frombytes(fe51* h, const uint8_t* s) { /* bit 255 ignored */
    uint64_t w[4];
    memcpy(w, s, 32);
    h->v[0] = w[0] & MASK51;
    h->v[1] = ((w[0] >> 51) | (w[1] << 13)) & MASK51;
    h->v[2] = ((w[1] >> 38) | (w[2] << 26)) & MASK51;
    h->v[3] = ((w[2] >> 25) | (w[3] << 39)) & MASK51;
    h->v[4] = (w[3] >> 12) & MASK51;
}
static inline int fe_isneg(const fe51* f) { uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1; }
static inline int fe_iszero(const fe51* f) { uint8_t s[32]; fe_tobytes(s, f); uint8_t o = 0; for (int i = 0; i < 32; i++) o |= s[i]; return o == 0; }
static inline int fe_eq(const fe51* f, const fe51* g) { uint8_t a[32], b[32]; fe_tobytes(a, f); fe_tobytes(b, g); return memcmp(a, b, 32) == 0; }
static inline void fe_abs(fe51* h, const fe51* f) { if (fe_isneg(f)) fe_neg(h, f); else *h = *f; }
static inline void fe_from_u64(fe51* h, uint64_t x) { fe_0(h); h->v[0] = x & MASK51; h->v[1] = x >> 51; }

/* curve constants, filled by ref_init() from their definitions */
extern fe51 REF_D, REF_D2, REF_SQRT_M1, REF_SQRT_AD_MINUS_ONE, REF_INVSQRT_A_MINUS_D, REF_ONE_MINUS_D_SQ, REF_D_MINUS_ONE_SQ;
extern pt REF_B, REF_BB;
void ref_init(void);

int fe_sqrt_ratio_m1(fe51* r, const fe51* u, const fe51* v);
void pt_identity(pt* r);
void pt_add(pt* r, const pt* p, const pt* q);
void pt_dbl(pt* r, const pt* p);
void pt_neg(pt* r, const pt* p);
void pt_compress(uint8_t out[32], const pt* p);
int pt_decompress(pt* r, const uint8_t in[32]);
void pt_from_uniform(pt* r, const uint8_t in[64]);
void pt_mul(pt* r, const pt* p, const uint8_t k[32]);                 /* k: 256-bit little-endian integer */
void pt_msm_vartime(pt* r, const scl* scalars, const pt* points, size_t n);

/* scalars */
void sc_from_bytes(scl* r, const uint8_t b[32]);       /* any 256-bit integer -> Montgomery form of it mod l */
void sc_from_wide(scl* r, const uint8_t b[64]);
void sc_to_bytes(uint8_t out[32], const scl* a);       /* canonical */
void sc_mul(scl* r, const scl* a, const scl* b);
void sc_add(scl* r, const scl* a, const scl* b);
void sc_sub(scl* r, const scl* a, const scl* b);
void sc_inv(scl* r, const scl* a);
void sc_from_u64(scl* r, uint64_t x);
void sc_pow(scl* r, const scl* a, uint64_t e);
int sc_is_canonical(const uint8_t b[32]);
int sc_is_zero(const scl* a);
extern scl SC_ONE, SC_ZERO;

#endif
