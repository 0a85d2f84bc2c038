// Twisted-Edwards (a = -1) group law on Curve25519 and the ristretto255 codec (RFC 9496) over fe.h.
// Replaces what the reference reaches through curve25519-dalek-ng: RistrettoPoint add (src/dapol/node.rs:76,
// src/proof/node.rs:58), compress (node.rs:35,67,68), decompress (src/proof/node.rs:88), from_uniform_bytes
// (bulletproofs generators).  Every formula is annotated with the fe.h TIGHT/LOOSE operand discipline.
#pragma once
#include "consts.h"
#include "fe.h"

namespace dapol {

struct ge_p3 {  // extended coordinates, x = X/Z, y = Y/Z, T = XY/Z; all four REDUCED (non-negative limbs)
    fe X, Y, Z, T;
};
struct ge_niels {  // affine precomputed point: (y+x, y-x, 2dxy), reduced
    fe ypx, ymx, xy2d;
};

DAPOL_HD void ge_identity(ge_p3& r) {
    fe_0(r.X);
    fe_1(r.Y);
    fe_1(r.Z);
    fe_0(r.T);
}
DAPOL_HD void ge_niels_identity(ge_niels& r) {
    fe_1(r.ypx);
    fe_1(r.ymx);
    fe_0(r.xy2d);
}
DAPOL_HD void ge_neg(ge_p3& r, const ge_p3& p) {
    fe t;
    fe_neg(t, p.X);
    fe_carry(r.X, t);               // keep the coordinates reduced
    r.Y = p.Y;
    r.Z = p.Z;
    fe_neg(t, p.T);
    fe_carry(r.T, t);
}

// r = p + (neg ? -q : q), q affine precomputed.  7 mul + 1 carry.
DAPOL_HD void ge_madd(ge_p3& r, const ge_p3& p, const ge_niels& q, bool neg) {
    fe ypx, ymx, A, B, C, D, E, F, G, H, qa = q.ymx, qb = q.ypx;
    fe_cswap(qa, qb, neg);          // -q swaps y+x <-> y-x and negates 2dxy
    fe_add(ypx, p.Y, p.X);          // loose(2)
    fe_sub(ymx, p.Y, p.X);          // tight
    fe_mul(A, ymx, qa);
    fe_mul(B, ypx, qb);
    fe_mul(C, p.T, q.xy2d);
    fe nC;
    fe_neg(nC, C);
    fe_cmov(C, nC, neg);            // tight either way
    fe_add(D, p.Z, p.Z);            // loose(2)
    fe_sub(E, B, A);                // tight
    fe_add(H, B, A);                // loose(2)
    fe_sub(F, D, C);                // loose(3)
    fe_add(G, D, C);                // loose(3)
    fe_carry(G, G);                 // -> reduced, usable as g
    fe_mul(r.X, F, E);
    fe_mul(r.Y, H, G);
    fe_mul(r.Z, F, G);
    fe_mul(r.T, H, E);
}

// r = p + q, both extended.  9 mul + 2 carry.
DAPOL_HD void ge_add(ge_p3& r, const ge_p3& p, const ge_p3& q) {
    fe ypx, ymx, qypx, qymx, A, B, C, D, E, F, G, H;
    fe_add(ypx, p.Y, p.X);          // loose
    fe_sub(ymx, p.Y, p.X);          // tight
    fe_addc(qypx, q.Y, q.X);        // reduced
    fe_sub(qymx, q.Y, q.X);         // tight
    fe_mul(A, ymx, qymx);
    fe_mul(B, ypx, qypx);
    fe_mul(C, p.T, q.T);
    fe_mul(C, C, FE_D2);
    fe_mul(D, p.Z, q.Z);
    fe_add(D, D, D);                // loose(2)
    fe_sub(E, B, A);
    fe_add(H, B, A);
    fe_sub(F, D, C);
    fe_add(G, D, C);
    fe_carry(G, G);
    fe_mul(r.X, F, E);
    fe_mul(r.Y, H, G);
    fe_mul(r.Z, F, G);
    fe_mul(r.T, H, E);
}

// r = 2p.  3 sq + 1 mul + 1 carry + (4 or 3) mul; T is only produced when want_t (it is not an input).
DAPOL_HD void ge_dbl(ge_p3& r, const ge_p3& p, bool want_t) {
    fe XX, YY, ZZ, E, F, G, H;
    fe_sq(XX, p.X);
    fe_sq(YY, p.Y);
    fe_sq(ZZ, p.Z);
    fe_mul(E, p.X, p.Y);
    fe_add(E, E, E);
    fe_carry(E, E);                 // E = 2XY reduced
    fe_sub(G, YY, XX);              // tight
    fe_add(H, XX, YY);
    fe_neg(H, H);                   // loose(2):  H = -(XX+YY)
    fe_sub(F, G, ZZ);
    fe_sub(F, F, ZZ);               // loose(3):  F = G - 2ZZ
    fe_mul(r.X, F, E);
    fe_mul(r.Y, H, G);
    fe_mul(r.Z, F, G);
    if (want_t) fe_mul(r.T, H, E);
}

// Projective precomputed point for repeated additions of the same variable point: (Y+X, Y-X, Z, 2dT), all reduced.
struct ge_cached {
    fe YpX, YmX, Z, T2d;
};
DAPOL_HD void ge_to_cached(ge_cached& r, const ge_p3& p) {
    fe t;
    fe_addc(r.YpX, p.Y, p.X);
    fe_sub(t, p.Y, p.X);
    fe_carry(r.YmX, t);
    r.Z = p.Z;
    fe_mul(r.T2d, p.T, FE_D2);
}
// r = p + (neg ? -q : q).  8 mul + 1 carry.
DAPOL_HD void ge_add_cached(ge_p3& r, const ge_p3& p, const ge_cached& q, bool neg) {
    fe ypx, ymx, A, B, C, D, E, F, G, H, qa = q.YmX, qb = q.YpX;
    fe_cswap(qa, qb, neg);
    fe_add(ypx, p.Y, p.X);          // loose(2)
    fe_sub(ymx, p.Y, p.X);          // tight
    fe_mul(A, ymx, qa);
    fe_mul(B, ypx, qb);
    fe_mul(C, p.T, q.T2d);
    fe nC;
    fe_neg(nC, C);
    fe_cmov(C, nC, neg);
    fe_mul(D, p.Z, q.Z);
    fe_add(D, D, D);                // loose(2)
    fe_sub(E, B, A);
    fe_add(H, B, A);
    fe_sub(F, D, C);
    fe_add(G, D, C);
    fe_carry(G, G);
    fe_mul(r.X, F, E);
    fe_mul(r.Y, H, G);
    fe_mul(r.Z, F, G);
    fe_mul(r.T, H, E);
}

DAPOL_HD void ge_to_niels(ge_niels& r, const fe& x, const fe& y) {  // affine x, y (reduced)
    fe t;
    fe_addc(r.ypx, y, x);
    fe_sub(t, y, x);
    fe_carry(r.ymx, t);
    fe_mul(t, x, y);
    fe_mul(r.xy2d, t, FE_D2);
}

// RFC 9496 4.2 SQRT_RATIO_M1.  u, v tight.  Returns was_square; r is the non-negative root (reduced).
DAPOL_HD_NOINLINE bool fe_sqrt_ratio_m1(fe& r, const fe& u, const fe& v) {
    fe v3, v7, t, check, nu, nui;
    fe_sq(v3, v);
    fe_mul(v3, v3, v);              // v^3
    fe_sq(v7, v3);
    fe_mul(v7, v7, v);              // v^7
    fe_mul(t, v7, u);               // u v^7
    fe_pow22523(t, t);
    fe_mul(t, t, v3);
    fe_mul(r, t, u);                // r = u v^3 (u v^7)^((p-5)/8)
    fe_sq(check, r);
    fe_mul(check, check, v);
    fe_neg(nu, u);
    fe_mul(nui, nu, FE_SQRT_M1);
    bool correct = fe_equal(check, u);
    bool flipped = fe_equal(check, nu);
    bool flipped_i = fe_equal(check, nui);
    fe rp;
    fe_mul(rp, r, FE_SQRT_M1);
    fe_cmov(r, rp, flipped | flipped_i);
    fe_abs(r, r);
    return correct | flipped;
}

// RFC 9496 4.3.2 Encode -> eight little-endian words.
DAPOL_HD_NOINLINE void ge_compress(uint32_t* out, const ge_p3& p) {
    fe u1, u2, t, invsqrt, den1, den2, z_inv, ix, iy, ench, x, y, den_inv, one;
    fe_add(t, p.Z, p.Y);            // loose
    fe_sub(u1, p.Z, p.Y);           // tight
    fe_mul(u1, t, u1);
    fe_mul(u2, p.X, p.Y);
    fe_sq(t, u2);
    fe_mul(t, t, u1);
    fe_1(one);
    fe_sqrt_ratio_m1(invsqrt, one, t);
    fe_mul(den1, invsqrt, u1);
    fe_mul(den2, invsqrt, u2);
    fe_mul(z_inv, den1, den2);
    fe_mul(z_inv, z_inv, p.T);
    fe_mul(ix, p.X, FE_SQRT_M1);
    fe_mul(iy, p.Y, FE_SQRT_M1);
    fe_mul(ench, den1, FE_INVSQRT_A_MINUS_D);
    fe_mul(t, p.T, z_inv);
    bool rotate = fe_isnegative(t);
    x = p.X;
    y = p.Y;
    den_inv = den2;
    fe_cmov(x, iy, rotate);
    fe_cmov(y, ix, rotate);
    fe_cmov(den_inv, ench, rotate);
    fe_mul(t, x, z_inv);
    fe ny;
    fe_neg(ny, y);
    fe_cmov(y, ny, fe_isnegative(t));
    fe_sub(t, p.Z, y);              // loose(2)
    fe_mul(t, t, den_inv);
    fe_abs(t, t);
    fe_towords(out, t);
}

// RFC 9496 4.3.1 Decode from eight words.  Returns false for non-canonical / invalid encodings.
DAPOL_HD_NOINLINE bool ge_decompress(ge_p3& r, const uint32_t* in) {
    fe s, ss, u1, u2, u2_sqr, v, t, invsqrt, den_x, den_y, one;
    fe_fromwords(s, in);
    uint32_t chk[8];
    fe_towords(chk, s);
    bool canonical = true;
    for (int i = 0; i < 8; i++) canonical &= (chk[i] == in[i]);   // also rejects bit 255 set and s >= p
    bool nonneg = (in[0] & 1) == 0;
    fe_sq(ss, s);
    fe_1(one);
    fe_sub(u1, one, ss);            // tight
    fe_addc(u2, one, ss);           // reduced
    fe_sq(u2_sqr, u2);
    fe_sq(t, u1);
    fe_mul(t, t, FE_D);
    fe_neg(t, t);
    fe_sub(v, t, u2_sqr);           // loose(2)
    fe_carry(v, v);
    fe_mul(t, v, u2_sqr);
    bool was_square = fe_sqrt_ratio_m1(invsqrt, one, t);
    fe_mul(den_x, invsqrt, u2);
    fe_mul(den_y, invsqrt, den_x);
    fe_mul(den_y, den_y, v);
    fe_mul(t, s, den_x);
    fe_add(t, t, t);
    fe_carry(t, t);
    fe_abs(r.X, t);
    fe_mul(r.Y, u1, den_y);
    fe_1(r.Z);
    fe_mul(r.T, r.X, r.Y);
    return canonical & nonneg & was_square & !fe_isnegative(r.T) & !fe_iszero(r.Y);
}

// RFC 9496 4.3.4 MAP (Elligator 2 for ristretto255); t reduced.
DAPOL_HD_NOINLINE void ge_elligator(ge_p3& out, const fe& t) {
    fe r, u, v, a, b, s, s_prime, c, N, w0, w1, w2, w3, one, tmp;
    fe_1(one);
    fe_sq(r, t);
    fe_mul(r, r, FE_SQRT_M1);       // r = i t^2
    fe_addc(u, r, one);
    fe_mul(u, u, FE_ONE_MINUS_D_SQ);
    fe_mul(a, r, FE_D);             // r d
    fe_add(a, a, one);
    fe_neg(a, a);
    fe_carry(a, a);                 // -1 - r d
    fe_addc(b, r, FE_D);            // r + d
    fe_mul(v, a, b);
    bool was_square = fe_sqrt_ratio_m1(s, u, v);
    fe_mul(s_prime, s, t);
    fe_abs(s_prime, s_prime);
    fe_neg(s_prime, s_prime);
    fe_cmov(s_prime, s, was_square);   // s_prime now holds the selected s
    s = s_prime;
    fe_neg(c, one);
    fe mone = c;
    c = r;
    fe_cmov(c, mone, was_square);   // c = was_square ? -1 : r
    fe_sub(tmp, r, one);            // tight
    fe_mul(N, c, tmp);
    fe_mul(N, N, FE_D_MINUS_ONE_SQ);
    fe_sub(N, N, v);                // tight
    fe_mul(w0, s, v);
    fe_add(w0, w0, w0);
    fe_carry(w0, w0);               // 2 s v
    fe_mul(w1, N, FE_SQRT_AD_MINUS_ONE);
    fe_sq(tmp, s);
    fe_sub(w2, one, tmp);           // tight
    fe_addc(w3, one, tmp);          // reduced
    fe_mul(out.X, w0, w3);
    fe_mul(out.Y, w2, w1);
    fe_mul(out.Z, w1, w3);
    fe_mul(out.T, w0, w2);
}

// RistrettoPoint::from_uniform_bytes on sixteen little-endian words (bit 255 of each half ignored).
DAPOL_HD void ge_from_uniform(ge_p3& out, const uint32_t* w16) {
    fe t1, t2;
    ge_p3 p1, p2;
    fe_fromwords(t1, w16);
    fe_fromwords(t2, w16 + 8);
    ge_elligator(p1, t1);
    ge_elligator(p2, t2);
    ge_add(out, p1, p2);
}

DAPOL_HD void ge_basepoint(ge_p3& r) {
    r.X = FE_BASE_X;
    r.Y = FE_BASE_Y;
    fe_1(r.Z);
    r.T = FE_BASE_T;
}

// ristretto equality: X1 Y2 == Y1 X2  or  Y1 Y2 == X1 X2
DAPOL_HD bool ge_equal(const ge_p3& p, const ge_p3& q) {
    fe a, b, c, d;
    fe_mul(a, p.X, q.Y);
    fe_mul(b, p.Y, q.X);
    fe_mul(c, p.Y, q.Y);
    fe_mul(d, p.X, q.X);
    return fe_equal(a, b) | fe_equal(c, d);
}

}  // namespace dapol
