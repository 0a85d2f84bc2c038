// Point arithmetic with ONE POINT PER FOUR LANES: lane 4g + k of a wavefront holds coordinate k (X, Y, Z, T) of group g's
// extended point.  The formulas of ge.h are two layers of four independent field products (doubling: X^2, Y^2, Z^2, (X+Y)^2, then
// F*E, H*G, F*G, H*E; mixed addition: (Y-X)(y-x), (Y+X)(y+x), 2Z, T*2dxy, then the same four), so a group does a point operation
// in two field products plus two exchanges inside the quad (DPP quad permutes: one instruction per limb, no LDS) instead of
// seven or eight products on one lane: 360 instructions deep instead of 800-980.  Throughput per point is no better -- four lanes
// are busy -- which is why only calls of a few proofs use it (k_rp_msm_quad): there the chain of W * (windows - 1) doublings on
// a lone wavefront is what the caller waits for.  Every lane of the wavefront must be active in these functions.
#pragma once
#include "tables.h"

namespace dapol {
#if defined(__HIPCC__)

template <int K>
__device__ __forceinline__ void quad_bcast_fe(fe& o, const fe& a) {      // coordinate K of the lane's group, on all four lanes
    for (int i = 0; i < FE_NL; i++) o.v[i] = __builtin_amdgcn_update_dpp(0, a.v[i], K * 0x55, 0xf, 0xf, false);
}
__device__ __forceinline__ void quad_identity(fe& c, int ql) {
    fe_0(c);
    if (ql == 1 || ql == 2) c.v[0] = 1;
}
// second layer, shared by all three operations: X3 = F E, Y3 = H G, Z3 = F G, T3 = H E  (E, G tight; F, H loose)
__device__ __forceinline__ void quad_finish(fe& c, int ql, const fe& E, const fe& F, const fe& G, const fe& H) {
    fe f, g;
    const bool odd = ql & 1, outer = ql == 0 || ql == 3;
    for (int i = 0; i < FE_NL; i++) { f.v[i] = odd ? H.v[i] : F.v[i]; g.v[i] = outer ? E.v[i] : G.v[i]; }
    fe_mul(c, f, g);
}
// group's point <- 2 * point  (ge_dbl's formulas with 2XY = (X+Y)^2 - X^2 - Y^2, so that the first layer is four squarings)
__device__ __forceinline__ void quad_dbl(fe& c, int ql) {
    fe X, Y, s, a, r1, XX, YY, ZZ, tt, E, F, G, H;
    quad_bcast_fe<0>(X, c);
    quad_bcast_fe<1>(Y, c);
    fe_addc(s, X, Y);
    for (int i = 0; i < FE_NL; i++) a.v[i] = ql == 3 ? s.v[i] : c.v[i];
    fe_sq(r1, a);
    quad_bcast_fe<0>(XX, r1);
    quad_bcast_fe<1>(YY, r1);
    quad_bcast_fe<2>(ZZ, r1);
    quad_bcast_fe<3>(tt, r1);
    fe_sub(E, tt, XX);
    fe_sub(E, E, YY);
    fe_carry(E, E);                 // E = 2XY, tight
    fe_sub(G, YY, XX);              // tight
    fe_add(H, XX, YY);
    fe_neg(H, H);                   // loose(2):  H = -(XX+YY)
    fe_sub(F, G, ZZ);
    fe_sub(F, F, ZZ);               // loose(3):  F = G - 2ZZ
    quad_finish(c, ql, E, F, G, H);
}
// group's point += (neg ? -q : q) for an affine-niels table entry q, of which lane 0 holds y-x (y+x if neg), lane 1 y+x (y-x if
// neg) and lane 3 2dxy in `qel` (lane 2's is ignored): ge_madd's formulas, 2Z as the fourth product of the first layer.
__device__ __forceinline__ void quad_madd(fe& c, int ql, const fe& qel, bool neg) {
    fe X, Y, ypx, ymx, f, g, r1, A, B, C, D, nC, E, F, G, H;
    quad_bcast_fe<0>(X, c);
    quad_bcast_fe<1>(Y, c);
    fe_add(ypx, Y, X);              // loose(2)
    fe_sub(ymx, Y, X);              // tight
    for (int i = 0; i < FE_NL; i++) {
        f.v[i] = ql == 0 ? ymx.v[i] : ql == 1 ? ypx.v[i] : c.v[i];
        g.v[i] = ql == 2 ? (i == 0 ? 2 : 0) : qel.v[i];
    }
    fe_mul(r1, f, g);
    quad_bcast_fe<0>(A, r1);
    quad_bcast_fe<1>(B, r1);
    quad_bcast_fe<2>(D, r1);
    quad_bcast_fe<3>(C, r1);
    fe_neg(nC, C);
    fe_cmov(C, nC, neg);            // tight either way
    fe_sub(E, B, A);                // tight
    fe_add(H, B, A);                // loose(2)
    fe_sub(F, D, C);                // loose(2)
    fe_add(G, D, C);
    fe_carry(G, G);                 // -> reduced, usable as g
    quad_finish(c, ql, E, F, G, H);
}
// group's point += another group's point, whose coordinates arrive lane by lane in `d` (ge_add's formulas; the constants 2 and 2d
// enter as a product of their own after the first layer)
__device__ __forceinline__ void quad_add(fe& c, int ql, const fe& d) {
    fe X1, Y1, X2, Y2, t1, t2, t3, t4, f, g, k, r1, r2, A, B, C, D, E, F, G, H;
    quad_bcast_fe<0>(X1, c);
    quad_bcast_fe<1>(Y1, c);
    quad_bcast_fe<0>(X2, d);
    quad_bcast_fe<1>(Y2, d);
    fe_sub(t1, Y1, X1);             // tight
    fe_add(t2, Y1, X1);             // loose
    fe_sub(t3, Y2, X2);             // tight
    fe_addc(t4, Y2, X2);            // reduced
    for (int i = 0; i < FE_NL; i++) {
        f.v[i] = ql == 0 ? t1.v[i] : ql == 1 ? t2.v[i] : c.v[i];
        g.v[i] = ql == 0 ? t3.v[i] : ql == 1 ? t4.v[i] : d.v[i];
        k.v[i] = ql == 3 ? FE_D2.v[i] : (i == 0 ? (ql == 2 ? 2 : 1) : 0);
    }
    fe_mul(r1, f, g);
    fe_mul(r2, r1, k);              // A | B | 2 Z1 Z2 | 2d T1 T2
    quad_bcast_fe<0>(A, r2);
    quad_bcast_fe<1>(B, r2);
    quad_bcast_fe<2>(D, r2);
    quad_bcast_fe<3>(C, r2);
    fe_sub(E, B, A);
    fe_add(H, B, A);
    fe_sub(F, D, C);
    fe_add(G, D, C);
    fe_carry(G, G);
    quad_finish(c, ql, E, F, G, H);
}

// The group's point as the multiplier of later additions ("cached" form, ge.h): lane 0 y-x, lane 1 y+x, lane 2 Z, lane 3 2dT.
__device__ __forceinline__ void quad_to_cached(fe& q, const fe& c, int ql) {
    fe X, Y, a, s, t;
    quad_bcast_fe<0>(X, c);
    quad_bcast_fe<1>(Y, c);
    fe_sub(a, Y, X);
    fe_carry(a, a);
    fe_addc(s, Y, X);
    fe_mul(t, c, FE_D2);
    for (int i = 0; i < FE_NL; i++) q.v[i] = ql == 0 ? a.v[i] : ql == 1 ? s.v[i] : ql == 2 ? c.v[i] : t.v[i];
}
// group's point += (neg ? -q : q), q in the cached form above, one element per lane (ge_add_cached's formulas)
__device__ __forceinline__ void quad_add_cached(fe& c, int ql, const fe& q, bool neg) {
    fe X, Y, ypx, ymx, qs, f, g, r1, A, B, C, D, nC, E, F, G, H;
    quad_bcast_fe<0>(X, c);
    quad_bcast_fe<1>(Y, c);
    fe_add(ypx, Y, X);              // loose(2)
    fe_sub(ymx, Y, X);              // tight
    for (int i = 0; i < FE_NL; i++) qs.v[i] = __builtin_amdgcn_update_dpp(0, q.v[i], 0xE1 /* quad_perm 1,0,2,3 */, 0xf, 0xf, false);
    for (int i = 0; i < FE_NL; i++) {
        f.v[i] = ql == 0 ? ymx.v[i] : ql == 1 ? ypx.v[i] : c.v[i];
        g.v[i] = neg ? qs.v[i] : q.v[i];                             // -q swaps y-x <-> y+x (and negates 2dT, below)
    }
    fe_mul(r1, f, g);
    quad_bcast_fe<0>(A, r1);
    quad_bcast_fe<1>(B, r1);
    quad_bcast_fe<2>(D, r1);
    quad_bcast_fe<3>(C, r1);
    fe_neg(nC, C);
    fe_cmov(C, nC, neg);
    fe_add(D, D, D);                // loose(2)
    fe_sub(E, B, A);
    fe_add(H, B, A);
    fe_sub(F, D, C);                // loose(3)
    fe_add(G, D, C);
    fe_carry(G, G);
    quad_finish(c, ql, E, F, G, H);
}

#endif
}  // namespace dapol
