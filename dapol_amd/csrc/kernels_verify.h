// Batched verifier: RangeProof::verify_multiple (bulletproofs 4.0.0) as called from src/range/mod.rs:83-119.
// One multiscalar check per proof:  sum_i g_i G_i + sum_i h_i H_i  (fixed-base: the prover's MSM kernel)
//   + A + x S + c x T1 + c x^2 T2 + sum u_k^2 L_k + sum u_k^-2 R_k + sum c z^(2+j) V_j   (the proof's own points)
//   + (-mu - c tau) B_blinding + (w (t_x - a b) + c (delta - t_x)) B   ==  identity.
#pragma once
#include "kernels_range.h"

namespace dapol {

enum { RV_MAX_ROUNDS = 20 };
struct VerifyState {                 // per proof
    sc y, z, y_inv, x, w, c, a, b, t_x, tau, mu;
    sc rho;                          // weight of this proof in a cross-proof batch (random linear combination)
    sc u[RV_MAX_ROUNDS], u_inv[RV_MAX_ROUNDS];
    uint32_t ok, pad_[3];
};
struct VerifyArgs {
    RangeArgs R;                     // reuses n, m, N, lgN, TP, B, Vc, dig, P0, P1, PA(=P2), out(=proof words), out_words, seed
    VerifyState* vs;
    uint8_t* verdict;                // [B]
    // Per-proof power tables (k_rv_tables), tab_stride scalars per proof:
    //   SH[2^hb] | SL[2^lb] : s_i = SH[i >> lb] * SL[i & (2^lb - 1)]        (lgN = hb + lb)
    //   YH[2^hb] | YL[2^lb] : y^-q = YH[q >> lb] * YL[q & (2^lb - 1)]
    //   ZZ[m]               : z^2 z^j
    sc* tabs;
    int hb, lb, tab_stride;
};
__host__ __device__ inline int rv_tab_entries(int lgN, int m) {
    int lb = lgN / 2, hb = lgN - lb;
    return 2 * (1 << hb) + 2 * (1 << lb) + m;
}

__device__ __forceinline__ bool words_canonical_scalar(const uint32_t* w) {   // w < l
    uint64_t borrow = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)w[i] - SC_L[i] - borrow;
        borrow = (x >> 32) & 1;
    }
    return borrow != 0;
}
__device__ __forceinline__ bool words_zero(const uint32_t* w) {
    uint32_t o = 0;
    for (int i = 0; i < 8; i++) o |= w[i];
    return o == 0;
}

// V1: parse + replay the transcript (lane per proof).
__global__ __launch_bounds__(64) void k_rv_transcript(VerifyArgs V) {
    const RangeArgs& A = V.R;
    size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B) return;
    VerifyState& vs = V.vs[b];
    const uint32_t* pr = A.out + b * A.out_words;
    bool ok = true;
    uint32_t w8[8];
    Strobe s;
    merlin_init(s, "", 0);
    merlin_append_bytes(s, "dom-sep", 7, "rangeproof v1", 13);
    merlin_append_u64(s, "n", 1, (uint64_t)A.n);
    merlin_append_u64(s, "m", 1, (uint64_t)A.m);
    for (int j = 0; j < A.m; j++) {
        ld8(w8, A.Vc + (b * A.m + j) * 8);
        merlin_append_words(s, "V", 1, w8, 8);
    }
    ld8(w8, pr);      ok &= !words_zero(w8); merlin_append_words(s, "A", 1, w8, 8);      // validate_and_append_point
    ld8(w8, pr + 8);  ok &= !words_zero(w8); merlin_append_words(s, "S", 1, w8, 8);
    challenge_scalar(vs.y, s, "y", 1);
    challenge_scalar(vs.z, s, "z", 1);
    ld8(w8, pr + 16); ok &= !words_zero(w8); merlin_append_words(s, "T_1", 3, w8, 8);
    ld8(w8, pr + 24); ok &= !words_zero(w8); merlin_append_words(s, "T_2", 3, w8, 8);
    challenge_scalar(vs.x, s, "x", 1);
    uint32_t tx[8], tau[8], mu[8], aw[8], bw[8];
    ld8(tx, pr + 32); ld8(tau, pr + 40); ld8(mu, pr + 48);
    ld8(aw, pr + 56 + 16 * A.lgN); ld8(bw, pr + 64 + 16 * A.lgN);
    ok &= words_canonical_scalar(tx) & words_canonical_scalar(tau) & words_canonical_scalar(mu) & words_canonical_scalar(aw) &
          words_canonical_scalar(bw);                                                   // Scalar::from_canonical_bytes
    append_scalar(s, "t_x", 3, tx);
    append_scalar(s, "t_x_blinding", 12, tau);
    append_scalar(s, "e_blinding", 10, mu);
    challenge_scalar(vs.w, s, "w", 1);
    merlin_append_bytes(s, "dom-sep", 7, "ipp v1", 6);
    merlin_append_u64(s, "n", 1, (uint64_t)A.N);
    for (int k = 0; k < A.lgN; k++) {
        ld8(w8, pr + 56 + 16 * k);     ok &= !words_zero(w8); merlin_append_words(s, "L", 1, w8, 8);
        ld8(w8, pr + 56 + 16 * k + 8); ok &= !words_zero(w8); merlin_append_words(s, "R", 1, w8, 8);
        challenge_scalar(vs.u[k], s, "u", 1);
        sc_invert_mont(vs.u_inv[k], vs.u[k]);
    }
    sc_invert_mont(vs.y_inv, vs.y);
    sc_to_mont(vs.t_x, tx); sc_to_mont(vs.tau, tau); sc_to_mont(vs.mu, mu); sc_to_mont(vs.a, aw); sc_to_mont(vs.b, bw);
    // batching scalar c: Scalar::random(rng) in the crate; here seed mode, domain 3, keyed by the proof's position
    uint32_t seed[8], wide[16];
    for (int i = 0; i < 8; i++) seed[i] = A.seed[i];
    seed_wide(wide, seed, 3u, (uint64_t)b, 0);
    sc_from_wide(vs.c, wide);
    if (sc_is_zero(vs.c)) sc_one_mont(vs.c);
    seed_wide(wide, seed, 5u, (uint64_t)b, 0);
    sc_from_wide(vs.rho, wide);
    if (sc_is_zero(vs.rho)) sc_one_mont(vs.rho);
    vs.ok = ok ? 1u : 0u;
}

// V2a: the power tables of every proof (lane per entry).  grid = B * ceil(tab_stride / 64) blocks.
__global__ __launch_bounds__(64) void k_rv_tables(VerifyArgs V) {
    const RangeArgs& A = V.R;
    const int bpp = (V.tab_stride + 63) >> 6;
    size_t b = blockIdx.x / bpp;
    int e = (int)(blockIdx.x % bpp) * 64 + threadIdx.x;
    if (e >= V.tab_stride) return;
    const VerifyState& vs = V.vs[b];
    const int nh = 1 << V.hb, nl = 1 << V.lb;
    sc r;
    if (e < nh + nl) {                                  // products of u_k^{+-1}: MSB of i <-> first round
        bool hi = e < nh;
        int x = hi ? e : e - nh, k0 = hi ? 0 : V.hb, nbits = hi ? V.hb : V.lb;
        sc_one_mont(r);
        for (int k = 0; k < nbits; k++) {
            bool bit = (x >> (nbits - 1 - k)) & 1;
            sc_montmul(r, r, bit ? vs.u[k0 + k] : vs.u_inv[k0 + k]);
        }
    } else if (e < 2 * (nh + nl)) {
        int x = e - nh - nl;
        sc_pow_mont(r, vs.y_inv, x < nh ? (uint32_t)x << V.lb : (uint32_t)(x - nh));
    } else {
        sc zz;
        sc_montmul(zz, vs.z, vs.z);
        sc_pow_mont(r, vs.z, (uint32_t)(e - 2 * (nh + nl)));
        sc_montmul(r, r, zz);
    }
    st_sc(V.tabs + b * (size_t)V.tab_stride + e, r);
}
// V2: g_i = -z - a s_i  (list 0, over G_i);  h_i = z + y^-i (z^2 z^j 2^i' - b s_(N-1-i))  (list 1, over H_i)  -> digits.
__device__ __forceinline__ void rv_gh_scalar(sc& r, const VerifyArgs& V, size_t b, int side, int q) {
    const RangeArgs& A = V.R;
    const VerifyState& vs = V.vs[b];
    const sc* T = V.tabs + b * (size_t)V.tab_stride;
    const int nh = 1 << V.hb, nl = 1 << V.lb;
    int i = side == 0 ? q : A.N - 1 - q;
    sc sh, sl, s, t;
    ld_sc(sh, T + (i >> V.lb));
    ld_sc(sl, T + nh + (i & (nl - 1)));
    sc_montmul(s, sh, sl);
    if (side == 0) {
        sc_montmul(t, vs.a, s);
        sc_add(t, t, vs.z);
        sc_neg(r, t);                                  // -z - a s_i
    } else {
        int j = q / A.n, ii = q - j * A.n;
        sc yh, yl, yi, zzj, two;
        ld_sc(yh, T + nh + nl + (q >> V.lb));
        ld_sc(yl, T + 2 * nh + nl + (q & (nl - 1)));
        sc_montmul(yi, yh, yl);
        ld_sc(zzj, T + 2 * (nh + nl) + j);
        sc_from_u64_mont(two, 1ull << ii);
        sc_montmul(t, zzj, two);                       // z^2 z^j 2^i'
        sc_montmul(s, vs.b, s);
        sc_sub(t, t, s);
        sc_montmul(t, yi, t);
        sc_add(r, vs.z, t);
    }
}
__global__ __launch_bounds__(64) void k_rv_scalars(VerifyArgs V) {
    const RangeArgs& A = V.R;
    int nch = A.TP >> 6;
    size_t b = blockIdx.x / nch;
    int ch = blockIdx.x % nch, l = threadIdx.x, side = l >> 5, q = 32 * ch + (l & 31), pos = 64 * ch + l;
    if (q >= A.N) { zero_digits(A, b, pos); return; }
    sc r;
    rv_gh_scalar(r, V, b, side, q);
    write_digits(A, b, pos, r);
}

// The t-th of a proof's own K = 4 + 2 lgN + m points and its scalar:  A, x S, c x T1, c x^2 T2, u_k^2 L_k, u_k^-2 R_k,
// c z^(2+j) V_j.
__device__ __forceinline__ void rv_own_point(uint32_t* w8, sc& sm, const VerifyArgs& V, size_t b, int t) {
    const RangeArgs& A = V.R;
    const VerifyState& vs = V.vs[b];
    const uint32_t* pr = A.out + b * A.out_words;
    if (t < 4) {
        ld8(w8, pr + 8 * t);
        if (t == 0) sc_one_mont(sm);
        else if (t == 1) sm = vs.x;
        else { sc_montmul(sm, vs.c, vs.x); if (t == 3) sc_montmul(sm, sm, vs.x); }
    } else if (t < 4 + A.lgN) {
        int k = t - 4;
        ld8(w8, pr + 56 + 16 * k);
        sc_montmul(sm, vs.u[k], vs.u[k]);
    } else if (t < 4 + 2 * A.lgN) {
        int k = t - 4 - A.lgN;
        ld8(w8, pr + 56 + 16 * k + 8);
        sc_montmul(sm, vs.u_inv[k], vs.u_inv[k]);
    } else {
        int j = t - 4 - 2 * A.lgN;
        ld8(w8, A.Vc + (b * A.m + j) * 8);
        sc zzj;
        ld_sc(zzj, V.tabs + b * (size_t)V.tab_stride + 2 * ((1 << V.hb) + (1 << V.lb)) + j);
        sc_montmul(sm, vs.c, zzj);
    }
}

// V4: the proof's own points (wave per proof): decompress, multiply (binary double-and-add), reduce.
__device__ __forceinline__ void ge_scalarmul_vartime(ge_p3& out, const ge_p3& p, const uint32_t* k8) {
    ge_cached c;
    ge_to_cached(c, p);
    ge_p3 acc;
    ge_identity(acc);
    for (int i = 252; i >= 0; i--) {
        ge_p3 t;
        ge_dbl(t, acc, true);
        acc = t;
        ge_add_cached(t, acc, c, false);
        if ((k8[i >> 5] >> (i & 31)) & 1) acc = t;
    }
    out = acc;
}
__global__ __launch_bounds__(64) void k_rv_varpoints(VerifyArgs V) {
    __shared__ int32_t lds[40 * 64];
    const RangeArgs& A = V.R;
    size_t b = blockIdx.x;
    int l = threadIdx.x;
    const int K = 4 + 2 * A.lgN + A.m;
    ge_p3 acc;
    ge_identity(acc);
    bool ok = true;
    for (int t = l; t < K; t += 64) {
        uint32_t w8[8], k8[8];
        sc sm;
        rv_own_point(w8, sm, V, b, t);
        ge_p3 p, q;
        ok &= ge_decompress(p, w8);
        sc_from_mont(k8, sm);
        ge_scalarmul_vartime(q, p, k8);
        ge_p3 r;
        ge_add(r, acc, q);
        acc = r;
    }
    wave_reduce_point(acc, lds, l, 64);
    if (l == 0) st_p3(A.PA + b * 40, acc);
    if (!ok) atomicAnd(&V.vs[b].ok, 0u);
}

// The scalars of B_blinding and B in one proof's check:  -mu - c tau   and   w (t_x - a b) + c (delta(y, z) - t_x).
__device__ __forceinline__ void rv_base_scalars(sc& bb, sc& bs, const VerifyState& vs, const RangeArgs& A) {
    sc one, zz, sumy, py, sumz, pz, sum2, delta, s1, s2;
    sc_one_mont(one);
    sc_montmul(zz, vs.z, vs.z);
    // sum_{i<N} y^i = (y^N - 1) / (y - 1)   (N a power of two: lgN squarings; y = 1 has probability 2^-252)
    py = vs.y;
    for (int i = 0; i < A.lgN; i++) sc_montmul(py, py, py);
    sc ym1, ym1_inv;
    sc_sub(ym1, vs.y, one);
    if (sc_is_zero(ym1)) { sc_from_u64_mont(sumy, (uint64_t)A.N); }
    else { sc_invert_mont(ym1_inv, ym1); sc_sub(py, py, one); sc_montmul(sumy, py, ym1_inv); }
    sc_zero(sumz); pz = one;
    for (int j = 0; j < A.m; j++) { sc_add(sumz, sumz, pz); sc_montmul(pz, pz, vs.z); }
    sc_zero(sum2);
    { sc p2s = one; for (int i = 0; i < A.n; i++) { sc_add(sum2, sum2, p2s); sc_add(p2s, p2s, p2s); } }
    sc_sub(s1, vs.z, zz); sc_montmul(delta, s1, sumy);
    sc_montmul(s2, zz, vs.z); sc_montmul(s2, s2, sum2); sc_montmul(s2, s2, sumz); sc_sub(delta, delta, s2);
    sc_montmul(s1, vs.c, vs.tau); sc_add(s1, s1, vs.mu); sc_neg(bb, s1);               // -mu - c tau
    sc_montmul(s1, vs.a, vs.b); sc_sub(s1, vs.t_x, s1); sc_montmul(bs, vs.w, s1);
    sc_sub(s2, delta, vs.t_x); sc_montmul(s2, vs.c, s2); sc_add(bs, bs, s2);            // w (t_x - ab) + c (delta - t_x)
}

// V5: assemble and test for the identity (lane per proof).
__global__ __launch_bounds__(64) void k_rv_finish(VerifyArgs V, TableView tbl) {
    const RangeArgs& A = V.R;
    size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B) return;
    const VerifyState& vs = V.vs[b];
    ge_p3 p0, p1, t;
    const int nsplit = A.nsplit > 1 ? A.nsplit : 1;
    ld_p3(p0, A.PA + b * 40);
    for (int sidx = 0; sidx < nsplit; sidx++) {            // partial sums of the (possibly split) fixed-base MSM
        ld_p3(p1, A.P0 + (b * nsplit + sidx) * 40);
        ge_add(t, p0, p1);
        ld_p3(p1, A.P1 + (b * nsplit + sidx) * 40);
        ge_add(p0, t, p1);
    }
    sc bb, bs;
    rv_base_scalars(bb, bs, vs, A);
    uint32_t k8[8], c8[8];
    sc_from_mont(k8, bb);
    tbl_fixed_mul_add(p0, tbl, tbl.row_Bb(0), k8);
    sc_from_mont(k8, bs);
    tbl_fixed_mul_add(p0, tbl, tbl.row_B(0), k8);
    ge_compress(c8, p0);
    V.verdict[b] = (vs.ok && words_zero(c8)) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Cross-proof batching (not in the reference, which checks every proof on its own): with
// random weights rho_p the sum over the batch of rho_p * (proof p's check) is one multiscalar test in which the shared
// bases G_i, H_i, B, B_blinding appear ONCE (their scalars summed over the proofs), so the 2nm-term fixed-base part is
// paid per batch instead of per proof.  What stays per proof is its own K points; they go through ONE Straus MSM over
// per-point 4-bit tables (the prover's tail kernel) instead of K separate double-and-add ladders.  A batch that passes
// means every proof passes (error 2^-250); a batch that fails is re-checked proof by proof by the path above, so the
// verdicts are always the per-proof ones.
enum { PT_WBITS = 4, PT_NWIN = 253 / PT_WBITS + 1, PT_ENTRIES = (1 << (PT_WBITS - 1)) + 1, PT_ROW_WORDS = PT_ENTRIES * 32 };
struct RlcArgs {
    VerifyArgs V;              // the batch: V.R.B proofs; V.R.dig / P0 / P1 / nsplit serve the ONE generator MSM
    int G;                     // proof groups of the generator-scalar accumulation
    sc* partial;               // [G][TP]  partial sums of rho_p * (g_i | h_i), digit-position order
    sc* bsum;                  // [B][2]   rho_p * (B_blinding scalar, B scalar)
    int K;                     // own points per proof
    size_t npts, Np;           // B * K points, as two lists of Np
    int TP2, ns2;              // digit-row length and wavefront splits of the point MSM
    int32_t* ptT;              // [2 Np][PT_ENTRIES][32]  per-point tables
    dig_t* dig2;               // [PT_NWIN][TP2]
    int32_t* Q0; int32_t* Q1;  // [ns2][40] partial sums of the point MSM
    uint32_t* flag;            // [0] = 1: the combined check is the identity; [1] = 1: some point failed to decode
};

// rho_p-weighted generator scalars, summed over the proofs p = g (mod G) of one group.  grid = (TP/64) * G blocks.
__global__ __launch_bounds__(64) void k_rvb_gh_partial(RlcArgs R) {
    const RangeArgs& A = R.V.R;
    int nch = A.TP >> 6;
    int g = blockIdx.x / nch, ch = blockIdx.x % nch, l = threadIdx.x, side = l >> 5, q = 32 * ch + (l & 31), pos = 64 * ch + l;
    sc acc;
    sc_zero(acc);
    if (q < A.N) {
        for (size_t p = g; p < A.B; p += R.G) {
            const VerifyState& vs = R.V.vs[p];
            if (!vs.ok) continue;
            sc r;
            rv_gh_scalar(r, R.V, p, side, q);
            sc_montmul(r, r, vs.rho);
            sc_add(acc, acc, r);
        }
    }
    st_sc(R.partial + (size_t)g * A.TP + pos, acc);
}
// ... summed over the groups -> the digits of the one generator MSM.  grid = TP/64 blocks.
__global__ __launch_bounds__(64) void k_rvb_gh_reduce(RlcArgs R) {
    const RangeArgs& A = R.V.R;
    int ch = blockIdx.x, l = threadIdx.x, q = 32 * ch + (l & 31), pos = 64 * ch + l;
    if (q >= A.N) { zero_digits(A, 0, pos); return; }
    sc acc, t;
    sc_zero(acc);
    for (int g = 0; g < R.G; g++) { ld_sc(t, R.partial + (size_t)g * A.TP + pos); sc_add(acc, acc, t); }
    write_digits(A, 0, pos, acc);
}
// rho_p-weighted scalars of B_blinding and B (lane per proof).
__global__ __launch_bounds__(64) void k_rvb_base_scalars(RlcArgs R) {
    const RangeArgs& A = R.V.R;
    size_t p = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= A.B) return;
    const VerifyState& vs = R.V.vs[p];
    sc bb, bs;
    sc_zero(bb); sc_zero(bs);
    if (vs.ok) {
        rv_base_scalars(bb, bs, vs, A);
        sc_montmul(bb, bb, vs.rho);
        sc_montmul(bs, bs, vs.rho);
    }
    st_sc(R.bsum + 2 * p, bb);
    st_sc(R.bsum + 2 * p + 1, bs);
}
// One lane per own point of the batch: decode, rho-weighted scalar -> digits, table row.  Slot t = list * Np + q.
__global__ __launch_bounds__(64) void k_rvb_points(RlcArgs R) {
    const RangeArgs& A = R.V.R;
    size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= 2 * R.Np) return;
    size_t list = t / R.Np, q = t - list * R.Np;
    dig_t* d = R.dig2 + 64 * (q >> 5) + (q & 31) + 32 * list;
    int32_t* row = R.ptT + t * (size_t)PT_ROW_WORDS;
    bool live = t < R.npts;
    size_t p = live ? t / R.K : 0;
    ge_p3 pt;
    ge_identity(pt);
    sc sm;
    sc_zero(sm);
    if (live && R.V.vs[p].ok) {
        uint32_t w8[8];
        rv_own_point(w8, sm, R.V, p, (int)(t - p * R.K));
        if (ge_decompress(pt, w8)) sc_montmul(sm, sm, R.V.vs[p].rho);
        else { atomicOr(&R.flag[1], 1u); ge_identity(pt); sc_zero(sm); }
    }
    uint32_t c[8];
    sc_from_mont(c, sm);
    const size_t TP2 = (size_t)R.TP2;
    sc_recode_w(PT_WBITS, PT_NWIN, c, [&](int i, int digit) { d[(size_t)i * TP2] = (dig_t)digit; });
    st_p3(row, pt);
    build_niels_row<PT_ENTRIES>(row);
}
// Sums everything and tests for the identity (one wavefront).
__global__ __launch_bounds__(64) void k_rvb_finish(RlcArgs R, TableView tbl) {
    __shared__ int32_t lds[40 * 64];
    const RangeArgs& A = R.V.R;
    int l = threadIdx.x;
    const int ns1 = A.nsplit > 1 ? A.nsplit : 1;
    ge_p3 acc, p, t;
    ge_identity(acc);
    for (int i = l; i < 2 * ns1 + 2 * R.ns2; i += 64) {
        const int32_t* src = i < ns1 ? A.P0 + (size_t)i * 40 : i < 2 * ns1 ? A.P1 + (size_t)(i - ns1) * 40
                           : i < 2 * ns1 + R.ns2 ? R.Q0 + (size_t)(i - 2 * ns1) * 40 : R.Q1 + (size_t)(i - 2 * ns1 - R.ns2) * 40;
        ld_p3(p, src);
        ge_add(t, acc, p);
        acc = t;
    }
    wave_reduce_point(acc, lds, l, 64);
    sc bb, bs, x;
    sc_zero(bb); sc_zero(bs);
    for (size_t i = l; i < A.B; i += 64) {
        ld_sc(x, R.bsum + 2 * i); sc_add(bb, bb, x);
        ld_sc(x, R.bsum + 2 * i + 1); sc_add(bs, bs, x);
    }
    uint32_t* lw = reinterpret_cast<uint32_t*>(lds);
    wave_reduce_sc(bb, lw, l);
    wave_reduce_sc(bs, lw, l);
    if (l == 0) {
        uint32_t k8[8], c8[8];
        sc_from_mont(k8, bb);
        tbl_fixed_mul_add(acc, tbl, tbl.row_Bb(0), k8);
        sc_from_mont(k8, bs);
        tbl_fixed_mul_add(acc, tbl, tbl.row_B(0), k8);
        ge_compress(c8, acc);
        R.flag[0] = words_zero(c8) ? 1u : 0u;
    }
}
// verdict[p] = the proof parsed (the batch check vouches for the rest)
__global__ void k_rvb_verdicts(VerifyArgs V) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < V.R.B) V.verdict[p] = V.vs[p].ok ? 1 : 0;
}

// MerkleProof::verify for single leaves with DapolProofNode::merge (src/proof/node.rs:56-69, src/proof/mod.rs:41-47):
// re-merge the leaf with its siblings (root side first in `pC/pH`) and compare with the root.  One lane per entity.
__global__ __launch_bounds__(64) void k_verify_paths(int dg, size_t b, int height, const uint64_t* leaf_idx, const uint32_t* leafC, const uint32_t* leafH,
                                                    const uint32_t* pC, const uint32_t* pH, const uint32_t* rootC, const uint32_t* rootH,
                                                    uint8_t* ok) {
    size_t e = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (e >= b) return;
    uint32_t c[8], h[8], sc_[8], sh[8], hn[8];
    ld8(c, leafC + e * 8);
    ld8(h, leafH + e * 8);
    ge_p3 acc, sp;
    bool good = ge_decompress(acc, c);
    uint64_t idx = leaf_idx[e];
    for (int k = 0; k < height; k++) {
        size_t slot = e * (size_t)height + (size_t)(height - 1 - k);
        ld8(sc_, pC + slot * 8);
        ld8(sh, pH + slot * 8);
        good &= ge_decompress(sp, sc_);                      // deserialisation rejects non-canonical points (proof/node.rs:88-94)
        if ((idx >> k) & 1) node_hash128(dg, hn, sc_, c, sh, h);
        else node_hash128(dg, hn, c, sc_, h, sh);
        ge_p3 t;
        ge_add(t, acc, sp);
        acc = t;
        ge_compress(c, acc);
        for (int i = 0; i < 8; i++) h[i] = hn[i];
    }
    for (int i = 0; i < 8; i++) good &= (c[i] == rootC[i]) & (h[i] == rootH[i]);
    ok[e] = good ? 1 : 0;
}
// commitments of one sub-proof from the path (pad parties: commit(0, 1) = B_blinding, src/range/padding.rs:176-180)
__global__ void k_gather_commitments(size_t b, int height, int start, int count, int m, const uint32_t* pC, const uint32_t* Bb_comp, uint32_t* Vc) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * (size_t)m) return;
    size_t e = t / m;
    int j = (int)(t - e * m);
    uint32_t c[8];
    if (j < count) ld8(c, pC + (e * (size_t)height + (size_t)(start + j)) * 8);
    else for (int i = 0; i < 8; i++) c[i] = Bb_comp[i];
    st8(Vc + t * 8, c);
}
__global__ void k_and_bytes(size_t n, uint8_t* acc, const uint8_t* x) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = acc[i] & x[i];
}

}  // namespace dapol
