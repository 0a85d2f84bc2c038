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
    sc u[RV_MAX_ROUNDS], u_inv[RV_MAX_ROUNDS];
    uint32_t ok, pad_[3];
};
struct VerifyArgs {
    RangeArgs R;                     // reuses n, m, N, lgN, TP, B, Vc, dig, P0, P1, PA(=P2), out(=proof words), out_words, seed
    VerifyState* vs;
    uint8_t* verdict;                // [B]
};

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
    vs.ok = ok ? 1u : 0u;
}

// V2: g_i = -z - a s_i  (list 0, over G_i);  h_i = z + y^-i (z^2 z^j 2^i' - b s_(N-1-i))  (list 1, over H_i)  -> digits.
__device__ __forceinline__ void rv_s(sc& r, const VerifyState& vs, int lgN, int i) {
    sc acc;
    sc_one_mont(acc);
    for (int k = 0; k < lgN; k++) {
        bool bit = (i >> (lgN - 1 - k)) & 1;           // MSB of i <-> first round
        sc_montmul(acc, acc, bit ? vs.u[k] : vs.u_inv[k]);
    }
    r = acc;
}
__global__ __launch_bounds__(64) void k_rv_scalars(VerifyArgs V) {
    const RangeArgs& A = V.R;
    int nch = A.TP >> 6;
    size_t b = blockIdx.x / nch;
    int ch = blockIdx.x % nch, l = threadIdx.x, side = l >> 5, q = 32 * ch + (l & 31), pos = 64 * ch + l;
    if (q >= A.N) { zero_digits(A, b, pos); return; }
    const VerifyState& vs = V.vs[b];
    sc s, t, r;
    if (side == 0) {
        rv_s(s, vs, A.lgN, q);
        sc_montmul(t, vs.a, s);
        sc_add(t, t, vs.z);
        sc_neg(r, t);                                  // -z - a s_i
    } else {
        int j = q / A.n, ii = q - j * A.n;
        rv_s(s, vs, A.lgN, A.N - 1 - q);
        sc zj, two, yi, zz;
        sc_montmul(zz, vs.z, vs.z);
        sc_pow_mont(zj, vs.z, (uint32_t)j);
        sc_from_u64_mont(two, 1ull << ii);
        sc_pow_mont(yi, vs.y_inv, (uint32_t)q);
        sc_montmul(t, zz, zj);
        sc_montmul(t, t, two);                         // z^2 z^j 2^i'
        sc_montmul(s, vs.b, s);
        sc_sub(t, t, s);
        sc_montmul(t, yi, t);
        sc_add(r, vs.z, t);
    }
    write_digits(A, b, pos, r);
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
    const VerifyState& vs = V.vs[b];
    const uint32_t* pr = A.out + b * A.out_words;
    const int K = 4 + 2 * A.lgN + A.m;
    ge_p3 acc;
    ge_identity(acc);
    bool ok = true;
    sc cx, cxx, czz;
    sc_montmul(cx, vs.c, vs.x);
    sc_montmul(cxx, cx, vs.x);
    sc_montmul(czz, vs.z, vs.z);
    sc_montmul(czz, czz, vs.c);
    for (int t = l; t < K; t += 64) {
        uint32_t w8[8], k8[8];
        sc sm;
        if (t < 4) {
            ld8(w8, pr + 8 * t);
            if (t == 0) sc_one_mont(sm); else if (t == 1) sm = vs.x; else if (t == 2) sm = cx; else sm = cxx;
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
            sc zj;
            sc_pow_mont(zj, vs.z, (uint32_t)j);
            sc_montmul(sm, czz, zj);
        }
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
    sc one, zz, sumy, py, sumz, pz, sum2, delta, s1, s2, bb, bs;
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
    uint32_t k8[8], c8[8];
    sc_from_mont(k8, bb);
    tbl_fixed_mul_add(p0, tbl, tbl.row_Bb(0), k8);
    sc_from_mont(k8, bs);
    tbl_fixed_mul_add(p0, tbl, tbl.row_B(0), k8);
    ge_compress(c8, p0);
    V.verdict[b] = (vs.ok && words_zero(c8)) ? 1 : 0;
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
