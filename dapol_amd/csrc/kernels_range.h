// Aggregated Bulletproofs range-proof prover, batched over independent proofs (one wavefront per proof in the
// heavy kernels, one lane per proof in the Fiat-Shamir kernels).  Restates bulletproofs 4.0.0
// RangeProof::prove_multiple_with_rng + InnerProductProof::create as called from src/range/mod.rs:48-78, with
// the algebra re-arranged for the GPU (same group elements, same bytes):
//   * a single prover plays every party, so blindings are pre-summed (A = sum A_j, ...);
//   * the generators are NEVER folded: round k's L_k / R_k are multiscalar products over the ORIGINAL shared
//     generators G_j, H_j with per-proof scalars a_i * s_j (s_j = the running product of u^{+-1}, the same
//     s-vector the verifier uses).  All point work is therefore fixed-base, served by the window tables
//     (tables.h), and the only per-proof state between rounds is scalar vectors in HBM;
//   * Q = w*B is never materialised: c_L*Q = (c_L*w)*B is one more fixed-base term.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "kernels_ctx_tree.h"
#include "ge_quad.h"

namespace dapol {

struct ProofState {             // per proof, lives in HBM between phase kernels
    uint64_t strobe[25];
    uint32_t pos, pos_begin;
    uint32_t err, pad_;
    sc y, z, y_inv, x, w, u, u_inv;     // Montgomery form
    sc a_bl, s_bl, t1, t2, t1_bl, t2_bl, t_x, cL, cR;
    uint32_t nkey[8];           // seed mode: this proof's nonce key (k_rp_nonce_key), bound to its statement
};

struct RangeArgs {
    int n, m, N, lgN, TP;       // bits per party, parties, n*m, log2 N, digit-row length (>= 64)
    int wbits, nwin;            // window width of the context's tables; windows per CANONICAL scalar (tv.nwin_c())
    size_t B;                   // proofs in this chunk
    // inputs
    const uint64_t* vals;       // [B][m]
    const uint32_t* blind;      // [B][m][8]   integers < 2^255 (reduced on load)
    const uint32_t* Vc;         // [B][m][8]   compressed value commitments
    // nonce source
    const uint32_t* seed;       // [8] device
    // Which proof is chunk-local proof b?  A call proves `sub_k` equal-sized sub-proofs per ROW (an entity of a policy's plan: its
    // individual proofs, or the equal parts of a split -- src/range/padding.rs:104-112, splitting.rs:110-123; 1 for a lone proof per
    // row): proof g = p0 + b of the call is sub-proof j = g % sub_k of row e = g / sub_k (proof_row below).  The row owns the RNG
    // stream and the tape row; the sub-proof's draws start sub_slots = m(2n+4) after its predecessor's.
    const uint64_t* stream_id;  // [rows] of the CALL (not of the chunk)
    uint64_t slot_base;         // first slot of sub-proof 0
    const uint32_t* tape;       // [rows][tape_stride][16] or null; sub-proof 0's slots start at the pointer
    uint32_t tape_stride;       // draws per row in `tape`: m(2n+4) for a lone proof, the whole entity's slots when the sub-proofs of a policy share one stream
    uint32_t sub_k, sub_slots;
    size_t p0;                  // call-wide index of this chunk's first proof
    // scratch
    sc* a; sc* b; sc* s1; sc* s2;       // [B][N] each
    // Coefficient TABLES (null: the s-vectors themselves are folded every round, as before round 3).  The coefficient of
    // generator j after k rounds depends only on the top k bits of j: s_G[j] = TG_k[j >> (lgN - k)], s_H[j] = y^-j * TH_k[...]
    // (the products of u_t^{+-1} the verifier calls s).  So the two N-entry vectors need not be rewritten every round: a proof
    // keeps 2^k-entry tables (k <= STAB_ROUNDS), extended by k_rp_fold, and s2 stays the constant y^-j.  Layout:
    // stab[b][buffer k & 1][side][STAB_N]; TG entries in plain form, TH entries in Montgomery form.
    sc* stab;
    dig_t* dig;                         // [B][nwin][TP] signed radix-2^wbits digits
    int nsplit;                         // MSM term-range splits per proof (0/1 = none); partial points at [b * nsplit + s]
    int use_hi;                         // small calls: the main MSM also walks only TableView::hi_split window steps (two lookups per term)
    int fs_parts;                       // small calls: wavefronts per proof in k_rp_poly / k_rp_lr (0 / 1 = one); partial sums in fs_part
    sc* fs_part;                        // [B][fs_parts][3]: t1, t2 (k_rp_poly), t_x (k_rp_lr) of each wavefront's share of the positions
    ProofState* st;                     // [B]
    int32_t* PA; int32_t* P0; int32_t* P1;   // [B][40] partial points
    int mat_round;                      // rounds folded before the materialisation (= never-fold rounds of the main argument)
    int tail_n;                         // T = length of the tail argument (32 / 64 / 128): 2T generators are materialised
    int32_t* tailT;                     // [B][2T][TAIL_ENTRIES][32]  per-proof window tables of the materialised folded generators
    sc* tail_a; sc* tail_b; sc* tail_s1; sc* tail_s2;   // [B][T] each: the vectors / coefficients of the tail argument
    uint32_t* out;                      // of the CALL: row e's sub-proof j at out + e * out_stride + j * out_words
    size_t out_stride;                  // words between rows (= out_words for a lone proof per row)
    int out_words;
    int out_round0;                     // rounds already written before this argument's round 0 (tail: lgN - 5)
};

// The tail of the hybrid inner-product argument is itself a never-fold argument of length T
// over the 2T materialised generators, served by a per-proof table of signed 5-bit windows in the same 128-byte
// affine-niels format as the context tables (rows G'_0..T-1, H'_0..T-1).
#ifndef DAPOL_TAIL_WBITS
#define DAPOL_TAIL_WBITS 5        // measured 4 vs 5 bits at 2^20 proofs: 39.5K vs 40.5K entities/s (profiles/r01_tail_wbits_ab.txt)
#endif
enum { TAIL_WBITS = DAPOL_TAIL_WBITS, TAIL_NWIN = 253 / TAIL_WBITS + 1, TAIL_ENTRIES = (1 << (TAIL_WBITS - 1)) + 1, TAIL_ROW_WORDS = TAIL_ENTRIES * 32 };
enum { MSM_PLAIN = 0, MSM_MATERIALIZE = 1, MSM_TAIL = 2 };
enum { STAB_ROUNDS = 6, STAB_N = 1 << STAB_ROUNDS };      // coefficient tables serve arguments that need at most TG_6 (64 entries)

__device__ __forceinline__ void proof_row(const RangeArgs& A, size_t b, size_t& e, uint32_t& j) {
    const size_t g = A.p0 + b;
    if (A.sub_k <= 1) { e = g; j = 0; }
    else { e = g / A.sub_k; j = (uint32_t)(g - e * A.sub_k); }
}
__device__ __forceinline__ uint32_t* proof_out(const RangeArgs& A, size_t b) {
    size_t e; uint32_t j;
    proof_row(A, b, e, j);
    return A.out + e * A.out_stride + (size_t)j * (size_t)A.out_words;
}
__device__ __forceinline__ void tape_wide(uint32_t* w16, const RangeArgs& A, size_t b, uint32_t slot) {
    size_t e; uint32_t j;
    proof_row(A, b, e, j);
    const uint32_t first = j * A.sub_slots;
    if (A.tape) {
        const uint4* p = reinterpret_cast<const uint4*>(A.tape + (e * (size_t)A.tape_stride + first + slot) * 16);
        for (int i = 0; i < 4; i++) { uint4 q = p[i]; w16[4 * i] = q.x; w16[4 * i + 1] = q.y; w16[4 * i + 2] = q.z; w16[4 * i + 3] = q.w; }
    } else {
        uint32_t key[8];
        for (int i = 0; i < 8; i++) key[i] = A.st[b].nkey[i];
        seed_wide(w16, key, 2u, A.stream_id[e], A.slot_base + first + slot);
    }
}
// Seed mode: the key of one proof's nonce stream.  The reference's prover draws fresh thread_rng randomness on every call;
// a deterministic stream must therefore be bound to the STATEMENT it blinds, or re-proving a leaf whose siblings changed
// (dapol_tree_update, another policy / aggregation factor) would reuse a_blinding, s_L, s_R ... against new challenges and
// leak the siblings' secrets.  key = chain over the parties' value commitments (31 per BLAKE3 chunk), started from
// seed -> (domain 6: stream id, first slot) -> (domain 7: bits, parties).  Same statement => same nonces => same proof.
__global__ __launch_bounds__(64) void k_rp_nonce_key(RangeArgs A) {
    size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B || A.tape) return;
    uint32_t key[8], w[16];
    for (int i = 0; i < 8; i++) key[i] = A.seed[i];
    size_t e; uint32_t sj;
    proof_row(A, b, e, sj);
    seed_wide(w, key, 6u, A.stream_id[e], A.slot_base + (uint64_t)sj * A.sub_slots);
    for (int i = 0; i < 8; i++) key[i] = w[i];
    seed_wide(w, key, 7u, (uint64_t)A.n, (uint64_t)A.m);
    for (int i = 0; i < 8; i++) key[i] = w[i];
    // One BLAKE3 chunk per 31 commitments, absorbed a block (= two 32-byte entries: the key, then the V_j) at a time.
    for (int j0 = 0; j0 < A.m; j0 += 31) {
        const int n_ent = 1 + ((A.m - j0 < 31) ? A.m - j0 : 31);
        const int n_blk = (n_ent + 1) / 2;
        uint32_t cv[8], blk[16], o[16];
        blake3_iv(cv);
        for (int i = 0; i < n_blk; i++) {
            for (int h = 0; h < 2; h++) {
                const int e = 2 * i + h;
                if (e == 0) { for (int k = 0; k < 8; k++) blk[k] = key[k]; }
                else if (e < n_ent) ld8(blk + 8 * h, A.Vc + (b * A.m + j0 + e - 1) * 8);
                else { for (int k = 0; k < 8; k++) blk[8 + k] = 0; }
            }
            const bool last = i == n_blk - 1;
            blake3_compress(o, cv, blk, 0, (last && (n_ent & 1)) ? 32u : 64u, (i == 0 ? B3_CHUNK_START : 0u) | (last ? (B3_CHUNK_END | B3_ROOT) : 0u));
            for (int k = 0; k < 8; k++) cv[k] = o[k];
        }
        for (int k = 0; k < 8; k++) key[k] = cv[k];
    }
    for (int i = 0; i < 8; i++) A.st[b].nkey[i] = key[i];
}
__device__ __forceinline__ void tape_scalar(sc& r, const RangeArgs& A, size_t b, uint32_t slot) {
    uint32_t w[16];
    tape_wide(w, A, b, slot);
    sc_from_wide(r, w);
}
__device__ __forceinline__ void ld_sc(sc& r, const sc* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
}
__device__ __forceinline__ void st_sc(sc* p, const sc& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
// digits of a canonical scalar (eight words, < l) for list position `pos` of proof b
__device__ __forceinline__ void write_digits_plain(const RangeArgs& A, size_t b, int pos, const uint32_t* c) {
    dig_t* d = A.dig + (size_t)b * A.nwin * A.TP + pos;
    const int TP = A.TP;
    sc_recode_w(A.wbits, A.nwin, c, [&](int i, int digit) { d[(size_t)i * TP] = (dig_t)digit; });
}
// digits of a Montgomery-form scalar
__device__ __forceinline__ void write_digits(const RangeArgs& A, size_t b, int pos, const sc& s_mont) {
    uint32_t c[8];
    sc_from_mont(c, s_mont);
    write_digits_plain(A, b, pos, c);
}
__device__ __forceinline__ void zero_digits(const RangeArgs& A, size_t b, int pos) {
    dig_t* d = A.dig + (size_t)b * A.nwin * A.TP + pos;
    for (int i = 0; i < A.nwin; i++) d[(size_t)i * A.TP] = 0;
}

// Coefficient of generator j (G side: isH = false) of proof b in round `round`, times the Montgomery-form vector entry v:
// the canonical product the MSM's digits are made of.  Table mode or vector mode (RangeArgs::stab).
__device__ __forceinline__ void coeff_times(sc& p, const RangeArgs& A, size_t b, int round, int j, bool isH, const sc& v) {
    if (A.stab) {
        const sc* T = A.stab + ((b * 2 + (size_t)(round & 1)) * 2 + (isH ? 1 : 0)) * STAB_N + (round ? (j >> (A.lgN - round)) : 0);
        sc t;
        ld_sc(t, T);
        if (isH) {
            sc yi, q;
            ld_sc(yi, A.s2 + b * A.N + j);            // y^-j, plain (k_rp_lr); ones in the tail argument
            sc_montmul(q, v, yi);                      // Montgomery x plain = plain v y^-j
            sc_montmul(p, q, t);                       // plain x Montgomery = plain
        } else sc_montmul(p, v, t);                    // Montgomery x plain
    } else {
        sc sv;
        ld_sc(sv, (isH ? A.s2 : A.s1) + b * A.N + j);  // plain form (k_rp_lr)
        sc_montmul(p, v, sv);                          // Montgomery x plain = the canonical product
    }
}
// The coefficient itself in plain form (the materialisation's scalars), after `round` rounds.
__device__ __forceinline__ void coeff_plain(sc& s, const RangeArgs& A, size_t b, int round, int j, bool isH) {
    if (A.stab) {
        const sc* T = A.stab + ((b * 2 + (size_t)(round & 1)) * 2 + (isH ? 1 : 0)) * STAB_N + (round ? (j >> (A.lgN - round)) : 0);
        sc t;
        ld_sc(t, T);
        if (isH) {
            sc yi;
            ld_sc(yi, A.s2 + b * A.N + j);
            sc_montmul(s, yi, t);                      // plain x Montgomery = plain
        } else s = t;
    } else ld_sc(s, (isH ? A.s2 : A.s1) + b * A.N + j);
}
__device__ __forceinline__ void stab_init(const RangeArgs& A, size_t b) {      // TG_0 = [1] (plain), TH_0 = [1] (Montgomery)
    if (!A.stab) return;
    sc one_p, one_m;
    sc_zero(one_p); one_p.v[0] = 1;
    sc_one_mont(one_m);
    st_sc(A.stab + (b * 2 * 2 + 0) * STAB_N, one_p);
    st_sc(A.stab + (b * 2 * 2 + 1) * STAB_N, one_m);
}

// List position -> generator.  A digit row holds two lists of N terms each: lanes 0-31 of the MSM wave walk
// list 0 (-> partial point P0), lanes 32-63 list 1 (-> P1); term q of a list sits at position 64*(q/32)+(q%32)
// (+32 for list 1).  round < 0 (the S commitment): list 0 = <s_L, G>, list 1 = <s_R, H>.  round k >= 0:
// list 0 = L_k = <a_L, G_R> + <b_R, H'_L>, list 1 = R_k = <a_R, G_L> + <b_L, H'_R>; the first N/2 terms of a
// list are its G terms.  Returns the generator's position j in [0, N) and whether it is an H generator.
__device__ __forceinline__ int term_generator(int round, int N, int lgN, int side, int q, bool& isH) {
    if (round < 0) {
        isH = side != 0;
        return q;
    }
    int lgh = lgN - 1 - round;            // log2(half)
    int half = 1 << lgh, Nh = N >> 1;
    isH = q >= Nh;
    int qq = isH ? q - Nh : q;
    bool upper = isH ? (side != 0) : (side == 0);   // L takes G_R (upper half) and H_L (lower half)
    // block b = qq / half, offset o = qq % half  ->  2 b half + o = qq + b half = qq + (qq & ~(half - 1))
    return qq + (qq & ~(half - 1)) + (upper ? half : 0);
}
__device__ __forceinline__ int gen_row(const TableView& t, int n, int j, bool isH) {
    int lgn = 31 - __clz(n);                  // n is 8, 16, 32 or 64
    int party = j >> lgn, bit = j & (n - 1);
    return isH ? t.row_H(party, bit) : t.row_G(party, bit);
}

// ----------------------------------------------------------------------------------------- wave reductions
// Sum of one point per lane over groups of `width` adjacent lanes (a power of two <= 64), by wavefront shuffles: every step
// brings the partner's 36 limbs over with __shfl_down (DPP / ds_bpermute, no LDS round trip, no barrier) and adds.  The lane
// whose index is a multiple of `width` ends with the group's sum; the other lanes hold partial sums nobody reads.
__device__ __forceinline__ void wave_reduce_point(ge_p3& acc, int width) {
    for (int off = width >> 1; off >= 1; off >>= 1) {
        ge_p3 o, r;
        for (int i = 0; i < FE_NL; i++) {
            o.X.v[i] = __shfl_down(acc.X.v[i], off, 64);
            o.Y.v[i] = __shfl_down(acc.Y.v[i], off, 64);
            o.Z.v[i] = __shfl_down(acc.Z.v[i], off, 64);
            o.T.v[i] = __shfl_down(acc.T.v[i], off, 64);
        }
        ge_add(r, acc, o);
        acc = r;
    }
}
// Sum of one scalar per lane over the wavefront, by wavefront shuffles (DPP / ds_bpermute; no LDS round trip, no barrier):
// the inner products <l, r>, c_L = <a_L, b_R>, c_R = <a_R, b_L> of the inner-product argument.  Lane 0 ends with the total.
__device__ __forceinline__ void wave_reduce_sc(sc& acc) {
    for (int off = 32; off >= 1; off >>= 1) {
        sc o, r;
        for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)__shfl_down((int)acc.v[i], off, 64);
        sc_add(r, acc, o);
        acc = r;
    }
}

// s * Base by the 32 lanes of a half-wavefront (s the same on all of them): the base points have one table row per window, so
// lane `sub` looks its window's digit up and a shuffle tree adds the (at most 32) entries -- one lookup and five additions deep
// instead of a chain of nwin.  The sum is valid on the lane with sub == 0.
__device__ __forceinline__ void tbl_fixed_mul_wave(ge_p3& out, const TableView& t, int row0, const uint32_t* s8, int sub) {
    int carry = 0, mine = 0;
    const int W = t.wbits, NW = t.nwin();
    for (int i = 0; i < NW; i++) {
        int o = i * W, wd = o >> 5, sh = o & 31;
        uint32_t lo = wd < 8 ? s8[wd] >> sh : 0u;
        uint32_t hi = (sh && wd + 1 < 8) ? (s8[wd + 1] << (32 - sh)) : 0u;
        int b = (int)((lo | hi) & ((1u << W) - 1)) + carry;
        carry = (b >= (1 << (W - 1)) && i < NW - 1) ? 1 : 0;
        if (i == sub) mine = b - (carry << W);
    }
    ge_identity(out);
    if (sub < NW) tbl_madd(out, t, row0 + sub, mine);
    wave_reduce_point(out, 32);
}

// ---------------------------------------------------------------------------- K0: nonces s_L, s_R + S digits
// grid = B * (TP/64) blocks of 64.  Lane (side, q): draws s_L[q] (side 0) or s_R[q] (side 1) from the tape in
// the crate's slot order, stores it (Montgomery) and writes its signed radix-256 digits for the S MSM.
__global__ __launch_bounds__(64) void k_rp_nonces(RangeArgs A) {
    int nch = A.TP >> 6;
    size_t b = blockIdx.x / nch;
    int ch = blockIdx.x % nch, l = threadIdx.x, side = l >> 5, q = 32 * ch + (l & 31), pos = 64 * ch + l;
    if (q >= A.N) { zero_digits(A, b, pos); return; }
    int j = q / A.n, ii = q - j * A.n;
    uint32_t slot = (uint32_t)(j * (2 * A.n + 2) + 2 + ii + (side ? A.n : 0));
    sc s;
    tape_scalar(s, A, b, slot);
    st_sc((side ? A.s2 : A.s1) + b * A.N + q, s);
    write_digits(A, b, pos, s);
}

// ------------------------------------------------------------------------- K1: A = sum_i (bit ? G_i : -H_i)
__global__ __launch_bounds__(64) void k_rp_A(RangeArgs A, TableView tbl) {
    size_t b = blockIdx.x;
    int l = threadIdx.x;
    ge_p3 acc;
    ge_identity(acc);
    for (int i = l; i < A.N; i += 64) {
        int j = i / A.n, ii = i - j * A.n;
        int bit = (int)((A.vals[b * A.m + j] >> ii) & 1ull);
        tbl_madd(acc, tbl, bit ? tbl.row_G(j, ii) : tbl.row_H(j, ii), bit ? 1 : -1);
    }
    wave_reduce_point(acc, 64);
    if (l == 0) st_p3(A.PA + b * 40, acc);
}

// The same sum on ONE lane per proof (round 6: large batches of short proofs).  A wavefront per proof spends 6 shuffle-additions on
// all 64 lanes to add up N / 64 terms per lane -- 448 lane-additions for the 64 terms of an individual proof; a lane walking its
// proof's N terms does N.  All lanes of a wavefront read the same two table entries per step (entry 1 of G_i's row or of H_i's).
__global__ __launch_bounds__(64) void k_rp_A_lane(RangeArgs A, TableView tbl) {
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B) return;
    ge_p3 acc;
    ge_identity(acc);
#pragma nounroll
    for (int j = 0; j < A.m; j++) {
        const uint64_t v = A.vals[b * A.m + j];
#pragma nounroll
        for (int ii = 0; ii < A.n; ii++) {
            const int bit = (int)((v >> ii) & 1ull);
            tbl_madd(acc, tbl, bit ? tbl.row_G(j, ii) : tbl.row_H(j, ii), bit ? 1 : -1);
        }
    }
    st_p3(A.PA + b * 40, acc);
}

// ------------------------------------------------------------------- K2: the fixed-base MSM, proof-stationary form
// A list (L or R of one proof) is owned by LPL lanes; a lane walks the nwin signed W-bit windows from the top with W shared
// doublings per window (Straus) and, per window, its terms of the list: digit -> one 128-byte entry of the generator's table row
// (a random HBM line: the rows of a 17-bit table are 8.4 MB each, 35 GB in all) -> one mixed addition.  The lanes' partial sums
// meet in a shuffle reduction (wave_reduce_point: no LDS, no barrier).  This is the form of calls of fewer than 8,192 proofs -- mid-size
// and small calls --, of the tail argument over a proof's own tables and of the verifier's generator MSM; calls of at least 8,192 proofs run
// the plain rounds and the materialisation generator-stationary instead (kernels_range_gs.h), which reads the same table rows
// out of the Infinity Cache.
#ifndef DAPOL_MSM_OCC
#define DAPOL_MSM_OCC 3          // resident wavefronts per SIMD the register allocation is bounded for
#endif
template <int MODE, int LPL>
__global__ __launch_bounds__(64, DAPOL_MSM_OCC) void k_rp_msm(RangeArgs A, TableView tbl, int round) {
    // LPL = lanes per list.  32: one proof per wavefront (64 terms per lane at N = 2048).  16 / 8: two / four proofs per
    // wavefront with 128 / 256 terms per lane, which amortises the W * nwin shared doublings (22 % of the instructions
    // at LPL = 32) over more mixed adds.  The digit layout is the same for every LPL.
    // MODE: MSM_PLAIN -> P0 / P1;  MSM_MATERIALIZE -> the 64 per-lane sums are kept (folded generators; T / 32 blocks
    // per proof, block c taking every (T/32)-th term);  MSM_TAIL -> the table is the PROOF's own (2N rows).
    constexpr int PPW = 32 / LPL;
    int l = threadIdx.x;
    int sub = l / (2 * LPL), ll = l % (2 * LPL), side = ll / LPL, ql = ll % LPL;
    // Large proofs (1024-party verification) are split over nsplit wavefronts per proof, each taking a slice of the
    // term range; the partial points are summed by the consumer.
    const int nsplit = A.nsplit > 1 ? A.nsplit : 1;
    const int split = (int)(blockIdx.x % nsplit);
    size_t b = (size_t)(blockIdx.x / nsplit) * PPW + sub;
    int niter_all = (A.N + LPL - 1) / LPL;
    int i_begin = (int)((long long)niter_all * split / nsplit), niter = (int)((long long)niter_all * (split + 1) / nsplit);
    int i_step = 1;
    int mat_split = 0;
    if constexpr (MODE == MSM_MATERIALIZE) {
        // Folded generator i of the length-T argument collects the original generators j = i (mod T).  Lane ql owns
        // the terms 32 * i + ql, so the class of term i is ql + 32 * (i mod K), K = T / 32: block c walks i = c (mod K).
        // Small calls split a block's terms over nsplit wavefronts (k_rp_sum_mat adds the partial sums).
        i_step = A.tail_n >> 5;
        mat_split = (int)(blockIdx.x % nsplit);
        const size_t blk = blockIdx.x / nsplit;
        b = blk / i_step;
        i_begin = (int)(blk % i_step);
        niter = niter_all;
    }
    bool valid = b < A.B;
    if (!valid) b = A.B - 1;
    if constexpr (MODE == MSM_TAIL) tbl.base += b * (size_t)(2 * A.N) * tbl.row_words();
    const dig_t* dig = A.dig + b * A.nwin * (size_t)A.TP + 32 * side;
    const int NW = A.nwin, W = A.wbits;
    // trips of this lane (PLAIN / TAIL): runs of four consecutive terms when the list divides that way, else one term per trip
    // Rows of the high halves (TableView::hi_split): only the low hi_split window steps are walked; a step looks a term up twice,
    // digit w in the generator's row and digit w + hi_split in its 2^(W hi_split) row.  Always for the materialisation (whose
    // lanes own few terms), and for the main MSM of small calls (A.use_hi), where the shared doublings are the latency.
    const int LW = (MODE != MSM_TAIL && tbl.hi_split && (MODE == MSM_MATERIALIZE || A.use_hi)) ? tbl.hi_split : NW;
    const int halves = LW < NW ? 2 : 1;
    const bool grp4 = halves == 1 && (A.N % (4 * LPL)) == 0;
    const int trips_all = grp4 ? A.N / LPL : niter_all, unit = grp4 ? 4 : 1;
    const int it_begin = (int)((long long)(trips_all / unit) * split / nsplit) * unit, it_end = (int)((long long)(trips_all / unit) * (split + 1) / nsplit) * unit;
    ge_p3 acc;
    ge_identity(acc);
    for (int w = LW - 1; w >= 0; w--) {
        if (w != LW - 1) {
            // ONE inlined copy of the doubling (T under a runtime flag).  Measured (profiles/r01_msm_variants.txt): a
            // second, T-less copy made the launch 17 % SLOWER (127 vs 109 ms) -- the loop body is ~25 KB of code and
            // the extra copy costs more in instruction-cache misses than the skipped multiplication saves.
            // In place: ge_dbl reads every input before it writes any output.
            for (int d = 0; d < W; d++) ge_dbl(acc, acc, d == W - 1);
        }
        const dig_t* dw = dig + (size_t)w * A.TP;
        if constexpr (MODE == MSM_MATERIALIZE) {
            // ONE copy of the addition: the two lookups of a term are two trips
            const int own = (niter - i_begin + i_step - 1) / i_step;        // this block's terms: i = i_begin + k * i_step
            const int k_begin = (int)((long long)own * mat_split / nsplit), k_end = (int)((long long)own * (mat_split + 1) / nsplit);
#pragma nounroll
            for (int it = 2 * k_begin; it < 2 * k_end; it += (halves == 2 ? 1 : 2)) {
                const int i = i_begin + (it >> 1) * i_step, hi = it & 1;
                if (hi && w + LW >= NW) continue;
                int q = LPL * i + ql;                       // (N is a multiple of 32 here: every q is a term)
                int d = (hi ? dw + (size_t)LW * A.TP : dw)[64 * (q >> 5) + (q & 31)];
                bool isH;
                int j = term_generator(round, A.N, A.lgN, side, q, isH);
                int row = gen_row(tbl, A.n, j, isH);
                tbl_madd(acc, tbl, hi ? tbl.row_hi(row) : row, d);
            }
        } else {
            // Which terms a lane sums does not change the result, so a lane takes its terms in runs of FOUR consecutive
            // positions where the list allows it (grp4): one 16-byte load brings four digits, and the rows of the digit matrix
            // are read in whole 16-byte pieces per lane instead of 4 bytes per trip (8x less digit traffic at two lanes per
            // list, where a 64-byte sector used to be fetched again for every pair of digits).  ONE copy of the addition: the
            // trips are not unrolled (instruction cache, see above); the four digits rotate through d4.
            dapol_v4i d4 = {0, 0, 0, 0};
#pragma nounroll
            for (int it2 = 2 * it_begin; it2 < 2 * it_end; it2 += (halves == 2 ? 1 : 2)) {
                const int it = it2 >> 1, hi = it2 & 1;
                if (hi && w + LW >= NW) continue;
                int q;
                if (grp4) {
                    q = 4 * (LPL * (it >> 2) + ql) + (it & 3);
                    if ((it & 3) == 0) d4 = *reinterpret_cast<const dapol_v4i*>(dw + 64 * (q >> 5) + (q & 31));
                } else {
                    q = LPL * it + ql;
                    if (q >= A.N) continue;
                    d4.x = (hi ? dw + (size_t)LW * A.TP : dw)[64 * (q >> 5) + (q & 31)];
                }
                const int d = d4.x;
                d4.x = d4.y; d4.y = d4.z; d4.z = d4.w;
                bool isH;
                int j = term_generator(round, A.N, A.lgN, side, q, isH);
                int row;
                if constexpr (MODE == MSM_TAIL) row = j + (isH ? A.N : 0);
                else row = gen_row(tbl, A.n, j, isH);
                tbl_madd(acc, tbl, hi ? tbl.row_hi(row) : row, d);
            }
        }
    }
    if constexpr (MODE == MSM_MATERIALIZE) {
        // Hybrid IPA: fed with the s-vector digits the lane's accumulator IS the folded generator G'_i (lanes 0-31) /
        // H'_i (lanes 32-63), i = ql + 32 c, of the round whose vectors have length T.  Keep it at the head of its table
        // row (k_rp_tail_table builds the row).
        static_assert(MODE != MSM_MATERIALIZE || LPL == 32, "materialisation needs one proof per wavefront");
        const int T = A.tail_n;
        const size_t g = b * (size_t)(2 * T) + (size_t)(side * T + ql + 32 * i_begin);
        if (nsplit > 1) st_p3(A.P0 + (g * nsplit + mat_split) * 40, acc);
        else st_p3(A.tailT + g * TAIL_ROW_WORDS, acc);
    } else {
        wave_reduce_point(acc, LPL);
        if (ql == 0 && valid) st_p3((side ? A.P1 : A.P0) + (b * nsplit + split) * 40, acc);
    }
}

// The main MSM of a call of a FEW proofs, one point per four lanes (ge_quad.h): a wavefront holds 16 accumulators -- 8 per list --
// each walking the window steps for its terms (one term per group when the proof is split over N / 8 wavefronts): the W shared
// doublings per window step and the two lookups per term (high-half rows) are two field products deep each instead of seven or
// eight.  PLAIN mode only; same digit matrix, same term -> generator map, partial points in the same records as k_rp_msm's
// (summed by k_rp_sum_splits).  A lone 64-bit, 32-party MSM: 0.25 -> see profiles/README.md.
__global__ __launch_bounds__(64) void k_rp_msm_quad(RangeArgs A, TableView tbl, int round) {
    const int l = threadIdx.x, ql = l & 3, grp = l >> 2, side = grp >> 3, sub = grp & 7;
    const int nsplit = A.nsplit > 1 ? A.nsplit : 1;
    const int split = (int)(blockIdx.x % nsplit);
    const size_t b = blockIdx.x / nsplit;                            // (the grid is exactly B * nsplit blocks)
    const dig_t* dig = A.dig + b * A.nwin * (size_t)A.TP + 32 * side;
    const int NW = A.nwin, W = A.wbits;
    const int LW = (tbl.hi_split && A.use_hi) ? tbl.hi_split : NW, halves = LW < NW ? 2 : 1;
    const int per = (A.N + 8 * nsplit - 1) / (8 * nsplit);           // terms per group
    fe c;
    quad_identity(c, ql);
    for (int w = LW - 1; w >= 0; w--) {
        if (w != LW - 1) {
#pragma nounroll
            for (int d = 0; d < W; d++) quad_dbl(c, ql);
        }
        const dig_t* dw = dig + (size_t)w * A.TP;
#pragma nounroll
        for (int it2 = 0; it2 < 2 * per; it2 += (halves == 2 ? 1 : 2)) {
            const int it = it2 >> 1, hi = it2 & 1;
            if (hi && w + LW >= NW) continue;
            const int q = 8 * (split + nsplit * it) + sub;
            int d = 0, row = 0;                                      // (no term left for this group: entry 0 of a row is the identity)
            if (q < A.N) {
                d = (hi ? dw + (size_t)LW * A.TP : dw)[64 * (q >> 5) + (q & 31)];
                bool isH;
                const int j = term_generator(round, A.N, A.lgN, side, q, isH);
                row = gen_row(tbl, A.n, j, isH);
                if (hi) row = tbl.row_hi(row);
            }
            const bool neg = d < 0;
            const int ad = neg ? -d : d;
            // this lane's element of the entry: lane 0 multiplies by y-x (y+x for a negative digit), lane 1 by y+x (y-x), lane 3 by 2dxy
            const int el = ql == 0 ? (neg ? 0 : 1) : ql == 1 ? (neg ? 1 : 0) : 2;
            const int32_t* e = tbl.base + ((uint64_t)(uint32_t)row * (uint32_t)tbl.row_words() + (uint32_t)(ad * TBL_ENTRY_WORDS)) + FE_NL * el;
            fe qel;
            for (int i = 0; i < FE_NL; i++) qel.v[i] = e[i];
            quad_madd(c, ql, qel, neg);
        }
    }
    for (int off = 16; off >= 4; off >>= 1) {                        // the 8 groups of a list -> its first
        fe o;
        for (int i = 0; i < FE_NL; i++) o.v[i] = __shfl_down(c.v[i], off, 64);
        quad_add(c, ql, o);
    }
    if (sub == 0) {
        int32_t* dst = (side ? A.P1 : A.P0) + (b * nsplit + split) * 40 + FE_NL * ql;
        for (int i = 0; i < FE_NL; i++) dst[i] = c.v[i];
    }
}

// Small calls split every proof's term range over several wavefronts (latency): P0 / P1 [b] = sum of the partials, one wavefront
// per (proof, side), the partials summed by a shuffle tree (nsplit a power of two).
__global__ __launch_bounds__(64) void k_rp_sum_splits(size_t B, int nsplit, const int32_t* PS0, const int32_t* PS1, int32_t* P0, int32_t* P1) {
    const size_t t = blockIdx.x;
    if (t >= 2 * B) return;
    const size_t b = t >> 1;
    const int l = threadIdx.x;
    const int32_t* src = ((t & 1) ? PS1 : PS0) + b * (size_t)nsplit * 40;
    ge_p3 acc;
    if (l < nsplit) ld_p3(acc, src + (size_t)l * 40);
    else ge_identity(acc);
    for (int s = l + 64; s < nsplit; s += 64) {                      // (k_rp_msm_quad: up to 256 partials)
        ge_p3 p, r;
        ld_p3(p, src + (size_t)s * 40);
        ge_add(r, acc, p);
        acc = r;
    }
    wave_reduce_point(acc, nsplit < 64 ? nsplit : 64);
    if (l == 0) st_p3(((t & 1) ? P1 : P0) + b * 40, acc);
}

// Split materialisation (small calls): folded generator g = sum of its nsplit partial sums, kept at the head of its table row.
__global__ __launch_bounds__(64) void k_rp_sum_mat(size_t n_gen, int nsplit, const int32_t* PS, int32_t* tailT) {
    size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= n_gen) return;
    const int32_t* src = PS + g * (size_t)nsplit * 40;
    ge_p3 acc, p, r;
    ld_p3(acc, src);
    for (int s = 1; s < nsplit; s++) {
        ld_p3(p, src + (size_t)s * 40);
        ge_add(r, acc, p);
        acc = r;
    }
    st_p3(tailT + g * TAIL_ROW_WORDS, acc);
}

template <int ENTRIES>
__device__ __forceinline__ void build_niels_row(int32_t* row) {
    ge_p3 base, mul;
    ld_p3(base, row);
    mul = base;
    fe pre[ENTRIES - 1];
#pragma unroll
    for (int e = 1; e < ENTRIES; e++) {
        if (e > 1) { ge_p3 t; ge_add(t, mul, base); mul = t; }
        int32_t* slot = row + e * 32;
        for (int i = 0; i < FE_NL; i++) { slot[i] = mul.X.v[i]; slot[FE_NL + i] = mul.Y.v[i]; slot[2 * FE_NL + i] = mul.Z.v[i]; }
        if (e == 1) pre[0] = mul.Z;
        else fe_mul(pre[e - 1], pre[e - 2], mul.Z);
    }
    fe inv;
    fe_invert(inv, pre[ENTRIES - 2]);
#pragma unroll
    for (int e = ENTRIES - 1; e >= 1; e--) {
        int32_t* slot = row + e * 32;
        fe X, Y, Z, zi, x, y;
        for (int i = 0; i < FE_NL; i++) { X.v[i] = slot[i]; Y.v[i] = slot[FE_NL + i]; Z.v[i] = slot[2 * FE_NL + i]; }
        if (e > 1) { fe_mul(zi, inv, pre[e - 2]); fe t; fe_mul(t, inv, Z); inv = t; }
        else zi = inv;
        fe_mul(x, X, zi);
        fe_mul(y, Y, zi);
        ge_niels q;
        ge_to_niels(q, x, y);
        niels_store_entry(slot, q);
    }
    ge_niels id;
    ge_niels_identity(id);
    niels_store_entry(row, id);
}

// One lane per materialised generator: its table row, and the tail argument's vectors: a, b = the first T entries of the
// folded vectors, coefficients s = 1.
__global__ __launch_bounds__(64) void k_rp_tail_table(RangeArgs A) {
    const int T = A.tail_n, bpp = (2 * T) >> 6;                  // blocks per proof
    size_t b = blockIdx.x / bpp;
    int g = (int)(blockIdx.x % bpp) * 64 + threadIdx.x;          // row: G'_g (g < T) or H'_(g-T)
    build_niels_row<TAIL_ENTRIES>(A.tailT + (b * (size_t)(2 * T) + g) * TAIL_ROW_WORDS);
    sc one, v;
    sc_zero(one); one.v[0] = 1;                              // plain 1 (the s-vectors are kept in plain form)
    int i = g < T ? g : g - T;
    if (g == 0) stab_init(A, b);                             // the tail argument starts from coefficient 1 on both sides
    st_sc((g < T ? A.tail_s1 : A.tail_s2) + b * T + i, one);
    ld_sc(v, (g < T ? A.a : A.b) + b * A.N + i);
    st_sc((g < T ? A.tail_a : A.tail_b) + b * T + i, v);
}

// Digits of the s-vectors themselves, in the S layout (list 0 = s_G over G, list 1 = s_H over H): input of the
// materialising MSM above.  grid = B * (TP/64) blocks of 64.
__global__ __launch_bounds__(64) void k_rp_mat_prep(RangeArgs A) {
    int nch = A.TP >> 6;
    size_t b = blockIdx.x / nch;
    int ch = blockIdx.x % nch, l = threadIdx.x, side = l >> 5, q = 32 * ch + (l & 31), pos = 64 * ch + l;
    if (q >= A.N) { zero_digits(A, b, pos); return; }
    sc s;
    coeff_plain(s, A, b, A.mat_round, q, side != 0);        // plain form
    write_digits_plain(A, b, pos, s.v);
}

// --------------------------------------------------------------------------------------- transcript helpers
__device__ __forceinline__ void st_load(Strobe& s, const ProofState& p) {
    for (int i = 0; i < 25; i++) s.s[i] = p.strobe[i];
    s.pos = p.pos;
    s.pos_begin = p.pos_begin;
}
__device__ __forceinline__ void st_store(ProofState& p, const Strobe& s) {
    for (int i = 0; i < 25; i++) p.strobe[i] = s.s[i];
    p.pos = s.pos;
    p.pos_begin = s.pos_begin;
}
__device__ __forceinline__ void st_load(WStrobe& s, const ProofState& p) {
    s.a = s.l < 25 ? p.strobe[s.l < 25 ? s.l : 0] : 0;
    s.pos = p.pos;
    s.pos_begin = p.pos_begin;
}
__device__ __forceinline__ void st_store(ProofState& p, const WStrobe& s) {
    if (s.l < 25) p.strobe[s.l] = s.a;
    if (s.l == 0) { p.pos = s.pos; p.pos_begin = s.pos_begin; }
}
template <class S>
__device__ __forceinline__ void challenge_scalar(sc& r, S& s, const char* label, int n) {
    uint32_t w[16];
    merlin_challenge_wide(s, label, n, w);
    sc_from_wide(r, w);
}
template <class S>
__device__ __forceinline__ void append_scalar(S& s, const char* label, int n, const uint32_t* canon8) {
    merlin_append_words(s, label, n, canon8, 8);
}
// t1 / t2 / t_x (which = 0 / 1 / 2) of proof b: the value k_rp_poly / k_rp_lr left in the state, or the sum of the wavefronts' shares
__device__ __forceinline__ void fs_scalar(sc& r, const RangeArgs& A, size_t b, int which) {
    const ProofState& ps = A.st[b];
    if (A.fs_parts > 1) {
        sc_zero(r);
        for (int w = 0; w < A.fs_parts; w++) {
            sc t;
            ld_sc(t, A.fs_part + (b * A.fs_parts + w) * 3 + which);
            sc_add(r, r, t);
        }
    } else {
        r = which == 0 ? ps.t1 : which == 1 ? ps.t2 : ps.t_x;
    }
}
// The lane-per-proof Fiat-Shamir kernels come in three shapes (template MODE):
//   0  one lane per proof (throughput: tens of thousands of proofs per launch);
//   1  PAIR: the two point computations of a proof (A and S; T_1, T_2; L_k, R_k) on two neighbouring lanes, the second encoding
//      handed to the first, which owns the transcript -- the lane's serial chain is what a small call waits for;
//   2  one WAVEFRONT per proof (a handful of proofs): the two point computations by the two half-wavefronts (blinding sums and the
//      windows of the fixed-base products spread over the lanes, lanes 0 and 32 add up and encode), the transcript held by the
//      wavefront (WStrobe, hash.h: 4 us per Keccak permutation instead of 23).
template <int MODE> struct FsShape {
    using strobe_t = typename std::conditional<MODE == 2, WStrobe, Strobe>::type;
    const int l;                 // lane in the block
    size_t b;                    // proof
    bool valid;
    int h_lo, h_hi;              // the halves (0: first point, 1: second) this lane computes
    __device__ FsShape(const RangeArgs& A) : l((int)threadIdx.x) {
        const size_t t_ = (size_t)blockIdx.x * 64 + threadIdx.x;
        b = MODE == 2 ? (size_t)blockIdx.x : (MODE == 1 ? t_ >> 1 : t_);
        valid = b < A.B;
        if (!valid) b = A.B - 1;
        if (MODE == 0) { h_lo = 0; h_hi = 2; }
        else if (MODE == 1) { h_lo = (int)(t_ & 1); h_hi = h_lo + 1; }
        else { h_lo = l >> 5; h_hi = h_lo + 1; }                               // half-wavefront h; its lane 0 ends with the point
    }
    // second encoding -> the lane that owns the transcript (MODE 1), or both -> every lane (MODE 2)
    __device__ void share(uint32_t* first8, uint32_t* second8) const {
        if (MODE == 1) { for (int i = 0; i < 8; i++) second8[i] = (uint32_t)__shfl_down((int)second8[i], 1, 64); }
        if (MODE == 2) {
            for (int i = 0; i < 8; i++) { first8[i] = (uint32_t)__shfl((int)first8[i], 0, 64); second8[i] = (uint32_t)__shfl((int)second8[i], 32, 64); }
        }
    }
    __device__ bool point_lane() const { return MODE == 2 ? (l & 31) == 0 : true; }
    __device__ bool owns_transcript() const { return MODE == 2 ? true : (MODE == 1 ? ((l & 1) == 0 && valid) : valid); }
    __device__ bool writes() const { return MODE == 2 ? l == 0 : true; }     // (among the lanes that own the transcript)
    __device__ void begin(strobe_t& s) const { if constexpr (MODE == 2) wstrobe_lanes(s, l); }
};
// Sum over the parties of one blinding stream (slot0 + j * stride): serial on a lane, or (MODE 2) spread over the 32 lanes of the
// half-wavefront and summed by shuffles (every lane of the half ends with the sum).
template <int MODE>
__device__ __forceinline__ void blinding_sum(sc& bl, const RangeArgs& A, size_t b, uint32_t slot0, uint32_t stride, int lane) {
    sc t;
    sc_zero(bl);
    if (MODE == 2) {
        for (int j = lane & 31; j < A.m; j += 32) {
            tape_scalar(t, A, b, slot0 + (uint32_t)j * stride);
            sc_add(bl, bl, t);
        }
        for (int off = 16; off >= 1; off >>= 1) {
            sc o, r;
            for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)__shfl_down((int)bl.v[i], off, 64);
            sc_add(r, bl, o);
            bl = r;
        }
        for (int i = 0; i < 8; i++) bl.v[i] = (uint32_t)__shfl((int)bl.v[i], lane & 32, 64);       // every lane of the half
    } else {
        for (int j = 0; j < A.m; j++) {
            tape_scalar(t, A, b, slot0 + (uint32_t)j * stride);
            sc_add(bl, bl, t);
        }
    }
}

// --------------------------------------------------------- F1: finish A and S, transcript up to y, z (lane/proof)
template <int MODE>
__global__ __launch_bounds__(64) void k_rp_finish1(RangeArgs A, TableView tbl) {
    FsShape<MODE> F(A);
    if (MODE == 0 && !F.valid) return;
    const size_t b = F.b;
    ProofState& ps = A.st[b];
    uint32_t c[8], Ac[8] = {0}, Sc[8] = {0};
    for (int h = F.h_lo; h < F.h_hi; h++) {
        sc bl;
        blinding_sum<MODE>(bl, A, b, (uint32_t)h, (uint32_t)(2 * A.n + 2), F.l);       // a_blinding (h = 0) / s_blinding of every party
        sc_from_mont(c, bl);
        ge_p3 p, q;
        if (MODE == 2) tbl_fixed_mul_wave(q, tbl, tbl.row_Bb(0), c, F.l & 31);
        if (!F.point_lane()) continue;
        if (h == 0) ld_p3(p, A.PA + b * 40);
        else {
            ge_p3 p0, p1;
            ld_p3(p0, A.P0 + b * 40);
            ld_p3(p1, A.P1 + b * 40);
            ge_add(p, p0, p1);
        }
        if (MODE == 2) { ge_p3 r; ge_add(r, p, q); p = r; }
        else tbl_fixed_mul_add(p, tbl, tbl.row_Bb(0), c);
        ge_compress(c, p);
        for (int i = 0; i < 8; i++) { if (h) Sc[i] = c[i]; else Ac[i] = c[i]; }
        if (F.valid) { if (h) ps.s_bl = bl; else ps.a_bl = bl; }
    }
    F.share(Ac, Sc);
    if (!F.owns_transcript()) return;
    uint32_t* out = proof_out(A, b);
    if (F.writes()) { st8(out, Ac); st8(out + 8, Sc); }
    typename FsShape<MODE>::strobe_t s;
    F.begin(s);
    merlin_init(s, LBL_APP_TRANSCRIPT);                                       // Transcript::new(&[])  (src/range/mod.rs:51,67)
    merlin_append_bytes(s, LBL_DOM_SEP, LBL_RANGEPROOF_DOMAIN);
    merlin_append_u64(s, LBL_N, (uint64_t)A.n);
    merlin_append_u64(s, LBL_M, (uint64_t)A.m);
    for (int j = 0; j < A.m; j++) {
        uint32_t v[8];
        ld8(v, A.Vc + (b * A.m + j) * 8);
        merlin_append_words(s, LBL_V, v, 8);
    }
    merlin_append_words(s, LBL_A, Ac, 8);
    merlin_append_words(s, LBL_S, Sc, 8);
    sc y, z, yi;
    challenge_scalar(y, s, LBL_Y);
    challenge_scalar(z, s, LBL_Z);
    sc_invert_vartime_mont(yi, y);                          // y is a public challenge
    if (F.writes()) { ps.y = y; ps.z = z; ps.y_inv = yi; ps.err = 0; }
    st_store(ps, s);
}

// ------------------------------------------------- K3: l0, l1, r0, r1 and t1, t2 (wave per proof)
// Party::apply_challenge_with_rng restated over the concatenated vectors: position i = (party j, bit ii).
__global__ __launch_bounds__(64) void k_rp_poly(RangeArgs A) {
    // small calls: fs_parts wavefronts per proof, wavefront w taking the positions [i_lo, i_hi) (whole multiples of 64)
    const int parts = A.fs_parts > 1 ? A.fs_parts : 1, part = (int)(blockIdx.x % parts);
    size_t b = blockIdx.x / parts;
    int l = threadIdx.x;
    const int iters = (A.N + 63) / 64, per = (iters + parts - 1) / parts;
    const int i_lo = 64 * per * part, i_hi = (64 * per * (part + 1) < A.N) ? 64 * per * (part + 1) : A.N;
    const ProofState& ps = A.st[b];
    sc y = ps.y, z = ps.z, one, yi, y64, t1, t2;
    sc_one_mont(one);
    sc_pow_mont(yi, y, (uint32_t)(i_lo + l));
    sc_pow_mont(y64, y, 64u);
    sc_zero(t1);
    sc_zero(t2);
    sc zm1;
    sc_sub(zm1, z, one);
    // A lane's positions i = l + 64 k share the bit index ii = l mod n (n divides 64) and step the party by 64 / n: z^(2+j) 2^ii
    // is a recurrence, not a power per position (a power is ~10 products; this loop was 0.28 ms of a lone proof's 6.4).
    sc zz, zstep, two;
    sc_pow_mont(zz, z, (uint32_t)(2 + (i_lo + l) / A.n));
    sc_pow_mont(zstep, z, (uint32_t)(64 / A.n));
    sc_from_u64_mont(two, 1ull << (l % A.n));
    sc_montmul(zz, zz, two);                                       // z^(2+j) 2^ii for this lane's first position
    for (int i = i_lo + l; i < i_hi; i += 64) {
        int j = i / A.n, ii = i - j * A.n;
        int bit = (int)((A.vals[b * A.m + j] >> ii) & 1ull);
        sc l0, l1, sR, r0, r1, t;
        const sc u = zz;
        sc_montmul(zz, zz, zstep);
        if (bit) sc_sub(l0, one, z); else sc_neg(l0, z);          // a_L - z
        ld_sc(l1, A.s1 + b * A.N + i);
        ld_sc(sR, A.s2 + b * A.N + i);
        sc_montmul(t, yi, bit ? z : zm1);                          // y^i (a_R + z)
        sc_add(r0, t, u);
        sc_montmul(r1, yi, sR);
        sc_montmul(t, l0, r1);
        sc_add(t1, t1, t);
        sc_montmul(t, l1, r0);
        sc_add(t1, t1, t);
        sc_montmul(t, l1, r1);
        sc_add(t2, t2, t);
        st_sc(A.b + b * A.N + i, r0);
        st_sc(A.s2 + b * A.N + i, r1);
        sc_montmul(yi, yi, y64);
    }
    wave_reduce_sc(t1);
    wave_reduce_sc(t2);
    if (l == 0) {
        if (parts > 1) { st_sc(A.fs_part + (b * parts + part) * 3 + 0, t1); st_sc(A.fs_part + (b * parts + part) * 3 + 1, t2); }
        else { A.st[b].t1 = t1; A.st[b].t2 = t2; }
    }
}

// ------------------------------------------------------------- F2: T1, T2 and the challenge x (lane/proof)
template <int MODE>
__global__ __launch_bounds__(64) void k_rp_finish2(RangeArgs A, TableView tbl) {
    FsShape<MODE> F(A);
    if (MODE == 0 && !F.valid) return;
    const size_t b = F.b;
    ProofState& ps = A.st[b];
    const uint32_t base = (uint32_t)(A.m * (2 * A.n + 2));
    uint32_t c[8], T1c[8] = {0}, T2c[8] = {0};
    for (int h = F.h_lo; h < F.h_hi; h++) {
        sc bl;
        blinding_sum<MODE>(bl, A, b, base + (uint32_t)h, 2u, F.l);            // t_1_blinding (h = 0) / t_2_blinding of every party
        ge_p3 p;
        if (MODE == 2) {
            ge_p3 q;
            sc tq;
            fs_scalar(tq, A, b, h);
            sc_from_mont(c, tq);
            tbl_fixed_mul_wave(p, tbl, tbl.row_B(0), c, F.l & 31);
            sc_from_mont(c, bl);
            tbl_fixed_mul_wave(q, tbl, tbl.row_Bb(0), c, F.l & 31);
            if (!F.point_lane()) continue;
            ge_p3 r;
            ge_add(r, p, q);
            p = r;
        } else {
            ge_identity(p);
            sc tq;
            fs_scalar(tq, A, b, h);
            sc_from_mont(c, tq);
            tbl_fixed_mul_add(p, tbl, tbl.row_B(0), c);
            sc_from_mont(c, bl);
            tbl_fixed_mul_add(p, tbl, tbl.row_Bb(0), c);
        }
        ge_compress(c, p);
        for (int i = 0; i < 8; i++) { if (h) T2c[i] = c[i]; else T1c[i] = c[i]; }
        if (F.valid) { if (h) ps.t2_bl = bl; else ps.t1_bl = bl; }
    }
    F.share(T1c, T2c);
    if (!F.owns_transcript()) return;
    uint32_t* out = proof_out(A, b);
    if (F.writes()) { st8(out + 16, T1c); st8(out + 24, T2c); }
    typename FsShape<MODE>::strobe_t s;
    F.begin(s);
    st_load(s, ps);
    merlin_append_words(s, LBL_T1, T1c, 8);
    merlin_append_words(s, LBL_T2, T2c, 8);
    sc x;
    challenge_scalar(x, s, LBL_X);
    if (F.writes()) {
        if (sc_is_zero(x)) ps.err = 1;                            // ProofError::MaliciousDealer in the crate
        ps.x = x;
    }
    st_store(ps, s);
}

// ------------------------------------- K4: l = l0 + l1 x, r = r0 + r1 x, t_x = <l, r>, s-vector init (wave/proof)
__global__ __launch_bounds__(64) void k_rp_lr(RangeArgs A) {
    const int parts = A.fs_parts > 1 ? A.fs_parts : 1, part = (int)(blockIdx.x % parts);      // (as k_rp_poly)
    size_t b = blockIdx.x / parts;
    int l = threadIdx.x;
    const int iters = (A.N + 63) / 64, per = (iters + parts - 1) / parts;
    const int i_lo = 64 * per * part, i_hi = (64 * per * (part + 1) < A.N) ? 64 * per * (part + 1) : A.N;
    const ProofState& ps = A.st[b];
    sc x = ps.x, z = ps.z, yinv = ps.y_inv, one, yi, y64, tx;
    sc_one_mont(one);
    sc_pow_mont(yi, yinv, (uint32_t)(i_lo + l));
    sc_pow_mont(y64, yinv, 64u);
    sc_zero(tx);
    for (int i = i_lo + l; i < i_hi; i += 64) {
        int j = i / A.n, ii = i - j * A.n;
        int bit = (int)((A.vals[b * A.m + j] >> ii) & 1ull);
        sc l0, l1, r0, r1, lv, rv, t;
        if (bit) sc_sub(l0, one, z); else sc_neg(l0, z);
        ld_sc(l1, A.s1 + b * A.N + i);
        ld_sc(r0, A.b + b * A.N + i);
        ld_sc(r1, A.s2 + b * A.N + i);
        sc_montmul(t, l1, x);
        sc_add(lv, l0, t);
        sc_montmul(t, r1, x);
        sc_add(rv, r0, t);
        sc_montmul(t, lv, rv);
        sc_add(tx, tx, t);
        st_sc(A.a + b * A.N + i, lv);
        st_sc(A.b + b * A.N + i, rv);
        // The coefficient vectors s_G, s_H live in PLAIN form (not Montgomery): montmul(x_mont, s_plain) = x * s is then the
        // canonical product k_rp_round_prep recodes, with no conversion per term; montmul(s_plain, u_mont) keeps them plain.
        sc pl;
        sc_zero(pl); pl.v[0] = 1;
        st_sc(A.s1 + b * A.N + i, pl);           // s_G[i] = 1
        sc_from_mont(pl.v, yi);
        st_sc(A.s2 + b * A.N + i, pl);           // s_H[i] = y^-i   (H' = y^-i H, folded into the scalar)
        sc_montmul(yi, yi, y64);
    }
    wave_reduce_sc(tx);
    if (l == 0) {
        if (part == 0) stab_init(A, b);
        if (parts > 1) st_sc(A.fs_part + (b * parts + part) * 3 + 2, tx);
        else A.st[b].t_x = tx;
    }
}

// ------------------------------- F3: t_x, tau_x, mu; challenge w; inner-product domain separator (lane/proof)
template <int MODE>                     // 0: a lane per proof; 2: a wavefront per proof (every lane computes the same scalars)
__global__ __launch_bounds__(64) void k_rp_finish3(RangeArgs A) {
    size_t b = MODE == 2 ? (size_t)blockIdx.x : (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B) return;
    const bool writes = MODE == 2 ? threadIdx.x == 0 : true;
    ProofState& ps = A.st[b];
    sc x = ps.x, z = ps.z, xx, zz, tau, mu, t, bl;
    sc_montmul(xx, x, x);
    sc_montmul(zz, z, z);
    sc_zero(tau);
    for (int j = 0; j < A.m; j++) {                               // sum_j z^(2+j) * v_blinding_j
        uint32_t w[8];
        ld8(w, A.blind + (b * A.m + j) * 8);
        sc_to_mont(bl, w);
        sc_montmul(t, zz, bl);
        sc_add(tau, tau, t);
        sc_montmul(zz, zz, z);
    }
    sc_montmul(t, ps.t1_bl, x);
    sc_add(tau, tau, t);
    sc_montmul(t, ps.t2_bl, xx);
    sc_add(tau, tau, t);
    sc_montmul(t, ps.s_bl, x);
    sc_add(mu, ps.a_bl, t);
    uint32_t c_tx[8], c_tau[8], c_mu[8];
    sc t_x;
    fs_scalar(t_x, A, b, 2);
    sc_from_mont(c_tx, t_x);
    sc_from_mont(c_tau, tau);
    sc_from_mont(c_mu, mu);
    uint32_t* out = proof_out(A, b);
    if (writes) {
        st8(out + 32, c_tx);
        st8(out + 40, c_tau);
        st8(out + 48, c_mu);
    }
    typename std::conditional<MODE == 2, WStrobe, Strobe>::type s;
    if constexpr (MODE == 2) wstrobe_lanes(s, (int)threadIdx.x);
    st_load(s, ps);
    append_scalar(s, LBL_TX, c_tx);
    append_scalar(s, LBL_TX_BLINDING, c_tau);
    append_scalar(s, LBL_E_BLINDING, c_mu);
    sc w;
    challenge_scalar(w, s, LBL_W);
    merlin_append_bytes(s, LBL_DOM_SEP, LBL_IPP_DOMAIN);
    merlin_append_u64(s, LBL_N, (uint64_t)A.N);
    if (writes) ps.w = w;
    st_store(ps, s);
}

// ------------------------------------------------- K5: round-k MSM scalars -> digits (grid B * TP/64 blocks)
__global__ __launch_bounds__(64) void k_rp_round_prep(RangeArgs A, int round) {
    int nch = A.TP >> 6;
    size_t b = blockIdx.x / nch;
    int ch = blockIdx.x % nch, l = threadIdx.x, side = l >> 5, q = 32 * ch + (l & 31), pos = 64 * ch + l;
    if (q >= A.N) { zero_digits(A, b, pos); return; }
    bool isH;
    int j = term_generator(round, A.N, A.lgN, side, q, isH);
    int lgh = A.lgN - 1 - round, half = 1 << lgh;
    int off = j & (half - 1);
    bool upper = (j >> lgh) & 1;              // generator sits in the upper half of its block
    // G_R pairs with a_L, G_L with a_R; H'_L pairs with b_R, H'_R with b_L
    int vi = upper ? off : off + half;
    sc v, p;
    ld_sc(v, (isH ? A.b : A.a) + b * A.N + vi);
    coeff_times(p, A, b, round, j, isH, v);
    write_digits_plain(A, b, pos, p.v);
}
// c_L = <a_L, b_R>, c_R = <a_R, b_L>  (wave per proof)
__global__ __launch_bounds__(64) void k_rp_round_ip(RangeArgs A, int round) {
    size_t b = blockIdx.x;
    int l = threadIdx.x, half = 1 << (A.lgN - 1 - round);
    sc cL, cR;
    sc_zero(cL);
    sc_zero(cR);
    for (int i = l; i < half; i += 64) {
        sc aL, aR, bL, bR, t;
        ld_sc(aL, A.a + b * A.N + i);
        ld_sc(aR, A.a + b * A.N + half + i);
        ld_sc(bL, A.b + b * A.N + i);
        ld_sc(bR, A.b + b * A.N + half + i);
        sc_montmul(t, aL, bR);
        sc_add(cL, cL, t);
        sc_montmul(t, aR, bL);
        sc_add(cR, cR, t);
    }
    wave_reduce_sc(cL);
    wave_reduce_sc(cR);
    if (l == 0) { A.st[b].cL = cL; A.st[b].cR = cR; }
}

// --------------------------------------------------------- F4: L_k, R_k, challenge u_k (lane/proof)
template <int MODE>
__global__ __launch_bounds__(64) void k_rp_round_finish(RangeArgs A, TableView tbl, int round) {
    FsShape<MODE> F(A);
    if (MODE == 0 && !F.valid) return;
    const size_t b = F.b;
    ProofState& ps = A.st[b];
    uint32_t c[8], Lc[8] = {0}, Rc[8] = {0};
    for (int h = F.h_lo; h < F.h_hi; h++) {
        ge_p3 p, q;
        sc t;
        sc_montmul(t, h ? ps.cR : ps.cL, ps.w);
        sc_from_mont(c, t);
        if (MODE == 2) tbl_fixed_mul_wave(q, tbl, tbl.row_B(0), c, F.l & 31);
        if (!F.point_lane()) continue;
        ld_p3(p, (h ? A.P1 : A.P0) + b * 40);
        if (MODE == 2) { ge_p3 r; ge_add(r, p, q); p = r; }
        else tbl_fixed_mul_add(p, tbl, tbl.row_B(0), c);      // + c_L * Q,  Q = w * B
        ge_compress(c, p);
        for (int i = 0; i < 8; i++) { if (h) Rc[i] = c[i]; else Lc[i] = c[i]; }
    }
    F.share(Lc, Rc);
    if (!F.owns_transcript()) return;
    uint32_t* out = proof_out(A, b) + 56 + 16 * (A.out_round0 + round);
    if (F.writes()) { st8(out, Lc); st8(out + 8, Rc); }
    typename FsShape<MODE>::strobe_t s;
    F.begin(s);
    st_load(s, ps);
    merlin_append_words(s, LBL_L, Lc, 8);
    merlin_append_words(s, LBL_R, Rc, 8);
    sc u, ui;
    challenge_scalar(u, s, LBL_U);
    sc_invert_vartime_mont(ui, u);                          // u_k is a public challenge
    if (F.writes()) { ps.u = u; ps.u_inv = ui; }
    st_store(ps, s);
}

// --------------------------------------------- K6: fold a, b and update the coefficients (elementwise)
// Vector mode (A.stab null): B * N lanes, lane j multiplies s_G[j], s_H[j] by u^{+-1} and, for j < half, folds a and b.
// Table mode: B * max(half, 2^(round + 2)) lanes; lane j < half folds a and b (both products of a fold in the 29-bit column domain,
// ONE Montgomery reduction per folded entry), and lanes j < 2^(round + 2) extend the coefficient tables by the new challenge:
// TG'[2t + bit] = TG[t] * (bit ? u : u^-1), TH'[2t + bit] = TH[t] * (bit ? u^-1 : u), into the other buffer.
__device__ __forceinline__ void fold_pair(sc& r, const sc& lo, const sc& hi, const uint32_t* ulo29, const uint32_t* uhi29) {
    uint32_t x[9];
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
    sc_split29(x, lo.v);
    sc_mac29(c, x, ulo29);
    sc_split29(x, hi.v);
    sc_mac29(c, x, uhi29);
    sc_redc29(r, c);                              // (lo * ulo + hi * uhi) / R mod l, canonical: what two products and a sum give
}
__global__ __launch_bounds__(256) void k_rp_fold(RangeArgs A, int round) {
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    int lgh = A.lgN - 1 - round, half = 1 << lgh;
    if (A.stab) {
        const int nt = (round + 1 <= STAB_ROUNDS) ? (2 << (round + 1)) : 0;          // table entries to write: 2^(round+1) per side
        const size_t per = (size_t)(half > nt ? half : nt);
        if (gid >= A.B * per) return;
        const size_t b = gid / per;
        const int j = (int)(gid - b * per);
        const ProofState& ps = A.st[b];
        const sc u = ps.u, ui = ps.u_inv;
        if (j < nt) {
            const int side = j >= (nt >> 1) ? 1 : 0, tp = side ? j - (nt >> 1) : j, t = tp >> 1, bit = tp & 1;
            const sc* src = A.stab + ((b * 2 + (size_t)(round & 1)) * 2 + side) * STAB_N;
            sc* dst = A.stab + ((b * 2 + (size_t)((round + 1) & 1)) * 2 + side) * STAB_N;
            sc x, y;
            ld_sc(x, src + t);
            sc_montmul(y, x, (bit != 0) != (side != 0) ? u : ui);        // G: bit ? u : u^-1;  H: bit ? u^-1 : u
            st_sc(dst + tp, y);
        }
        if (j < half) {
            uint32_t u29[9], ui29[9];
            sc_split29(u29, u.v);
            sc_split29(ui29, ui.v);
            sc lo, hi, r;
            ld_sc(lo, A.a + b * A.N + j);
            ld_sc(hi, A.a + b * A.N + half + j);
            fold_pair(r, lo, hi, u29, ui29);
            st_sc(A.a + b * A.N + j, r);              // a' = a_L u + a_R u^-1
            ld_sc(lo, A.b + b * A.N + j);
            ld_sc(hi, A.b + b * A.N + half + j);
            fold_pair(r, lo, hi, ui29, u29);
            st_sc(A.b + b * A.N + j, r);              // b' = b_L u^-1 + b_R u
        }
        return;
    }
    if (gid >= A.B * (size_t)A.N) return;
    size_t b = gid / A.N;
    int j = (int)(gid - b * A.N);
    const ProofState& ps = A.st[b];
    sc u = ps.u, ui = ps.u_inv;
    bool upper = (j >> lgh) & 1;
    sc s, t;
    ld_sc(s, A.s1 + b * A.N + j);                 // G' = u^-1 G_L + u G_R
    sc_montmul(t, s, upper ? u : ui);
    st_sc(A.s1 + b * A.N + j, t);
    ld_sc(s, A.s2 + b * A.N + j);                 // H' = u H_L + u^-1 H_R
    sc_montmul(t, s, upper ? ui : u);
    st_sc(A.s2 + b * A.N + j, t);
    if (j < half) {
        sc lo, hi, r;
        ld_sc(lo, A.a + b * A.N + j);
        ld_sc(hi, A.a + b * A.N + half + j);
        sc_montmul(lo, lo, u);
        sc_montmul(hi, hi, ui);
        sc_add(r, lo, hi);
        st_sc(A.a + b * A.N + j, r);              // a' = a_L u + a_R u^-1
        ld_sc(lo, A.b + b * A.N + j);
        ld_sc(hi, A.b + b * A.N + half + j);
        sc_montmul(lo, lo, ui);
        sc_montmul(hi, hi, u);
        sc_add(r, lo, hi);
        st_sc(A.b + b * A.N + j, r);              // b' = b_L u^-1 + b_R u
    }
}

// ------------------------------------------------------------------------------ F5: final a, b (lane/proof)
__global__ __launch_bounds__(64) void k_rp_final(RangeArgs A, uint32_t* err_flag) {
    size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B) return;
    sc a, bb;
    uint32_t c[8];
    ld_sc(a, A.a + b * A.N);
    ld_sc(bb, A.b + b * A.N);
    uint32_t* out = proof_out(A, b) + 56 + 16 * (A.out_round0 + A.lgN);
    sc_from_mont(c, a);
    st8(out, c);
    sc_from_mont(c, bb);
    st8(out + 8, c);
    if (A.st[b].err) atomicOr(err_flag, 1u);
}

// Value commitments of the parties (Party::new: V = commit(v, v_blinding)) when the caller has none yet.
__global__ __launch_bounds__(256) void k_rp_commit_V(TableView tbl, size_t n, const uint64_t* v, const uint32_t* r, uint32_t* Vc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rw[8], c[8];
    ld8(rw, r + i * 8);
    rw[7] &= 0x7fffffffu;
    ge_p3 acc;
    ge_identity(acc);
    tbl_fixed_mul_add_u64(acc, tbl, tbl.row_B(0), v[i]);
    tbl_fixed_mul_add(acc, tbl, tbl.row_Bb(0), rw);
    ge_compress(c, acc);
    st8(Vc + i * 8, c);
}

// Scratch owned by the context, grown on demand.
struct RangeScratch {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace dapol
