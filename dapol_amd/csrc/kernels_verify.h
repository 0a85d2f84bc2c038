// Batched verifier: RangeProof::verify_multiple (bulletproofs 4.0.0) as called from src/range/mod.rs:83-119.
// One multiscalar check per proof:  sum_i g_i G_i + sum_i h_i H_i  (fixed-base: the prover's MSM kernel)
//   + A + x S + c x T1 + c x^2 T2 + sum u_k^2 L_k + sum u_k^-2 R_k + sum c z^(2+j) V_j   (the proof's own points)
//   + (-mu - c tau) B_blinding + (w (t_x - a b) + c (delta - t_x)) B   ==  identity.
#pragma once
#include "kernels_range.h"

namespace dapol {

enum { RV_MAX_ROUNDS = 20, RV_MAX_LGM = 11 };
struct VerifyState {                 // per proof
    sc y, z, y_inv, x, w, c, a, b, t_x, tau, mu;
    sc ym1_inv, zm1_inv;             // 1 / (y - 1), 1 / (z - 1) for the closed-form power sums (0 when y or z is 1)
    sc rho;                          // weight of this proof in a cross-proof batch (random linear combination)
    sc u[RV_MAX_ROUNDS], u_inv[RV_MAX_ROUNDS];
    sc ypow[RV_MAX_ROUNDS], zpow[RV_MAX_LGM];   // y^-(2^k), z^(2^k): a table entry is the product over the set bits of its exponent
    uint32_t ok, st_pos, st_pos_begin, pad_;
    uint64_t st[25];                 // STROBE state after the m commitments (k_rv_absorb_V -> k_rv_transcript)
    uint32_t dg[8];                  // digest of the whole statement + proof (transcript after a, b): input of the batch weights
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
    int wave_transcript;             // 1: k_rv_absorb_V has absorbed the commitments, k_rv_transcript resumes from vs.st
};
__host__ __device__ inline int rv_tab_entries(int lgN, int m) {
    int lb = lgN / 2, hb = lgN - lb;
    return 2 * (1 << hb) + 2 * (1 << lb) + m;
}
// Cross-proof batching appends product tables (k_rvb_tables2) that bring the rho-weighted generator scalars down to one
// (G side) and two (H side) products per proof and generator:
//   SHa[h] = -rho a SH[h] | PHb[h] = -rho b YH[h] SH[nh-1-h] | QH[h] = rho YH[h] z^2 z^jh(h) 2^ih(h) | PL[l] = YL[l] SL[nl-1-l] |
//   QL[l] = YL[l] z^jl(l) 2^il(l) | RZ = rho z,      where the party j = jh + jl and the bit i' = ih + il of position
//   q = h 2^lb + l split between the halves (see rv_q_split).
__host__ __device__ inline int rv_tab_entries_rlc(int lgN, int m) {
    int lb = lgN / 2, hb = lgN - lb;
    return rv_tab_entries(lgN, m) + 3 * (1 << hb) + 2 * (1 << lb) + 1;
}
// Position q = h 2^lb + l of an n-bit, m-party statement: party j = q / n and bit i' = q mod n, as sums of a part that
// depends on h only and a part that depends on l only.
__host__ __device__ inline void rv_q_split(int lgn, int lb, int h, int l, int& jh, int& ih, int& jl, int& il) {
    if (lb >= lgn) { jh = h << (lb - lgn); ih = 0; jl = l >> lgn; il = l & ((1 << lgn) - 1); }
    else { jh = h >> (lgn - lb); ih = (h & ((1 << (lgn - lb)) - 1)) << lb; jl = 0; il = l; }
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

// V0 (many parties): the m commitments, absorbed by ONE WAVEFRONT per proof.
// The replay below keeps one lane per proof -- right for inclusion proofs (m = 32, tens of thousands of proofs), but a
// 1,024-party proof appends 1,024 x 41 bytes = 253 STROBE blocks before its first challenge, and a lane needs ~12,000
// instructions per Keccak-f[1600]: 17 ms of pure latency for configs[4] however small the batch.  Two observations
// make this phase parallel:
//  * nothing in it squeezes, so its byte stream is a closed-form function of the offset k: V number j = k / 41
//    contributes  [pos_begin, M|A, 'V', LE32(32), pos_begin', A, 32 bytes]  (merlin_frame + strobe_begin_op above), and the
//    two position bytes follow from k alone: STROBE sets pos_begin = pos + 1 at a begin_op and clears it in run_f, i.e.
//    old_begin(k) = ((pos0 + k_prev) mod 166) + 1 if k_prev and k fall into the same 166-byte block, else 0.
//    Every lane therefore assembles its own 8 bytes of a block without looking at any other;
//  * the permutation runs with one 64-bit state word per lane (lane = x + 5 y): theta, rho/pi and chi are nine lane
//    permutations (ds_bpermute) and a dozen ALU operations per round instead of ~500.
// The state it leaves in vs.st / st_pos / st_pos_begin is bit for bit what the lane-per-proof loop would hold.
// PHASES (round 5): the kernel absorbs the stream blocks that lie entirely inside commitments [0, j1) and were not absorbed by the
// phases before it (which had commitments [0, j0)), carrying the state through vs.st -- so that a host whose commitments are still
// on their way over PCIe can start the replay on the first quarter of every proof's commitments while the next quarter is being
// copied (dapol_range_verify_batch).  One phase (j0 = 0, j1 = m) is the whole thing.
// The transcript's head -- Transcript::new, the range proof's domain separator, n and m -- is the same for every proof of a call:
// the HOST replays it once (rv_transcript_head; hash.h's one-lane Strobe compiles for both sides) and hands the state over in the
// kernel arguments (round 5: replayed by every wavefront of every phase it cost ~35 us a phase -- 23 for Strobe128::new's
// permutation on a lone lane, the rest for byte-wise writes into a 25-word array indexed by a run-time position).
struct VHead {
    uint64_t st[25];
    uint32_t pos, pos_begin;
};
inline VHead rv_transcript_head(int n, int m) {
    Strobe s;
    merlin_init(s, LBL_APP_TRANSCRIPT);
    merlin_append_bytes(s, LBL_DOM_SEP, LBL_RANGEPROOF_DOMAIN);
    merlin_append_u64(s, LBL_N, (uint64_t)n);
    merlin_append_u64(s, LBL_M, (uint64_t)m);
    VHead h;
    for (int i = 0; i < 25; i++) h.st[i] = s.s[i];
    h.pos = s.pos;
    h.pos_begin = s.pos_begin;
    return h;
}
__global__ __launch_bounds__(64) void k_rv_absorb_V(VerifyArgs V, VHead H, int j0, int j1) {
    const RangeArgs& A = V.R;
    const size_t b = blockIdx.x;
    const int l = threadIdx.x;
    // A latency chain on a lone wavefront, while the side stream's decode fills the same SIMDs with throughput work: ask the
    // arbiter for the issue slots (alone on an idle chip a block costs 5.5 us, tools/ubench_absorb.hip; beside the decode, 7.3).
    __builtin_amdgcn_s_setprio(3);
    const uint32_t pos0 = H.pos, pb0 = H.pos_begin;
    VerifyState& vs = V.vs[b];
    uint64_t a = l < 25 ? (j0 > 0 ? vs.st[l] : H.st[l]) : 0;
    KeccakLanes K;
    keccak_lanes_init(K, l);
    const uint32_t total = RV_V_BYTES * (uint32_t)A.m, end_abs = pos0 + total, nfull = end_abs / STROBE_R;
    const bool last_phase = j1 >= A.m;
    // blocks whose every byte belongs to a commitment below j: floor((41 j + pos0) / 166)
    const uint32_t beta_begin = j0 > 0 ? (RV_V_BYTES * (uint32_t)j0 + pos0) / STROBE_R : 0u;
    const uint32_t beta_end = last_phase ? nfull : (RV_V_BYTES * (uint32_t)j1 + pos0) / STROBE_R;
    // This lane's eight bytes of block beta, without branches and without a loop over the bytes (round 5: the byte loop was ~200
    // instructions and eight dependent byte loads per block on the replay's critical path): absorb_block_word (hash.h, where the
    // stream's form is written down and from where the host-side test replays it) from three aligned words, which block_load
    // requests a block AHEAD, so that they travel while the block before is permuted.
    struct Raw { uint32_t d0, d1, d2; };
    const uint32_t* Vw = A.Vc + b * (size_t)A.m * 8;
    const uint32_t last_w = 8 * (uint32_t)A.m - 1;
    auto block_load = [&](uint32_t beta) -> Raw {
        const uint32_t w0 = absorb_first_word(absorb_window(beta, (uint32_t)l, pos0));        // words past the proof's last one hold no live byte
        return Raw{Vw[w0 < last_w ? w0 : last_w], Vw[w0 + 1 < last_w ? w0 + 1 : last_w], Vw[w0 + 2 < last_w ? w0 + 2 : last_w]};
    };
    auto block_word = [&](uint32_t beta, const Raw& R) -> uint64_t { return absorb_block_word(beta, (uint32_t)l, pos0, pb0, end_abs, R.d0, R.d1, R.d2); };
    const bool any = beta_begin < beta_end || last_phase;
    Raw rw = any ? block_load(beta_begin) : Raw{0, 0, 0};
    uint64_t w = any ? block_word(beta_begin, rw) : 0;
    for (uint32_t beta = beta_begin; beta < beta_end; beta++) {
        a ^= w;
        if (l == 20) a ^= absorb_runf_word(beta, pos0, pb0);       // run_f: pos_begin at byte 166, 0x04 and 0x80 at byte 167
        const bool more = beta + 1 < beta_end || last_phase;
        if (more) rw = block_load(beta + 1);                      // the next block's words travel while this one is permuted
        a = keccak_f1600_wave<24>(a, K, l);                       // (never a block this phase's commitments do not cover)
        if (more) w = block_word(beta + 1, rw);
    }
    if (last_phase) a ^= w;                                       // the bytes after the last permutation
    if (l < 25) vs.st[l] = a;
    if (l == 0 && last_phase) {
        uint32_t pe, pbe;
        absorb_end_position(pos0, (uint32_t)A.m, pe, pbe);
        vs.st_pos = pe;
        vs.st_pos_begin = pbe;
    }
}

// V1: parse + replay the transcript (lane per proof).
// WAVE = 1 (calls of a few proofs): one wavefront per proof, every lane replaying the same transcript with the state spread over
// the lanes (WStrobe, hash.h) -- 4 us per Keccak permutation instead of 23; all lanes compute, and store, the same values.
__device__ __forceinline__ void rv_state_from_absorb(Strobe& s, const VerifyState& vs) {
    for (int i = 0; i < 25; i++) s.s[i] = vs.st[i];
}
__device__ __forceinline__ void rv_state_from_absorb(WStrobe& s, const VerifyState& vs) { s.a = s.l < 25 ? vs.st[s.l < 25 ? s.l : 0] : 0; }
template <int WAVE>
__global__ __launch_bounds__(64) void k_rv_transcript(VerifyArgs V) {
    const RangeArgs& A = V.R;
    size_t b = WAVE ? (size_t)blockIdx.x : (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B) return;
    VerifyState& vs = V.vs[b];
    const uint32_t* pr = A.out + b * A.out_words;
    bool ok = true;
    uint32_t w8[8];
    typename std::conditional<WAVE != 0, WStrobe, Strobe>::type s;
    if constexpr (WAVE != 0) {
        wstrobe_lanes(s, (int)threadIdx.x);
        __builtin_amdgcn_s_setprio(3);                           // (a latency chain beside the side stream's throughput kernels, as k_rv_absorb_V)
    }
    if (V.wave_transcript) {                                     // k_rv_absorb_V has done the head and the commitments
        rv_state_from_absorb(s, vs);
        s.pos = vs.st_pos;
        s.pos_begin = vs.st_pos_begin;
    } else {
        merlin_init(s, LBL_APP_TRANSCRIPT);
        merlin_append_bytes(s, LBL_DOM_SEP, LBL_RANGEPROOF_DOMAIN);
        merlin_append_u64(s, LBL_N, (uint64_t)A.n);
        merlin_append_u64(s, LBL_M, (uint64_t)A.m);
        for (int j = 0; j < A.m; j++) {
            ld8(w8, A.Vc + (b * A.m + j) * 8);
            merlin_append_words(s, LBL_V, w8, 8);
        }
    }
    ld8(w8, pr);      ok &= !words_zero(w8); merlin_append_words(s, LBL_A, w8, 8);      // validate_and_append_point
    ld8(w8, pr + 8);  ok &= !words_zero(w8); merlin_append_words(s, LBL_S, w8, 8);
    challenge_scalar(vs.y, s, LBL_Y);
    challenge_scalar(vs.z, s, LBL_Z);
    ld8(w8, pr + 16); ok &= !words_zero(w8); merlin_append_words(s, LBL_T1, w8, 8);
    ld8(w8, pr + 24); ok &= !words_zero(w8); merlin_append_words(s, LBL_T2, w8, 8);
    challenge_scalar(vs.x, s, LBL_X);
    uint32_t tx[8], tau[8], mu[8], aw[8], bw[8];
    ld8(tx, pr + 32); ld8(tau, pr + 40); ld8(mu, pr + 48);
    ld8(aw, pr + 56 + 16 * A.lgN); ld8(bw, pr + 64 + 16 * A.lgN);
    ok &= words_canonical_scalar(tx) & words_canonical_scalar(tau) & words_canonical_scalar(mu) & words_canonical_scalar(aw) &
          words_canonical_scalar(bw);                                                   // Scalar::from_canonical_bytes
    append_scalar(s, LBL_TX, tx);
    append_scalar(s, LBL_TX_BLINDING, tau);
    append_scalar(s, LBL_E_BLINDING, mu);
    challenge_scalar(vs.w, s, LBL_W);
    merlin_append_bytes(s, LBL_DOM_SEP, LBL_IPP_DOMAIN);
    merlin_append_u64(s, LBL_N, (uint64_t)A.N);
    // y^-1, (y-1)^-1, (z-1)^-1 and the u_k^-1 by Montgomery's trick: one inversion (~265 products) instead of lgN + 3.  The
    // running products of the u_k wait in the u_inv slots.  A zero among them (probability 2^-250) would zero every inverse,
    // so that case inverts one by one.
    sc one, ym1, zm1, p_zm1, run;
    sc_one_mont(one);
    sc_sub(ym1, vs.y, one);
    sc_sub(zm1, vs.z, one);
    sc_montmul(p_zm1, vs.y, ym1);                                // running product before z - 1
    sc_montmul(run, p_zm1, zm1);
    bool any_zero = sc_is_zero(vs.y) | sc_is_zero(ym1) | sc_is_zero(zm1);
    for (int k = 0; k < A.lgN; k++) {
        ld8(w8, pr + 56 + 16 * k);     ok &= !words_zero(w8); merlin_append_words(s, LBL_L, w8, 8);
        ld8(w8, pr + 56 + 16 * k + 8); ok &= !words_zero(w8); merlin_append_words(s, LBL_R, w8, 8);
        sc u;
        challenge_scalar(u, s, LBL_U);
        vs.u[k] = u;
        any_zero |= sc_is_zero(u);
        vs.u_inv[k] = run;                                       // y (y-1) (z-1) u_0 ... u_(k-1)
        sc_montmul(run, run, u);
    }
    if (any_zero) {
        for (int k = 0; k < A.lgN; k++) sc_invert_vartime_mont(vs.u_inv[k], vs.u[k]);
        sc_invert_vartime_mont(vs.y_inv, vs.y);
        sc_invert_vartime_mont(vs.ym1_inv, ym1);
        sc_invert_vartime_mont(vs.zm1_inv, zm1);
    } else {
        sc inv, t;
        sc_invert_vartime_mont(inv, run);
        for (int k = A.lgN - 1; k >= 0; k--) {
            sc u = vs.u[k];
            t = vs.u_inv[k];
            sc_montmul(t, t, inv);                               // (y .. u_(k-1)) / (y .. u_k) = 1 / u_k
            vs.u_inv[k] = t;
            sc_montmul(inv, inv, u);
        }
        sc_montmul(t, inv, p_zm1); vs.zm1_inv = t;  sc_montmul(inv, inv, zm1);
        sc_montmul(t, inv, vs.y);  vs.ym1_inv = t;  sc_montmul(inv, inv, ym1);
        vs.y_inv = inv;
    }
    {   // power-of-two powers for k_rv_tables
        sc p = vs.y_inv;
        for (int k = 0; k < A.lgN; k++) { vs.ypow[k] = p; sc_montsq(p, p); }
        p = vs.z;
        for (int k = 0; k < RV_MAX_LGM; k++) { vs.zpow[k] = p; sc_montsq(p, p); }
    }
    sc_to_mont(vs.t_x, tx); sc_to_mont(vs.tau, tau); sc_to_mont(vs.mu, mu); sc_to_mont(vs.a, aw); sc_to_mont(vs.b, bw);
    // Digest of everything this proof's check depends on: the transcript has absorbed n, m, every commitment and every proof
    // element except the final a, b -- append those and squeeze.  The batching scalars (k_rv_weights) are derived from it, so
    // they are bound to the data they weigh: soundness-critical, see k_rv_weights.
    {
        uint32_t wide[16];
        append_scalar(s, LBL_OWN_A, aw);
        append_scalar(s, LBL_OWN_B, bw);
        merlin_challenge_wide(s, LBL_OWN_BATCH, wide);
        for (int i = 0; i < 8; i++) vs.dg[i] = wide[i];
    }
    vs.ok = ok ? 1u : 0u;
}

// Digest of a batch of proofs: BLAKE3 tree over the per-proof digests, 31 per chunk; (level, count, group) lead every chunk.
// in: n digests of 8 words, `stride` words apart; out: ceil(n / 31) digests, packed.
__global__ __launch_bounds__(64) void k_rv_digest_level(size_t n, int level, const uint32_t* in, size_t stride, uint32_t* out) {
    size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= (n + 30) / 31) return;
    // one BLAKE3 chunk = the head and up to 31 digests, absorbed a 64-byte block (two 32-byte entries) at a time
    const uint32_t head[8] = {(uint32_t)level, (uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)g, (uint32_t)((uint64_t)g >> 32), 0x6c6f7061u, 0, 0};
    const size_t first = g * 31, cnt = (n - first < 31) ? n - first : 31;
    const int n_ent = 1 + (int)cnt, n_blk = (n_ent + 1) / 2;
    uint32_t cv[8], blk[16], o[16];
    blake3_iv(cv);
    for (int i = 0; i < n_blk; i++) {
        for (int h = 0; h < 2; h++) {
            const int e = 2 * i + h;
            for (int k = 0; k < 8; k++) blk[8 * h + k] = e == 0 ? head[k] : (e < n_ent ? in[(first + e - 1) * stride + k] : 0u);
        }
        const bool last = i == n_blk - 1;
        blake3_compress(o, cv, blk, 0, (last && (n_ent & 1)) ? 32u : 64u, (i == 0 ? B3_CHUNK_START : 0u) | (last ? (B3_CHUNK_END | B3_ROOT) : 0u));
        for (int k = 0; k < 8; k++) cv[k] = o[k];
    }
    for (int k = 0; k < 8; k++) out[g * 8 + k] = cv[k];
}
// The verifier's random scalars (lane per proof): c combines the two equations of one proof (Scalar::random(thread_rng) in
// the crate's verify_multiple), rho weighs the proof inside a cross-proof batch (not in the reference).  SOUNDNESS: a weight
// an adversary can predict can be cancelled -- proof q carries A' = A + kappa B with kappa = -rho_p c_p eps / rho_q against
// the residual eps of an out-of-range proof p, and the batch sums to the identity.  Both are therefore derived from
// key = BLAKE3(verify_seed | D), D = the digest of EVERY proof and commitment of the batch (D == nullptr: of this proof
// alone, for the per-proof check): changing any byte of any proof changes every weight (Fiat-Shamir), and with a secret
// seed (the library draws one from the OS when the caller passes none) they are unpredictable outright.
__global__ __launch_bounds__(64) void k_rv_weights(VerifyArgs V, const uint32_t* D) {
    const RangeArgs& A = V.R;
    size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= A.B) return;
    VerifyState& vs = V.vs[b];
    uint32_t w8[8], key[8], wide[16];
    Digest d;
    dg_init(d, DG_BLAKE3);
    for (int i = 0; i < 8; i++) w8[i] = A.seed[i];
    dg_update_words(d, w8, 8);
    for (int i = 0; i < 8; i++) w8[i] = D ? D[i] : vs.dg[i];
    dg_update_words(d, w8, 8);
    dg_final(d, key);
    seed_wide(wide, key, 3u, (uint64_t)b, (uint64_t)A.B);
    sc_from_wide(vs.c, wide);
    if (sc_is_zero(vs.c)) sc_one_mont(vs.c);
    seed_wide(wide, key, 5u, (uint64_t)b, (uint64_t)A.B);
    sc_from_wide(vs.rho, wide);
    if (sc_is_zero(vs.rho)) sc_one_mont(vs.rho);
}

// V2a: the power tables of every proof (lane per entry).  grid = B * ceil(tab_stride / 64) blocks.
__global__ __launch_bounds__(64) void k_rv_tables(VerifyArgs V) {
    const RangeArgs& A = V.R;
    const int bpp = (V.tab_stride + 63) >> 6;
    size_t b = blockIdx.x / bpp;
    int e = (int)(blockIdx.x % bpp) * 64 + threadIdx.x;
    const int nh = 1 << V.hb, nl = 1 << V.lb;
    if (e >= 2 * (nh + nl) + A.m) return;               // (a batch's stride also holds the product tables of k_rvb_tables2)
    const VerifyState& vs = V.vs[b];
    sc r;
    if (e < nh + nl) {                                  // products of u_k^{+-1}: MSB of i <-> first round
        bool hi = e < nh;
        int x = hi ? e : e - nh, k0 = hi ? 0 : V.hb, nbits = hi ? V.hb : V.lb;
        sc_one_mont(r);
        for (int k = 0; k < nbits; k++) {
            bool bit = (x >> (nbits - 1 - k)) & 1;
            sc_montmul(r, r, bit ? vs.u[k0 + k] : vs.u_inv[k0 + k]);
        }
    } else if (e < 2 * (nh + nl)) {                     // y^-(x 2^lb) | y^-x: product of the y^-(2^k) of the set bits
        int x = e - nh - nl;
        uint32_t ex = x < nh ? (uint32_t)x << V.lb : (uint32_t)(x - nh);
        sc_one_mont(r);
        for (int k = 0; ex; k++, ex >>= 1)
            if (ex & 1) sc_montmul(r, r, vs.ypow[k]);
    } else {                                            // z^2 z^j
        uint32_t ex = (uint32_t)(e - 2 * (nh + nl));
        r = vs.zpow[1];
        for (int k = 0; ex; k++, ex >>= 1)
            if (ex & 1) sc_montmul(r, r, vs.zpow[k]);
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
// ... its encoding alone (needs nothing from the transcript: the bucket method decodes the points while the replay runs)
__device__ __forceinline__ void rv_own_point_bytes(uint32_t* w8, const VerifyArgs& V, size_t b, int t) {
    const RangeArgs& A = V.R;
    const uint32_t* pr = A.out + b * A.out_words;
    if (t < 4) ld8(w8, pr + 8 * t);
    else if (t < 4 + A.lgN) ld8(w8, pr + 56 + 16 * (t - 4));
    else if (t < 4 + 2 * A.lgN) ld8(w8, pr + 56 + 16 * (t - 4 - A.lgN) + 8);
    else ld8(w8, A.Vc + (b * A.m + (t - 4 - 2 * A.lgN)) * 8);
}
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

// V4: the proof's own points (wave per proof): decompress, multiply, reduce.  A lane multiplies its point by signed radix-16
// double-and-add over its own table {P, 2P, ..., 8P} of cached points in LDS (laid out [entry][word][lane]: a lane reads only
// its own column, no bank conflicts, no barrier): 252 doublings + 64 + 7 additions instead of the 253 + 253 of the bit-serial
// ladder it replaces (a lane cannot skip an addition another lane of the wavefront needs).  Variable time: the scalars are public.
enum { VP_ENTRIES = 8, VP_TAB_WORDS = VP_ENTRIES * 4 * FE_NL * 64 };
__device__ __forceinline__ void vp_store(int32_t* tab, int e, int lane, const ge_cached& c) {
    int32_t* t = tab + (size_t)e * 4 * FE_NL * 64 + lane;
    for (int i = 0; i < FE_NL; i++) {
        t[(0 * FE_NL + i) * 64] = c.YpX.v[i];
        t[(1 * FE_NL + i) * 64] = c.YmX.v[i];
        t[(2 * FE_NL + i) * 64] = c.Z.v[i];
        t[(3 * FE_NL + i) * 64] = c.T2d.v[i];
    }
}
__device__ __forceinline__ void vp_load(ge_cached& c, const int32_t* tab, int e, int lane) {
    const int32_t* t = tab + (size_t)e * 4 * FE_NL * 64 + lane;
    for (int i = 0; i < FE_NL; i++) {
        c.YpX.v[i] = t[(0 * FE_NL + i) * 64];
        c.YmX.v[i] = t[(1 * FE_NL + i) * 64];
        c.Z.v[i] = t[(2 * FE_NL + i) * 64];
        c.T2d.v[i] = t[(3 * FE_NL + i) * 64];
    }
}
__device__ __forceinline__ void ge_scalarmul_vartime(ge_p3& out, const ge_p3& p, const uint32_t* k8 /* < 2^253 */, int32_t* tab, int lane) {
    ge_cached c1, c;
    ge_p3 m = p, t;
    ge_to_cached(c1, p);
    vp_store(tab, 0, lane, c1);
#pragma nounroll
    for (int e = 1; e < VP_ENTRIES; e++) {
        ge_add_cached(t, m, c1, false);
        m = t;
        ge_to_cached(c, m);
        vp_store(tab, e, lane, c);
    }
    // signed nibbles in [-8, 7] (two's complement, packed): nibble + carry >= 8 borrows 16 from the next one up
    uint32_t r8[8], carry = 0;
    for (int w = 0; w < 8; w++) {
        uint32_t word = 0;
        for (int j = 0; j < 8; j++) {
            const uint32_t nib = ((k8[w] >> (4 * j)) & 15u) + carry;
            carry = nib >= 8u ? 1u : 0u;
            word |= (nib & 15u) << (4 * j);
        }
        r8[w] = word;
    }                                                        // (k < 2^253: the top nibble is at most 2, nothing is carried out)
    ge_p3 acc;
    ge_identity(acc);
#pragma nounroll
    for (int i = 63; i >= 0; i--) {
        if (i != 63)
            for (int d = 0; d < 4; d++) { ge_dbl(t, acc, d == 3); acc = t; }
        uint32_t word = r8[0];
        for (int j = 1; j < 8; j++) word = ((i >> 3) == j) ? r8[j] : word;
        const int nib = (int)((word >> (4 * (i & 7))) & 15u), dg = nib >= 8 ? nib - 16 : nib, mag = dg < 0 ? -dg : dg;
        vp_load(c, tab, mag ? mag - 1 : 0, lane);
        ge_add_cached(t, acc, c, dg < 0);
        if (mag) acc = t;
    }
    out = acc;
}
__global__ __launch_bounds__(64) void k_rv_varpoints(VerifyArgs V) {
    const RangeArgs& A = V.R;
    size_t b = blockIdx.x;
    int l = threadIdx.x;
    const int K = 4 + 2 * A.lgN + A.m;
    __shared__ int32_t vp_tab[VP_TAB_WORDS];
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
        ge_scalarmul_vartime(q, p, k8, vp_tab, l);
        ge_p3 r;
        ge_add(r, acc, q);
        acc = r;
    }
    wave_reduce_point(acc, 64);
    if (l == 0) st_p3(A.PA + b * 40, acc);
    if (!ok) atomicAnd(&V.vs[b].ok, 0u);
}

// The same sum for calls of a few proofs, one point per FOUR lanes (ge_quad.h): 16 points per wavefront, ceil(K / 16) wavefronts
// per proof, the ladder's doublings and additions two field products deep; the table of a group's multiples holds one
// cached-form element per lane ([entry][limb][lane] in LDS, 18 KB).  Partial sums per wavefront -> k_rv_sum_parts.
__global__ __launch_bounds__(64) void k_rv_varpoints_quad(VerifyArgs V, int32_t* parts /*[B][nw][40]*/, int nw) {
    const RangeArgs& A = V.R;
    const size_t b = blockIdx.x / nw;
    const int wv = (int)(blockIdx.x % nw), l = threadIdx.x, ql = l & 3, grp = l >> 2;
    const int K = 4 + 2 * A.lgN + A.m, t = 16 * wv + grp;
    __shared__ int32_t tab[VP_ENTRIES * FE_NL * 64];
    const bool live = t < K;
    uint32_t w8[8], k8[8] = {0};
    fe c;
    bool ok = true;
    quad_identity(c, ql);
    if (live) {                                                      // (the four lanes of a group decode the same point)
        sc sm;
        rv_own_point(w8, sm, V, b, t);
        ge_p3 p;
        ok = ge_decompress(p, w8);
        sc_from_mont(k8, sm);
        for (int i = 0; i < FE_NL; i++) c.v[i] = ql == 0 ? p.X.v[i] : ql == 1 ? p.Y.v[i] : ql == 2 ? p.Z.v[i] : p.T.v[i];
    }
    // table: e * P, e = 1..8, cached form
    fe q1, q, m = c;
    quad_to_cached(q1, c, ql);
    for (int i = 0; i < FE_NL; i++) tab[(0 * FE_NL + i) * 64 + l] = q1.v[i];
#pragma nounroll
    for (int e = 1; e < VP_ENTRIES; e++) {
        quad_add_cached(m, ql, q1, false);
        quad_to_cached(q, m, ql);
        for (int i = 0; i < FE_NL; i++) tab[(e * FE_NL + i) * 64 + l] = q.v[i];
    }
    uint32_t r8[8], carry = 0;                                       // signed nibbles in [-8, 7], as in ge_scalarmul_vartime
    for (int w = 0; w < 8; w++) {
        uint32_t word = 0;
        for (int j = 0; j < 8; j++) {
            const uint32_t nib = ((k8[w] >> (4 * j)) & 15u) + carry;
            carry = nib >= 8u ? 1u : 0u;
            word |= (nib & 15u) << (4 * j);
        }
        r8[w] = word;
    }
    fe acc;
    quad_identity(acc, ql);
#pragma nounroll
    for (int i = 63; i >= 0; i--) {
        if (i != 63)
            for (int d = 0; d < 4; d++) quad_dbl(acc, ql);
        uint32_t word = r8[0];
        for (int j = 1; j < 8; j++) word = ((i >> 3) == j) ? r8[j] : word;
        const int nib = (int)((word >> (4 * (i & 7))) & 15u), dg = nib >= 8 ? nib - 16 : nib, mag = dg < 0 ? -dg : dg;
        const int e = mag ? mag - 1 : 0;
        for (int k = 0; k < FE_NL; k++) q.v[k] = tab[(e * FE_NL + k) * 64 + l];
        fe nx = acc;
        quad_add_cached(nx, ql, q, dg < 0);
        for (int k = 0; k < FE_NL; k++) acc.v[k] = mag ? nx.v[k] : acc.v[k];
    }
    for (int off = 32; off >= 4; off >>= 1) {                        // the 16 groups -> the first
        fe o;
        for (int i = 0; i < FE_NL; i++) o.v[i] = __shfl_down(acc.v[i], off, 64);
        quad_add(acc, ql, o);
    }
    if (grp == 0) {
        int32_t* dst = parts + (b * nw + wv) * 40 + FE_NL * ql;
        for (int i = 0; i < FE_NL; i++) dst[i] = acc.v[i];
    }
    if (!ok) atomicAnd(&V.vs[b].ok, 0u);
}
// PA[b] = sum of the nw partial sums of proof b (one wavefront per proof)
__global__ __launch_bounds__(64) void k_rv_sum_parts(size_t B, int nw, const int32_t* parts, int32_t* PA) {
    const size_t b = blockIdx.x;
    const int l = threadIdx.x;
    if (b >= B) return;
    ge_p3 acc;
    ge_identity(acc);
    for (int s = l; s < nw; s += 64) {
        ge_p3 p, r;
        ld_p3(p, parts + (b * nw + s) * 40);
        ge_add(r, acc, p);
        acc = r;
    }
    wave_reduce_point(acc, 64);
    if (l == 0) st_p3(PA + b * 40, acc);
}

// The scalars of B_blinding and B in one proof's check:  -mu - c tau   and   w (t_x - a b) + c (delta(y, z) - t_x).
__device__ __forceinline__ void rv_base_scalars(sc& bb, sc& bs, const VerifyState& vs, const RangeArgs& A) {
    sc one, zz, sumy, py, sumz, pz, sum2, delta, s1, s2;
    sc_one_mont(one);
    sc_montmul(zz, vs.z, vs.z);
    // sum_{i<N} y^i = (y^N - 1) / (y - 1) and sum_{j<m} z^j = (z^m - 1) / (z - 1): N and m are powers of two, so lg squarings
    // each; the inverses come from the transcript replay's batch inversion (y = 1 or z = 1 has probability 2^-252).
    sc ym1, zm1;
    py = vs.y;
    for (int i = 0; i < A.lgN; i++) sc_montmul(py, py, py);
    sc_sub(ym1, vs.y, one);
    if (sc_is_zero(ym1)) { sc_from_u64_mont(sumy, (uint64_t)A.N); }
    else { sc_sub(py, py, one); sc_montmul(sumy, py, vs.ym1_inv); }
    pz = vs.z;
    for (int mm = A.m; mm > 1; mm >>= 1) sc_montmul(pz, pz, pz);
    sc_sub(zm1, vs.z, one);
    if (sc_is_zero(zm1)) { sc_from_u64_mont(sumz, (uint64_t)A.m); }
    else { sc_sub(pz, pz, one); sc_montmul(sumz, pz, vs.zm1_inv); }
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
// Bucket method: 11-bit signed windows.  23 x 11 = 253 bits exactly, so there is no carry-only top window -- with 12 bits the
// 22nd window held nothing but the recoding carry and its bucket 1 half of all points (one 200 ms bucket).  The top window
// keeps its carry (sc_recode_w): its digit is 0..1024 (bit 252 of a canonical scalar comes with zeros below it).
enum { RVP_C = 11, RVP_NW = 23, RVP_NB = 1 << (RVP_C - 1), RVP_S = 16 };   // window bits, windows, buckets per window, lanes per bucket
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
    // Bucket method for large batches (k_rvp_*): the points as affine niels, their 11-bit signed digits, and per window the
    // points sorted by |digit| (counting sort: hist -> offs -> cursor)
    int32_t* pN;               // [npts][32]
    int16_t* pdig;             // [RVP_NW][npts]
    uint32_t* psorted;         // [RVP_NW][npts]   point index | sign << 31
    uint32_t* phist;           // [RVP_NW][RVP_NB + 1]   (then offs and cursor, same shape)
    uint32_t* poffs;
    uint32_t* pcursor;
    int32_t* pbsum;            // [RVP_NW][RVP_NB][40]   bucket sums
    // Lazy generator scalars (k_rvb_gh_lazy): the product tables once more as nine 29-bit limbs per entry (12 words), and
    // per proof group the sum of rho z
    uint32_t* tab29;           // [B][3 nh + 3 nl][12]:  SHa | PHb | QH | PL | QL | SL
    sc* rzg;                   // [G]
    int lazy;                  // 1: k_rvb_tables2 also writes tab29
};

// The product tables of the batch (layout: rv_tab_entries_rlc).  Lane per entry; grid = B * ceil(extra / 64) blocks; runs
// after k_rv_tables, whose entries it multiplies together.
__global__ __launch_bounds__(64) void k_rvb_tables2(RlcArgs R) {
    const VerifyArgs& V = R.V;
    const RangeArgs& A = V.R;
    const int nh = 1 << V.hb, nl = 1 << V.lb, base = 2 * (nh + nl) + A.m, extra = 3 * nh + 2 * nl + 1;
    const int extra2 = extra + (R.lazy ? nl : 0);                // + the limb copy of SL
    const int bpp = (extra2 + 63) >> 6;
    size_t b = blockIdx.x / bpp;
    int e = (int)(blockIdx.x % bpp) * 64 + threadIdx.x;
    if (e >= extra2) return;
    uint32_t* T29 = R.lazy ? R.tab29 + b * (size_t)(3 * nh + 3 * nl) * 12 : nullptr;
    auto put29 = [&](int slot, const sc& x) {                   // limb copy of a table entry (canonical Montgomery value < L)
        uint32_t lm[9];
        sc_split29(lm, x.v);
        uint32_t* o = T29 + (size_t)slot * 12;
        for (int i = 0; i < 9; i++) o[i] = lm[i];
    };
    if (e >= extra) {                                            // SL, limbs only
        sc x;
        if (R.V.vs[b].ok) ld_sc(x, V.tabs + b * (size_t)V.tab_stride + nh + (e - extra));
        else sc_zero(x);
        put29(3 * nh + 2 * nl + (e - extra), x);
        return;
    }
    const VerifyState& vs = V.vs[b];
    sc* T = V.tabs + b * (size_t)V.tab_stride;
    const sc *SH = T, *SL = T + nh, *YH = T + nh + nl, *YL = T + 2 * nh + nl, *ZZ = T + 2 * (nh + nl);
    const int lgn = 31 - __clz(A.n);
    int jh, ih, jl, il;
    sc r, t;
    if (!vs.ok) {                                                // a proof that did not parse takes no part in the batch sum
        sc_zero(r);
        st_sc(T + base + e, r);
        if (R.lazy && e < 3 * nh + 2 * nl) put29(e, r);
        return;
    }
    if (e < nh) {                                                // SHa
        ld_sc(t, SH + e);
        sc_montmul(r, vs.rho, vs.a);
        sc_montmul(r, r, t);
        sc_neg(r, r);
    } else if (e < 2 * nh) {                                     // PHb
        int h = e - nh;
        ld_sc(r, YH + h);
        ld_sc(t, SH + (nh - 1 - h));
        sc_montmul(r, r, t);
        sc_montmul(t, vs.rho, vs.b);
        sc_montmul(r, r, t);
        sc_neg(r, r);
    } else if (e < 3 * nh) {                                     // QH
        int h = e - 2 * nh;
        rv_q_split(lgn, V.lb, h, 0, jh, ih, jl, il);
        ld_sc(r, YH + h);
        ld_sc(t, ZZ + jh);
        sc_montmul(r, r, t);
        sc_from_u64_mont(t, 1ull << ih);
        sc_montmul(r, r, t);
        sc_montmul(r, r, vs.rho);
    } else if (e < 3 * nh + nl) {                                // PL
        int l = e - 3 * nh;
        ld_sc(r, YL + l);
        ld_sc(t, SL + (nl - 1 - l));
        sc_montmul(r, r, t);
    } else if (e < 3 * nh + 2 * nl) {                            // QL
        int l = e - 3 * nh - nl;
        rv_q_split(lgn, V.lb, 0, l, jh, ih, jl, il);
        ld_sc(r, YL + l);
        sc_pow_mont(t, vs.z, (uint32_t)jl);
        sc_montmul(r, r, t);
        sc_from_u64_mont(t, 1ull << il);
        sc_montmul(r, r, t);
    } else {                                                     // RZ
        sc_montmul(r, vs.rho, vs.z);
    }
    st_sc(T + base + e, r);
    if (R.lazy && e < 3 * nh + 2 * nl) put29(e, r);              // tab29 keeps the order SHa | PHb | QH | PL | QL
}
// Sum of rho z over the proofs of a group (one wavefront per group) -- the constant term of every generator scalar.
__global__ __launch_bounds__(64) void k_rvb_rzsum(RlcArgs R) {
    const VerifyArgs& V = R.V;
    const RangeArgs& A = V.R;
    const int g = blockIdx.x, l = threadIdx.x;
    const size_t o_rz = (size_t)2 * ((1 << V.hb) + (1 << V.lb)) + A.m + 3 * (1 << V.hb) + 2 * (1 << V.lb);
    sc acc, x;
    sc_zero(acc);
    for (size_t p = g + (size_t)l * R.G; p < A.B; p += (size_t)64 * R.G) {
        ld_sc(x, V.tabs + p * (size_t)V.tab_stride + o_rz);
        sc_add(acc, acc, x);
    }
    wave_reduce_sc(acc);
    if (l == 0) st_sc(R.rzg + g, acc);
}
// The sums of k_rvb_gh_partial with the reduction taken out of the loop: the operands wait as 29-bit limbs, a trip is the
// 81 multiply-adds of the column product (162 on the H side), and the Montgomery reduction runs once per six products
// (the columns hold six products of values below L; the reduced value stays below 2 L).  ~110 instructions per product
// instead of 263.  Same grid and position mapping as k_rvb_gh_partial (nch > 1).
__global__ __launch_bounds__(64) void k_rvb_gh_lazy(RlcArgs R) {
    const VerifyArgs& V = R.V;
    const RangeArgs& A = V.R;
    const int nch = A.TP >> 6;
    const int g = blockIdx.x / nch, ch = blockIdx.x % nch, l = threadIdx.x, side = ch & 1, q = 64 * (ch >> 1) + l;
    const int pos = 64 * (q >> 5) + (q & 31) + 32 * side;
    const int nh = 1 << V.hb, nl = 1 << V.lb;
    const int qh = q >> V.lb, ql = q & (nl - 1);
    const size_t per29 = (size_t)(3 * nh + 3 * nl) * 12;
    const size_t o_x1 = (size_t)(side == 0 ? qh : 2 * nh + qh) * 12;                                 // SHa | QH
    const size_t o_y1 = (size_t)(side == 0 ? 3 * nh + 2 * nl + ql : 3 * nh + nl + ql) * 12;          // SL  | QL
    const size_t o_x2 = (size_t)(nh + qh) * 12, o_y2 = (size_t)(3 * nh + ql) * 12;                   //     | PHb, PL
    auto ld29 = [&](uint32_t* d, const uint32_t* src) {
        const uint4* p4 = reinterpret_cast<const uint4*>(src);
        uint4 a = p4[0], b = p4[1];
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w; d[8] = src[8];
    };
    sc acc, r;
    sc_zero(acc);
    uint64_t c[18];
    for (int k = 0; k < 18; k++) c[k] = 0;
    int pending = 0;
    const int need = side ? 2 : 1;
    for (size_t p = g; p < A.B; p += R.G) {
        if (pending + need > 6) {
            sc_redc29(r, c);
            sc_add(acc, acc, r);
            for (int k = 0; k < 18; k++) c[k] = 0;
            pending = 0;
        }
        const uint32_t* T29 = R.tab29 + p * per29;
        uint32_t x[9], y[9];
        ld29(x, T29 + o_x1);
        ld29(y, T29 + o_y1);
        sc_mac29(c, x, y);
        if (side) {
            ld29(x, T29 + o_x2);
            ld29(y, T29 + o_y2);
            sc_mac29(c, x, y);
        }
        pending += need;
    }
    if (pending) { sc_redc29(r, c); sc_add(acc, acc, r); }
    sc rz;
    ld_sc(rz, R.rzg + g);
    if (side == 0) sc_sub(acc, acc, rz); else sc_add(acc, acc, rz);
    st_sc(R.partial + (size_t)g * A.TP + pos, acc);
}
// rho_p-weighted generator scalars, summed over the proofs p = g (mod G) of one group.  grid = (TP/64) * G blocks.
//   G side: rho (-z - a s_i)                        = SHa[i_hi] SL[i_lo] - RZ
//   H side: rho (z + y^-q (z^2 z^j 2^i' - b s_(N-1-q))) = QH[q_hi] QL[q_lo] + PHb[q_hi] PL[q_lo] + RZ
__global__ __launch_bounds__(64) void k_rvb_gh_partial(RlcArgs R) {
    const VerifyArgs& V = R.V;
    const RangeArgs& A = V.R;
    int nch = A.TP >> 6;
    int g = blockIdx.x / nch, ch = blockIdx.x % nch, l = threadIdx.x, side = l >> 5, q = 32 * ch + (l & 31);
    if (nch > 1) {                                               // N >= 64: a wavefront works on ONE side (64 positions of it), so
        side = ch & 1;                                           // the one-product and the two-product branch never share a wave
        q = 64 * (ch >> 1) + l;
    }
    const int pos = 64 * (q >> 5) + (q & 31) + 32 * side;
    const int nh = 1 << V.hb, nl = 1 << V.lb, base = 2 * (nh + nl) + A.m;
    const int qh = q >> V.lb, ql = q & (nl - 1);
    sc acc;
    sc_zero(acc);
    if (q < A.N) {
        // Two proofs per trip, all loads first: every proof's tables are another 15-65 KB region of HBM, so a trip is one memory
        // latency plus three products.  (A proof that failed to parse has zero tables -- k_rvb_tables2 -- and adds nothing.)
        const size_t o_rz = base + 3 * nh + 2 * nl;
        const size_t o_x1 = side == 0 ? (size_t)base + qh : (size_t)base + 2 * nh + qh;            // SHa | QH
        const size_t o_y1 = side == 0 ? (size_t)nh + ql : (size_t)base + 3 * nh + nl + ql;         // SL  | QL
        const size_t o_x2 = (size_t)base + nh + qh, o_y2 = (size_t)base + 3 * nh + ql;             //     | PHb, PL
        for (size_t p = g; p < A.B; p += 2 * (size_t)R.G) {
            const size_t pb = p + R.G;
            const bool two = pb < A.B;
            const sc* Ta = V.tabs + p * (size_t)V.tab_stride;
            const sc* Tb = V.tabs + (two ? pb : p) * (size_t)V.tab_stride;
            sc xa, ya, za, xb, yb, zb, ua, va, ub, vb, r;
            ld_sc(za, Ta + o_rz); ld_sc(xa, Ta + o_x1); ld_sc(ya, Ta + o_y1);
            ld_sc(zb, Tb + o_rz); ld_sc(xb, Tb + o_x1); ld_sc(yb, Tb + o_y1);
            if (side != 0) { ld_sc(ua, Ta + o_x2); ld_sc(va, Ta + o_y2); ld_sc(ub, Tb + o_x2); ld_sc(vb, Tb + o_y2); }
            sc_montmul(r, xa, ya);
            if (side == 0) sc_sub(r, r, za);
            else { sc_montmul(ua, ua, va); sc_add(r, r, ua); sc_add(r, r, za); }
            sc_add(acc, acc, r);
            if (two) {
                sc_montmul(r, xb, yb);
                if (side == 0) sc_sub(r, r, zb);
                else { sc_montmul(ub, ub, vb); sc_add(r, r, ub); sc_add(r, r, zb); }
                sc_add(acc, acc, r);
            }
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
// acc (lane 0's point) <- 2^nd acc, with the wavefront's first four lanes doing every doubling TOGETHER (ge_quad.h: lane k holds
// coordinate k; a doubling is one squaring + one product deep instead of 3 + 4 on a lone lane).  The window sums of the bucket method
// and of the doubling-free generator MSM end in such chains -- up to 252 doublings on a lone lane of a lone wavefront, the serial
// tail of a verification pass.  Every lane of the wavefront must call this (the other groups double copies; nobody reads them).
__device__ __forceinline__ void wave_dbl_chain(ge_p3& acc, int nd, int l) {
    if (nd <= 0) return;
    const int ql = l & 3;
    fe c;
    for (int i = 0; i < FE_NL; i++) {
        const int x = __shfl(acc.X.v[i], 0, 64), y = __shfl(acc.Y.v[i], 0, 64), z = __shfl(acc.Z.v[i], 0, 64), t = __shfl(acc.T.v[i], 0, 64);
        c.v[i] = ql == 0 ? x : ql == 1 ? y : ql == 2 ? z : t;
    }
#pragma nounroll
    for (int i = 0; i < nd; i++) quad_dbl(c, ql);
    for (int i = 0; i < FE_NL; i++) {
        acc.X.v[i] = __shfl(c.v[i], 0, 64); acc.Y.v[i] = __shfl(c.v[i], 1, 64);
        acc.Z.v[i] = __shfl(c.v[i], 2, 64); acc.T.v[i] = __shfl(c.v[i], 3, 64);
    }
}
// The ONE generator MSM of a large batch without a doubling in its loop (round 5).  As k_rp_msm<0, .> with B = 1 it was 1,024
// wavefronts walking 22 window steps of 12 shared doublings each for ~44 additions per lane: 252 doublings per lane, six times the
// additions (0.97 ms, 1.8 ms beside the own-point branch).  Here a wavefront owns (window w, slice s of the 2N terms): additions only
// (k_rvb_gen_sweep), the slices of a window are summed and scaled by 2^(W w) by one wavefront per window (k_rvb_gen_windows), and
// k_rvb_finish adds the nwin window points like any other partial sums.  The doublings that remain are ONE chain of W (nwin - 1) on
// a lone lane per window -- latency that runs beside the own-point branch, not work.  Same digits, same rows, same point.
__global__ __launch_bounds__(64) void k_rvb_gen_sweep(RangeArgs A, TableView tbl, int NS, int32_t* part /* [nwin][NS][40] */) {
    const int w = blockIdx.x / NS, s = blockIdx.x % NS, l = threadIdx.x;
    const dig_t* dw = A.dig + (size_t)w * A.TP;
    ge_p3 acc;
    ge_identity(acc);
#pragma nounroll
    for (int t = s * 64 + l; t < 2 * A.N; t += NS * 64) {
        const int side = t >= A.N ? 1 : 0, q = t - side * A.N;
        const int d = dw[64 * (q >> 5) + (q & 31) + 32 * side];
        bool isH;
        const int j = term_generator(-1, A.N, A.lgN, side, q, isH);
        tbl_madd(acc, tbl, gen_row(tbl, A.n, j, isH), d);
    }
    wave_reduce_point(acc, 64);
    if (l == 0) st_p3(part + ((size_t)w * NS + s) * 40, acc);
}
__global__ __launch_bounds__(64) void k_rvb_gen_windows(RangeArgs A, int NS, const int32_t* part) {
    const int w = blockIdx.x, l = threadIdx.x;
    ge_p3 acc, p, t;
    ge_identity(acc);
    for (int i = l; i < NS; i += 64) { ld_p3(p, part + ((size_t)w * NS + i) * 40); ge_add(t, acc, p); acc = t; }
    wave_reduce_point(acc, 64);
    wave_dbl_chain(acc, A.wbits * w, l);
    if (l == 0) {
        st_p3(A.P0 + (size_t)w * 40, acc);
        ge_identity(t);
        st_p3(A.P1 + (size_t)w * 40, t);
    }
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
// ------------------------------------------------------------------------------------------------------------------
// The batch's own points by the bucket method.  With per-point tables (k_rvb_points + Straus) a point costs ~80 additions:
// 8 for its table, the normalisation, and one per 4-bit window.  A batch is ONE sum over all its points -- 1.08 M of them
// for 1,024 proofs x 1,024 parties, 3 M for 65,536 inclusion proofs -- which is where Pippenger's method pays: per
// 11-bit window every point is added once into the bucket of its digit (23 additions per point in total), and the 1,024
// buckets of a window are combined with running sums.  Steps: decode (early, second stream) -> digits + histogram -> offsets
// (wave per window) -> scatter (counting sort by |digit|) -> bucket sums (8 lanes per bucket, tree-reduced) -> window sums
// (wave per window: sum_k k B_k by running sums over 16-bucket segments, then 2^(11 w)).  The order inside a bucket is
// whatever the atomics give; the sum -- and so the verdict -- does not depend on it.
// Decode (lane per point).  Independent of the challenges, so it runs on a second stream beside the transcript replay, whose
// lane- or wavefront-per-proof kernels leave most of the chip idle.  Word 30 of the entry records a failed decode.
__global__ __launch_bounds__(64) void k_rvp_decode(RlcArgs R) {
    size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= R.npts) return;
    size_t p = t / R.K;
    uint32_t w8[8];
    rv_own_point_bytes(w8, R.V, p, (int)(t - p * R.K));
    ge_p3 pt;
    bool good = ge_decompress(pt, w8);
    if (!good) ge_identity(pt);
    fe a, b2, c2;                                                 // affine niels (the decoded point has Z = 1)
    fe_addc(a, pt.Y, pt.X);
    fe_sub(b2, pt.Y, pt.X);
    fe_carry(b2, b2);
    fe_mul(c2, pt.T, FE_D2);
    int32_t* o = R.pN + t * 32;
    ge_niels qn;
    qn.ypx = a; qn.ymx = b2; qn.xy2d = c2;
    niels_store_entry(o, qn);
    o[30] = good ? 0 : 1;
}
// Scalars -> digits + histogram, after the replay.  A point that did not decode sends the chunk to the proof-by-proof check if its
// proof is still in the running.
// Round 5: the 25 M increments of a 1,024 x 1,024-party batch used to be 25 M device-scope atomics -- which on this chip go past
// the (per-XCD, mutually incoherent) L2s to the memory side, 28 bytes of fabric traffic each, to 23,575 counters that are hit a
// thousand times each (706 MB written for 50 MB of digits, profiles/r07k_verify_pmc.txt).  A block now owns a contiguous run of
// RVP_RUN points and counts them in LDS (23 x 1,025 counters of 16 bits, two to a word: a run holds 4,096 points, 46 KB); what
// reaches memory is one add per non-empty counter and block.
enum { RVP_BLOCK = 256, RVP_RUN = 4096, RVP_HW = (RVP_NW * (RVP_NB + 1) + 1) / 2 };   // threads / points per block (digits and scatter kernels), LDS words
static_assert(RVP_RUN < 65536, "16-bit run counters");
__global__ __launch_bounds__(RVP_BLOCK) void k_rvp_digits(RlcArgs R) {
    __shared__ uint32_t hist[RVP_HW];
    for (int i = threadIdx.x; i < RVP_HW; i += RVP_BLOCK) hist[i] = 0;
    __syncthreads();
    const size_t np = R.npts, t0 = (size_t)blockIdx.x * RVP_RUN, t1 = t0 + RVP_RUN < np ? t0 + RVP_RUN : np;
    for (size_t t = t0 + threadIdx.x; t < t1; t += RVP_BLOCK) {
        size_t p = t / R.K;
        sc sm;
        sc_zero(sm);
        if (R.V.vs[p].ok) {
            if (R.pN[t * 32 + 30]) atomicOr(&R.flag[1], 1u);
            else {
                uint32_t w8[8];
                rv_own_point(w8, sm, R.V, p, (int)(t - p * R.K));
                sc_montmul(sm, sm, R.V.vs[p].rho);
            }
        }
        uint32_t c[8];
        sc_from_mont(c, sm);
        sc_recode_w(RVP_C, RVP_NW, c, [&](int i, int digit) {
            R.pdig[(size_t)i * np + t] = (int16_t)digit;
            if (digit) {
                const int k = i * (RVP_NB + 1) + (digit < 0 ? -digit : digit);
                atomicAdd(&hist[k >> 1], 1u << (16 * (k & 1)));
            }
        });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RVP_NW * (RVP_NB + 1); i += RVP_BLOCK) {
        const uint32_t c = (hist[i >> 1] >> (16 * (i & 1))) & 0xffffu;
        if (c) atomicAdd(&R.phist[i], c);
    }
}
// offs[w][d] = number of points of window w with 0 < |digit| < d; cursor = offs.  One wavefront per window.
__global__ __launch_bounds__(64) void k_rvp_scan(RlcArgs R) {
    __shared__ uint32_t tot[64];
    const int w = blockIdx.x, l = threadIdx.x, per = (RVP_NB + 1 + 63) / 64;
    const uint32_t* h = R.phist + (size_t)w * (RVP_NB + 1);
    int lo = l * per, hi = lo + per < RVP_NB + 1 ? lo + per : RVP_NB + 1;
    uint32_t sum = 0;
    for (int i = lo; i < hi; i++) sum += i ? h[i] : 0u;
    tot[l] = sum;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < l; i++) base += tot[i];
    for (int i = lo; i < hi; i++) {
        R.poffs[(size_t)w * (RVP_NB + 1) + i] = base;
        R.pcursor[(size_t)w * (RVP_NB + 1) + i] = base;
        base += i ? h[i] : 0u;
    }
}
// Counting-sort scatter.  Round 5, two changes against 25 M returning device-scope atomics + 25 M stray 4-byte stores (1.25 GB
// written for a 100 MB array: a line of psorted was touched from all eight XCDs, whose L2s cannot merge each other's bytes):
//  * a block owns (a run of RVP_RUN points) x (the windows of ONE XCD: w = xcd, xcd + 8, ...; blockIdx.x % 8 is the XCD a block lands
//    on) -- every line of a window's slice of psorted is then written from one L2 only, which merges the 4-byte stores into lines;
//  * per window the block counts its run in LDS first, reserves its share of every bucket with ONE returning add per non-empty
//    bucket, and ranks its points inside that share with LDS atomics.
// The order inside a bucket is whatever the atomics give; the bucket's sum does not depend on it.
__global__ __launch_bounds__(RVP_BLOCK) void k_rvp_scatter(RlcArgs R) {
    __shared__ uint32_t cnt[RVP_NB + 1];
    const int xcd = blockIdx.x & 7;
    const size_t np = R.npts, t0 = (size_t)(blockIdx.x >> 3) * RVP_RUN, t1 = t0 + RVP_RUN < np ? t0 + RVP_RUN : np;
    for (int w = xcd; w < RVP_NW; w += 8) {
        for (int i = threadIdx.x; i <= RVP_NB; i += RVP_BLOCK) cnt[i] = 0;
        __syncthreads();
        const int16_t* dg = R.pdig + (size_t)w * np;
        for (size_t t = t0 + threadIdx.x; t < t1; t += RVP_BLOCK) {
            const int d = dg[t];
            if (d) atomicAdd(&cnt[d < 0 ? -d : d], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i <= RVP_NB; i += RVP_BLOCK) {   // cnt[d] := where this block's share of bucket d begins
            const uint32_t c = cnt[i];
            cnt[i] = c ? atomicAdd(&R.pcursor[(size_t)w * (RVP_NB + 1) + i], c) : 0u;
        }
        __syncthreads();
        uint32_t* out = R.psorted + (size_t)w * np;
        for (size_t t = t0 + threadIdx.x; t < t1; t += RVP_BLOCK) {
            const int d = dg[t];
            if (!d) continue;
            const uint32_t pos = atomicAdd(&cnt[d < 0 ? -d : d], 1u);
            out[pos] = (uint32_t)t | (d < 0 ? 0x80000000u : 0u);
        }
        __syncthreads();
    }
}
// Bucket sums: RVP_S adjacent lanes share a bucket (every RVP_S-th point each), then a tree reduction.  grid = NW * NB * S / 64.
__global__ __launch_bounds__(64) void k_rvp_buckets(RlcArgs R) {
    const int l = threadIdx.x;
    const size_t gid = (size_t)blockIdx.x * 64 + l;
    const int s = (int)(gid % RVP_S), bidx = (int)((gid / RVP_S) % RVP_NB), w = (int)(gid / ((size_t)RVP_S * RVP_NB));
    const uint32_t start = R.poffs[(size_t)w * (RVP_NB + 1) + bidx + 1], n = R.phist[(size_t)w * (RVP_NB + 1) + bidx + 1];
    const uint32_t* srt = R.psorted + (size_t)w * R.npts + start;
    ge_p3 acc;
    ge_identity(acc);
    for (uint32_t i = s; i < n; i += RVP_S) {
        uint32_t e = srt[i];
        ge_niels q;
        niels_load_entry(q, R.pN + (size_t)(e & 0x7fffffffu) * 32);
        ge_madd(acc, acc, q, (e >> 31) != 0);
    }
    wave_reduce_point(acc, RVP_S);
    if (s == 0) st_p3(R.pbsum + ((size_t)w * RVP_NB + bidx) * 40, acc);
}
// Window sums: W_w = sum_k k B_k, scaled by 2^(C w), into Q0[w] (Q1[w] = identity) for k_rvb_finish.  One wavefront per
// window: lane l owns the L = NB / 64 buckets of weights l L + 1 .. l L + L.
__global__ __launch_bounds__(64) void k_rvp_window(RlcArgs R) {
    const int w = blockIdx.x, l = threadIdx.x, L = RVP_NB / 64;
    ge_p3 run, aseg, b, t;
    ge_identity(run);
    ge_identity(aseg);
    for (int k = L - 1; k >= 0; k--) {                            // running sums from the segment's heaviest bucket down:
        ld_p3(b, R.pbsum + ((size_t)w * RVP_NB + (size_t)l * L + k) * 40);
        ge_add(t, run, b); run = t;                               // run  = B_(lL+k) + .. + B_(lL+L-1)
        ge_add(t, aseg, run); aseg = t;                           // aseg = sum_k (k + 1) B_(lL+k) when the loop ends
    }
    // W = sum_l aseg_l + L * sum_l l * C_l (C_l = run): the second sum bit by bit of l, most significant first
    ge_p3 U;
    ge_identity(U);
    for (int j = 5; j >= 0; j--) {
        ge_p3 m;
        if ((l >> j) & 1) m = run; else ge_identity(m);
        wave_reduce_point(m, 64);
        if (l == 0) { ge_dbl(t, U, true); ge_add(U, t, m); }
    }
    wave_reduce_point(aseg, 64);
    if (l == 0) {
        for (int i = 1; i < L; i <<= 1) { ge_dbl(t, U, 2 * i >= L); U = t; }  // times L
        ge_add(t, aseg, U);
    }
    wave_dbl_chain(t, RVP_C * w, l);                              // times 2^(C w): the four-lane doubling chain (lane 0's point)
    if (l == 0) {
        st_p3(R.Q0 + (size_t)w * 40, t);
        ge_identity(t);
        st_p3(R.Q1 + (size_t)w * 40, t);
    }
}
// Sums everything and tests for the identity (one wavefront).
__global__ __launch_bounds__(64) void k_rvb_finish(RlcArgs R, TableView tbl) {
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
    wave_reduce_point(acc, 64);
    sc bb, bs, x;
    sc_zero(bb); sc_zero(bs);
    for (size_t i = l; i < A.B; i += 64) {
        ld_sc(x, R.bsum + 2 * i); sc_add(bb, bb, x);
        ld_sc(x, R.bsum + 2 * i + 1); sc_add(bs, bs, x);
    }
    wave_reduce_sc(bb);
    wave_reduce_sc(bs);
    // bb * B_blinding + bs * B: the low half-wavefront looks bb's windows up in B_blinding's per-window rows, the high half bs's in
    // B's (tbl_fixed_mul_wave: one lookup and a five-deep shuffle tree per product, nwin <= 32 as W >= WBITS_MIN = 8) -- instead of
    // 2 nwin dependent additions on lane 0.
    {
        sc sel;
        for (int i = 0; i < 8; i++) {
            const uint32_t b0 = (uint32_t)__shfl((int)bb.v[i], 0, 64), s0 = (uint32_t)__shfl((int)bs.v[i], 0, 64);
            sel.v[i] = l < 32 ? b0 : s0;
        }
        uint32_t k8[8];
        sc_from_mont(k8, sel);
        ge_p3 term, other;
        tbl_fixed_mul_wave(term, tbl, l < 32 ? tbl.row_Bb(0) : tbl.row_B(0), k8, l & 31);      // valid on lanes 0 and 32
        for (int i = 0; i < FE_NL; i++) {
            other.X.v[i] = __shfl(term.X.v[i], 32, 64); other.Y.v[i] = __shfl(term.Y.v[i], 32, 64);
            other.Z.v[i] = __shfl(term.Z.v[i], 32, 64); other.T.v[i] = __shfl(term.T.v[i], 32, 64);
        }
        if (l == 0) { ge_add(t, acc, term); ge_add(acc, t, other); }
    }
    if (l == 0) {
        uint32_t c8[8];
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
// HW = words of a node hash (8: the 32-byte digests; 16: Blake2b-512 -- hash arrays then hold 16 words per node).
template <int HW>
__device__ __forceinline__ void ldhw(uint32_t* w, const uint32_t* p) { ld8(w, p); if (HW == 16) ld8(w + 8, p + 8); }
template <int HW>
__global__ __launch_bounds__(64) void k_verify_paths(int dg, size_t b, int height, const uint64_t* leaf_idx, const uint32_t* leafC, const uint32_t* leafH,
                                                    const uint32_t* pC, const uint32_t* pH, const uint32_t* rootC, const uint32_t* rootH,
                                                    int leaf_first, uint8_t* ok) {
    size_t e = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (e >= b) return;
    uint32_t c[8], h[HW], sc_[8], sh[HW], hn[HW];
    ld8(c, leafC + e * 8);
    ldhw<HW>(h, leafH + e * HW);
    ge_p3 acc, sp;
    bool good = ge_decompress(acc, c);
    uint64_t idx = leaf_idx[e];
    for (int k = 0; k < height; k++) {
        size_t slot = e * (size_t)height + (size_t)(leaf_first ? k : height - 1 - k);
        ld8(sc_, pC + slot * 8);
        ldhw<HW>(sh, pH + slot * HW);
        good &= ge_decompress(sp, sc_);                      // deserialisation rejects non-canonical points (proof/node.rs:88-94)
        if ((idx >> k) & 1) node_hash_parent_w<HW>(dg, hn, sc_, c, sh, h);
        else node_hash_parent_w<HW>(dg, hn, c, sc_, h, sh);
        ge_p3 t;
        ge_add(t, acc, sp);
        acc = t;
        ge_compress(c, acc);
        for (int i = 0; i < HW; i++) h[i] = hn[i];
    }
    for (int i = 0; i < 8; i++) good &= (c[i] == rootC[i]);
    for (int i = 0; i < HW; i++) good &= (h[i] == rootH[i]);
    ok[e] = good ? 1 : 0;
}
// commitments of one sub-proof from the path (pad parties: commit(0, 1) = B_blinding, src/range/padding.rs:176-180)
// The same check with ONE WAVEFRONT per proof, for calls of few proofs (a user checking their own inclusion proof): the lane
// version above is a chain of `height` decompressions, additions and encodings -- 4.9 ms for one height-32 path.  Here lane k
// decompresses sibling k, an inclusive prefix sum over the lanes (shuffles) gives every ancestor's commitment at once, each lane
// encodes its own, and only the hash chain stays serial (every lane walks it in step; its inputs come by shuffle).
template <int HW>
__global__ __launch_bounds__(64) void k_verify_paths_wave(int dg, size_t b, int height, const uint64_t* leaf_idx, const uint32_t* leafC,
                                                         const uint32_t* leafH, const uint32_t* pC, const uint32_t* pH, const uint32_t* rootC,
                                                         const uint32_t* rootH, int leaf_first, uint8_t* ok) {
    const size_t e = blockIdx.x;
    const int l = threadIdx.x;
    if (e >= b) return;
    uint32_t c[8], h[HW], sc_[8] = {0}, sh[HW] = {0}, cn[8];
    ld8(c, leafC + e * 8);
    ldhw<HW>(h, leafH + e * HW);
    const bool live = l < height;
    ge_p3 acc, lf;
    bool good = ge_decompress(lf, c);
    ge_identity(acc);
    if (live) {
        size_t slot = e * (size_t)height + (size_t)(leaf_first ? l : height - 1 - l);
        ld8(sc_, pC + slot * 8);
        ldhw<HW>(sh, pH + slot * HW);
        good &= ge_decompress(acc, sc_);                     // deserialisation rejects non-canonical points (proof/node.rs:88-94)
    }
    {
        ge_p3 r;
        ge_add(r, acc, lf);
        if (l == 0) acc = r;                                 // lane k's prefix sum = leaf + siblings 0..k = the ancestor at level k + 1
    }
    for (int off = 1; off < height; off <<= 1) {
        ge_p3 o, r;
        for (int i = 0; i < FE_NL; i++) {
            o.X.v[i] = __shfl_up(acc.X.v[i], off, 64);
            o.Y.v[i] = __shfl_up(acc.Y.v[i], off, 64);
            o.Z.v[i] = __shfl_up(acc.Z.v[i], off, 64);
            o.T.v[i] = __shfl_up(acc.T.v[i], off, 64);
        }
        ge_add(r, acc, o);
        if (l >= off) acc = r;
    }
    ge_compress(cn, acc);
    const uint64_t idx = leaf_idx[e];
    for (int k = 0; k < height; k++) {
        uint32_t sk[8], shk[HW], nx[8], hn[HW];
        for (int i = 0; i < 8; i++) {
            sk[i] = (uint32_t)__shfl((int)sc_[i], k, 64);
            nx[i] = (uint32_t)__shfl((int)cn[i], k, 64);
        }
        for (int i = 0; i < HW; i++) shk[i] = (uint32_t)__shfl((int)sh[i], k, 64);
        if ((idx >> k) & 1) node_hash_parent_w<HW>(dg, hn, sk, c, shk, h);
        else node_hash_parent_w<HW>(dg, hn, c, sk, h, shk);
        for (int i = 0; i < 8; i++) c[i] = nx[i];
        for (int i = 0; i < HW; i++) h[i] = hn[i];
    }
    for (int i = 0; i < 8; i++) good &= (c[i] == rootC[i]);
    for (int i = 0; i < HW; i++) good &= (h[i] == rootH[i]);
    const bool all_good = __all(good ? 1 : 0) != 0;
    if (l == 0) ok[e] = all_good ? 1 : 0;
}
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
// Grouped sub-proofs (round 6, host_verify.inc: verify_policy_device): proof g = e * k + j of a group takes the commitments
// start + j * m ... of its entity (all m of them real: a group's sub-proofs are full), its words are gathered from the entity's
// blob into a contiguous batch, and the entity's verdict is the AND over its k sub-proofs.
__global__ void k_gather_commitments_grouped(size_t b, int height, int start, int m, int k, const uint32_t* pC, uint32_t* Vc) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * (size_t)k * (size_t)m) return;
    const size_t g = t / m, e = g / k;
    const int jj = (int)(t - g * m), j = (int)(g - e * k);
    uint32_t c[8];
    ld8(c, pC + (e * (size_t)height + (size_t)(start + j * m + jj)) * 8);
    st8(Vc + t * 8, c);
}
__global__ void k_gather_words_grouped(size_t b, int k, size_t pw, size_t entity_words, size_t word_off, const uint32_t* in, uint32_t* out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * (size_t)k * pw) return;
    const size_t g = t / pw, w = t - g * pw, e = g / k, j = g - e * k;
    out[t] = in[e * entity_words + word_off + j * pw + w];
}
__global__ void k_and_bytes_grouped(size_t b, int k, uint8_t* acc, const uint8_t* x) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b) return;
    uint8_t a = acc[e];
    for (int j = 0; j < k; j++) a &= x[e * (size_t)k + j];
    acc[e] = a;
}
__global__ void k_and_bytes(size_t n, uint8_t* acc, const uint8_t* x) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = acc[i] & x[i];
}

}  // namespace dapol
