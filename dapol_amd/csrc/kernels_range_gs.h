// Generator-stationary form of the prover's fixed-base MSM (large calls).  Same group elements as k_rp_msm, other order:
//
//   k_rp_msm (proof-stationary): a lane owns a slice of ONE proof's list and walks window-outer / term-inner; every lookup of
//   every lane is a random 128-byte line somewhere in the 35 GB of tables -- an HBM gather.
//
//   here: a lane owns ONE accumulator -- (proof p, window w) of the list being swept -- and the whole chip sweeps the list's
//   generators in the same order, a tile of rows per launch.  All lanes of all wavefronts of a launch read the SAME few rows
//   (2^(W-1)+1 lines of 128 B each: 8.4 MB at 17 bits), cb * nwin lookups per row, so the tile sits in the Infinity Cache
//   (256 MB) and a row comes from HBM once per MSM instead of once per lookup.  No doubling in the loop: the nwin window sums
//   of a list are combined afterwards (k_rp_gs_combine: W (nwin-1) doublings + nwin-1 additions per list, 0.8 % of the
//   additions of a 2,048-term list).  The accumulators are carried between the tiles in HBM (coalesced 36-word SoA records:
//   288 B of traffic per lane and tile against 128 B per lookup).
//   Measured before it was built into the library (tools/ubench_msm_order.hip, profiles/r03_msm_order_ubench.txt): 2,118 ns
//   of SIMD time per wavefront-addition in 32-row tiles against 2,445 for the proof-stationary order (-13 %) -- the chip is
//   held at its power cap either way, and the sweep spends less of the budget on HBM (shader clock 2.1 GHz against 1.72).
//
// Digit layout of this form (written by the *_gs producers below): sweep position sp = side * N + q (q = term of the list,
// term_generator's q) -- four consecutive positions of one accumulator lane are one 16-byte element:
//     dig4[(sp >> 2) * L + lane][sp & 3],   L = cb * nwin,   lane = w * cb + p
// so a wavefront's digit load is 1 KB contiguous and so is every store of a producer (lanes run over the proofs).
#pragma once
#include "kernels_range.h"

namespace dapol {

// Signed radix-2^W digits of ONE canonical scalar (the term at sweep position sp of proof p): component sp & 3 of the quad's
// 16-byte element, one 4-byte store per window.  Same recoding as sc_recode_w; the scalar is shifted down a window at a time
// (static register indexes only).
__device__ __forceinline__ void write_digits_gs(const RangeArgs& A, size_t p, int sp, const uint32_t* c) {
    const size_t L = A.B * (size_t)A.nwin;
    int32_t* out = reinterpret_cast<int32_t*>(A.dig) + ((((size_t)(sp >> 2)) * L + p) << 2) + (sp & 3);
    const size_t wstride = A.B * (size_t)4;
    const int W = A.wbits, NW = A.nwin, half = 1 << (W - 1);
    const uint32_t mask = (1u << W) - 1;
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = c[i];
    int carry = 0;
#pragma nounroll
    for (int w = 0; w < NW; w++) {
        const int b = (int)(x[0] & mask) + carry;
        carry = (b >= half && w != NW - 1) ? 1 : 0;
        out[(size_t)w * wstride] = b - (carry << W);
#pragma unroll
        for (int i = 0; i < 7; i++) x[i] = (x[i] >> W) | (x[i + 1] << (32 - W));
        x[7] >>= W;
    }
}

// Thread -> (proof p, sweep position sp).  The four positions of a quad run fastest: four adjacent lanes read the four scalars
// of ONE 128-byte line of a proof's vector (the line is fetched once, by one pair of load instructions) and write the four
// components of ONE 16-byte digit element; then the proofs, so that a wavefront's store is 256 contiguous bytes per window.
// grid = ceil(cb * 2N / 64) blocks of 64.
// The producers walk their blocks in a GRID-STRIDE loop (GS_BLOCKS): launched with as many blocks as there are, each does one;
// launched with fewer (host_range.inc, DAPOL_PRODUCER_WAVES: an experiment that caps how many wavefront slots an HBM-bound
// producer may hold beside another chunk's VALU-bound sweep), each walks several.
#define GS_BLOCKS(A, vb) for (size_t vb = blockIdx.x, nb_ = ((size_t)(A).B * (size_t)(2 * (A).N) + 63) / 64; vb < nb_; vb += gridDim.x)
__device__ __forceinline__ bool gs_thread(const RangeArgs& A, size_t vblock, size_t& p, int& sp) {
    const size_t t = vblock * 64 + threadIdx.x;
    const size_t quad = t >> 2, sp4 = quad / A.B;
    p = quad - sp4 * A.B;
    sp = (int)(4 * sp4) + (int)(t & 3);
    return sp4 < (size_t)(2 * A.N) / 4;
}

// K0 (generator-stationary layout): nonces s_L, s_R + the S commitment's digits
__global__ __launch_bounds__(64) void k_rp_nonces_gs(RangeArgs A) {
    GS_BLOCKS(A, vb) {
        size_t p;
        int sp;
        if (!gs_thread(A, vb, p, sp)) continue;
        const int side = sp >= A.N ? 1 : 0, q = sp - side * A.N;
        const int j = q / A.n, ii = q - j * A.n;
        const uint32_t slot = (uint32_t)(j * (2 * A.n + 2) + 2 + ii + (side ? A.n : 0));
        sc s;
        tape_scalar(s, A, p, slot);
        st_sc((side ? A.s2 : A.s1) + p * A.N + q, s);
        uint32_t c[8];
        sc_from_mont(c, s);
        write_digits_gs(A, p, sp, c);
    }
}

// K5 (generator-stationary layout): round-k MSM scalars -> digits
__global__ __launch_bounds__(64) void k_rp_round_prep_gs(RangeArgs A, int round) {
    GS_BLOCKS(A, vb) {
        size_t p;
        int sp;
        if (!gs_thread(A, vb, p, sp)) continue;
        const int side = sp >= A.N ? 1 : 0, q = sp - side * A.N;
        const int lgh = A.lgN - 1 - round, half = 1 << lgh;
        bool isH;
        const int j = term_generator(round, A.N, A.lgN, side, q, isH);
        const int off = j & (half - 1);
        const bool upper = (j >> lgh) & 1;
        const int vi = upper ? off : off + half;           // G_R pairs with a_L, G_L with a_R; H'_L with b_R, H'_R with b_L
        sc v, pr;
        ld_sc(v, (isH ? A.b : A.a) + p * A.N + vi);
        coeff_times(pr, A, p, round, j, isH, v);           // the canonical product a_i * s_j (coefficient tables or vectors)
        write_digits_gs(A, p, sp, pr.v);
    }
}

// The sweep: rows q0 .. q0 + nq - 1 (terms of list `side`, nq a multiple of 4) added into every accumulator lane.
// first: the accumulators start at the identity; else they are loaded from acc (SoA: word k of lane l at acc[k * Ltot + l]).
// SLICES (calls of a few thousand proofs, whose B * nwin lanes cannot fill the chip): gridDim.y slices of a list, slice s sweeping
// rows s * slice_rows + q0 ... into its own accumulators (lane s * L + l of Ltot = L * gridDim.y); k_rp_gs_sum_slices adds them up.
#ifndef DAPOL_GS_OCC
#define DAPOL_GS_OCC 4
#endif
__global__ __launch_bounds__(64, DAPOL_GS_OCC) void k_rp_msm_gs(RangeArgs A, TableView tbl, int round, int side, int q0, int nq, int first, int32_t* __restrict__ accs, int slice_rows) {
    const size_t L = A.B * (size_t)A.nwin, Ltot = L * gridDim.y;
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (lane >= L) return;
    q0 += (int)blockIdx.y * slice_rows;
    ge_p3 acc;
    int32_t* ap = accs + (size_t)blockIdx.y * L + lane;
    if (first) ge_identity(acc);
    else {
#pragma unroll
        for (int i = 0; i < FE_NL; i++) {
            acc.X.v[i] = ap[(size_t)i * Ltot]; acc.Y.v[i] = ap[(size_t)(FE_NL + i) * Ltot];
            acc.Z.v[i] = ap[(size_t)(2 * FE_NL + i) * Ltot]; acc.T.v[i] = ap[(size_t)(3 * FE_NL + i) * Ltot];
        }
    }
    const dapol_v4i* dg = reinterpret_cast<const dapol_v4i*>(A.dig) + ((size_t)(side * A.N + q0) >> 2) * L + lane;
    // The digits of the NEXT four rows are requested while the current four are being added: the digit load (a cold, streamed
    // 16 bytes) is then never what a lane's table lookup waits for.
    dapol_v4i d4 = dg[0], dn = d4;
#pragma nounroll
    for (int i = 0; i < nq; i++) {
        if ((i & 3) == 0 && i + 4 < nq) dn = dg[(size_t)((i >> 2) + 1) * L];
        const int d = d4.x;
        d4.x = d4.y; d4.y = d4.z; d4.z = d4.w;
        if ((i & 3) == 3) d4 = dn;
        bool isH;
        const int j = term_generator(round, A.N, A.lgN, side, q0 + i, isH);      // (uniform over a block)
        tbl_madd(acc, tbl, gen_row(tbl, A.n, j, isH), d);
    }
#pragma unroll
    for (int i = 0; i < FE_NL; i++) {
        ap[(size_t)i * Ltot] = acc.X.v[i]; ap[(size_t)(FE_NL + i) * Ltot] = acc.Y.v[i];
        ap[(size_t)(2 * FE_NL + i) * Ltot] = acc.Z.v[i]; ap[(size_t)(3 * FE_NL + i) * Ltot] = acc.T.v[i];
    }
}

// The sweep of SHORT lists (round 6: large batches of proofs of few parties -- a policy's individual proofs, 64 generators a side).
// With so few terms per list the Horner combine is no longer small change (252 addition-equivalents against 64 x 15 additions), so
// the lane owns (proof, window w < LW = TableView::hi_split) and looks every term up TWICE -- digit w in the generator's row, digit
// w + LW in its 2^(W LW) row, as k_rp_msm does for small calls -- which halves the window sums to combine (119 doublings + 7
// additions per list).  Same digit matrix (L_dig = B * nwin lanes), accumulators SoA over Ltot = B * LW * gridDim.y lanes.  A copy
// of the kernel above rather than a flag in it: that one is the headline's and stays as measured.
__global__ __launch_bounds__(64, DAPOL_GS_OCC) void k_rp_msm_gs_hi(RangeArgs A, TableView tbl, int round, int side, int q0, int nq, int first, int32_t* __restrict__ accs,
                                                                  int slice_rows, int LW) {
    const size_t L = A.B * (size_t)LW, Ltot = L * gridDim.y, Ld = A.B * (size_t)A.nwin;
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (lane >= L) return;
    q0 += (int)blockIdx.y * slice_rows;
    const int w = (int)(lane / A.B);
    const bool has_hi = w + LW < A.nwin;                                     // (15 windows, LW = 8: window 7 has no partner)
    ge_p3 acc;
    int32_t* ap = accs + (size_t)blockIdx.y * L + lane;
    if (first) ge_identity(acc);
    else {
#pragma unroll
        for (int i = 0; i < FE_NL; i++) {
            acc.X.v[i] = ap[(size_t)i * Ltot]; acc.Y.v[i] = ap[(size_t)(FE_NL + i) * Ltot];
            acc.Z.v[i] = ap[(size_t)(2 * FE_NL + i) * Ltot]; acc.T.v[i] = ap[(size_t)(3 * FE_NL + i) * Ltot];
        }
    }
    const dapol_v4i* dg = reinterpret_cast<const dapol_v4i*>(A.dig) + ((size_t)(side * A.N + q0) >> 2) * Ld + lane;
    const dapol_v4i* dh = dg + (has_hi ? (size_t)LW * A.B : 0);
    dapol_v4i d4 = dg[0], h4 = dh[0], dn = d4, hn = h4;
    // ONE copy of the addition (instruction cache): a term's two lookups are two trips
#pragma nounroll
    for (int i2 = 0; i2 < 2 * nq; i2++) {
        const int i = i2 >> 1, hi = i2 & 1;
        if (!hi && (i & 3) == 0 && i + 4 < nq) { dn = dg[(size_t)((i >> 2) + 1) * Ld]; hn = dh[(size_t)((i >> 2) + 1) * Ld]; }
        int d;
        if (!hi) { d = d4.x; d4.x = d4.y; d4.y = d4.z; d4.z = d4.w; }
        else {
            d = h4.x;
            h4.x = h4.y; h4.y = h4.z; h4.z = h4.w;
            if ((i & 3) == 3) { d4 = dn; h4 = hn; }
            if (!has_hi) continue;                                           // (uniform over a wavefront but for the one that straddles two windows)
        }
        bool isH;
        const int j = term_generator(round, A.N, A.lgN, side, q0 + i, isH);      // (uniform over a block)
        const int row = gen_row(tbl, A.n, j, isH);
        tbl_madd(acc, tbl, hi ? tbl.row_hi(row) : row, d);
    }
#pragma unroll
    for (int i = 0; i < FE_NL; i++) {
        ap[(size_t)i * Ltot] = acc.X.v[i]; ap[(size_t)(FE_NL + i) * Ltot] = acc.Y.v[i];
        ap[(size_t)(2 * FE_NL + i) * Ltot] = acc.Z.v[i]; ap[(size_t)(3 * FE_NL + i) * Ltot] = acc.T.v[i];
    }
}

// Slices of a sweep -> slice 0: one lane per (side, window, proof), nslice - 1 additions.  accs: both sides' accumulators,
// side_words apart, each SoA over Ltot = L * nslice lanes.
__global__ __launch_bounds__(64) void k_rp_gs_sum_slices(RangeArgs A, int32_t* __restrict__ accs, size_t side_words, int nslice, int LW) {
    const size_t L = A.B * (size_t)LW, Ltot = L * (size_t)nslice;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= 2 * L) return;
    const int side = t >= L ? 1 : 0;
    int32_t* ap = accs + (size_t)side * side_words + (t - (size_t)side * L);
    auto load = [&](ge_p3& r, int s) {
        const int32_t* q = ap + (size_t)s * L;
        for (int i = 0; i < FE_NL; i++) {
            r.X.v[i] = q[(size_t)i * Ltot]; r.Y.v[i] = q[(size_t)(FE_NL + i) * Ltot];
            r.Z.v[i] = q[(size_t)(2 * FE_NL + i) * Ltot]; r.T.v[i] = q[(size_t)(3 * FE_NL + i) * Ltot];
        }
    };
    ge_p3 acc, o, r;
    load(acc, 0);
#pragma nounroll
    for (int s = 1; s < nslice; s++) {
        load(o, s);
        ge_add(r, acc, o);
        acc = r;
    }
    for (int i = 0; i < FE_NL; i++) {
        ap[(size_t)i * Ltot] = acc.X.v[i]; ap[(size_t)(FE_NL + i) * Ltot] = acc.Y.v[i];
        ap[(size_t)(2 * FE_NL + i) * Ltot] = acc.Z.v[i]; ap[(size_t)(3 * FE_NL + i) * Ltot] = acc.T.v[i];
    }
}

// P_side[p] = sum_w 2^(W w) * acc_side[w * cb + p]: Horner from the top window, one lane per (side, proof) -- both lists of a
// round in one launch (accs: the two sides' accumulators, side_words apart; SoA over L * nslice lanes, the sums in slice 0).
// LW = window sums per list: nwin, or TableView::hi_split when the sweep looked every term up twice (k_rp_msm_gs_hi).
__global__ __launch_bounds__(64) void k_rp_gs_combine(RangeArgs A, const int32_t* __restrict__ accs, size_t side_words, int nslice, int LW) {
    const size_t Ltot = A.B * (size_t)LW * (size_t)nslice;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= 2 * A.B) return;
    const int side = t >= A.B ? 1 : 0;
    const size_t p = t - (size_t)side * A.B;
    accs += (size_t)side * side_words;
    auto load = [&](ge_p3& r, int w) {
        const int32_t* ap = accs + (size_t)w * A.B + p;
        for (int i = 0; i < FE_NL; i++) {
            r.X.v[i] = ap[(size_t)i * Ltot]; r.Y.v[i] = ap[(size_t)(FE_NL + i) * Ltot];
            r.Z.v[i] = ap[(size_t)(2 * FE_NL + i) * Ltot]; r.T.v[i] = ap[(size_t)(3 * FE_NL + i) * Ltot];
        }
    };
    ge_p3 acc, t2, r;
    load(acc, LW - 1);
#pragma nounroll
    for (int w = LW - 2; w >= 0; w--) {
#pragma nounroll
        for (int d = 0; d < A.wbits; d++) ge_dbl(acc, acc, d == A.wbits - 1);
        load(t2, w);
        ge_add(r, acc, t2);
        acc = r;
    }
    st_p3((side ? A.P1 : A.P0) + p * 40, acc);
}

// ------------------------------------------------------------------------------------------------ materialisation
// The hybrid argument's 2T folded generators (k_rp_msm<MSM_MATERIALIZE>: G'_i = sum over q = i (mod T) of s_q G_q, H'_i alike)
// in the same order of work: the terms of one folded generator ("class" i of a side) are one tile of N / T rows -- plus the tile
// of their high-half rows when the context has them (TableView::hi_split: window w and window w + hi_split in the same lane) --
// and a lane owns (proof, window) of that class.  A class's LW window sums are combined by k_rp_mat_gs_horner, MAT_GROUP classes
// per launch so that the Horner chains (W (LW - 1) doublings, the same work the proof-stationary kernel does) fill the chip.
// Digits: sweep position sp = side * N + i * (N / T) + k  <->  term q = i + k T, same dig4 layout as above.
enum { MAT_GROUP = 16 };
// accumulator slots of a chunk (MAT_GROUP classes, or 2 lists x up to GS_MAX_SLICES slices); lanes that fill the chip (3,072 wavefronts)
enum { GS_ACC_SLOTS = 32, GS_MAX_SLICES = 16, GS_FULL_LANES = 3072 * 64 };
static_assert(GS_ACC_SLOTS >= MAT_GROUP && GS_ACC_SLOTS >= 2 * GS_MAX_SLICES, "accumulator slots");

// grid = ceil(cb * 2N / 64) blocks of 64
__global__ __launch_bounds__(64) void k_rp_mat_prep_gs(RangeArgs A) {
    GS_BLOCKS(A, vb) {
        size_t p;
        int sp;
        if (!gs_thread(A, vb, p, sp)) continue;
        const int side = sp >= A.N ? 1 : 0, rem = sp - side * A.N;
        const int per = A.N / A.tail_n, cls = rem / per, k = rem - cls * per;
        sc s;
        coeff_plain(s, A, p, A.mat_round, cls + k * A.tail_n, side != 0);          // plain form
        write_digits_gs(A, p, sp, s.v);
    }
}

// One class, one half (0: the generators' own rows, digits of window w; 1: their high-half rows, digits of window w + LW).
// accs: this class's slot, SoA over LM = cb * LW lanes (lane = w * cb + p).
// gridDim.y classes per launch (cls + blockIdx.y, each in its own slot) when the lanes of one class cannot fill the chip.
__global__ __launch_bounds__(64, DAPOL_GS_OCC) void k_rp_mat_gs(RangeArgs A, TableView tbl, int side, int cls, int hi, int LW, int32_t* __restrict__ accs) {
    const size_t LM = A.B * (size_t)LW, L = A.B * (size_t)A.nwin;
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (lane >= LM) return;
    cls += (int)blockIdx.y;
    accs += (size_t)blockIdx.y * 4 * FE_NL * LM;
    const int w = (int)(lane / A.B);
    const size_t p = lane - (size_t)w * A.B;
    const int wd = w + (hi ? LW : 0);
    if (wd >= A.nwin) return;                          // (no such window: the accumulator stays as the first half left it)
    ge_p3 acc;
    int32_t* ap = accs + lane;
    if (!hi) ge_identity(acc);
    else {
#pragma unroll
        for (int i = 0; i < FE_NL; i++) {
            acc.X.v[i] = ap[(size_t)i * LM]; acc.Y.v[i] = ap[(size_t)(FE_NL + i) * LM];
            acc.Z.v[i] = ap[(size_t)(2 * FE_NL + i) * LM]; acc.T.v[i] = ap[(size_t)(3 * FE_NL + i) * LM];
        }
    }
    const int per = A.N / A.tail_n;
    const dapol_v4i* dg = reinterpret_cast<const dapol_v4i*>(A.dig) + ((size_t)(side * A.N + cls * per) >> 2) * L + (size_t)wd * A.B + p;
    dapol_v4i d4 = dg[0], dn = d4;                       // (digits one quad ahead, as in k_rp_msm_gs)
#pragma nounroll
    for (int k = 0; k < per; k++) {
        if ((k & 3) == 0 && k + 4 < per) dn = dg[(size_t)((k >> 2) + 1) * L];
        const int d = d4.x;
        d4.x = d4.y; d4.y = d4.z; d4.z = d4.w;
        if ((k & 3) == 3) d4 = dn;
        const int row = gen_row(tbl, A.n, cls + k * A.tail_n, side != 0);
        tbl_madd(acc, tbl, hi ? tbl.row_hi(row) : row, d);
    }
#pragma unroll
    for (int i = 0; i < FE_NL; i++) {
        ap[(size_t)i * LM] = acc.X.v[i]; ap[(size_t)(FE_NL + i) * LM] = acc.Y.v[i];
        ap[(size_t)(2 * FE_NL + i) * LM] = acc.Z.v[i]; ap[(size_t)(3 * FE_NL + i) * LM] = acc.T.v[i];
    }
}

// Folded generator of (proof p, class cls0 + c) = sum_w 2^(W w) * (window sum w): Horner, one lane per (class, proof); the point
// goes to the head of its row of the proof's tail table (k_rp_tail_table builds the row), as the proof-stationary kernel leaves it.
__global__ __launch_bounds__(64) void k_rp_mat_gs_horner(RangeArgs A, int side, int cls0, int ncls, int LW, const int32_t* __restrict__ accs) {
    const size_t LM = A.B * (size_t)LW;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= (size_t)ncls * A.B) return;
    const int c = (int)(t / A.B);
    const size_t p = t - (size_t)c * A.B;
    const int32_t* slot = accs + (size_t)c * 4 * FE_NL * LM;
    auto load = [&](ge_p3& r, int w) {
        const int32_t* ap = slot + (size_t)w * A.B + p;
        for (int i = 0; i < FE_NL; i++) {
            r.X.v[i] = ap[(size_t)i * LM]; r.Y.v[i] = ap[(size_t)(FE_NL + i) * LM];
            r.Z.v[i] = ap[(size_t)(2 * FE_NL + i) * LM]; r.T.v[i] = ap[(size_t)(3 * FE_NL + i) * LM];
        }
    };
    ge_p3 acc, q, r;
    load(acc, LW - 1);
#pragma nounroll
    for (int w = LW - 2; w >= 0; w--) {
#pragma nounroll
        for (int d = 0; d < A.wbits; d++) ge_dbl(acc, acc, d == A.wbits - 1);
        load(q, w);
        ge_add(r, acc, q);
        acc = r;
    }
    const int T = A.tail_n;
    st_p3(A.tailT + (p * (size_t)(2 * T) + (size_t)(side * T + cls0 + c)) * TAIL_ROW_WORDS, acc);
}

}  // namespace dapol
