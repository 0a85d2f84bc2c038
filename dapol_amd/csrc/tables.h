// Fixed-base window tables in HBM (L2 / Infinity-Cache resident) and the lookups the kernels use.
//
// Layout (W-bit signed windows): a ROW holds the 2^(W-1)+1 multiples k*P, k = 0..2^(W-1), of one
// base point P as 128-byte entries
// (affine niels form: y+x, y-x, 2dxy as 3 x 9 int32 limbs, 5 words of padding -> seven 16-byte loads per lookup,
// one cache line).  Entry 0 is the identity so a zero digit needs no branch; digits are signed (-128..128).
//   rows [0, 64*P)            G[party][bit]      (P = max_parties)       bulletproofs BulletproofGens G chain
//   rows [64*P, 128*P)        H[party][bit]                              ... H chain
//   rows [128*P, 128*P+nwin)        2^(W w) * B_blinding, w = 0..nwin-1   PedersenGens::default().B_blinding
//   rows [128*P+nwin, 128*P+2nwin)  2^(W w) * B,          w = 0..nwin-1   PedersenGens::default().B
// The per-window rows of B / B_blinding make single-base commitments doubling-free (32 mixed adds); the G/H
// rows are used Straus-style (shared doublings across the terms a lane owns).
#pragma once
#include "ge.h"

namespace dapol {

// The window width W is a property of the context (chosen at dapol_ctx_create from a table-memory budget, or the
// DAPOL_WBITS environment variable): 8 <= W <= 20.  Measured on MI355X (profiles/r01_wbits_ab*.txt, 2^16 proofs
// n=64 m=32, k_rp_msm per launch): 155 / 144 / 134 / 128 ms at W = 10 / 12 / 13 / 14, proofs byte-identical.  Past
// W = 10 the tables (273 MB -> 4.4 GB at 14 bits) no longer fit the Infinity Cache and the lookups become random
// 128-byte HBM gathers (~3 TB/s).  With the final kernel: 96 / 91 / 102 ms at W = 16 / 17 / 19 -- 17 bits (15 windows,
// 34.6 GB of tables for 32 parties) is the sweet spot; at 19 bits the 138 GB of tables cost more than the saved window.
enum { TBL_ENTRY_WORDS = 32, WBITS_MIN = 8, WBITS_MAX = 20, WBITS_AUTO_MAX = 17 };
typedef int32_t dig_t;       // digits of up to 20 bits (int16 would cap the width at 16)

struct TableView {
    const int32_t* base;   // device pointer
    int32_t max_parties;
    int32_t wbits;         // W
    int32_t digest;        // node hash D of the context: DG_BLAKE3 (0) or DG_BLAKE2S (1)
    // High-half rows (0 = none): a second row per G / H generator holding the multiples of 2^(W * hi_split) * P, so that a sum
    // whose accumulators cannot share doublings (the materialisation of the folded generators: one accumulator per lane, 255
    // doublings for 480 additions) walks hi_split windows instead of nwin_c(): the scalar is split s = s_lo + 2^(W hi_split) s_hi
    // and both halves are looked up in the same window step.  Row of generator row r: n_rows() + r.
    int32_t hi_split;
    // Windows of an UNREDUCED 255-bit integer (Scalar::from_bits leaf blindings): the top window must hold its value
    // plus the recoding carry within 2^(W-1), i.e. be at most W-1 bits wide -> 255/W + 1 windows.  Canonical scalars
    // (< 2^253, everything in the digit matrices) need 253/W + 1.
    __host__ __device__ int nwin() const { return 255 / wbits + 1; }
    __host__ __device__ int nwin_c() const { return 253 / wbits + 1; }
    __host__ __device__ int nwin64() const { return (64 + wbits - 1) / wbits + 1; }     // 64-bit value (+1: recoding carry)
    __host__ __device__ int entries() const { return (1 << (wbits - 1)) + 1; }
    __host__ __device__ size_t row_words() const { return (size_t)entries() * TBL_ENTRY_WORDS; }
    __host__ __device__ int row_G(int party, int bit) const { return party * 64 + bit; }
    __host__ __device__ int row_H(int party, int bit) const { return 64 * max_parties + party * 64 + bit; }
    __host__ __device__ int row_Bb(int w) const { return 128 * max_parties + w; }
    __host__ __device__ int row_B(int w) const { return 128 * max_parties + nwin() + w; }
    __host__ __device__ int n_rows() const { return 128 * max_parties + 2 * nwin(); }
    __host__ __device__ int n_rows_total() const { return n_rows() + (hi_split ? 128 * max_parties : 0); }
    __host__ __device__ int row_hi(int gh_row) const { return n_rows() + gh_row; }
};

#if defined(__HIPCC__)
// A 128-byte entry holds ypx | ymx | xy2d as 3 x FE_NL = 27 limbs (words 0..26); words 27..31 are padding (word 30 doubles
// as a flag in the verifier's decoded-point arrays).  Seven 16-byte loads fetch it.
__host__ __device__ __forceinline__ void niels_store_entry(int32_t* e, const ge_niels& q) {
    for (int i = 0; i < FE_NL; i++) { e[i] = q.ypx.v[i]; e[FE_NL + i] = q.ymx.v[i]; e[2 * FE_NL + i] = q.xy2d.v[i]; }
    for (int i = 3 * FE_NL; i < TBL_ENTRY_WORDS; i++) e[i] = 0;
}
// Plain (temporal) loads on purpose.  Tried: non-temporal loads (global_load_dwordx4 ... nt), on the theory that a random
// gather over 35 GB has no line worth keeping -- 32 % SLOWER and +55 % HBM bytes (profiles/r02_digits_nt_ab.txt): the seven
// loads of one entry are the same 128-byte line, and only a line that stays in L2 between them is fetched from HBM once.
typedef int dapol_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void niels_load_entry(ge_niels& q, const int32_t* e) {
    const dapol_v4i* p = reinterpret_cast<const dapol_v4i*>(e);
    dapol_v4i a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5], a6 = p[6];
    q.ypx.v[0] = a0.x; q.ypx.v[1] = a0.y; q.ypx.v[2] = a0.z; q.ypx.v[3] = a0.w;
    q.ypx.v[4] = a1.x; q.ypx.v[5] = a1.y; q.ypx.v[6] = a1.z; q.ypx.v[7] = a1.w;
    q.ypx.v[8] = a2.x;
    q.ymx.v[0] = a2.y; q.ymx.v[1] = a2.z; q.ymx.v[2] = a2.w;
    q.ymx.v[3] = a3.x; q.ymx.v[4] = a3.y; q.ymx.v[5] = a3.z; q.ymx.v[6] = a3.w;
    q.ymx.v[7] = a4.x; q.ymx.v[8] = a4.y;
    q.xy2d.v[0] = a4.z; q.xy2d.v[1] = a4.w;
    q.xy2d.v[2] = a5.x; q.xy2d.v[3] = a5.y; q.xy2d.v[4] = a5.z; q.xy2d.v[5] = a5.w;
    q.xy2d.v[6] = a6.x; q.xy2d.v[7] = a6.y; q.xy2d.v[8] = a6.z;
}
// Load entry |d| of `row`; the sign is applied by ge_madd.
__device__ __forceinline__ void tbl_load(ge_niels& q, const TableView& t, int row, int absd) {
    // one 64-bit multiply-add for the entry's word offset (row and |d| are non-negative; the offset stays below 2^40)
    niels_load_entry(q, t.base + ((uint64_t)(uint32_t)row * (uint32_t)t.row_words() + (uint32_t)(absd * TBL_ENTRY_WORDS)));
}

// acc += d * (row's base point), d in [-128, 128]
__device__ __forceinline__ void tbl_madd(ge_p3& acc, const TableView& t, int row, int d) {
    ge_niels q;
    int ad = d < 0 ? -d : d;
    tbl_load(q, t, row, ad);
    ge_madd(acc, acc, q, d < 0);          // in place: ge_madd reads all of p before it writes r
}

// acc += s * Base for a 255-bit integer s (eight words), using the NWIN per-window rows starting at row0.
__device__ __forceinline__ void tbl_fixed_mul_add(ge_p3& acc, const TableView& t, int row0, const uint32_t* s8) {
    int carry = 0;
    const int W = t.wbits, NW = t.nwin();
    for (int i = 0; i < NW; i++) {
        int o = i * W, wd = o >> 5, sh = o & 31;
        uint32_t lo = wd < 8 ? s8[wd] >> sh : 0u;
        uint32_t hi = (sh && wd + 1 < 8) ? (s8[wd + 1] << (32 - sh)) : 0u;
        int b = (int)((lo | hi) & ((1u << W) - 1)) + carry;
        carry = (b >= (1 << (W - 1)) && i < NW - 1) ? 1 : 0;
        tbl_madd(acc, t, row0 + i, b - (carry << W));
    }
}
// acc += v * B for a 64-bit v (NWIN64 signed windows)
__device__ __forceinline__ void tbl_fixed_mul_add_u64(ge_p3& acc, const TableView& t, int row0, uint64_t v) {
    int carry = 0;
    const int W = t.wbits, NW = t.nwin64();
    for (int i = 0; i < NW; i++) {
        int o = i * W;
        int b = (o < 64 ? (int)((v >> o) & ((1u << W) - 1)) : 0) + carry;
        carry = b >= (1 << (W - 1)) ? 1 : 0;
        tbl_madd(acc, t, row0 + i, b - (carry << W));
    }
}
#endif

}  // namespace dapol
