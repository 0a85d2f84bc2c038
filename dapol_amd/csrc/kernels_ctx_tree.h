// Context-creation kernels (generators + window tables, derived on the GPU) and the sparse-Merkle tree kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "hash.h"
#include "sc.h"
#include "tables.h"

namespace dapol {

// ------------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ void ld8(uint32_t* w, const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void st8(uint32_t* p, const uint32_t* w) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// An extended point in memory: a record of P3_WORDS = 40 words (160 bytes, the stride every buffer uses) holding
// X | Y | Z | T as 4 x FE_NL = 36 limbs; the last four words are unused.
enum { P3_WORDS = 40 };
__device__ __forceinline__ void ld_p3(ge_p3& p, const int32_t* src) {
    const int4* q = reinterpret_cast<const int4*>(src);
    int32_t w[4 * FE_NL];
    for (int i = 0; i < FE_NL; i++) { int4 a = q[i]; w[4 * i] = a.x; w[4 * i + 1] = a.y; w[4 * i + 2] = a.z; w[4 * i + 3] = a.w; }
    for (int i = 0; i < FE_NL; i++) { p.X.v[i] = w[i]; p.Y.v[i] = w[FE_NL + i]; p.Z.v[i] = w[2 * FE_NL + i]; p.T.v[i] = w[3 * FE_NL + i]; }
}
__device__ __forceinline__ void st_p3(int32_t* dst, const ge_p3& p) {
    int32_t w[4 * FE_NL];
    for (int i = 0; i < FE_NL; i++) { w[i] = p.X.v[i]; w[FE_NL + i] = p.Y.v[i]; w[2 * FE_NL + i] = p.Z.v[i]; w[3 * FE_NL + i] = p.T.v[i]; }
    int4* q = reinterpret_cast<int4*>(dst);
    for (int i = 0; i < FE_NL; i++) q[i] = make_int4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ----------------------------------------------------------------------------- context: generator chains
// bulletproofs 4.0.0 GeneratorsChain: SHAKE256("GeneratorsChain" || label), label = 'G'|'H' || u32le(party);
// successive 64-byte reads.  One lane per chain (2 * max_parties chains), 64 reads each.
__device__ __forceinline__ void absorb_label(Sponge& sp, const char* s, int n) {
    for (int i = 0; i < n; i++) sponge_absorb_byte(sp, (uint8_t)s[i]);
}
__global__ void k_ctx_chains(uint32_t* uniform /*[2P][64][16]*/, int P) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * P) return;
    int which = c / P, party = c % P;
    Sponge sp;
    sponge_init(sp, 136);
    absorb_label(sp, LBL_GENERATORS_CHAIN);
    sponge_absorb_byte(sp, which ? LBL_GENS_H : LBL_GENS_G);
    for (int i = 0; i < 4; i++) sponge_absorb_byte(sp, (uint8_t)((uint32_t)party >> (8 * i)));
    sponge_finish(sp, 0x1F);
    uint32_t* out = uniform + (size_t)c * 64 * 16;
    for (int i = 0; i < 64 * 16; i++) {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) w |= (uint32_t)sponge_squeeze_byte(sp) << (8 * k);
        out[i] = w;
    }
}
// One lane per generator: RistrettoPoint::from_uniform_bytes.
__global__ void k_ctx_points(int32_t* base_pts /*[rows][40]*/, const uint32_t* uniform, int P) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 128 * P) return;
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = uniform[(size_t)r * 16 + i];
    ge_p3 p;
    ge_from_uniform(p, w);
    st_p3(base_pts + (size_t)r * 40, p);
}
// PedersenGens::default(): B = ristretto basepoint, B_blinding = hash_from_bytes::<Sha3_512>(B.compress()).
// Lane 0 -> 2^(W w) B_blinding rows, lane 1 -> 2^(W w) B rows.
__global__ void k_ctx_pedersen(int32_t* base_pts, int P, int wbits, int nwin) {
    int t = threadIdx.x;
    if (t >= 2) return;
    ge_p3 p;
    ge_basepoint(p);
    if (t == 0) {
        uint32_t c[8], w[16];
        ge_compress(c, p);
        Sponge sp;
        sponge_init(sp, 72);
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) sponge_absorb_byte(sp, (uint8_t)(c[i] >> (8 * k)));
        sponge_finish(sp, 0x06);
        for (int i = 0; i < 16; i++) {
            uint32_t x = 0;
            for (int k = 0; k < 4; k++) x |= (uint32_t)sponge_squeeze_byte(sp) << (8 * k);
            w[i] = x;
        }
        ge_from_uniform(p, w);
    }
    int row0 = 128 * P + (t == 0 ? 0 : nwin);
    for (int w = 0; w < nwin; w++) {
        st_p3(base_pts + (size_t)(row0 + w) * 40, p);
        for (int d = 0; d < wbits; d++) { ge_p3 q; ge_dbl(q, p, true); p = q; }
    }
}
// High-half base points: 2^(shift) * (G / H generator), one lane per generator (TableView::hi_split).
__global__ void k_ctx_hi_points(int32_t* base_pts, int n_gh, int n_rows, int shift) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_gh) return;
    ge_p3 p, q;
    ld_p3(p, base_pts + (size_t)r * 40);
    for (int d = 0; d < shift; d++) { ge_dbl(q, p, true); p = q; }
    st_p3(base_pts + (size_t)(n_rows + r) * 40, p);
}
// One lane per table entry: k * base, normalised to affine niels form.
__global__ void k_ctx_table(int32_t* table, const int32_t* base_pts, int n_rows, int wbits, int entries) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)n_rows * entries) return;
    int row = (int)(gid / entries), k = (int)(gid % entries);
    ge_niels q;
    if (k == 0) {
        ge_niels_identity(q);
    } else {
        ge_p3 base, acc, t;
        ld_p3(base, base_pts + (size_t)row * 40);
        ge_identity(acc);
        for (int b = wbits - 1; b >= 0; b--) {
            ge_dbl(t, acc, true);
            acc = t;
            if ((k >> b) & 1) { ge_add(t, acc, base); acc = t; }
        }
        fe zi, x, y;
        fe_invert(zi, acc.Z);
        fe_mul(x, acc.X, zi);
        fe_mul(y, acc.Y, zi);
        ge_to_niels(q, x, y);
    }
    int32_t* e = table + ((size_t)row * entries + (size_t)k) * TBL_ENTRY_WORDS;
    niels_store_entry(e, q);
}
__global__ void k_ctx_compress(uint32_t* comp /*[rows][8]*/, const int32_t* base_pts, int n_rows) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    ge_p3 p;
    ld_p3(p, base_pts + (size_t)r * 40);
    uint32_t c[8];
    ge_compress(c, p);
    st8(comp + (size_t)r * 8, c);
}

// ------------------------------------------------------------------------------------------- commitments
// DapolNode::new (src/dapol/node.rs:29-45): C = v*B + r*B_blinding, h = BLAKE3(compress(C)).  One lane per node.
// r: eight words, bit 255 ignored (Scalar::from_bits); may be >= l.
__global__ __launch_bounds__(256) void k_commit_hash(TableView tbl, size_t n, const uint64_t* v, uint32_t* r, uint32_t* C,
                                                     uint32_t* H, int32_t* ext) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rw[8];
    ld8(rw, r + i * 8);
    if (rw[7] >> 31) {             // Scalar::from_bits clears bit 255; keep the stored copy consistent
        rw[7] &= 0x7fffffffu;
        r[i * 8 + 7] = rw[7];
    }
    ge_p3 acc;
    ge_identity(acc);
    tbl_fixed_mul_add_u64(acc, tbl, tbl.row_B(0), v[i]);
    tbl_fixed_mul_add(acc, tbl, tbl.row_Bb(0), rw);
    uint32_t c[8], h[8];
    ge_compress(c, acc);
    node_hash32(tbl.digest, h, c);
    st8(C + i * 8, c);
    st8(H + i * 8, h);
    if (ext) st_p3(ext + i * 40, acc);
}

// --------------------------------------------------------------------------------------------------- scan
// The level sizes live on the DEVICE (cnt[k] = real nodes of level k): every kernel of the build reads its bound from there
// and is launched over the host-side upper bound, so the host never waits for a level (no per-level synchronisation).
// flag[i] = 1 if node i starts a new parent (its index >> 1 differs from its predecessor's).
__global__ void k_tree_flags(const uint32_t* n_ptr, const uint64_t* idx, uint32_t* flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    flag[i] = (i == 0 || (idx[i] >> 1) != (idx[i - 1] >> 1)) ? 1u : 0u;
}
// Inclusive scan of 256 values held one per thread; returns the inclusive value, *total = sum over the block.
__device__ __forceinline__ uint32_t block_scan256(uint32_t x, uint32_t* wave_tot /*[4] shared*/, uint32_t* total) {
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t add = 0, all = 0;
    for (int k = 0; k < 4; k++) { if (k < wv) add += wave_tot[k]; all += wave_tot[k]; }
    __syncthreads();
    *total = all;
    return x + add;
}
// Block-local inclusive scan of 1024 elements per 256-thread block.
__global__ __launch_bounds__(256) void k_scan_block(const uint32_t* n_ptr, const uint32_t* in, uint32_t* out, uint32_t* block_sums) {
    __shared__ uint32_t wave_tot[4];
    const size_t n = *n_ptr;
    size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
    if ((size_t)blockIdx.x * 1024 >= n) return;                      // (whole block: the launch covers the host-side bound)
    uint32_t a[4];
    for (int k = 0; k < 4; k++) a[k] = (base + k < n) ? in[base + k] : 0u;
    a[1] += a[0]; a[2] += a[1]; a[3] += a[2];
    uint32_t tot, incl = block_scan256(a[3], wave_tot, &tot);
    uint32_t excl = incl - a[3];
    for (int k = 0; k < 4; k++)
        if (base + k < n) out[base + k] = a[k] + excl;
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// Exclusive scan of the block sums by ONE block of 256 threads (256 sums per trip); the grand total = size of the next level.
__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t* n_ptr, uint32_t* block_sums, uint32_t* total) {
    __shared__ uint32_t wave_tot[4];
    const size_t nb = ((size_t)*n_ptr + 1023) / 1024;
    uint32_t carry = 0;
    for (size_t base = 0; base < nb; base += 256) {
        size_t i = base + threadIdx.x;
        uint32_t t = i < nb ? block_sums[i] : 0u, tot;
        uint32_t incl = block_scan256(t, wave_tot, &tot);
        if (i < nb) block_sums[i] = carry + incl - t;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
// pos[i] = parent slot of node i; head[q] = first child of parent q.
__global__ void k_scan_finish(const uint32_t* n_ptr, const uint32_t* flag, uint32_t* pos, const uint32_t* block_offs, uint32_t* head) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_ptr) return;
    uint32_t p = pos[i] + block_offs[i >> 10] - 1;
    pos[i] = p;
    if (flag[i]) head[p] = (uint32_t)i;
}

// ------------------------------------------------------------------------------------------- tree levels
struct LevelView {
    size_t n;
    uint64_t* idx;
    uint32_t* C;
    uint32_t* H;
    uint64_t* v;
    uint32_t* r;
    uint32_t* padC;
    uint32_t* padH;
    uint32_t* padr;
    uint8_t* has_pad;
    uint32_t* parent;
    int32_t* ext;   // extended points of this level (only alive while the next level is being built)
};

// One lane per PARENT q (smtree build restated level-synchronously): children = the real node head[q] and either
// the adjacent real node or a padding node made on the spot (Paddable::padding, src/dapol/node.rs:86-88, with the
// positional seed-mode blinding); parent = Mergeable::merge (node.rs:64-80).
// TAPE mode (pad_tape != null; dapol_tree_build_tape): the padding node's 64-byte draw is read from the caller's tape instead of being
// derived from the seed.  Tape order = (level bottom-up, index ascending) over the padding nodes.  Parent q of this level has the
// real children head[q] (and head[q] + 1 when they are a pair), so 2 q - head[q] parents before it own a padding child -- the rank
// inside the level -- and the levels below hold sum (2 cnt[j + 1] - cnt[j]) padding nodes: no scan, no second pass.
struct PadTape {
    const uint32_t* draws;     // [n_draws][16] or null (seed mode)
    uint32_t n_draws;
    uint32_t* short_flag;      // set when the tree needs more draws than the tape holds
};
// SPLIT = 1 (round 6 experiment, DAPOL_TREE_SPLIT=1): the padding children of the level were made by k_tree_pad_level just before --
// their records are read from padC / padH / padr and their points from extpad[q] -- so that neither kernel holds a fixed-base
// product, two encodings and three hashes in one register allocation.
template <int SPLIT>
__global__ __launch_bounds__(256) void k_tree_merge(TableView tbl, LevelView cur, LevelView nxt, const uint32_t* head, int level,
                                                    const uint32_t* pad_seed /*8 words*/, const uint32_t* cnt /*[levels + 1], device*/, PadTape tape,
                                                    const int32_t* extpad) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t cur_n = cnt[level], nxt_n = cnt[level + 1];           // (cur.n / nxt.n are only host-side bounds during the build)
    if (q >= nxt_n) return;
    size_t i = head[q];
    uint64_t my_idx = cur.idx[i];
    bool pair = (i + 1 < cur_n) && (cur.idx[i + 1] == (my_idx ^ 1ull));
    uint32_t cA[8], hA[8], rA[8], cB[8], hB[8], rB[8];
    uint64_t vA = cur.v[i], vB = 0;
    ge_p3 pA, pB;
    ld8(cA, cur.C + i * 8);
    ld8(hA, cur.H + i * 8);
    ld8(rA, cur.r + i * 8);
    ld_p3(pA, cur.ext + i * 40);
    if (pair) {
        ld8(cB, cur.C + (i + 1) * 8);
        ld8(hB, cur.H + (i + 1) * 8);
        ld8(rB, cur.r + (i + 1) * 8);
        vB = cur.v[i + 1];
        ld_p3(pB, cur.ext + (i + 1) * 40);
        cur.has_pad[i] = 0;
        cur.has_pad[i + 1] = 0;
        cur.parent[i + 1] = (uint32_t)q;
    } else if (SPLIT) {
        ld8(cB, cur.padC + i * 8);
        ld8(hB, cur.padH + i * 8);
        ld8(rB, cur.padr + i * 8);
        ld_p3(pB, extpad + q * 40);
        cur.has_pad[i] = 1;
    } else {
        uint32_t seed[8], wide[16];
        if (tape.draws) {
            uint64_t rank = 2 * (uint64_t)q - (uint64_t)i;
            for (int j = 0; j < level; j++) rank += 2 * (uint64_t)cnt[j + 1] - (uint64_t)cnt[j];
            if (rank < tape.n_draws) { for (int k = 0; k < 16; k++) wide[k] = tape.draws[rank * 16 + k]; }
            else { atomicOr(tape.short_flag, 1u); for (int k = 0; k < 16; k++) wide[k] = 0; }
        } else {
            for (int k = 0; k < 8; k++) seed[k] = pad_seed[k];
            seed_wide(wide, seed, 1u, (uint64_t)level, my_idx ^ 1ull);
        }
        sc rm;
        sc_from_wide(rm, wide);
        sc_from_mont(rB, rm);
        ge_identity(pB);
        tbl_fixed_mul_add(pB, tbl, tbl.row_Bb(0), rB);
        ge_compress(cB, pB);
        node_hash32(tbl.digest, hB, cB);
        st8(cur.padC + i * 8, cB);
        st8(cur.padH + i * 8, hB);
        st8(cur.padr + i * 8, rB);
        cur.has_pad[i] = 1;
    }
    cur.parent[i] = (uint32_t)q;
    bool a_is_left = pair || ((my_idx & 1ull) == 0);
    // r = (r_L + r_R) mod l  (dalek Scalar add reduces even unreduced inputs)
    sc ma, mb, ms;
    sc_to_mont(ma, rA);
    sc_to_mont(mb, rB);
    sc_add(ms, ma, mb);
    uint32_t rp[8], cp[8], hp[8];
    sc_from_mont(rp, ms);
    ge_p3 pp;
    ge_add(pp, pA, pB);
    ge_compress(cp, pp);
    if (a_is_left) node_hash128(tbl.digest, hp, cA, cB, hA, hB);
    else node_hash128(tbl.digest, hp, cB, cA, hB, hA);
    nxt.idx[q] = my_idx >> 1;
    nxt.v[q] = vA + vB;           // u64 wrap == release-mode Rust (node.rs:72)
    st8(nxt.r + q * 8, rp);
    st8(nxt.C + q * 8, cp);
    st8(nxt.H + q * 8, hp);
    if (nxt.ext) st_p3(nxt.ext + q * 40, pp);
}

// The padding children of one level on their own (seed mode): Paddable::padding for every parent q whose second child is not real.
__global__ __launch_bounds__(256) void k_tree_pad_level(TableView tbl, LevelView cur, const uint32_t* head, int level, const uint32_t* pad_seed,
                                                        const uint32_t* cnt, int32_t* extpad) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t cur_n = cnt[level], nxt_n = cnt[level + 1];
    if (q >= nxt_n) return;
    size_t i = head[q];
    uint64_t my_idx = cur.idx[i];
    if ((i + 1 < cur_n) && (cur.idx[i + 1] == (my_idx ^ 1ull))) return;
    uint32_t seed[8], wide[16], rB[8], cB[8], hB[8];
    for (int k = 0; k < 8; k++) seed[k] = pad_seed[k];
    seed_wide(wide, seed, 1u, (uint64_t)level, my_idx ^ 1ull);
    sc rm;
    sc_from_wide(rm, wide);
    sc_from_mont(rB, rm);
    ge_p3 pB;
    ge_identity(pB);
    tbl_fixed_mul_add(pB, tbl, tbl.row_Bb(0), rB);
    ge_compress(cB, pB);
    node_hash32(tbl.digest, hB, cB);
    st8(cur.padC + i * 8, cB);
    st8(cur.padH + i * 8, hB);
    st8(cur.padr + i * 8, rB);
    st_p3(extpad + q * 40, pB);
}

// ------------------------------------------------------------------------------------- incremental update
// smtree's `update` (src/dapol/mod.rs:210-213 -> SparseMerkleTree::update) re-merges ONE root-to-leaf path.  Replacing the
// liabilities of k leaves that already exist changes no structure: the same nodes, parents and padding siblings (those are keyed
// by position).  What changes along every path is a sum: an ancestor's value, blinding and commitment move by the sum of the
// deltas of the updated leaves below it --
//     v' = v + sum dv (u64 wrap, as Mergeable::merge),   r' = r + sum dr (mod l),   C' = C + sum dP  (group addition)
// -- and every ancestor is independent of every other, so all levels run side by side (U2); only the hash chain is sequential
// over the levels (U3, one block stepping through them).  The result equals dapol_tree_build over the new leaf set bit for bit
// (ristretto encodings are canonical; tests/test_gpu_parity.py::test_update_equals_build).
//   U0  k_tree_upd_find     per update: find the leaf and its node positions at every level (parent pointers); a missing index
//                           sends the whole batch to the rebuild before anything has been written
//   U1  k_tree_upd_leaves   per update: the leaf's new node, dP = P_new - P_old, dv, dr
//   U2  k_tree_upd_nodes    per (update, level >= 1): the first update of a run that shares the node adds the run's deltas to it
//   U3  k_tree_upd_hash     one block, level by level: re-hash the touched parents from their (updated) children
struct TreeUpdArgs {
    size_t k;                  // updates, sorted by leaf index, distinct
    int height;
    const uint64_t* idx;       // [k]
    const uint64_t* v;         // [k]
    const uint32_t* r;         // [k][8]
    uint32_t* pos;             // [k][height + 1]   node position of update j at every level
    int32_t* dP;               // [k][40]           P_new - P_old, extended
    uint64_t* dv;              // [k]
    uint32_t* dr;              // [k][8]            Montgomery form
    uint32_t* missing;         // != 0: some leaf index is not in the tree (the caller rebuilds instead)
    uint8_t* found;            // [k] (may be null): per update, whether its leaf exists
    const uint32_t* first;     // [k] (may be null = 1): lowest level whose EXISTING node this update changes by its delta
};
// U0: where the updated leaves are (nothing is written to the tree until every index has been found)
__global__ __launch_bounds__(64) void k_tree_upd_find(const LevelView* views, TreeUpdArgs U) {
    const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= U.k) return;
    const LevelView L0 = views[0];
    const uint64_t want = U.idx[j];
    size_t lo = 0, hi = L0.n;                       // lower bound in the sorted leaf indexes
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (L0.idx[mid] < want) lo = mid + 1; else hi = mid;
    }
    if (lo >= L0.n || L0.idx[lo] != want) { atomicOr(U.missing, 1u); if (U.found) U.found[j] = 0; return; }
    if (U.found) U.found[j] = 1;
    uint32_t* pos = U.pos + j * (size_t)(U.height + 1);
    size_t p = lo;
    pos[0] = (uint32_t)p;
    for (int k = 0; k < U.height; k++) { p = views[k].parent[p]; pos[k + 1] = (uint32_t)p; }
}
__global__ __launch_bounds__(64) void k_tree_upd_leaves(TableView tbl, const LevelView* views, TreeUpdArgs U) {
    const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= U.k) return;
    const LevelView L0 = views[0];
    const size_t lo = U.pos[j * (size_t)(U.height + 1)];
    uint32_t rn[8], ro[8], c[8], h[8];
    ld8(rn, U.r + j * 8);
    rn[7] &= 0x7fffffffu;                          // Scalar::from_bits
    ld8(ro, L0.r + lo * 8);
    ld8(c, L0.C + lo * 8);
    ge_p3 pn, po, npo, d;
    ge_identity(pn);
    tbl_fixed_mul_add_u64(pn, tbl, tbl.row_B(0), U.v[j]);
    tbl_fixed_mul_add(pn, tbl, tbl.row_Bb(0), rn);
    (void)ge_decompress(po, c);                    // the tree's own encoding: always decodes
    ge_neg(npo, po);
    ge_add(d, pn, npo);
    st_p3(U.dP + j * 40, d);
    sc mn, mo, md;
    sc_to_mont(mn, rn);
    sc_to_mont(mo, ro);
    sc_sub(md, mn, mo);
    for (int i = 0; i < 8; i++) U.dr[j * 8 + i] = md.v[i];
    U.dv[j] = U.v[j] - L0.v[lo];
    ge_compress(c, pn);
    node_hash32(tbl.digest, h, c);
    L0.v[lo] = U.v[j];
    st8(L0.r + lo * 8, rn);
    st8(L0.C + lo * 8, c);
    st8(L0.H + lo * 8, h);
}
__global__ __launch_bounds__(64) void k_tree_upd_nodes(const LevelView* views, TreeUpdArgs U) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= U.k * (size_t)U.height) return;
    const size_t j = t / (size_t)U.height;
    const int k = 1 + (int)(t - j * (size_t)U.height);
    const size_t stride = (size_t)U.height + 1;
    if (U.first && k < (int)U.first[j]) return;                     // (an inserted leaf: its own new nodes below that level)
    const uint32_t p = U.pos[j * stride + k];
    if (j > 0 && U.pos[(j - 1) * stride + k] == p) return;          // an earlier update of the run owns this node
    ge_p3 sum, q, s;
    ld_p3(sum, U.dP + j * 40);
    uint64_t dv = U.dv[j];
    sc dr, t2;
    for (int i = 0; i < 8; i++) dr.v[i] = U.dr[j * 8 + i];
    for (size_t jj = j + 1; jj < U.k && U.pos[jj * stride + k] == p; jj++) {
        ld_p3(q, U.dP + jj * 40);
        ge_add(s, sum, q);
        sum = s;
        dv += U.dv[jj];
        for (int i = 0; i < 8; i++) t2.v[i] = U.dr[jj * 8 + i];
        sc_add(dr, dr, t2);
    }
    const LevelView L = views[k];
    uint32_t c[8], rr[8];
    ld8(c, L.C + (size_t)p * 8);
    ge_p3 old;
    (void)ge_decompress(old, c);
    ge_add(s, old, sum);
    ge_compress(c, s);
    st8(L.C + (size_t)p * 8, c);
    L.v[p] += dv;
    ld8(rr, L.r + (size_t)p * 8);
    sc mr;
    sc_to_mont(mr, rr);
    sc_add(mr, mr, dr);
    sc_from_mont(rr, mr);
    st8(L.r + (size_t)p * 8, rr);
}
// grid = 1 block of up to 1,024 threads; thread j = update j (k <= 1,024), stepping through the levels together.
__global__ __launch_bounds__(1024) void k_tree_upd_hash(int digest, const LevelView* views, TreeUpdArgs U, int level_begin, int level_end) {
    const size_t j = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    const size_t stride = (size_t)U.height + 1;
    for (int k = level_begin; k < level_end; k++) {                // children at level k -> parent at level k + 1
        if (j < U.k) {
            const uint32_t pp = U.pos[j * stride + k + 1];
            if (j == 0 || U.pos[(j - 1) * stride + k + 1] != pp) {
                const LevelView L = views[k];
                const size_t i = U.pos[j * stride + k];
                const uint64_t my = L.idx[i];
                uint32_t cA[8], hA[8], cB[8], hB[8], hp[8];
                ld8(cA, L.C + i * 8); ld8(hA, L.H + i * 8);
                const bool left = (my & 1ull) == 0;
                if (L.has_pad[i]) { ld8(cB, L.padC + i * 8); ld8(hB, L.padH + i * 8); }
                else { const size_t s = left ? i + 1 : i - 1; ld8(cB, L.C + s * 8); ld8(hB, L.H + s * 8); }
                if (left) node_hash128(digest, hp, cA, cB, hA, hB);
                else node_hash128(digest, hp, cB, cA, hB, hA);
                st8(views[k + 1].H + (size_t)pp * 8, hp);
            }
        }
        if (gridDim.x == 1) __syncthreads();                       // (several blocks: one launch per level instead)
    }
}

// ------------------------------------------------------------------------------------- incremental insert
// A NEW leaf x changes the structure, but only along its own path and only below the first ancestor that exists already: the
// nodes (x >> t) for t < m are new -- each the merge of the one below with a padding sibling made for its position -- and at
// level m - 1 the new node's sibling S is a real node whose padding sibling (at the new node's position) goes away.  From level m
// upwards the ancestors exist and move by a delta, exactly as in the replacement path: dP = P_chain_top - P_old_pad, dr alike,
// dv = v.  The level arrays are compact and sorted, so a level that gains nodes is rewritten out of place (k_tree_relayout: every
// existing node moves up by the number of insertions before it, parent pointers follow the next level's moves); no commitment is
// recomputed for it.  Leaves whose new chains would share a node (two new leaves under one new subtree) are left to the rebuild.
//   I1  k_tree_ins_plan   per new leaf: chain length m, insertion position at every level below m, position of the first
//                         existing ancestor; flags chains that share a node
//   I2  k_tree_relayout   per level that gains nodes: the level's arrays in their new order (existing nodes only)
//   I3  k_tree_ins_chain  a wavefront per new leaf, lane t = chain node t: padding siblings, prefix sums of their points,
//                         encodings; writes the new nodes, drops S's padding sibling, leaves the delta for the levels above
//   then U2 / U3 of the replacement path (k_tree_upd_nodes from level m, k_tree_upd_hash from the leaves).
struct TreeInsPlan {
    size_t k;
    int height;
    const uint64_t* idx;       // [k] sorted, distinct, all NEW
    uint32_t* m;               // [k] chain length (1..height)
    uint32_t* inspos;          // [k][height + 1]: t < m: lower-bound position in level t (old layout); [m]: position of the existing ancestor
    uint32_t* conflict;        // != 0: two chains share a new node
};
__global__ __launch_bounds__(64) void k_tree_ins_plan(const LevelView* views, TreeInsPlan P) {
    const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= P.k) return;
    const uint64_t x = P.idx[j];
    uint32_t* out = P.inspos + j * (size_t)(P.height + 1);
    int m = P.height;
    for (int t = 0; t <= P.height; t++) {
        const LevelView L = views[t];
        const uint64_t want = t < 64 ? x >> t : 0;
        size_t lo = 0, hi = L.n;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (L.idx[mid] < want) lo = mid + 1; else hi = mid;
        }
        out[t] = (uint32_t)lo;
        if (lo < L.n && L.idx[lo] == want) { m = t; break; }
    }
    P.m[j] = (uint32_t)m;
    if (m == 0) atomicOr(P.conflict, 2u);          // (the leaf exists: the caller partitions wrongly)
    if (j > 0) {                                    // shares a new node with its left neighbour?
        const uint64_t y = P.idx[j - 1];
        for (int t = 0; t < m; t++)
            if ((t < 64 ? x >> t : 0) == (t < 64 ? y >> t : 0)) { atomicOr(P.conflict, 1u); break; }
    }
}
// Existing nodes of one level to their new positions.  ins_pos: sorted old-layout lower-bound positions of the nodes this level
// gains (n_ins of them); next_ins_pos / n_next: the same for the level above (parent pointers move with it).
__device__ __forceinline__ uint32_t count_le(const uint32_t* a, uint32_t n, uint32_t x) {      // # of a[i] <= x, a sorted
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_tree_relayout(LevelView src, LevelView dst, size_t n_old, const uint32_t* ins_pos, uint32_t n_ins,
                                                       const uint32_t* next_ins_pos, uint32_t n_next, int is_root_level) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_old) return;
    const size_t d = i + count_le(ins_pos, n_ins, (uint32_t)i);
    dst.idx[d] = src.idx[i];
    dst.v[d] = src.v[i];
    uint32_t w[8];
    ld8(w, src.r + i * 8); st8(dst.r + d * 8, w);
    ld8(w, src.C + i * 8); st8(dst.C + d * 8, w);
    ld8(w, src.H + i * 8); st8(dst.H + d * 8, w);
    dst.has_pad[d] = src.has_pad[i];
    if (src.has_pad[i]) {
        ld8(w, src.padC + i * 8); st8(dst.padC + d * 8, w);
        ld8(w, src.padH + i * 8); st8(dst.padH + d * 8, w);
        ld8(w, src.padr + i * 8); st8(dst.padr + d * 8, w);
    }
    if (!is_root_level) {
        const uint32_t pq = src.parent[i];
        dst.parent[d] = pq + (n_next ? count_le(next_ins_pos, n_next, pq) : 0u);
    }
}
struct TreeInsArgs {
    size_t k;
    int height;
    const uint64_t* idx; const uint64_t* v; const uint32_t* r;     // the new leaves
    const uint32_t* m;                                              // [k]
    const uint32_t* newpos;                                         // [k][height + 1]: t < m: position of chain node t in the NEW layout; [m]: of the existing ancestor
    uint32_t* pos;                                                  // [k][height + 1] out: node positions at every level (new layout), for U2 / U3
    int32_t* dP; uint64_t* dv; uint32_t* dr;                        // out: the delta for levels >= m
    const uint32_t* pad_seed;                                       // [8]
};
// one wavefront per new leaf; lane t < m handles chain node t
__global__ __launch_bounds__(64) void k_tree_ins_chain(TableView tbl, const LevelView* views, TreeInsArgs I) {
    const size_t j = blockIdx.x;
    const int t = threadIdx.x, m = (int)I.m[j];
    const uint64_t x = I.idx[j];
    const uint32_t* np = I.newpos + j * (size_t)(I.height + 1);
    // the leaf's own point and reduced blinding (every lane: cheaper than a broadcast of 36 + 8 words)
    uint32_t rl[8];
    ld8(rl, I.r + j * 8);
    rl[7] &= 0x7fffffffu;                          // Scalar::from_bits
    ge_p3 P0;
    ge_identity(P0);
    tbl_fixed_mul_add_u64(P0, tbl, tbl.row_B(0), I.v[j]);
    tbl_fixed_mul_add(P0, tbl, tbl.row_Bb(0), rl);
    sc r0;
    sc_to_mont(r0, rl);
    // lane t: the padding sibling of chain node t (needed for t < m - 1), Paddable::padding at (t, (x >> t) ^ 1)
    const bool has_pad = t < m - 1;
    ge_p3 Pp;
    ge_identity(Pp);
    sc rp;
    sc_zero(rp);
    uint32_t rB[8] = {0}, cB[8] = {0}, hB[8] = {0};
    if (has_pad) {
        uint32_t seed[8], wide[16];
        for (int q = 0; q < 8; q++) seed[q] = I.pad_seed[q];
        seed_wide(wide, seed, 1u, (uint64_t)t, (x >> t) ^ 1ull);
        sc_from_wide(rp, wide);
        sc_from_mont(rB, rp);
        tbl_fixed_mul_add(Pp, tbl, tbl.row_Bb(0), rB);
        ge_compress(cB, Pp);
        node_hash32(tbl.digest, hB, cB);
    }
    // inclusive prefix sums over the lanes: S_t = sum_{u <= t} pad(u) (points and blindings); chain node t = leaf + S_{t-1}
    ge_p3 S = Pp;
    sc rs = rp;
    for (int off = 1; off < 64; off <<= 1) {
        ge_p3 o, q;
        sc ro, rq;
        for (int i = 0; i < FE_NL; i++) {
            o.X.v[i] = __shfl_up(S.X.v[i], off, 64); o.Y.v[i] = __shfl_up(S.Y.v[i], off, 64);
            o.Z.v[i] = __shfl_up(S.Z.v[i], off, 64); o.T.v[i] = __shfl_up(S.T.v[i], off, 64);
        }
        for (int i = 0; i < 8; i++) ro.v[i] = (uint32_t)__shfl_up((int)rs.v[i], off, 64);
        if (t >= off) { ge_add(q, S, o); S = q; sc_add(rq, rs, ro); rs = rq; }
    }
    ge_p3 E;                                        // exclusive: S_{t-1}
    sc re;
    for (int i = 0; i < FE_NL; i++) {
        E.X.v[i] = __shfl_up(S.X.v[i], 1, 64); E.Y.v[i] = __shfl_up(S.Y.v[i], 1, 64);
        E.Z.v[i] = __shfl_up(S.Z.v[i], 1, 64); E.T.v[i] = __shfl_up(S.T.v[i], 1, 64);
    }
    for (int i = 0; i < 8; i++) re.v[i] = (uint32_t)__shfl_up((int)rs.v[i], 1, 64);
    if (t >= m) return;
    ge_p3 N;
    sc rn;
    if (t == 0) { N = P0; rn = r0; }
    else { ge_add(N, P0, E); sc_add(rn, r0, re); }
    uint32_t cN[8], hN[8], rN[8];
    ge_compress(cN, N);
    sc_from_mont(rN, rn);
    const LevelView L = views[t];
    const size_t d = np[t];
    L.idx[d] = x >> t;
    L.v[d] = I.v[j];
    st8(L.r + d * 8, t == 0 ? rl : rN);             // a leaf keeps its blinding as given (possibly >= l); parents hold the reduced sum
    st8(L.C + d * 8, cN);
    if (t == 0) { node_hash32(tbl.digest, hN, cN); st8(L.H + d * 8, hN); }       // (parents' hashes: k_tree_upd_hash, level by level)
    L.has_pad[d] = has_pad ? 1 : 0;
    if (has_pad) { st8(L.padC + d * 8, cB); st8(L.padH + d * 8, hB); st8(L.padr + d * 8, rB); }
    L.parent[d] = np[t + 1];
    uint32_t* pos = I.pos + j * (size_t)(I.height + 1);
    pos[t] = (uint32_t)d;
    if (t == m - 1) {
        // the sibling S of the chain's top node is real and adjacent in the sorted level; its padding sibling goes away
        const size_t s = ((x >> t) & 1ull) ? d - 1 : d + 1;
        uint32_t oc[8], orr[8];
        ld8(oc, L.padC + s * 8);
        ld8(orr, L.padr + s * 8);
        L.has_pad[s] = 0;
        ge_p3 po, npo, dd;
        (void)ge_decompress(po, oc);
        ge_neg(npo, po);
        ge_add(dd, N, npo);
        st_p3(I.dP + j * 40, dd);
        sc mo, md;
        sc_to_mont(mo, orr);
        sc_sub(md, rn, mo);
        for (int i = 0; i < 8; i++) I.dr[j * 8 + i] = md.v[i];
        I.dv[j] = I.v[j];
        // positions of the existing ancestors, by the (already moved) parent pointers
        size_t p = np[m];
        pos[m] = (uint32_t)p;
        for (int q = m; q < I.height; q++) { p = views[q].parent[p]; pos[q + 1] = (uint32_t)p; }
    }
}

// ------------------------------------------------------------------------------------- small trees, by phases
// k_tree_merge makes a level in one pass, which is right for millions of nodes (one trip through HBM) and wrong for a tree of a
// few thousand (the reference's `build` criterion group, benches/dapol.rs:24-57; dapol_tree_update): a level is then ONE lane's
// chain -- a padding node's fixed-base product and encoding, the parent's encoding, the hashes: 0.16 ms -- times `height`.  Only
// three things really are sequential over the levels: who is whose parent (integers), the parents' POINTS (one addition per
// level) and the hash chain; the fixed-base products and the inverse square roots of all levels can run side by side.  So:
//   S  k_tree_structure_small   one block walks all levels: parent slots, pairs, index of every parent          (integers)
//   P  k_tree_padding_all       every padding node of every level: blinding, point, encoding, hash              (one launch)
//   M  k_tree_sum_level         level by level: v, r and the extended point of every parent                     (height launches)
//   C  k_tree_compress_all      the encodings of all parents of all levels                                      (one launch)
//   H  k_tree_hash_level        level by level: the parents' hashes                                             (height launches)
// The arrays they leave are exactly k_tree_merge's.
enum { TREE_SMALL_MAX = 8192, TREE_SMALL_PER = 8 };   // leaves; nodes per thread of the structure block (1024 threads)

__global__ __launch_bounds__(1024) void k_tree_structure_small(int height, const LevelView* views, uint32_t* cnt) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint64_t sidx[2][TREE_SMALL_MAX];                     // the level's node indexes and the next level's (128 KB of LDS)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t cur_n = cnt[0];
    {
        const uint64_t* idx0 = views[0].idx;
        for (uint32_t i = (uint32_t)t; i < cur_n; i += 1024) sidx[0][i] = idx0[i];
    }
    __syncthreads();
    for (int k = 0; k < height; k++) {
        const LevelView cur = views[k], nxt = views[k + 1];
        const uint64_t* src = sidx[k & 1];
        uint64_t* dst = sidx[(k + 1) & 1];
        const uint32_t base = (uint32_t)t * TREE_SMALL_PER;
        uint64_t v[TREE_SMALL_PER + 1];
        for (int e = 0; e <= TREE_SMALL_PER; e++) v[e] = (base + e < cur_n) ? src[base + e] : ~0ull;
        const uint64_t prev = (base > 0 && base <= cur_n) ? src[base - 1] : ~0ull;
        uint32_t f = 0, local = 0;                                   // bit e: node base + e starts a new parent
        for (int e = 0; e < TREE_SMALL_PER; e++) {
            const uint32_t i = base + e;
            const uint64_t before = e ? v[e - 1] : prev;
            if (i < cur_n && (i == 0 || (v[e] >> 1) != (before >> 1))) { f |= 1u << e; local++; }
        }
        uint32_t x = local;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wave_tot[wv] = x;
        __syncthreads();
        uint32_t add = 0, all = 0;
        for (int w = 0; w < 16; w++) { if (w < wv) add += wave_tot[w]; all += wave_tot[w]; }
        uint32_t q = x - local + add;                                // parents started before this thread's nodes
        for (int e = 0; e < TREE_SMALL_PER; e++) {
            const uint32_t i = base + e;
            if (i >= cur_n) break;
            if ((f >> e) & 1u) {
                const bool pair = (i + 1 < cur_n) && v[e + 1] == (v[e] ^ 1ull);
                cur.parent[i] = q;
                cur.has_pad[i] = pair ? 0 : 1;
                nxt.idx[q] = v[e] >> 1;
                dst[q] = v[e] >> 1;
                q++;
            } else {                                                 // the second node of a pair: its sibling's parent
                cur.parent[i] = q - 1;
                cur.has_pad[i] = 0;
            }
        }
        cur_n = all;
        if (t == 0) cnt[k + 1] = all;
        __syncthreads();                                             // dst is the next level's src; wave_tot is reused
    }
}
// Which level a flattened node number belongs to (lvl_off[k] = nodes below level k by the host-side bounds).
__device__ __forceinline__ int tree_level_of(size_t t, const uint32_t* lvl_off, int nlev) {
    int k = 0;
    while (k + 1 < nlev && t >= lvl_off[k + 1]) k++;
    return k;
}
__global__ __launch_bounds__(64) void k_tree_padding_all(TableView tbl, const LevelView* views, int height, const uint32_t* cnt, const uint32_t* lvl_off,
                                                        const uint32_t* pad_seed, int32_t* extpad) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= lvl_off[height]) return;                                // (the root level has no sibling)
    const int level = tree_level_of(t, lvl_off, height);
    const size_t i = t - lvl_off[level];
    const LevelView cur = views[level];
    if (i >= cnt[level] || !cur.has_pad[i]) return;
    uint32_t seed[8], wide[16], rB[8], cB[8], hB[8];
    for (int k = 0; k < 8; k++) seed[k] = pad_seed[k];
    seed_wide(wide, seed, 1u, (uint64_t)level, cur.idx[i] ^ 1ull);  // Paddable::padding with the positional blinding (k_tree_merge)
    sc rm;
    sc_from_wide(rm, wide);
    sc_from_mont(rB, rm);
    ge_p3 pB;
    ge_identity(pB);
    tbl_fixed_mul_add(pB, tbl, tbl.row_Bb(0), rB);
    ge_compress(cB, pB);
    node_hash32(tbl.digest, hB, cB);
    st8(cur.padC + i * 8, cB);
    st8(cur.padH + i * 8, hB);
    st8(cur.padr + i * 8, rB);
    st_p3(extpad + t * 40, pB);
}
__global__ __launch_bounds__(64) void k_tree_sum_level(LevelView cur, LevelView nxt, int level, const uint32_t* cnt, const int32_t* ext_cur,
                                                      const int32_t* extpad_cur, int32_t* ext_nxt) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= cnt[level]) return;
    const uint32_t q = cur.parent[i];
    if (i > 0 && cur.parent[i - 1] == q) return;                    // the second node of a pair
    const bool pair = !cur.has_pad[i];
    uint32_t rA[8], rB[8], rp[8];
    ge_p3 pA, pB, pp;
    ld8(rA, cur.r + i * 8);
    ld_p3(pA, ext_cur + i * 40);
    uint64_t vB = 0;
    if (pair) {
        ld8(rB, cur.r + (i + 1) * 8);
        ld_p3(pB, ext_cur + (i + 1) * 40);
        vB = cur.v[i + 1];
    } else {
        ld8(rB, cur.padr + i * 8);
        ld_p3(pB, extpad_cur + i * 40);
    }
    sc ma, mb, ms;                                                   // r = (r_L + r_R) mod l  (node.rs:75)
    sc_to_mont(ma, rA);
    sc_to_mont(mb, rB);
    sc_add(ms, ma, mb);
    sc_from_mont(rp, ms);
    ge_add(pp, pA, pB);
    nxt.v[q] = cur.v[i] + vB;                                        // u64 wrap == release-mode Rust (node.rs:72)
    st8(nxt.r + (size_t)q * 8, rp);
    st_p3(ext_nxt + (size_t)q * 40, pp);
}
__global__ __launch_bounds__(64) void k_tree_compress_all(const LevelView* views, int height, const uint32_t* cnt, const uint32_t* lvl_off,
                                                         const int32_t* ext_all) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x + lvl_off[1];       // levels 1 .. height
    if (t >= lvl_off[height + 1]) return;
    const int level = tree_level_of(t, lvl_off, height + 1);
    const size_t q = t - lvl_off[level];
    if (q >= cnt[level]) return;
    ge_p3 p;
    uint32_t c[8];
    ld_p3(p, ext_all + t * 40);
    ge_compress(c, p);
    st8(views[level].C + q * 8, c);
}
__global__ __launch_bounds__(64) void k_tree_hash_level(int digest, LevelView cur, LevelView nxt, int level, const uint32_t* cnt) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= cnt[level]) return;
    const uint32_t q = cur.parent[i];
    if (i > 0 && cur.parent[i - 1] == q) return;
    const bool pair = !cur.has_pad[i];
    uint32_t cA[8], hA[8], cB[8], hB[8], hp[8];
    ld8(cA, cur.C + i * 8);
    ld8(hA, cur.H + i * 8);
    if (pair) { ld8(cB, cur.C + (i + 1) * 8); ld8(hB, cur.H + (i + 1) * 8); }
    else { ld8(cB, cur.padC + i * 8); ld8(hB, cur.padH + i * 8); }
    if (pair || (cur.idx[i] & 1ull) == 0) node_hash128(digest, hp, cA, cB, hA, hB);       // Mergeable::merge (node.rs:64-80)
    else node_hash128(digest, hp, cB, cA, hB, hA);
    st8(nxt.H + (size_t)q * 8, hp);
}

// Validation of the leaf index array: strictly increasing and below 2^height (smtree panics otherwise).
__global__ void k_tree_check_leaves(size_t n, const uint64_t* idx, int index_bits, int levels, uint32_t* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool b = (index_bits < 64 && (idx[i] >> index_bits) != 0) || (i > 0 && idx[i] <= idx[i - 1]);
    if (levels < 64 && (idx[i] >> levels) != (idx[0] >> levels)) b = true;     // all leaves in one shard subtree
    if (b) atomicOr(bad, 1u);
}

// Leaf lookup + sibling gather, one lane per (proof, level).  Output party order: root side first (or leaf first, see k_tree_path_level).
struct PathOut {
    uint32_t* C;     // [b][height][8] or null
    uint32_t* H;
    uint64_t* v;     // [b][height]
    uint32_t* r;     // [b][height][8]
};
__global__ void k_tree_find_leaves(size_t b, const uint64_t* want, size_t n, const uint64_t* idx, uint32_t* pos, uint32_t* missing) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    uint64_t w = want[t];
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (idx[mid] < w) lo = mid + 1; else hi = mid;
    }
    if (lo < n && idx[lo] == w) pos[t] = (uint32_t)lo;
    else { pos[t] = 0xffffffffu; atomicOr(missing, 1u); }
}
// Walks each proof's leaf up to the root: writes the sibling met at every level (one launch for the whole path; a lane's
// chain is `height` dependent gathers, the lanes of a launch cover the proofs).
__global__ __launch_bounds__(64) void k_tree_path_walk(size_t b, uint32_t* pos, const LevelView* views, int height, int n_upper, int leaf_first, PathOut out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    uint32_t p = pos[t];
    if (p == 0xffffffffu) return;
    for (int level = 0; level < height; level++) {
        const LevelView lv = views[level];
        // sibling order of a proof (dapol_wire_config.siblings_leaf_first): root side first (slot 0 = the root's child) or leaf first
        size_t slot = t * (size_t)(height + n_upper) + (size_t)(leaf_first ? level : n_upper + height - 1 - level);
        uint32_t c[8], h[8], r[8];
        uint64_t v = 0;
        if (lv.has_pad[p]) {
            ld8(c, lv.padC + (size_t)p * 8);
            ld8(h, lv.padH + (size_t)p * 8);
            ld8(r, lv.padr + (size_t)p * 8);
        } else {
            size_t s = (lv.idx[p] & 1ull) ? (size_t)p - 1 : (size_t)p + 1;
            ld8(c, lv.C + s * 8);
            ld8(h, lv.H + s * 8);
            ld8(r, lv.r + s * 8);
            v = lv.v[s];
        }
        if (out.C) st8(out.C + slot * 8, c);
        if (out.H) st8(out.H + slot * 8, h);
        if (out.v) out.v[slot] = v;
        if (out.r) st8(out.r + slot * 8, r);
        p = lv.parent[p];
    }
    pos[t] = p;
}
// Siblings above a shard root are the same for every leaf of the shard: broadcast them into slots [0, n_upper).
__global__ void k_tree_path_upper(size_t b, int height, int n_upper, int leaf_first, const uint32_t* uC, const uint32_t* uH, const uint64_t* uv,
                                  const uint32_t* ur, PathOut out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * (size_t)n_upper) return;
    size_t e = t / n_upper;
    int u = (int)(t - e * n_upper);
    size_t slot = e * (size_t)(height + n_upper) + (size_t)(leaf_first ? height + (n_upper - 1 - u) : u);     // u = 0: the root's child
    uint32_t w[8];
    if (out.C) { ld8(w, uC + u * 8); st8(out.C + slot * 8, w); }
    if (out.H) { ld8(w, uH + u * 8); st8(out.H + slot * 8, w); }
    if (out.v) out.v[slot] = uv[u];
    if (out.r) { ld8(w, ur + u * 8); st8(out.r + slot * 8, w); }
}

// ------------------------------------------------------------------------------------- 64-byte node hashes (Blake2b)
// Dapol<blake2::Blake2b, R> on the new_blank + build path (src/tests.rs:100-106; mod.rs:196-208 has no digest check).  Everything
// of a node but its hash -- structure, values, blindings, commitments, padding nodes -- is independent of the digest, so a
// context with a 64-byte digest builds (and updates) its trees with the kernels above and then lays the hash chain over them:
// per level an array of 16-word hashes for the real nodes and one for their padding siblings (WideView), filled bottom-up, one
// launch per level.  The 8-word H arrays of such a tree are not used.
struct WideView {
    uint32_t* H;       // [n][16]  hash of real node i
    uint32_t* padH;    // [n][16]  hash of the padding sibling of real node i (where has_pad[i])
};
__device__ __forceinline__ void ld16(uint32_t* w, const uint32_t* p) { ld8(w, p); ld8(w + 8, p + 8); }
__device__ __forceinline__ void st16(uint32_t* p, const uint32_t* w) { st8(p, w); st8(p + 8, w + 8); }
// leaves: H = D(C)  (DapolNode::new, src/dapol/node.rs:34-36)
__global__ __launch_bounds__(256) void k_wide_hash_leaves(size_t n, const uint32_t* C, uint32_t* H16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c[8], h[16];
    ld8(c, C + i * 8);
    blake2b_hash32(h, c);
    st16(H16 + i * 16, h);
}
// level k -> k + 1: one lane per child that owns its parent (the left child of a real pair, or the real child of a padded pair):
// the padding sibling's hash D(C_pad), then Mergeable::merge's D(C_L || C_R || H_L || H_R) (node.rs:66-77).
__global__ __launch_bounds__(256) void k_wide_hash_level(LevelView cur, WideView wcur, WideView wnxt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cur.n) return;
    const bool left = (cur.idx[i] & 1ull) == 0, padded = cur.has_pad[i] != 0;
    if (!padded && !left) return;                          // the right child of a real pair: its left neighbour does the parent
    uint32_t cA[8], cB[8], hA[16], hB[16], hp[16];
    ld8(cA, cur.C + i * 8);
    ld16(hA, wcur.H + i * 16);
    if (padded) {
        ld8(cB, cur.padC + i * 8);
        blake2b_hash32(hB, cB);
        st16(wcur.padH + i * 16, hB);
    } else {
        ld8(cB, cur.C + (i + 1) * 8);
        ld16(hB, wcur.H + (i + 1) * 16);
    }
    if (left) blake2b_hash192(hp, cA, cB, hA, hB);
    else blake2b_hash192(hp, cB, cA, hB, hA);
    st16(wnxt.H + (size_t)cur.parent[i] * 16, hp);
}
// the 64-byte hashes of the siblings along b paths (k_tree_path_walk's walk; pos = the leaves' positions, left untouched)
__global__ __launch_bounds__(64) void k_wide_path_walk(size_t b, const uint32_t* pos, const LevelView* views, const WideView* wviews, int height, int n_upper,
                                                      int leaf_first, uint32_t* outH16) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    uint32_t p = pos[t];
    if (p == 0xffffffffu) return;
    for (int level = 0; level < height; level++) {
        const LevelView lv = views[level];
        const WideView wv = wviews[level];
        const size_t slot = t * (size_t)(height + n_upper) + (size_t)(leaf_first ? level : n_upper + height - 1 - level);
        uint32_t h[16];
        if (lv.has_pad[p]) ld16(h, wv.padH + (size_t)p * 16);
        else ld16(h, wv.H + ((lv.idx[p] & 1ull) ? (size_t)p - 1 : (size_t)p + 1) * 16);
        st16(outH16 + slot * 16, h);
        p = lv.parent[p];
    }
}

}  // namespace dapol
