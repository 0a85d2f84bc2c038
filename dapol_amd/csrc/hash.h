// Hash primitives of the proving path: BLAKE3 compression (node hash D = blake3::Hasher, benches/dapol.rs:38,
// and the seed-mode randomness PRF), Keccak-f[1600] with STROBE-128 / Merlin (Transcript::new(&[]) at
// src/range/mod.rs:51,67,86,105), SHAKE256 (bulletproofs GeneratorsChain) and SHA3-512 (PedersenGens B_blinding).
#pragma once
#include "labels.h"
#include <stdint.h>
#include "fe.h"

namespace dapol {

// ------------------------------------------------------------------------------------------------ BLAKE3
enum : uint32_t { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8, B3_KEYED_HASH = 16 };

DAPOL_HD uint32_t rotr32(uint32_t x, int n) {
    return (x >> n) | (x << (32 - n));
}

#define B3_G(a, b, c, d, mx, my)      \
    a = a + b + (mx);                 \
    d = rotr32(d ^ a, 16);            \
    c = c + d;                        \
    b = rotr32(b ^ c, 12);            \
    a = a + b + (my);                 \
    d = rotr32(d ^ a, 8);             \
    c = c + d;                        \
    b = rotr32(b ^ c, 7);

// One BLAKE3 compression.  out16 receives the full 16-word output (first 8 = chaining value / hash).
DAPOL_HD void blake3_compress(uint32_t* out16, const uint32_t* cv, const uint32_t* m_in, uint64_t counter, uint32_t block_len,
                              uint32_t flags) {
    uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
    uint32_t s8 = 0x6A09E667u, s9 = 0xBB67AE85u, s10 = 0x3C6EF372u, s11 = 0xA54FF53Au;
    uint32_t s12 = (uint32_t)counter, s13 = (uint32_t)(counter >> 32), s14 = block_len, s15 = flags;
    uint32_t m[16];
    for (int i = 0; i < 16; i++) m[i] = m_in[i];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        B3_G(s0, s4, s8, s12, m[0], m[1]);
        B3_G(s1, s5, s9, s13, m[2], m[3]);
        B3_G(s2, s6, s10, s14, m[4], m[5]);
        B3_G(s3, s7, s11, s15, m[6], m[7]);
        B3_G(s0, s5, s10, s15, m[8], m[9]);
        B3_G(s1, s6, s11, s12, m[10], m[11]);
        B3_G(s2, s7, s8, s13, m[12], m[13]);
        B3_G(s3, s4, s9, s14, m[14], m[15]);
        if (r < 6) {
            uint32_t t[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
            for (int i = 0; i < 16; i++) m[i] = t[i];
        }
    }
    out16[0] = s0 ^ s8;  out16[1] = s1 ^ s9;  out16[2] = s2 ^ s10; out16[3] = s3 ^ s11;
    out16[4] = s4 ^ s12; out16[5] = s5 ^ s13; out16[6] = s6 ^ s14; out16[7] = s7 ^ s15;
    out16[8] = s8 ^ cv[0];   out16[9] = s9 ^ cv[1];   out16[10] = s10 ^ cv[2]; out16[11] = s11 ^ cv[3];
    out16[12] = s12 ^ cv[4]; out16[13] = s13 ^ cv[5]; out16[14] = s14 ^ cv[6]; out16[15] = s15 ^ cv[7];
}
#undef B3_G

DAPOL_HD void blake3_iv(uint32_t* cv) {
    cv[0] = 0x6A09E667u; cv[1] = 0xBB67AE85u; cv[2] = 0x3C6EF372u; cv[3] = 0xA54FF53Au;
    cv[4] = 0x510E527Fu; cv[5] = 0x9B05688Cu; cv[6] = 0x1F83D9ABu; cv[7] = 0x5BE0CD19u;
}

// Leaf node hash: BLAKE3(C) for one 32-byte compressed commitment (src/dapol/node.rs:34-36).
DAPOL_HD void blake3_hash32(uint32_t* out8, const uint32_t* c8) {
    uint32_t cv[8], m[16], o[16];
    blake3_iv(cv);
    for (int i = 0; i < 8; i++) m[i] = c8[i];
    for (int i = 8; i < 16; i++) m[i] = 0;
    blake3_compress(o, cv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
    for (int i = 0; i < 8; i++) out8[i] = o[i];
}
// Parent node hash: BLAKE3(C_L || C_R || H_L || H_R), 128 bytes = two blocks of one chunk (node.rs:66-77).
DAPOL_HD void blake3_hash128(uint32_t* out8, const uint32_t* cl, const uint32_t* cr, const uint32_t* hl, const uint32_t* hr) {
    uint32_t cv[8], m[16], o[16];
    blake3_iv(cv);
    for (int i = 0; i < 8; i++) { m[i] = cl[i]; m[8 + i] = cr[i]; }
    blake3_compress(o, cv, m, 0, 64, B3_CHUNK_START);
    for (int i = 0; i < 8; i++) { cv[i] = o[i]; m[i] = hl[i]; m[8 + i] = hr[i]; }
    blake3_compress(o, cv, m, 0, 64, B3_CHUNK_END | B3_ROOT);
    for (int i = 0; i < 8; i++) out8[i] = o[i];
}
// Seed-mode randomness: 64 bytes = first XOF block of BLAKE3-keyed(seed, LE32 domain | LE64 a | LE64 b).
DAPOL_HD void seed_wide(uint32_t* out16, const uint32_t* seed8, uint32_t domain, uint64_t a, uint64_t b) {
    uint32_t m[16];
    for (int i = 0; i < 16; i++) m[i] = 0;
    m[0] = domain; m[1] = (uint32_t)a; m[2] = (uint32_t)(a >> 32); m[3] = (uint32_t)b; m[4] = (uint32_t)(b >> 32);
    blake3_compress(out16, seed8, m, 0, 20, B3_KEYED_HASH | B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
}

// Streaming digest D over a byte string assembled from parts (leaf derivation, src/dapol/mod.rs:323-441):
// D = BLAKE3 (any length: the chunk chaining values of inputs beyond 1024 bytes go through the caller's stack, see
// dg_init_long) or Blake2s-256 (the reference's KAT digest, src/dapol/tests.rs:13,21).
// Usage: dg_init / dg_init_long, dg_update* (bytes), dg_final -> eight little-endian words.
enum : int { DG_BLAKE3 = 0, DG_BLAKE2S = 1, DG_BLAKE2B = 2 };   // (Blake2b: node hashes only, 64 bytes -- see "wide node hashes" below)
enum : int { B3_STACK_DEPTH = 24 };   // subtree chaining values of up to 2^24 chunks = 16 GiB (inputs here are < 2^32 + 600 bytes)
struct Digest {
    uint32_t h[8];
    uint32_t buf[16];
    uint32_t buflen;      // bytes in buf
    uint32_t total;       // bytes compressed so far (before buf); for BLAKE3: within the current chunk
    int kind;
    bool overflow;        // BLAKE3 input longer than one chunk on a digest without a stack (dg_init)
    uint32_t chunks;      // BLAKE3: chunks completed so far (= the current chunk's counter)
    uint32_t sp;          // BLAKE3: chaining values on the stack
    uint32_t* stack;      // BLAKE3: [B3_STACK_DEPTH][8] owned by the caller, or nullptr (single-chunk inputs only)
};
DAPOL_HD void blake2s_compress(uint32_t* h, const uint32_t* m, uint32_t t, bool last) {
    const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    const uint8_t SIG[10][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                 {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                 {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                 {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                 {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    uint32_t v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y)                \
    v[a] = v[a] + v[b] + (x);                \
    v[d] = rotr32(v[d] ^ v[a], 16);          \
    v[c] = v[c] + v[d];                      \
    v[b] = rotr32(v[b] ^ v[c], 12);          \
    v[a] = v[a] + v[b] + (y);                \
    v[d] = rotr32(v[d] ^ v[a], 8);           \
    v[c] = v[c] + v[d];                      \
    v[b] = rotr32(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
        const uint8_t* s = SIG[r];
        B2G(0, 4, 8, 12, m[s[0]], m[s[1]]) B2G(1, 5, 9, 13, m[s[2]], m[s[3]]) B2G(2, 6, 10, 14, m[s[4]], m[s[5]]) B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
        B2G(0, 5, 10, 15, m[s[8]], m[s[9]]) B2G(1, 6, 11, 12, m[s[10]], m[s[11]]) B2G(2, 7, 8, 13, m[s[12]], m[s[13]]) B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef B2G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
}
// Node hashes for D = Blake2s-256 (Dapol<blake2::Blake2s, _>, src/dapol/tests.rs:21,33): the same byte strings as
// the BLAKE3 ones above, 32 bytes = one block, 128 bytes = two.
DAPOL_HD void blake2s_hash32(uint32_t* out8, const uint32_t* c8) {
    uint32_t h[8], m[16];
    blake3_iv(h);
    h[0] ^= 0x01010020u;
    for (int i = 0; i < 8; i++) m[i] = c8[i];
    for (int i = 8; i < 16; i++) m[i] = 0;
    blake2s_compress(h, m, 32, true);
    for (int i = 0; i < 8; i++) out8[i] = h[i];
}
DAPOL_HD void blake2s_hash128(uint32_t* out8, const uint32_t* cl, const uint32_t* cr, const uint32_t* hl, const uint32_t* hr) {
    uint32_t h[8], m[16];
    blake3_iv(h);
    h[0] ^= 0x01010020u;
    for (int i = 0; i < 8; i++) { m[i] = cl[i]; m[8 + i] = cr[i]; }
    blake2s_compress(h, m, 64, false);
    for (int i = 0; i < 8; i++) { m[i] = hl[i]; m[8 + i] = hr[i]; }
    blake2s_compress(h, m, 128, true);
    for (int i = 0; i < 8; i++) out8[i] = h[i];
}
// The node hash of the context's digest D (DG_BLAKE3 / DG_BLAKE2S): leaf = D(C) (src/dapol/node.rs:34-36),
// parent = D(C_L || C_R || H_L || H_R) (node.rs:66-77, src/proof/node.rs:58-64).
// On a 64-byte-digest context (DG_BLAKE2B) the 8-word hash arrays are NOT the node hashes -- the chain is laid over the built tree by
// tree_hash_wide, 16 words per node -- so the kernels that fill them as they go write a recognisable poison instead of hashing with
// some other digest: no work for a value nobody may read, and a kernel that consumed the wrong array would stand out.
DAPOL_HD void node_hash32(int kind, uint32_t* out8, const uint32_t* c8) {
    if (kind == DG_BLAKE2S) blake2s_hash32(out8, c8);
    else if (kind == DG_BLAKE2B) { for (int i = 0; i < 8; i++) out8[i] = 0xB2B2B2B2u; }
    else blake3_hash32(out8, c8);
}
DAPOL_HD void node_hash128(int kind, uint32_t* out8, const uint32_t* cl, const uint32_t* cr, const uint32_t* hl, const uint32_t* hr) {
    if (kind == DG_BLAKE2S) blake2s_hash128(out8, cl, cr, hl, hr);
    else if (kind == DG_BLAKE2B) { for (int i = 0; i < 8; i++) out8[i] = 0xB2B2B2B2u; }
    else blake3_hash128(out8, cl, cr, hl, hr);
}

// ---------------------------------------------------------------------------------------- wide node hashes (Blake2b-512)
// The reference's own integration test also runs Dapol<blake2::Blake2b, _> through new_blank + build + prove + verify
// (src/tests.rs:100-106; only Dapol::new checks D::output_size() == 32, src/dapol/mod.rs:101-103): node hashes of 64 bytes,
// leaf = Blake2b(C) (32 bytes: one block), parent = Blake2b(C_L || C_R || H_L || H_R) (192 bytes: a full block + 64 bytes).
// A hash is 16 little-endian 32-bit words here, like every other byte string of the library.
DAPOL_HD uint64_t rotr64(uint64_t x, int n) {
    return (x >> n) | (x << (64 - n));
}
DAPOL_HD void blake2b_compress(uint64_t* h, const uint64_t* m, uint64_t t, bool last) {
    const uint64_t IV[8] = {0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
                            0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull};
    const uint8_t SIG[10][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                 {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                 {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                 {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                 {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    uint64_t v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
    v[12] ^= t;                                    // (inputs here are < 2^64 bytes: the high counter word stays 0)
    if (last) v[14] = ~v[14];
#define B2BG(a, b, c, d, x, y)               \
    v[a] = v[a] + v[b] + (x);                \
    v[d] = rotr64(v[d] ^ v[a], 32);          \
    v[c] = v[c] + v[d];                      \
    v[b] = rotr64(v[b] ^ v[c], 24);          \
    v[a] = v[a] + v[b] + (y);                \
    v[d] = rotr64(v[d] ^ v[a], 16);          \
    v[c] = v[c] + v[d];                      \
    v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++) {
        const uint8_t* s = SIG[r % 10];
        B2BG(0, 4, 8, 12, m[s[0]], m[s[1]]) B2BG(1, 5, 9, 13, m[s[2]], m[s[3]]) B2BG(2, 6, 10, 14, m[s[4]], m[s[5]]) B2BG(3, 7, 11, 15, m[s[6]], m[s[7]])
        B2BG(0, 5, 10, 15, m[s[8]], m[s[9]]) B2BG(1, 6, 11, 12, m[s[10]], m[s[11]]) B2BG(2, 7, 8, 13, m[s[12]], m[s[13]]) B2BG(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef B2BG
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
}
DAPOL_HD void blake2b_init512(uint64_t* h) {
    const uint64_t IV[8] = {0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
                            0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull};
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010040ull;                         // parameter block: digest 64 bytes, no key, fanout 1, depth 1
}
DAPOL_HD uint64_t w64(const uint32_t* w, int i) { return (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32); }
DAPOL_HD void blake2b_hash32(uint32_t* out16, const uint32_t* c8) {
    uint64_t h[8], m[16];
    blake2b_init512(h);
    for (int i = 0; i < 4; i++) m[i] = w64(c8, i);
    for (int i = 4; i < 16; i++) m[i] = 0;
    blake2b_compress(h, m, 32, true);
    for (int i = 0; i < 8; i++) { out16[2 * i] = (uint32_t)h[i]; out16[2 * i + 1] = (uint32_t)(h[i] >> 32); }
}
DAPOL_HD void blake2b_hash192(uint32_t* out16, const uint32_t* cl, const uint32_t* cr, const uint32_t* hl16, const uint32_t* hr16) {
    uint64_t h[8], m[16];
    blake2b_init512(h);
    for (int i = 0; i < 4; i++) { m[i] = w64(cl, i); m[4 + i] = w64(cr, i); }
    for (int i = 0; i < 8; i++) m[8 + i] = w64(hl16, i);
    blake2b_compress(h, m, 128, false);
    for (int i = 0; i < 8; i++) { m[i] = w64(hr16, i); m[8 + i] = 0; }
    blake2b_compress(h, m, 192, true);
    for (int i = 0; i < 8; i++) { out16[2 * i] = (uint32_t)h[i]; out16[2 * i + 1] = (uint32_t)(h[i] >> 32); }
}
// Words of one node hash of digest `kind`, and the node hash over HW-word hashes: HW = 8 -> the 32-byte digests above (by kind),
// HW = 16 -> Blake2b-512.  Kernels that carry hashes are templates over HW; the host picks the instantiation from the context.
DAPOL_HD int dg_hash_words(int kind) { return kind == DG_BLAKE2B ? 16 : 8; }
template <int HW>
DAPOL_HD void node_hash_leaf_w(int kind, uint32_t* out, const uint32_t* c8) {
    if (HW == 16) blake2b_hash32(out, c8);
    else node_hash32(kind, out, c8);
}
template <int HW>
DAPOL_HD void node_hash_parent_w(int kind, uint32_t* out, const uint32_t* cl, const uint32_t* cr, const uint32_t* hl, const uint32_t* hr) {
    if (HW == 16) blake2b_hash192(out, cl, cr, hl, hr);
    else node_hash128(kind, out, cl, cr, hl, hr);
}

DAPOL_HD void dg_init(Digest& d, int kind) {
    blake3_iv(d.h);                        // both digests start from the SHA-256 IV ...
    if (kind == DG_BLAKE2S) d.h[0] ^= 0x01010020u;   // ... Blake2s xors the parameter block (digest 32, fanout 1, depth 1)
    for (int i = 0; i < 16; i++) d.buf[i] = 0;
    d.buflen = 0;
    d.total = 0;
    d.kind = kind;
    d.overflow = false;
    d.chunks = 0;
    d.sp = 0;
    d.stack = nullptr;
}
// A digest whose BLAKE3 input may exceed one 1024-byte chunk: stack = B3_STACK_DEPTH * 8 words of the caller's.
DAPOL_HD void dg_init_long(Digest& d, int kind, uint32_t* stack) {
    dg_init(d, kind);
    d.stack = stack;
}
// BLAKE3 parent node: chaining value of (left || right)
DAPOL_HD void blake3_parent(uint32_t* out16, const uint32_t* left8, const uint32_t* right8, uint32_t flags) {
    uint32_t cv[8], m[16];
    blake3_iv(cv);
    for (int i = 0; i < 8; i++) { m[i] = left8[i]; m[8 + i] = right8[i]; }
    blake3_compress(out16, cv, m, 0, 64, B3_PARENT | flags);
}
DAPOL_HD void dg_flush(Digest& d) {        // compress a FULL, non-final block
    if (d.kind == DG_BLAKE3) {
        uint32_t o[16];
        const bool chunk_end = d.total + 64 >= 1024;   // more input follows (the flush is lazy), so this chunk is not the last one
        blake3_compress(o, d.h, d.buf, d.chunks, 64, (d.total == 0 ? B3_CHUNK_START : 0u) | (chunk_end ? B3_CHUNK_END : 0u));
        if (!chunk_end) {
            for (int i = 0; i < 8; i++) d.h[i] = o[i];
            d.total += 64;
        } else {
            // the chunk's chaining value joins the stack of subtree roots: one merge per trailing zero bit of the new chunk count
            d.chunks++;
            if (!d.stack) d.overflow = true;
            else {
                for (uint32_t t = d.chunks; (t & 1u) == 0 && d.sp > 0; t >>= 1) {
                    d.sp--;
                    uint32_t p[16];
                    blake3_parent(p, d.stack + 8 * d.sp, o, 0);
                    for (int i = 0; i < 8; i++) o[i] = p[i];
                }
                if (d.sp < (uint32_t)B3_STACK_DEPTH) { for (int i = 0; i < 8; i++) d.stack[8 * d.sp + i] = o[i]; d.sp++; }
                else d.overflow = true;
            }
            blake3_iv(d.h);
            d.total = 0;
        }
    } else {
        blake2s_compress(d.h, d.buf, d.total + 64, false);
        d.total += 64;
    }
    d.buflen = 0;
    for (int i = 0; i < 16; i++) d.buf[i] = 0;
}
DAPOL_HD void dg_update_byte(Digest& d, uint8_t b) {
    if (d.buflen == 64) dg_flush(d);       // lazily, so that the last block stays in the buffer for dg_final
    d.buf[d.buflen >> 2] |= (uint32_t)b << (8 * (d.buflen & 3));
    d.buflen++;
}
DAPOL_HD void dg_update(Digest& d, const uint8_t* p, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) dg_update_byte(d, p[i]);
}
DAPOL_HD void dg_update_words(Digest& d, const uint32_t* w, int nwords) {
    for (int i = 0; i < nwords; i++)
        for (int k = 0; k < 4; k++) dg_update_byte(d, (uint8_t)(w[i] >> (8 * k)));
}
DAPOL_HD void dg_final(Digest& d, uint32_t* out8) {
    if (d.kind == DG_BLAKE3) {
        uint32_t o[16];
        const uint32_t flags = (d.total == 0 ? B3_CHUNK_START : 0u) | B3_CHUNK_END;
        if (d.sp == 0) {
            blake3_compress(o, d.h, d.buf, d.chunks, d.buflen, flags | B3_ROOT);       // a single chunk is its own root
        } else {
            blake3_compress(o, d.h, d.buf, d.chunks, d.buflen, flags);                 // the last chunk, then up the right edge of the tree
            while (d.sp > 0) {
                d.sp--;
                uint32_t p[16];
                blake3_parent(p, d.stack + 8 * d.sp, o, d.sp == 0 ? B3_ROOT : 0u);
                for (int i = 0; i < 8; i++) o[i] = p[i];
            }
        }
        for (int i = 0; i < 8; i++) out8[i] = o[i];
    } else {
        blake2s_compress(d.h, d.buf, d.total + d.buflen, true);
        for (int i = 0; i < 8; i++) out8[i] = d.h[i];
    }
}

// ------------------------------------------------------------------------------------------- Keccak-f[1600]
DAPOL_HD uint64_t rotl64(uint64_t x, int n) {
    return (x << n) | (x >> (64 - n));
}

DAPOL_HD_NOINLINE void keccak_f1600(uint64_t* A) {
    const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                             0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                             0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                             0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                             0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                             0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    uint64_t a00 = A[0], a01 = A[1], a02 = A[2], a03 = A[3], a04 = A[4], a05 = A[5], a06 = A[6], a07 = A[7], a08 = A[8], a09 = A[9],
             a10 = A[10], a11 = A[11], a12 = A[12], a13 = A[13], a14 = A[14], a15 = A[15], a16 = A[16], a17 = A[17], a18 = A[18],
             a19 = A[19], a20 = A[20], a21 = A[21], a22 = A[22], a23 = A[23], a24 = A[24];
    for (int r = 0; r < 24; r++) {
        uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                 c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
        uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1),
                 d4 = c3 ^ rotl64(c0, 1);
        a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
        a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
        a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
        a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
        a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
        // rho + pi: B[y][2x+3y] = rot(A[x][y])
        uint64_t b00 = a00, b10 = rotl64(a01, 1), b20 = rotl64(a02, 62), b05 = rotl64(a03, 28), b15 = rotl64(a04, 27);
        uint64_t b16 = rotl64(a05, 36), b01 = rotl64(a06, 44), b11 = rotl64(a07, 6), b21 = rotl64(a08, 55), b06 = rotl64(a09, 20);
        uint64_t b07 = rotl64(a10, 3), b17 = rotl64(a11, 10), b02 = rotl64(a12, 43), b12 = rotl64(a13, 25), b22 = rotl64(a14, 39);
        uint64_t b23 = rotl64(a15, 41), b08 = rotl64(a16, 45), b18 = rotl64(a17, 15), b03 = rotl64(a18, 21), b13 = rotl64(a19, 8);
        uint64_t b14 = rotl64(a20, 18), b24 = rotl64(a21, 2), b09 = rotl64(a22, 61), b19 = rotl64(a23, 56), b04 = rotl64(a24, 14);
        a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
        a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
        a00 ^= RC[r];
    }
    A[0] = a00; A[1] = a01; A[2] = a02; A[3] = a03; A[4] = a04; A[5] = a05; A[6] = a06; A[7] = a07; A[8] = a08; A[9] = a09;
    A[10] = a10; A[11] = a11; A[12] = a12; A[13] = a13; A[14] = a14; A[15] = a15; A[16] = a16; A[17] = a17; A[18] = a18; A[19] = a19;
    A[20] = a20; A[21] = a21; A[22] = a22; A[23] = a23; A[24] = a24;
}

// Generic sponge on a 25-lane state (used for SHAKE256 / SHA3-512 at context creation).
struct Sponge {
    uint64_t s[25];
    uint32_t pos, rate;
};
DAPOL_HD void sponge_init(Sponge& sp, uint32_t rate) {
    for (int i = 0; i < 25; i++) sp.s[i] = 0;
    sp.pos = 0;
    sp.rate = rate;
}
DAPOL_HD void sponge_absorb_byte(Sponge& sp, uint8_t b) {
    sp.s[sp.pos >> 3] ^= (uint64_t)b << (8 * (sp.pos & 7));
    if (++sp.pos == sp.rate) {
        keccak_f1600(sp.s);
        sp.pos = 0;
    }
}
DAPOL_HD void sponge_finish(Sponge& sp, uint8_t domain) {  // 0x1F = SHAKE, 0x06 = SHA3
    sp.s[sp.pos >> 3] ^= (uint64_t)domain << (8 * (sp.pos & 7));
    sp.s[(sp.rate - 1) >> 3] ^= 0x80ull << (8 * ((sp.rate - 1) & 7));
    keccak_f1600(sp.s);
    sp.pos = 0;
}
DAPOL_HD uint8_t sponge_squeeze_byte(Sponge& sp) {
    if (sp.pos == sp.rate) {
        keccak_f1600(sp.s);
        sp.pos = 0;
    }
    uint8_t b = (uint8_t)(sp.s[sp.pos >> 3] >> (8 * (sp.pos & 7)));
    sp.pos++;
    return b;
}

// --------------------------------------------------------------------------------- STROBE-128 / Merlin 3.0.0
struct Strobe {
    uint64_t s[25];
    uint32_t pos, pos_begin;
};
enum : uint8_t { SF_I = 1, SF_A = 2, SF_C = 4, SF_T = 8, SF_M = 16, SF_K = 32 };
enum : uint32_t { STROBE_R = 166 };

DAPOL_HD void strobe_run_f(Strobe& st) {
    st.s[st.pos >> 3] ^= (uint64_t)st.pos_begin << (8 * (st.pos & 7));
    st.s[(st.pos + 1) >> 3] ^= 0x04ull << (8 * ((st.pos + 1) & 7));
    st.s[(STROBE_R + 1) >> 3] ^= 0x80ull << (8 * ((STROBE_R + 1) & 7));
    keccak_f1600(st.s);
    st.pos = 0;
    st.pos_begin = 0;
}
DAPOL_HD void strobe_absorb_byte(Strobe& st, uint8_t b) {
    st.s[st.pos >> 3] ^= (uint64_t)b << (8 * (st.pos & 7));
    if (++st.pos == STROBE_R) strobe_run_f(st);
}
DAPOL_HD uint8_t strobe_squeeze_byte(Strobe& st) {
    uint32_t sh = 8 * (st.pos & 7);
    uint8_t b = (uint8_t)(st.s[st.pos >> 3] >> sh);
    st.s[st.pos >> 3] &= ~(0xffull << sh);
    if (++st.pos == STROBE_R) strobe_run_f(st);
    return b;
}
DAPOL_HD void strobe_reset(Strobe& st) {                   // the state before Strobe128::new's permutation
    for (int i = 0; i < 25; i++) st.s[i] = 0;
    st.pos = 0;
    st.pos_begin = 0;
}

#if defined(__HIPCC__)
// The same STROBE state held by ONE WAVEFRONT: lane x + 5 y keeps word (x, y) of the Keccak state, the permutation is nine lane
// permutations and ~30 ALU operations per round instead of ~500 instructions on a lane (kernels_verify.h, k_rv_absorb_V),
// and every lane follows the byte stream in step (all 64 lanes call every function below with the same arguments).  Used by the
// calls of few proofs, where a lane's 23 us per permutation is what the caller waits for.
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
struct KeccakLanes {                 // per-lane sources of the round's permutations, as ds_bpermute byte addresses (lane << 2)
    int th1, th2, th3, th4, xp1, xp2, pi_src, cm_src, cp_src;
    uint32_t rot_sh;                 // rho: the source word's rotation amount mod 32 ...
    bool rot_swap;                   // ... and whether it is >= 32 (the halves swap first)
};
__device__ __forceinline__ void keccak_lanes_init(KeccakLanes& K, int l) {
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    if (l >= 25) { K.th1 = K.th2 = K.th3 = K.th4 = K.xp1 = K.xp2 = K.pi_src = K.cm_src = K.cp_src = l << 2; K.rot_sh = 0; K.rot_swap = false; return; }
    int x = l % 5, y = l / 5, row = 5 * y;
    K.th1 = ((l + 5) % 25) << 2; K.th2 = ((l + 10) % 25) << 2; K.th3 = ((l + 15) % 25) << 2; K.th4 = ((l + 20) % 25) << 2;
    K.xp1 = (row + (x + 1) % 5) << 2; K.xp2 = (row + (x + 2) % 5) << 2;
    // pi: B[x'][y'] = rot(A[x][y]) with (x', y') = (y, 2x + 3y): lane (x', y') pulls from y = x', x = 3 (y' - 3 x') mod 5
    int sy = x, sx = (3 * ((y - 3 * x) % 5 + 5)) % 5;
    const int src = sx + 5 * sy;
    K.pi_src = src << 2;
    K.cm_src = ((sx + 4) % 5) << 2;  // any lane of column sx - 1 / sx + 1 holds that column's parity: row 0
    K.cp_src = ((sx + 1) % 5) << 2;
    int r = 0;
    for (int i = 0; i < 25; i++) r = (i == src) ? ROT[i] : r;
    K.rot_sh = (uint32_t)r & 31u;
    K.rot_swap = r >= 32;
}
struct KeccakWord { uint32_t lo, hi; };
__device__ __forceinline__ KeccakWord keccak_pull(int addr, KeccakWord v) {
    return KeccakWord{(uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.lo), (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.hi)};
}
// Three dependent permutation stages per round: (1) the column parities and, alongside, every lane's pi source word;
// (2) the two parities theta needs for the SOURCE's column -- theta, rho and pi are then applied at the destination;
// (3) chi's two row neighbours.  A wavefront that replays a transcript is alone on its SIMD, so a round costs its latency:
// three ds_bpermute round trips + ~30 dependent ALU instructions = ~490 cycles (tools/ubench_keccak.hip,
// profiles/r09b_keccak_ubench.txt: 4.9 us per permutation; 6.2 us before round 5, when the round constant was a scalar load
// inside a lane-0 branch at the END of the round -- its latency on the critical path 24 times -- and rho two 64-bit shifts).
// Now: the constant is loaded at the top of the round and applied through a lane mask, rho is two v_alignbit on (possibly
// swapped) halves, and TRIP rounds share a loop trip.
// TRIP = rounds per loop trip: 2 where the permutation is inlined at many places of a transcript replay, 24 (no loop: 4 % fewer
// cycles per round) where one call site is the whole kernel (k_rv_absorb_V).
template <int TRIP = 2>
__device__ __forceinline__ uint64_t keccak_f1600_wave(uint64_t a64, const KeccakLanes& K, int l) {
    const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                             0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                             0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                             0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                             0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                             0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    KeccakWord a{(uint32_t)a64, (uint32_t)(a64 >> 32)};
    const uint32_t lane0 = l == 0 ? 0xffffffffu : 0u;
    const uint32_t s = (32u - K.rot_sh) & 31u;
#pragma unroll TRIP
    for (int r = 0; r < 24; r++) {
        const uint64_t rc = RC[r];
        KeccakWord as = keccak_pull(K.pi_src, a);
        const KeccakWord t1 = keccak_pull(K.th1, a), t2 = keccak_pull(K.th2, a), t3 = keccak_pull(K.th3, a), t4 = keccak_pull(K.th4, a);
        const KeccakWord c{a.lo ^ t1.lo ^ t2.lo ^ t3.lo ^ t4.lo, a.hi ^ t1.hi ^ t2.hi ^ t3.hi ^ t4.hi};       // C[x] on every lane of column x
        const KeccakWord cm = keccak_pull(K.cm_src, c), cp = keccak_pull(K.cp_src, c);
        as.lo ^= cm.lo ^ __builtin_amdgcn_alignbit(cp.lo, cp.hi, 31);                                      // D = C[x-1] ^ rotl(C[x+1], 1)
        as.hi ^= cm.hi ^ __builtin_amdgcn_alignbit(cp.hi, cp.lo, 31);
        const uint32_t l0 = K.rot_swap ? as.hi : as.lo, h0 = K.rot_swap ? as.lo : as.hi;                   // rotl by 32, then by rot_sh < 32
        const KeccakWord b{K.rot_sh ? __builtin_amdgcn_alignbit(l0, h0, s) : l0, K.rot_sh ? __builtin_amdgcn_alignbit(h0, l0, s) : h0};
        const KeccakWord b1 = keccak_pull(K.xp1, b), b2 = keccak_pull(K.xp2, b);
        a.lo = b.lo ^ (~b1.lo & b2.lo) ^ ((uint32_t)rc & lane0);
        a.hi = b.hi ^ (~b1.hi & b2.hi) ^ ((uint32_t)(rc >> 32) & lane0);
    }
    return ((uint64_t)a.hi << 32) | a.lo;
}
struct WStrobe {
    uint64_t a;                      // this lane's state word (lanes 25-63 carry zeros that nothing reads)
    uint32_t pos, pos_begin;
    int l;
    KeccakLanes K;
};
__device__ __forceinline__ void wstrobe_lanes(WStrobe& st, int lane) {
    st.l = lane;
    keccak_lanes_init(st.K, lane);
}
__device__ __forceinline__ void wstrobe_xor(WStrobe& st, uint32_t at, uint64_t byte) {
    if ((uint32_t)st.l == (at >> 3)) st.a ^= byte << (8 * (at & 7));
}
__device__ __forceinline__ void strobe_run_f(WStrobe& st) {
    wstrobe_xor(st, st.pos, st.pos_begin);
    wstrobe_xor(st, st.pos + 1, 0x04);
    wstrobe_xor(st, STROBE_R + 1, 0x80);
    st.a = keccak_f1600_wave(st.a, st.K, st.l);
    st.pos = 0;
    st.pos_begin = 0;
}
__device__ __forceinline__ void strobe_absorb_byte(WStrobe& st, uint8_t b) {
    wstrobe_xor(st, st.pos, b);
    if (++st.pos == STROBE_R) strobe_run_f(st);
}
// (the position is the same on every lane: the word that holds it is READ from its lane -- v_readlane, a few cycles -- not pulled
// through the LDS crossbar like a per-lane source would have to be)
__device__ __forceinline__ uint64_t wstrobe_word(const WStrobe& st, uint32_t word) {
    const int src = __builtin_amdgcn_readfirstlane((int)word);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(st.a >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)st.a, src);
}
__device__ __forceinline__ uint8_t strobe_squeeze_byte(WStrobe& st) {
    const uint32_t sh = 8 * (st.pos & 7);
    const uint8_t b = (uint8_t)(wstrobe_word(st, st.pos >> 3) >> sh);
    if ((uint32_t)st.l == (st.pos >> 3)) st.a &= ~(0xffull << sh);
    if (++st.pos == STROBE_R) strobe_run_f(st);
    return b;
}
__device__ __forceinline__ void strobe_reset(WStrobe& st) {
    st.a = 0;
    st.pos = 0;
    st.pos_begin = 0;
}
__device__ __forceinline__ void strobe_permute_raw(WStrobe& st) { st.a = keccak_f1600_wave(st.a, st.K, st.l); }
#endif
DAPOL_HD void strobe_permute_raw(Strobe& st) { keccak_f1600(st.s); }

// ---------------------------------------------------------------- the m commitments of a proof as a closed-form byte stream
// append_message(b"V", commitment) m times adds m records of 41 bytes to the STROBE stream, nothing in between squeezes, and
// STROBE's two position bytes are functions of the offset alone -- so word `lane` of 166-byte block `beta` can be put together
// without looking at any other (k_rv_absorb_V, kernels_verify.h: one wavefront per proof, lane = state word).  Written for both
// sides: the kernel calls these, and tests/host_shim.cpp replays a whole stream with them against the byte-wise Strobe.
//   record j = [pos_begin, M|A, 'V', LE32(32), pos_begin', A, 32 data bytes]  (merlin_frame + strobe_begin_op above)
// A lane's window (8 bytes) starts at stream offset k = 166 beta + 8 lane - pos0 = 41 j + t; the framing (9 bytes) and the data
// (32 bytes) are both longer than the window, so it is (data, framing), (framing, data), all data or all framing:
//   t >= 9: min(41 - t, 8) data bytes of commitment j from its byte t - 9, then the framing of record j + 1 from its byte 0;
//   t <  9: the framing of record j from its byte t (min(9 - t, 8) bytes), then commitment j from its byte 0.
// Data bytes are consecutive in memory even across commitments, so three aligned 32-bit words hold the (at most) 8 of them.
// The two position bytes: the begin_op they belong to follows the previous one by 34 (t = 0) or 7 (t = 7) bytes, so STROBE's old
// pos_begin is (q - d) + 1 when that one lies in the same block (q >= d) and 0 after a run_f; the very first record inherits
// pb0.  Dead bytes: before pos0 (block 0: the stream starts at pos0, i.e. the window is the stream's head shifted up), from
// byte 166 of the block on (lane 20's last two; lanes 21-24), and after the end of the stream.
enum { RV_V_BYTES = 41 };            // stream bytes per commitment
struct AbsorbWindow { uint32_t lead, j, t; };                       // dead leading bytes; record and byte of the first live one
DAPOL_HD AbsorbWindow absorb_window(uint32_t beta, uint32_t lane, uint32_t pos0) {
    const uint32_t at0 = beta * STROBE_R + 8 * lane;
    const uint32_t lead = at0 < pos0 ? (pos0 - at0 < 8u ? pos0 - at0 : 8u) : 0u;
    const uint32_t k = at0 + lead > pos0 ? at0 + lead - pos0 : 0u, j = k / RV_V_BYTES;    // (a window wholly before pos0: lead = 8, all dead)
    return AbsorbWindow{lead, j, k - j * RV_V_BYTES};
}
// index of the first of the three aligned words of the proof's commitments ([m][8] words) the window's data bytes lie in
DAPOL_HD uint32_t absorb_first_word(const AbsorbWindow& W) { return (32 * W.j + (W.t >= 9 ? W.t - 9 : 0u)) >> 2; }
DAPOL_HD uint64_t absorb_shl_bytes(uint64_t x, uint32_t n) { return n >= 8 ? 0ull : x << (8 * n); }
DAPOL_HD uint64_t absorb_low_bytes(uint32_t n) { return n >= 8 ? ~0ull : (1ull << (8 * n)) - 1; }
// word `lane` of block `beta`: d0..d2 = the words at absorb_first_word() + 0, 1, 2 (any value where they lie past the last
// commitment: no live byte comes from there); end_abs = pos0 + 41 m
DAPOL_HD uint64_t absorb_block_word(uint32_t beta, uint32_t lane, uint32_t pos0, uint32_t pb0, uint32_t end_abs, uint32_t d0, uint32_t d1, uint32_t d2) {
    const AbsorbWindow W = absorb_window(beta, lane, pos0);
    const uint32_t q0 = 8 * lane, at0 = beta * STROBE_R + q0, t = W.t;
    const bool data_first = t >= 9;
    const uint32_t sh = 8 * ((data_first ? t - 9 : 0u) & 3u);                               // (32 j is a multiple of 4)
    const uint64_t w01 = ((uint64_t)d1 << 32) | d0, w12 = ((uint64_t)d2 << 32) | d1;
    const uint64_t raw = ((uint64_t)(uint32_t)(w12 >> sh) << 32) | (uint32_t)(w01 >> sh);   // two funnel shifts (v_alignbit)
    // the framing record in (or after) the window: number jf, its byte 0 at in-block position qf0 (negative: before the window)
    const uint32_t jf = data_first ? W.j + 1 : W.j;
    const int qf0 = (int)(q0 + W.lead) + (data_first ? (int)(RV_V_BYTES - t) : -(int)t), qf7 = qf0 + 7;
    const uint32_t ob0 = jf == 0 ? pb0 : (qf0 >= 34 ? (uint32_t)(qf0 - 33) : 0u), ob7 = qf7 >= 7 ? (uint32_t)(qf7 - 6) : 0u;
    const uint64_t fr = (uint64_t)ob0 | 0x20561200ull | ((uint64_t)ob7 << 56), f8 = (uint64_t)SF_A;   // bytes 0..7 of the record, byte 8
    uint64_t sw;
    if (data_first) {
        const uint32_t nd = RV_V_BYTES - t;                                                 // 1..32 data bytes left in commitment j
        sw = (raw & absorb_low_bytes(nd)) | absorb_shl_bytes(fr, nd);
    } else {
        const uint64_t frs = t >= 8 ? f8 : (fr >> (8 * t)) | (t ? f8 << (8 * (8 - t)) : 0ull);
        const uint32_t nf = 9 - t;                                                          // 1..9 framing bytes first
        sw = (frs & absorb_low_bytes(nf)) | absorb_shl_bytes(raw, nf);
    }
    const uint32_t in_block = q0 < STROBE_R ? STROBE_R - q0 : 0u, in_stream = at0 < end_abs ? end_abs - at0 : 0u;
    return absorb_shl_bytes(sw, W.lead) & absorb_low_bytes(in_block < in_stream ? in_block : in_stream);
}
// what run_f adds to word 20 after block beta: pos_begin at byte 166, 0x04 and 0x80 at byte 167
DAPOL_HD uint64_t absorb_runf_word(uint32_t beta, uint32_t pos0, uint32_t pb0) {
    const uint32_t k_end = beta * STROBE_R + (STROBE_R - 1) - pos0, je = k_end / RV_V_BYTES, te = k_end - je * RV_V_BYTES;
    const uint32_t kb = RV_V_BYTES * je + (te >= 7 ? 7u : 0u), ab = pos0 + kb;               // the last begin_op at or before the block's end
    const uint32_t pbe = ab >= beta * STROBE_R ? ab % STROBE_R + 1 : (beta == 0 ? pb0 : 0u);
    return ((uint64_t)pbe << 48) | (0x84ull << 56);
}
// position state after the whole stream (what the lane-per-proof replay holds in pos / pos_begin)
DAPOL_HD void absorb_end_position(uint32_t pos0, uint32_t m, uint32_t& pos, uint32_t& pos_begin) {
    const uint32_t end_abs = pos0 + RV_V_BYTES * m, nfull = end_abs / STROBE_R, kb = RV_V_BYTES * (m - 1) + 7;   // the last begin_op
    pos = end_abs % STROBE_R;
    pos_begin = (pos0 + kb) / STROBE_R == nfull ? (pos0 + kb) % STROBE_R + 1 : 0u;
}

// Everything above the byte level is written once for both holders of the state (S = Strobe: one lane; S = WStrobe: a wavefront).
template <class S>
DAPOL_HD void strobe_begin_op(S& st, uint8_t flags) {
    uint8_t old_begin = (uint8_t)st.pos_begin;
    st.pos_begin = st.pos + 1;
    strobe_absorb_byte(st, old_begin);
    strobe_absorb_byte(st, flags);
    if ((flags & (SF_C | SF_K)) && st.pos != 0) strobe_run_f(st);
}
template <class S>
DAPOL_HD void strobe_init(S& st, const char* label, int n) {  // Strobe128::new
    strobe_reset(st);
    const uint8_t hdr[18] = {1, STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (int i = 0; i < 18; i++) { strobe_absorb_byte(st, hdr[i]); }      // (18 < R: plain XORs into the zero state)
    st.pos = 0;
    strobe_permute_raw(st);
    st.pos_begin = 0;
    strobe_begin_op(st, SF_M | SF_A);
    for (int i = 0; i < n; i++) strobe_absorb_byte(st, (uint8_t)label[i]);
}

// Merlin: label framing shared by append_message / challenge_bytes.
template <class S>
DAPOL_HD void merlin_frame(S& st, const char* label, int label_len, uint32_t data_len) {
    strobe_begin_op(st, SF_M | SF_A);
    for (int i = 0; i < label_len; i++) strobe_absorb_byte(st, (uint8_t)label[i]);
    for (int i = 0; i < 4; i++) strobe_absorb_byte(st, (uint8_t)(data_len >> (8 * i)));   // meta_ad(len, more=true)
}
template <class S>
DAPOL_HD void merlin_init(S& st, const char* app_label, int n) {  // Transcript::new(label)
    strobe_init(st, LBL_STROBE_PROTO);
    merlin_frame(st, LBL_DOM_SEP, (uint32_t)n);
    strobe_begin_op(st, SF_A);
    for (int i = 0; i < n; i++) strobe_absorb_byte(st, (uint8_t)app_label[i]);
}
template <class S>
DAPOL_HD void merlin_append_bytes(S& st, const char* label, int label_len, const char* msg, int n) {
    merlin_frame(st, label, label_len, (uint32_t)n);
    strobe_begin_op(st, SF_A);
    for (int i = 0; i < n; i++) strobe_absorb_byte(st, (uint8_t)msg[i]);
}
template <class S>
DAPOL_HD void merlin_append_words(S& st, const char* label, int label_len, const uint32_t* w, int nwords) {
    merlin_frame(st, label, label_len, (uint32_t)(4 * nwords));
    strobe_begin_op(st, SF_A);
    for (int i = 0; i < nwords; i++)
        for (int k = 0; k < 4; k++) strobe_absorb_byte(st, (uint8_t)(w[i] >> (8 * k)));
}
#if defined(__HIPCC__)
// The wavefront's state takes a message word in one step when its four bytes end inside the block: the lane(s) whose state word
// overlaps bytes [pos, pos + 4) XOR their part of it (a dozen instructions instead of four byte steps of as many each).
__device__ __forceinline__ void strobe_absorb_word(WStrobe& st, uint32_t w) {
    if (st.pos + 4 <= STROBE_R) {
        const int s = 8 * (int)st.pos - 64 * st.l;                   // bit offset of the message word in this lane's state word
        if (s > -32 && s < 64) st.a ^= s >= 0 ? (uint64_t)w << s : (uint64_t)w >> -s;
        st.pos += 4;
        if (st.pos == STROBE_R) strobe_run_f(st);
    } else {
        for (int k = 0; k < 4; k++) strobe_absorb_byte(st, (uint8_t)(w >> (8 * k)));
    }
}
__device__ __forceinline__ void merlin_append_words(WStrobe& st, const char* label, int label_len, const uint32_t* w, int nwords) {
    merlin_frame(st, label, label_len, (uint32_t)(4 * nwords));
    strobe_begin_op(st, SF_A);
    for (int i = 0; i < nwords; i++) strobe_absorb_word(st, w[i]);
}
#endif
template <class S>
DAPOL_HD void merlin_append_u64(S& st, const char* label, int label_len, uint64_t x) {
    uint32_t w[2] = {(uint32_t)x, (uint32_t)(x >> 32)};
    merlin_append_words(st, label, label_len, w, 2);
}
// challenge_bytes(label, 64) as sixteen little-endian words (input of Scalar::from_bytes_mod_order_wide)
template <class S>
DAPOL_HD void merlin_challenge_wide(S& st, const char* label, int label_len, uint32_t* w16) {
    merlin_frame(st, label, label_len, 64);
    strobe_begin_op(st, SF_I | SF_A | SF_C);
    for (int i = 0; i < 16; i++) {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) w |= (uint32_t)strobe_squeeze_byte(st) << (8 * k);
        w16[i] = w;
    }
}

}  // namespace dapol
