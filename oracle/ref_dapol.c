/* oracle/ref_dapol.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference proving path in plain C.
 *
 * "Parity pinned" status: the reference crate cannot be built here (no cargo/rustc; bulletproofs 4.0.0,
 * curve25519-dalek-ng 4.1.1, merlin 3.0.0, smtree 0.1.2, blake3 0.3.8 are not vendored).  This file restates
 * their published algorithms and is checked (tests/test_oracle_c.py) bit-for-bit against oracle/pyref.py, which
 * is pinned to RFC 9496 vectors, the Merlin / STROBE conformance vectors, BLAKE3 official vectors and the
 * reference's own known answers (src/dapol/tests.rs:24,30-85; src/range/mod.rs:18).
 *
 * It follows the reference's STRUCTURE, including the costs the GPU path removes:
 *   node algebra ............ src/dapol/node.rs:29-45 (new), :64-80 (merge: re-compresses both children), :86-88
 *   sparse tree ............. smtree build as driven by src/dapol/mod.rs:196-208
 *   range proof ............. src/range/mod.rs:48-78 -> bulletproofs party/dealer protocol + InnerProductProof::create
 *                             (generators folded round by round; BulletproofGens::new per call when `faithful`)
 *   verification ............ src/range/mod.rs:83-119 -> verify_multiple (single multiscalar check)
 * Randomness: explicit tapes (include/dapol_hip.h "Randomness contract"), never thread_rng.
 */
#include <stdio.h>
#include <stdlib.h>
#include "ref_math.h"
#ifdef _OPENMP
#include <omp.h>
#endif

/* Thread count of the OpenMP loops below (bench.py times the baseline single-threaded and on all usable cores). */
void ref_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------- Keccak */
static const uint64_t KRC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
    0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL,
    0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL,
    0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL,
    0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static inline uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static void keccak_f(uint64_t a[25]) {
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], KROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KRC[r];
    }
}
typedef struct { uint64_t s[25]; unsigned pos, rate; } sponge;
static void sp_init(sponge* sp, unsigned rate) { memset(sp, 0, sizeof *sp); sp->rate = rate; }
static void sp_absorb(sponge* sp, const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        sp->s[sp->pos >> 3] ^= (uint64_t)d[i] << (8 * (sp->pos & 7));
        if (++sp->pos == sp->rate) { keccak_f(sp->s); sp->pos = 0; }
    }
}
static void sp_finish(sponge* sp, uint8_t dom) {
    sp->s[sp->pos >> 3] ^= (uint64_t)dom << (8 * (sp->pos & 7));
    sp->s[(sp->rate - 1) >> 3] ^= 0x80ULL << (8 * ((sp->rate - 1) & 7));
    keccak_f(sp->s);
    sp->pos = 0;
}
static void sp_squeeze(sponge* sp, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (sp->pos == sp->rate) { keccak_f(sp->s); sp->pos = 0; }
        out[i] = (uint8_t)(sp->s[sp->pos >> 3] >> (8 * (sp->pos & 7)));
        sp->pos++;
    }
}
/* STROBE-128 / Merlin 3.0.0 */
typedef struct { uint8_t st[200]; unsigned pos, pos_begin; } strobe;
enum { SF_I = 1, SF_A = 2, SF_C = 4, SF_M = 16, SF_K = 32, STROBE_R = 166 };
static void st_f(strobe* s) {
    s->st[s->pos] ^= (uint8_t)s->pos_begin;
    s->st[s->pos + 1] ^= 0x04;
    s->st[STROBE_R + 1] ^= 0x80;
    uint64_t a[25];
    memcpy(a, s->st, 200);
    keccak_f(a);
    memcpy(s->st, a, 200);
    s->pos = 0; s->pos_begin = 0;
}
static void st_absorb(strobe* s, const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { s->st[s->pos++] ^= d[i]; if (s->pos == STROBE_R) st_f(s); }
}
static void st_squeeze(strobe* s, uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { d[i] = s->st[s->pos]; s->st[s->pos++] = 0; if (s->pos == STROBE_R) st_f(s); }
}
static void st_begin(strobe* s, uint8_t flags) {
    uint8_t hdr[2] = {(uint8_t)s->pos_begin, flags};
    s->pos_begin = s->pos + 1;
    st_absorb(s, hdr, 2);
    if ((flags & (SF_C | SF_K)) && s->pos != 0) st_f(s);
}
static void tr_frame(strobe* s, const char* label, uint32_t len) {
    st_begin(s, SF_M | SF_A);
    st_absorb(s, (const uint8_t*)label, strlen(label));
    uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    st_absorb(s, l4, 4);
}
static void tr_append(strobe* s, const char* label, const uint8_t* m, uint32_t n) {
    tr_frame(s, label, n);
    st_begin(s, SF_A);
    st_absorb(s, m, n);
}
static void tr_append_u64(strobe* s, const char* label, uint64_t x) { uint8_t b[8]; memcpy(b, &x, 8); tr_append(s, label, b, 8); }
static void tr_init(strobe* s, const uint8_t* app, uint32_t n) {
    memset(s, 0, sizeof *s);
    static const uint8_t hdr[18] = {1, STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    memcpy(s->st, hdr, 18);
    uint64_t a[25];
    memcpy(a, s->st, 200); keccak_f(a); memcpy(s->st, a, 200);
    st_begin(s, SF_M | SF_A);
    st_absorb(s, (const uint8_t*)"Merlin v1.0", 11);
    tr_append(s, "dom-sep", app, n);
}
static void tr_challenge(strobe* s, const char* label, scl* out) {
    uint8_t b[64];
    tr_frame(s, label, 64);
    st_begin(s, SF_I | SF_A | SF_C);
    st_squeeze(s, b, 64);
    sc_from_wide(out, b);
}
static void tr_append_scalar(strobe* s, const char* label, const scl* x) { uint8_t b[32]; sc_to_bytes(b, x); tr_append(s, label, b, 32); }

/* ------------------------------------------------------------------------------------------- BLAKE3 */
static const uint32_t B3IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const int B3PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
static inline uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define G3(a, b, c, d, x, y) s[a] += s[b] + (x); s[d] = ror32(s[d] ^ s[a], 16); s[c] += s[d]; s[b] = ror32(s[b] ^ s[c], 12); \
                             s[a] += s[b] + (y); s[d] = ror32(s[d] ^ s[a], 8); s[c] += s[d]; s[b] = ror32(s[b] ^ s[c], 7);
static void b3_compress(uint32_t out[16], const uint32_t cv[8], const uint32_t blk[16], uint64_t ctr, uint32_t len, uint32_t flags) {
    uint32_t s[16], m[16], t[16];
    memcpy(s, cv, 32); memcpy(s + 8, B3IV, 16);
    s[12] = (uint32_t)ctr; s[13] = (uint32_t)(ctr >> 32); s[14] = len; s[15] = flags;
    memcpy(m, blk, 64);
    for (int r = 0; r < 7; r++) {
        G3(0, 4, 8, 12, m[0], m[1]) G3(1, 5, 9, 13, m[2], m[3]) G3(2, 6, 10, 14, m[4], m[5]) G3(3, 7, 11, 15, m[6], m[7])
        G3(0, 5, 10, 15, m[8], m[9]) G3(1, 6, 11, 12, m[10], m[11]) G3(2, 7, 8, 13, m[12], m[13]) G3(3, 4, 9, 14, m[14], m[15])
        for (int i = 0; i < 16; i++) t[i] = m[B3PERM[i]];
        memcpy(m, t, 64);
    }
    for (int i = 0; i < 8; i++) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
/* single-chunk BLAKE3 (<= 1024 bytes) */
static void blake3_hash(uint8_t out[32], const uint8_t* data, size_t n) {
    uint32_t cv[8], blk[16], o[16];
    memcpy(cv, B3IV, 32);
    size_t nblocks = n == 0 ? 1 : (n + 63) / 64;
    for (size_t i = 0; i < nblocks; i++) {
        size_t len = (i == nblocks - 1) ? n - 64 * i : 64;
        uint8_t buf[64] = {0};
        memcpy(buf, data + 64 * i, len);
        memcpy(blk, buf, 64);
        uint32_t fl = (i == 0 ? 1u : 0u) | (i == nblocks - 1 ? (2u | 8u) : 0u);
        b3_compress(o, cv, blk, 0, (uint32_t)len, fl);
        memcpy(cv, o, 32);
    }
    memcpy(out, cv, 32);
}
void ref_seed_wide(uint8_t out[64], const uint8_t seed[32], uint32_t dom, uint64_t a, uint64_t b) {
    uint32_t key[8], blk[16] = {0}, o[16];
    memcpy(key, seed, 32);
    blk[0] = dom; blk[1] = (uint32_t)a; blk[2] = (uint32_t)(a >> 32); blk[3] = (uint32_t)b; blk[4] = (uint32_t)(b >> 32);
    b3_compress(o, key, blk, 0, 20, 16u | 1u | 2u | 8u);
    memcpy(out, o, 64);
}

/* ------------------------------------------------------------------------------------------- generators */
static int g_bb_ready = 0;
static void ensure_bb(void) {      /* PedersenGens::default(): B_blinding = hash_from_bytes::<Sha3_512>(B.compress()) */
    ref_init();
    if (g_bb_ready) return;
    uint8_t c[32], h[64];
    pt_compress(c, &REF_B);
    sponge sp;
    sp_init(&sp, 72);
    sp_absorb(&sp, c, 32);
    sp_finish(&sp, 0x06);
    sp_squeeze(&sp, h, 64);
    pt_from_uniform(&REF_BB, h);
    g_bb_ready = 1;
}
/* BulletproofGens::new(n_cap = 64, m): G[j][i], H[j][i] */
static void bp_gens_new(pt* G, pt* H, int n, int m) {
    for (int j = 0; j < m; j++)
        for (int which = 0; which < 2; which++) {
            sponge sp;
            sp_init(&sp, 136);
            sp_absorb(&sp, (const uint8_t*)"GeneratorsChain", 15);
            uint8_t label[5] = {(uint8_t)(which ? 'H' : 'G'), (uint8_t)j, (uint8_t)(j >> 8), (uint8_t)(j >> 16), (uint8_t)(j >> 24)};
            sp_absorb(&sp, label, 5);
            sp_finish(&sp, 0x1F);
            for (int i = 0; i < n; i++) {
                uint8_t u[64];
                sp_squeeze(&sp, u, 64);
                pt_from_uniform(&(which ? H : G)[j * n + i], u);
            }
        }
}
typedef struct { int n, m; pt *G, *H; } gens_cache;
static gens_cache g_cache[16];
static int g_ncache = 0;
static void get_gens(pt** G, pt** H, int n, int m, int faithful, pt** to_free_g, pt** to_free_h) {
    *to_free_g = *to_free_h = NULL;
    if (!faithful) {
        #pragma omp critical(gens)
        {
            int hit = -1;
            for (int i = 0; i < g_ncache; i++) if (g_cache[i].n == n && g_cache[i].m == m) hit = i;
            if (hit < 0 && g_ncache < 16) {
                hit = g_ncache;
                g_cache[hit].n = n; g_cache[hit].m = m;
                g_cache[hit].G = (pt*)malloc(sizeof(pt) * n * m); g_cache[hit].H = (pt*)malloc(sizeof(pt) * n * m);
                bp_gens_new(g_cache[hit].G, g_cache[hit].H, n, m);
                g_ncache++;
            }
            if (hit >= 0) { *G = g_cache[hit].G; *H = g_cache[hit].H; }
        }
        if (*G) return;
    }
    *G = *to_free_g = (pt*)malloc(sizeof(pt) * n * m);
    *H = *to_free_h = (pt*)malloc(sizeof(pt) * n * m);
    bp_gens_new(*G, *H, n, m);          /* the reference does this on every prove / verify call (src/range/mod.rs:50,66) */
}
void ref_generator(int which, int party, int bit, uint8_t out[32]) {
    ensure_bb();
    if (which == 0) { pt_compress(out, &REF_B); return; }
    if (which == 1) { pt_compress(out, &REF_BB); return; }
    pt *G = NULL, *H = NULL, *fg, *fh;
    get_gens(&G, &H, 64, party + 1, 1, &fg, &fh);
    pt_compress(out, &(which == 2 ? G : H)[party * 64 + bit]);
    free(fg); free(fh);
}

/* PedersenGens::default() recomputed from scratch (what the reference pays on every DapolNode::new, node.rs:31, and
 * every prove / verify call, src/range/mod.rs:49,65,84,103); only the `faithful` cost model calls this. */
static void pedersen_default(pt* bb) {
    uint8_t c[32], h[64];
    pt_compress(c, &REF_B);
    sponge sp;
    sp_init(&sp, 72);
    sp_absorb(&sp, c, 32);
    sp_finish(&sp, 0x06);
    sp_squeeze(&sp, h, 64);
    pt_from_uniform(bb, h);
}
static void commit_with(pt* out, uint64_t v, const uint8_t r32[32], const pt* bb) {   /* PedersenGens::commit = v B + r B_blinding */
    uint8_t vb[32] = {0};
    memcpy(vb, &v, 8);
    pt a, b;
    pt_mul(&a, &REF_B, vb);
    pt_mul(&b, bb, r32);
    pt_add(out, &a, &b);
}
static void commit(pt* out, uint64_t v, const uint8_t r32[32]) { commit_with(out, v, r32, &REF_BB); }

/* ------------------------------------------------------------------------------------------- tape */
typedef struct { const uint8_t* tape; const uint8_t* seed; uint64_t stream, slot_base; uint8_t key[32]; } tape_t;
static void tape_scalar(scl* out, const tape_t* t, uint32_t slot) {
    uint8_t w[64];
    if (t->tape) memcpy(w, t->tape + 64 * (size_t)slot, 64);
    else ref_seed_wide(w, t->key, 2, t->stream, t->slot_base + slot);
    sc_from_wide(out, w);
}
/* Seed mode: the nonce key of one proof, bound to its statement (include/dapol_hip.h "Randomness contract"):
 * seed -> (domain 6: stream id, first slot) -> (domain 7: bits, parties) -> chained BLAKE3 over the value commitments,
 * 31 per chunk.  Vc: [m][32]. */
static void tape_rekey(tape_t* t, int n, int m, const uint8_t* Vc) {
    if (t->tape) return;
    uint8_t w[64], buf[1024];
    ref_seed_wide(w, t->seed, 6, t->stream, t->slot_base);
    memcpy(t->key, w, 32);
    ref_seed_wide(w, t->key, 7, (uint64_t)n, (uint64_t)m);
    memcpy(t->key, w, 32);
    for (int j0 = 0; j0 < m; j0 += 31) {
        int cnt = m - j0 < 31 ? m - j0 : 31;
        memcpy(buf, t->key, 32);
        memcpy(buf + 32, Vc + 32 * (size_t)j0, 32 * (size_t)cnt);
        blake3_hash(t->key, buf, 32 + 32 * (size_t)cnt);
    }
}

/* ------------------------------------------------------------------------------------------- prover */
static void ipp_create(strobe* tr, uint8_t* out, const pt* Q, const scl* Hfac, pt* G, pt* H, scl* a, scl* b, size_t n) {
    tr_append(tr, "dom-sep", (const uint8_t*)"ipp v1", 6);
    tr_append_u64(tr, "n", n);
    int first = 1;
    scl* sc_tmp = (scl*)malloc(sizeof(scl) * (2 * n + 1));
    pt* pt_tmp = (pt*)malloc(sizeof(pt) * (2 * n + 1));
    while (n != 1) {
        n /= 2;
        scl cL = SC_ZERO, cR = SC_ZERO, t;
        for (size_t i = 0; i < n; i++) {
            sc_mul(&t, &a[i], &b[n + i]); sc_add(&cL, &cL, &t);
            sc_mul(&t, &a[n + i], &b[i]); sc_add(&cR, &cR, &t);
        }
        /* L = <a_L, G_R> + <b_R * Hfac_L, H_L> + c_L Q   (H factors only in the first round) */
        for (size_t i = 0; i < n; i++) {
            sc_tmp[i] = a[i]; pt_tmp[i] = G[n + i];
            if (first) sc_mul(&sc_tmp[n + i], &b[n + i], &Hfac[i]); else sc_tmp[n + i] = b[n + i];
            pt_tmp[n + i] = H[i];
        }
        sc_tmp[2 * n] = cL; pt_tmp[2 * n] = *Q;
        pt Lp, Rp;
        pt_msm_vartime(&Lp, sc_tmp, pt_tmp, 2 * n + 1);
        for (size_t i = 0; i < n; i++) {
            sc_tmp[i] = a[n + i]; pt_tmp[i] = G[i];
            if (first) sc_mul(&sc_tmp[n + i], &b[i], &Hfac[n + i]); else sc_tmp[n + i] = b[i];
            pt_tmp[n + i] = H[n + i];
        }
        sc_tmp[2 * n] = cR; pt_tmp[2 * n] = *Q;
        pt_msm_vartime(&Rp, sc_tmp, pt_tmp, 2 * n + 1);
        pt_compress(out, &Lp); pt_compress(out + 32, &Rp);
        tr_append(tr, "L", out, 32); tr_append(tr, "R", out + 32, 32);
        out += 64;
        scl u, ui;
        tr_challenge(tr, "u", &u);
        sc_inv(&ui, &u);
        for (size_t i = 0; i < n; i++) {
            scl x, y2;
            sc_mul(&x, &a[i], &u); sc_mul(&y2, &ui, &a[n + i]); sc_add(&a[i], &x, &y2);
            sc_mul(&x, &b[i], &ui); sc_mul(&y2, &u, &b[n + i]); sc_add(&b[i], &x, &y2);
            scl s2[2];
            pt p2[2];
            s2[0] = ui; s2[1] = u; p2[0] = G[i]; p2[1] = G[n + i];
            pt_msm_vartime(&G[i], s2, p2, 2);
            if (first) { sc_mul(&s2[0], &u, &Hfac[i]); sc_mul(&s2[1], &ui, &Hfac[n + i]); } else { s2[0] = u; s2[1] = ui; }
            p2[0] = H[i]; p2[1] = H[n + i];
            pt_msm_vartime(&H[i], s2, p2, 2);
        }
        first = 0;
    }
    sc_to_bytes(out, &a[0]);
    sc_to_bytes(out + 32, &b[0]);
    free(sc_tmp); free(pt_tmp);
}

/* RangeProof::prove_multiple_with_rng (party / dealer protocol played by one process).  Returns 0 on success. */
int ref_range_prove(int n, int m, const uint64_t* v, const uint8_t* r32, const uint8_t* seed, uint64_t stream, uint64_t slot_base,
                    const uint8_t* tape, int faithful, uint8_t* proof_out) {
    ensure_bb();
    if (!(n == 8 || n == 16 || n == 32 || n == 64) || m < 1 || (m & (m - 1))) return 1;
    size_t N = (size_t)n * m;
    tape_t tp = {tape, seed, stream, slot_base, {0}};
    pt *G, *H, *fg, *fh;
    G = H = NULL;
    get_gens(&G, &H, n, m, faithful, &fg, &fh);
    if (faithful) { pt bb; pedersen_default(&bb); (void)bb; }  /* PedersenGens::default() once per prove call */
    strobe tr;
    tr_init(&tr, (const uint8_t*)"", 0);                      /* Transcript::new(&[]) */
    tr_append(&tr, "dom-sep", (const uint8_t*)"rangeproof v1", 13);
    tr_append_u64(&tr, "n", n);
    tr_append_u64(&tr, "m", m);
    scl* sL = (scl*)malloc(sizeof(scl) * N); scl* sR = (scl*)malloc(sizeof(scl) * N);
    scl* l0 = (scl*)malloc(sizeof(scl) * N); scl* r0 = (scl*)malloc(sizeof(scl) * N); scl* r1 = (scl*)malloc(sizeof(scl) * N);
    scl* a_bl = (scl*)malloc(sizeof(scl) * m); scl* s_bl = (scl*)malloc(sizeof(scl) * m);
    pt A, S;
    pt_identity(&A); pt_identity(&S);
    scl* msm_s = (scl*)malloc(sizeof(scl) * (2 * n + 1));
    pt* msm_p = (pt*)malloc(sizeof(pt) * (2 * n + 1));
    uint8_t* Vall = (uint8_t*)malloc(32 * (size_t)m);
    for (int j = 0; j < m; j++) {                             /* Party::new: V_j = commit(v_j, v_blinding_j) */
        pt V;
        commit(&V, v[j], r32 + 32 * j);
        pt_compress(Vall + 32 * (size_t)j, &V);
    }
    tape_rekey(&tp, n, m, Vall);
    for (int j = 0; j < m; j++) {                             /* assign_position_with_rng */
        uint8_t ab[32];
        const uint8_t* Vc = Vall + 32 * (size_t)j;
        tr_append(&tr, "V", Vc, 32);
        uint32_t base = (uint32_t)(j * (2 * n + 2));
        tape_scalar(&a_bl[j], &tp, base);
        sc_to_bytes(ab, &a_bl[j]);
        pt Aj;
        pt_mul(&Aj, &REF_BB, ab);
        for (int i = 0; i < n; i++) {
            if ((v[j] >> i) & 1) pt_add(&Aj, &Aj, &G[j * n + i]);
            else { pt nh; pt_neg(&nh, &H[j * n + i]); pt_add(&Aj, &Aj, &nh); }
        }
        tape_scalar(&s_bl[j], &tp, base + 1);
        for (int i = 0; i < n; i++) tape_scalar(&sL[j * n + i], &tp, base + 2 + i);
        for (int i = 0; i < n; i++) tape_scalar(&sR[j * n + i], &tp, base + 2 + n + i);
        msm_s[0] = s_bl[j]; msm_p[0] = REF_BB;
        for (int i = 0; i < n; i++) { msm_s[1 + i] = sL[j * n + i]; msm_p[1 + i] = G[j * n + i]; msm_s[1 + n + i] = sR[j * n + i]; msm_p[1 + n + i] = H[j * n + i]; }
        pt Sj;
        pt_msm_vartime(&Sj, msm_s, msm_p, 2 * n + 1);         /* constant-time multiscalar_mul in the crate */
        pt_add(&A, &A, &Aj);
        pt_add(&S, &S, &Sj);
    }
    free(msm_s); free(msm_p); free(Vall);
    uint8_t* o = proof_out;
    pt_compress(o, &A); pt_compress(o + 32, &S);
    tr_append(&tr, "A", o, 32); tr_append(&tr, "S", o + 32, 32);
    scl y, z, zz, one = SC_ONE;
    tr_challenge(&tr, "y", &y);
    tr_challenge(&tr, "z", &z);
    sc_mul(&zz, &z, &z);
    /* Party::apply_challenge_with_rng */
    scl t1s = SC_ZERO, t2s = SC_ZERO, t0s = SC_ZERO, t1bl = SC_ZERO, t2bl = SC_ZERO;
    scl* t1blv = (scl*)malloc(sizeof(scl) * m); scl* t2blv = (scl*)malloc(sizeof(scl) * m);
    scl* ozz = (scl*)malloc(sizeof(scl) * m);
    pt T1, T2;
    pt_identity(&T1); pt_identity(&T2);
    for (int j = 0; j < m; j++) {
        scl exp_y, exp_2 = one, offz, t0 = SC_ZERO, t1 = SC_ZERO, t2 = SC_ZERO;
        sc_pow(&exp_y, &y, (uint64_t)j * n);
        sc_pow(&offz, &z, (uint64_t)j);
        sc_mul(&ozz[j], &zz, &offz);
        for (int i = 0; i < n; i++) {
            size_t k = (size_t)j * n + i;
            scl aL = ((v[j] >> i) & 1) ? one : SC_ZERO, aR, t, u;
            sc_sub(&aR, &aL, &one);
            sc_sub(&l0[k], &aL, &z);
            sc_add(&t, &aR, &z); sc_mul(&t, &exp_y, &t); sc_mul(&u, &ozz[j], &exp_2); sc_add(&r0[k], &t, &u);
            sc_mul(&r1[k], &exp_y, &sR[k]);
            sc_mul(&t, &l0[k], &r0[k]); sc_add(&t0, &t0, &t);
            sc_mul(&t, &l0[k], &r1[k]); sc_add(&t1, &t1, &t);
            sc_mul(&t, &sL[k], &r0[k]); sc_add(&t1, &t1, &t);
            sc_mul(&t, &sL[k], &r1[k]); sc_add(&t2, &t2, &t);
            sc_mul(&exp_y, &exp_y, &y);
            sc_add(&exp_2, &exp_2, &exp_2);
        }
        uint32_t base = (uint32_t)(m * (2 * n + 2) + 2 * j);
        tape_scalar(&t1blv[j], &tp, base);
        tape_scalar(&t2blv[j], &tp, base + 1);
        uint8_t sb[32], bb[32];
        pt p1, p2, c;
        sc_to_bytes(sb, &t1); sc_to_bytes(bb, &t1blv[j]);
        pt_mul(&p1, &REF_B, sb); pt_mul(&p2, &REF_BB, bb); pt_add(&c, &p1, &p2); pt_add(&T1, &T1, &c);
        sc_to_bytes(sb, &t2); sc_to_bytes(bb, &t2blv[j]);
        pt_mul(&p1, &REF_B, sb); pt_mul(&p2, &REF_BB, bb); pt_add(&c, &p1, &p2); pt_add(&T2, &T2, &c);
        sc_add(&t0s, &t0s, &t0); sc_add(&t1s, &t1s, &t1); sc_add(&t2s, &t2s, &t2);
        sc_add(&t1bl, &t1bl, &t1blv[j]); sc_add(&t2bl, &t2bl, &t2blv[j]);
    }
    pt_compress(o + 64, &T1); pt_compress(o + 96, &T2);
    tr_append(&tr, "T_1", o + 64, 32); tr_append(&tr, "T_2", o + 96, 32);
    scl x, xx;
    tr_challenge(&tr, "x", &x);
    int rc = sc_is_zero(&x) ? 2 : 0;
    sc_mul(&xx, &x, &x);
    /* Party::apply_challenge + Dealer::assemble_shares */
    scl t_x, tau = SC_ZERO, mu = SC_ZERO, t;
    sc_mul(&t, &t1s, &x); sc_add(&t_x, &t0s, &t); sc_mul(&t, &t2s, &xx); sc_add(&t_x, &t_x, &t);
    for (int j = 0; j < m; j++) {
        scl bl;
        sc_from_bytes(&bl, r32 + 32 * j);
        sc_mul(&t, &ozz[j], &bl); sc_add(&tau, &tau, &t);
        sc_mul(&t, &s_bl[j], &x); sc_add(&t, &t, &a_bl[j]); sc_add(&mu, &mu, &t);
    }
    sc_mul(&t, &t1bl, &x); sc_add(&tau, &tau, &t);
    sc_mul(&t, &t2bl, &xx); sc_add(&tau, &tau, &t);
    scl* lv = l0; scl* rv = r0;
    for (size_t k = 0; k < N; k++) {
        sc_mul(&t, &sL[k], &x); sc_add(&lv[k], &l0[k], &t);
        sc_mul(&t, &r1[k], &x); sc_add(&rv[k], &r0[k], &t);
    }
    sc_to_bytes(o + 128, &t_x); sc_to_bytes(o + 160, &tau); sc_to_bytes(o + 192, &mu);
    tr_append(&tr, "t_x", o + 128, 32); tr_append(&tr, "t_x_blinding", o + 160, 32); tr_append(&tr, "e_blinding", o + 192, 32);
    scl w;
    tr_challenge(&tr, "w", &w);
    uint8_t wb[32];
    sc_to_bytes(wb, &w);
    pt Q;
    pt_mul(&Q, &REF_B, wb);
    scl yinv, *Hfac = (scl*)malloc(sizeof(scl) * N);
    sc_inv(&yinv, &y);
    Hfac[0] = one;
    for (size_t k = 1; k < N; k++) sc_mul(&Hfac[k], &Hfac[k - 1], &yinv);
    pt* Gw = (pt*)malloc(sizeof(pt) * N); pt* Hw = (pt*)malloc(sizeof(pt) * N);
    memcpy(Gw, G, sizeof(pt) * N); memcpy(Hw, H, sizeof(pt) * N);
    ipp_create(&tr, o + 224, &Q, Hfac, Gw, Hw, lv, rv, N);
    free(Gw); free(Hw); free(Hfac); free(sL); free(sR); free(l0); free(r0); free(r1); free(a_bl); free(s_bl);
    free(t1blv); free(t2blv); free(ozz); free(fg); free(fh);
    return rc;
}

size_t ref_range_proof_size(int n, int m) {
    int lg = 0;
    while ((1 << lg) < n * m) lg++;
    return 32 * (size_t)(9 + 2 * lg);
}

/* b proofs, parallel over proofs with OpenMP when built with -fopenmp */
int ref_range_prove_batch(int n, int m, size_t b, const uint64_t* v, const uint8_t* r32, const uint8_t* seed, const uint64_t* stream,
                          uint64_t slot_base, const uint8_t* tape, int faithful, uint8_t* out) {
    size_t ps = ref_range_proof_size(n, m), slots = (size_t)m * (2 * (size_t)n + 4);
    int bad = 0;
    ensure_bb();
    #pragma omp parallel for schedule(dynamic)
    for (long i = 0; i < (long)b; i++) {
        int rc = ref_range_prove(n, m, v + (size_t)i * m, r32 + (size_t)i * m * 32, seed, stream ? stream[i] : 0, slot_base,
                                 tape ? tape + (size_t)i * slots * 64 : NULL, faithful, out + (size_t)i * ps);
        if (rc) bad = rc;
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------- verifier */
/* RangeProof::from_bytes + verify_multiple.  c32 = the verifier's batching scalar (any non-zero value). */
int ref_range_verify(int n, int m, const uint8_t* proof, size_t proof_len, const uint8_t* V32, const uint8_t c32[32], int faithful) {
    ensure_bb();
    if (!(n == 8 || n == 16 || n == 32 || n == 64) || m < 1 || (m & (m - 1))) return 0;
    if (proof_len % 32 || proof_len < 7 * 32) return 0;
    size_t nel = proof_len / 32 - 7;
    if (nel < 2 || (nel - 2) % 2) return 0;
    size_t lg = (nel - 2) / 2, N = (size_t)n * m;
    if (lg >= 32 || ((size_t)1 << lg) != N) return 0;
    const uint8_t* o = proof;
    for (int k = 4; k < 7; k++) if (!sc_is_canonical(o + 32 * k)) return 0;
    if (!sc_is_canonical(o + proof_len - 64) || !sc_is_canonical(o + proof_len - 32)) return 0;
    static const uint8_t zero32[32] = {0};
    pt *G = NULL, *H = NULL, *fg, *fh;
    get_gens(&G, &H, n, m, faithful, &fg, &fh);
    strobe tr;
    tr_init(&tr, (const uint8_t*)"", 0);
    tr_append(&tr, "dom-sep", (const uint8_t*)"rangeproof v1", 13);
    tr_append_u64(&tr, "n", n);
    tr_append_u64(&tr, "m", m);
    for (int j = 0; j < m; j++) tr_append(&tr, "V", V32 + 32 * j, 32);
    int ok = 1;
    if (!memcmp(o, zero32, 32) || !memcmp(o + 32, zero32, 32)) ok = 0;
    tr_append(&tr, "A", o, 32); tr_append(&tr, "S", o + 32, 32);
    scl y, z, zz, x, w, c, t_x, tau, mu, a, b;
    tr_challenge(&tr, "y", &y); tr_challenge(&tr, "z", &z);
    sc_mul(&zz, &z, &z);
    if (!memcmp(o + 64, zero32, 32) || !memcmp(o + 96, zero32, 32)) ok = 0;
    tr_append(&tr, "T_1", o + 64, 32); tr_append(&tr, "T_2", o + 96, 32);
    tr_challenge(&tr, "x", &x);
    tr_append(&tr, "t_x", o + 128, 32); tr_append(&tr, "t_x_blinding", o + 160, 32); tr_append(&tr, "e_blinding", o + 192, 32);
    tr_challenge(&tr, "w", &w);
    sc_from_bytes(&c, c32);
    sc_from_bytes(&t_x, o + 128); sc_from_bytes(&tau, o + 160); sc_from_bytes(&mu, o + 192);
    sc_from_bytes(&a, o + proof_len - 64); sc_from_bytes(&b, o + proof_len - 32);
    tr_append(&tr, "dom-sep", (const uint8_t*)"ipp v1", 6);
    tr_append_u64(&tr, "n", N);
    scl* u = (scl*)malloc(sizeof(scl) * (lg + 1)); scl* ui = (scl*)malloc(sizeof(scl) * (lg + 1));
    for (size_t k = 0; k < lg; k++) {
        const uint8_t* Lp = o + 224 + 64 * k;
        if (!memcmp(Lp, zero32, 32) || !memcmp(Lp + 32, zero32, 32)) ok = 0;
        tr_append(&tr, "L", Lp, 32); tr_append(&tr, "R", Lp + 32, 32);
        tr_challenge(&tr, "u", &u[k]);
        sc_inv(&ui[k], &u[k]);
    }
    size_t npts = 4 + 2 * lg + 2 + 2 * N + m;
    scl* sc_v = (scl*)malloc(sizeof(scl) * npts);
    pt* pt_v = (pt*)malloc(sizeof(pt) * npts);
    size_t q = 0;
    scl cx, cxx, t, one = SC_ONE;
    sc_mul(&cx, &c, &x); sc_mul(&cxx, &cx, &x);
    const uint8_t* pbytes[4] = {o, o + 32, o + 64, o + 96};
    scl pscal[4] = {one, x, cx, cxx};
    for (int k = 0; k < 4; k++) { if (!pt_decompress(&pt_v[q], pbytes[k])) ok = 0; sc_v[q++] = pscal[k]; }
    scl allinv = one;
    for (size_t k = 0; k < lg; k++) {
        if (!pt_decompress(&pt_v[q], o + 224 + 64 * k)) ok = 0;
        sc_mul(&sc_v[q++], &u[k], &u[k]);
        sc_mul(&allinv, &allinv, &ui[k]);
    }
    for (size_t k = 0; k < lg; k++) {
        if (!pt_decompress(&pt_v[q], o + 224 + 64 * k + 32)) ok = 0;
        sc_mul(&sc_v[q++], &ui[k], &ui[k]);
    }
    /* s vector */
    scl* s = (scl*)malloc(sizeof(scl) * N);
    s[0] = allinv;
    for (size_t i = 1; i < N; i++) {
        int lgi = 63 - __builtin_clzll((unsigned long long)i);
        size_t kk = (size_t)1 << lgi;
        scl usq;
        sc_mul(&usq, &u[(lg - 1) - lgi], &u[(lg - 1) - lgi]);
        sc_mul(&s[i], &s[i - kk], &usq);
    }
    /* B_blinding and B terms */
    sc_mul(&t, &c, &tau); sc_add(&t, &t, &mu); sc_sub(&sc_v[q], &SC_ZERO, &t); pt_v[q++] = REF_BB;
    scl ab, sumy = SC_ZERO, sum2, sumz = SC_ZERO, py = one, pz = one, delta, bs;
    sc_mul(&ab, &a, &b);
    for (size_t i = 0; i < N; i++) { sc_add(&sumy, &sumy, &py); sc_mul(&py, &py, &y); }
    for (int j = 0; j < m; j++) { sc_add(&sumz, &sumz, &pz); sc_mul(&pz, &pz, &z); }
    sum2 = SC_ZERO;
    { scl p2 = one; for (int i = 0; i < n; i++) { sc_add(&sum2, &sum2, &p2); sc_add(&p2, &p2, &p2); } }
    sc_sub(&t, &z, &zz); sc_mul(&delta, &t, &sumy);
    sc_mul(&t, &zz, &z); sc_mul(&t, &t, &sum2); sc_mul(&t, &t, &sumz); sc_sub(&delta, &delta, &t);
    sc_sub(&t, &t_x, &ab); sc_mul(&bs, &w, &t);
    sc_sub(&t, &delta, &t_x); sc_mul(&t, &c, &t); sc_add(&sc_v[q], &bs, &t); pt_v[q++] = REF_B;
    /* g and h terms */
    scl mz, yinv, eyi = one;
    sc_sub(&mz, &SC_ZERO, &z);
    sc_inv(&yinv, &y);
    for (size_t i = 0; i < N; i++) { sc_mul(&t, &a, &s[i]); sc_sub(&sc_v[q], &mz, &t); pt_v[q++] = G[i]; }
    for (size_t i = 0; i < N; i++) {
        scl z2, p2, zj;
        sc_pow(&zj, &z, i / n);
        sc_from_u64(&p2, 1ULL << (i % n));
        sc_mul(&z2, &zj, &p2); sc_mul(&z2, &zz, &z2);
        sc_mul(&t, &b, &s[N - 1 - i]); sc_sub(&t, &z2, &t); sc_mul(&t, &eyi, &t); sc_add(&sc_v[q], &z, &t);
        pt_v[q++] = H[i];
        sc_mul(&eyi, &eyi, &yinv);
    }
    scl zjc;
    sc_mul(&zjc, &c, &zz);
    for (int j = 0; j < m; j++) {
        if (!pt_decompress(&pt_v[q], V32 + 32 * j)) ok = 0;
        sc_v[q++] = zjc;
        sc_mul(&zjc, &zjc, &z);
    }
    int verdict = 0;
    if (ok) {
        pt chk;
        uint8_t cb[32];
        pt_msm_vartime(&chk, sc_v, pt_v, q);
        pt_compress(cb, &chk);
        verdict = memcmp(cb, zero32, 32) == 0;       /* identity encodes as 32 zero bytes */
    }
    free(u); free(ui); free(sc_v); free(pt_v); free(s); free(fg); free(fh);
    return verdict;
}

/* ------------------------------------------------------------------------------------------- node algebra + tree */
typedef struct { uint64_t idx, v; uint8_t r[32], C[32], H[32]; pt com; uint8_t pad; } node;

static void node_new(node* nd, uint64_t idx, uint64_t v, const uint8_t r32[32], int faithful) {   /* DapolNode::new */
    nd->idx = idx; nd->v = v; nd->pad = 0;
    memcpy(nd->r, r32, 32);
    nd->r[31] &= 0x7f;
    if (faithful) { pt bb; pedersen_default(&bb); commit_with(&nd->com, v, nd->r, &bb); }   /* PedersenGens::default() per call (node.rs:31) */
    else commit(&nd->com, v, nd->r);
    pt_compress(nd->C, &nd->com);
    blake3_hash(nd->H, nd->C, 32);
}
static void node_merge(node* p, const node* l, const node* r, int faithful) {                      /* Mergeable::merge */
    uint8_t buf[128];
    if (faithful) { pt_compress(buf, &l->com); pt_compress(buf + 32, &r->com); }   /* node.rs:67-68 re-compress */
    else { memcpy(buf, l->C, 32); memcpy(buf + 32, r->C, 32); }
    memcpy(buf + 64, l->H, 32); memcpy(buf + 96, r->H, 32);
    blake3_hash(p->H, buf, 128);
    p->v = l->v + r->v;
    scl a, b, s;
    sc_from_bytes(&a, l->r); sc_from_bytes(&b, r->r); sc_add(&s, &a, &b); sc_to_bytes(p->r, &s);
    pt_add(&p->com, &l->com, &r->com);
    pt_compress(p->C, &p->com);
    p->idx = l->idx >> 1; p->pad = 0;
}
static void node_padding(node* nd, const uint8_t seed[32], int level, uint64_t idx, int faithful) { /* Paddable::padding */
    uint8_t w[64], r[32];
    scl s;
    ref_seed_wide(w, seed, 1, (uint64_t)level, idx);
    sc_from_wide(&s, w);
    sc_to_bytes(r, &s);
    node_new(nd, idx, 0, r, faithful);
    nd->pad = 1;
}
void ref_commit_hash(size_t n, const uint64_t* v, const uint8_t* r32, uint8_t* C, uint8_t* H) {
    ensure_bb();
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        node nd;
        node_new(&nd, 0, v[i], r32 + 32 * (size_t)i, 0);
        memcpy(C + 32 * (size_t)i, nd.C, 32); memcpy(H + 32 * (size_t)i, nd.H, 32);
    }
}

typedef struct { int height; size_t* n; node** lv; node** pad; } ref_tree;    /* lv[k][i] real, pad[k][i] = padding sibling of lv[k][i] (idx==~0 if none) */

ref_tree* ref_tree_build(int height, size_t n, const uint64_t* idx, const uint64_t* v, const uint8_t* r32, const uint8_t seed[32], int faithful) {
    ensure_bb();
    for (size_t i = 0; i < n; i++) if ((height < 64 && (idx[i] >> height)) || (i && idx[i] <= idx[i - 1])) return NULL;
    if (n == 0) return NULL;
    ref_tree* t = (ref_tree*)calloc(1, sizeof *t);
    t->height = height;
    t->n = (size_t*)calloc(height + 1, sizeof(size_t));
    t->lv = (node**)calloc(height + 1, sizeof(node*));
    t->pad = (node**)calloc(height + 1, sizeof(node*));
    t->n[0] = n;
    t->lv[0] = (node*)malloc(sizeof(node) * n);
    #pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) node_new(&t->lv[0][i], idx[i], v[i], r32 + 32 * (size_t)i, 0);
    for (int k = 0; k < height; k++) {
        size_t cn = t->n[k], pn = 0;
        node* cur = t->lv[k];
        size_t* head = (size_t*)malloc(sizeof(size_t) * cn);
        for (size_t i = 0; i < cn; i++) if (i == 0 || (cur[i].idx >> 1) != (cur[i - 1].idx >> 1)) head[pn++] = i;
        t->n[k + 1] = pn;
        t->lv[k + 1] = (node*)malloc(sizeof(node) * pn);
        t->pad[k] = (node*)malloc(sizeof(node) * cn);
        for (size_t i = 0; i < cn; i++) t->pad[k][i].idx = ~0ULL;
        #pragma omp parallel for schedule(static)
        for (long q = 0; q < (long)pn; q++) {
            size_t i = head[q];
            int pair = (i + 1 < cn) && cur[i + 1].idx == (cur[i].idx ^ 1);
            if (pair) node_merge(&t->lv[k + 1][q], &cur[i], &cur[i + 1], faithful);
            else {
                node* pd = &t->pad[k][i];
                node_padding(pd, seed, k, cur[i].idx ^ 1, faithful);
                if (cur[i].idx & 1) node_merge(&t->lv[k + 1][q], pd, &cur[i], faithful);
                else node_merge(&t->lv[k + 1][q], &cur[i], pd, faithful);
            }
        }
        free(head);
    }
    return t;
}
void ref_tree_free(ref_tree* t) {
    if (!t) return;
    for (int k = 0; k <= t->height; k++) { free(t->lv[k]); free(t->pad[k]); }
    free(t->lv); free(t->pad); free(t->n); free(t);
}
void ref_tree_root(const ref_tree* t, uint8_t C[32], uint8_t H[32], uint64_t* v, uint8_t r[32]) {
    const node* rt = &t->lv[t->height][0];
    memcpy(C, rt->C, 32); memcpy(H, rt->H, 32); *v = rt->v; memcpy(r, rt->r, 32);
}
uint64_t ref_tree_node_count(const ref_tree* t) {
    uint64_t c = 0;
    for (int k = 0; k <= t->height; k++) {
        c += t->n[k];
        if (k < t->height) for (size_t i = 0; i < t->n[k]; i++) if (t->pad[k][i].idx != ~0ULL) c++;
    }
    return c;
}
/* siblings of one leaf, root side first; returns 0 if the leaf is absent */
int ref_tree_path(const ref_tree* t, uint64_t leaf, uint8_t* sC, uint8_t* sH, uint64_t* sv, uint8_t* sr) {
    size_t lo = 0, hi = t->n[0];
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (t->lv[0][mid].idx < leaf) lo = mid + 1; else hi = mid; }
    if (lo >= t->n[0] || t->lv[0][lo].idx != leaf) return 0;
    size_t pos = lo;
    for (int k = 0; k < t->height; k++) {
        const node* cur = t->lv[k];
        const node* sib;
        if (t->pad[k][pos].idx != ~0ULL) sib = &t->pad[k][pos];
        else sib = (cur[pos].idx & 1) ? &cur[pos - 1] : &cur[pos + 1];
        int slot = t->height - 1 - k;
        memcpy(sC + 32 * slot, sib->C, 32); memcpy(sH + 32 * slot, sib->H, 32); sv[slot] = sib->v; memcpy(sr + 32 * slot, sib->r, 32);
        /* parent position: number of distinct parents before */
        uint64_t pidx = cur[pos].idx >> 1;
        size_t l2 = 0, h2 = t->n[k + 1];
        while (l2 < h2) { size_t mid = (l2 + h2) / 2; if (t->lv[k + 1][mid].idx < pidx) l2 = mid + 1; else h2 = mid; }
        pos = l2;
    }
    return 1;
}

/* Padding-policy inclusion proofs (src/range/padding.rs:88-118 with aggregation_factor = height, the bench's
 * choice benches/dapol.rs:155) for `count` leaves; parallel over entities.  out: [count][proof_size(n_bits, m)]. */
int ref_prove_entities_padding(const ref_tree* t, size_t count, const uint64_t* leaves, int n_bits, const uint8_t nonce_seed[32],
                               int faithful, uint8_t* out) {
    int H = t->height, m = 1;
    while (m < H) m <<= 1;
    size_t ps = ref_range_proof_size(n_bits, m);
    int bad = 0;
    #pragma omp parallel for schedule(dynamic)
    for (long e = 0; e < (long)count; e++) {
        uint8_t* sC = (uint8_t*)malloc(32 * H); uint8_t* sH = (uint8_t*)malloc(32 * H); uint8_t* sr = (uint8_t*)calloc(32, m);
        uint64_t* sv = (uint64_t*)calloc(m, 8);
        if (!ref_tree_path(t, leaves[e], sC, sH, sv, sr)) bad = 9;
        else {
            for (int j = H; j < m; j++) { sv[j] = 0; memset(sr + 32 * j, 0, 32); sr[32 * j] = 1; }   /* (0, Scalar::one()) */
            int rc = ref_range_prove(n_bits, m, sv, sr, nonce_seed, leaves[e], 0, NULL, faithful, out + (size_t)e * ps);
            if (rc) bad = rc;
        }
        free(sC); free(sH); free(sr); free(sv);
    }
    return bad;
}
