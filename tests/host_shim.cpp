// Test-only shim: compiles the PRODUCT device headers (dapol_amd/csrc/*.h) for the host so that the limb
// arithmetic, codec, scalar field and hash framing can be checked against oracle/pyref.py without a GPU.
// Built by tests/conftest.py with g++; never linked into libdapol_hip.so.
#include <cstring>
#include "../dapol_amd/csrc/ge.h"
#include "../dapol_amd/csrc/hash.h"
#include "../dapol_amd/csrc/sc.h"
using namespace dapol;

static void ld(uint32_t* w, const uint8_t* b, int n) { memcpy(w, b, 4 * n); }
static void st(uint8_t* b, const uint32_t* w, int n) { memcpy(b, w, 4 * n); }

extern "C" {
// op: 0 mul, 1 sq, 2 invert, 3 add-then-mul(f=a+b loose, g=b), 4 (a-b)*(a-b) via sq of tight difference
void t_fe_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    fe x, y, r;
    fe_frombytes(x, a);
    fe_frombytes(y, b);
    if (op == 0) fe_mul(r, x, y);
    if (op == 1) fe_sq(r, x);
    if (op == 2) fe_invert(r, x);
    if (op == 3) { fe t; fe_add(t, x, y); fe_add(t, t, x); fe_mul(r, t, y); }
    if (op == 4) { fe t; fe_sub(t, x, y); fe_sq(r, t); }
    if (op == 5) { fe t; fe_sub(t, x, y); fe_sub(t, t, y); fe_neg(t, t); fe_carry(r, t); }
    if (op == 6) { fe_add(r, x, y); }                       // canonical encoding of an UNCARRIED sum of two reduced values
    if (op == 7) { fe t; fe_neg(t, y); fe_sub(r, x, t); }   // x - (-y): what fe_equal(check, -u) encodes
    if (op == 8) { fe t; fe_neg(t, x); fe_sub(r, t, y); }   // -x - y
    fe_tobytes(out, r);
}
// k*B by double-and-add over ge_dbl / ge_add, k = 256-bit little-endian integer
static void scalarmul(ge_p3& acc, const ge_p3& base, const uint8_t* k) {
    ge_identity(acc);
    for (int i = 255; i >= 0; i--) {
        ge_p3 t;
        ge_dbl(t, acc, true);
        acc = t;
        if ((k[i >> 3] >> (i & 7)) & 1) { ge_add(t, acc, base); acc = t; }
    }
}
void t_basemul(const uint8_t* k, uint8_t* out) {
    ge_p3 b, acc;
    ge_basepoint(b);
    scalarmul(acc, b, k);
    uint32_t w[8];
    ge_compress(w, acc);
    st(out, w, 8);
}
// k*P via madd with the affine niels form of P (P given compressed); exercises ge_to_niels + ge_madd(+/-)
void t_maddmul(const uint8_t* pc, const uint8_t* k, int neg, uint8_t* out) {
    uint32_t w[8];
    ld(w, pc, 8);
    ge_p3 p, acc, t;
    ge_decompress(p, w);
    ge_niels q;
    ge_to_niels(q, p.X, p.Y);   // decompress returns Z = 1
    ge_identity(acc);
    for (int i = 255; i >= 0; i--) {
        ge_dbl(t, acc, true);
        acc = t;
        if ((k[i >> 3] >> (i & 7)) & 1) { ge_madd(t, acc, q, neg != 0); acc = t; }
    }
    ge_compress(w, acc);
    st(out, w, 8);
}
int t_decompress(const uint8_t* in, uint8_t* out) {
    uint32_t w[8];
    ld(w, in, 8);
    ge_p3 p;
    bool ok = ge_decompress(p, w);
    ge_compress(w, p);
    st(out, w, 8);
    return ok;
}
void t_from_uniform(const uint8_t* in64, uint8_t* out) {
    uint32_t w[16], o[8];
    ld(w, in64, 16);
    ge_p3 p;
    ge_from_uniform(p, w);
    ge_compress(o, p);
    st(out, o, 8);
}
// scalar ops on canonical / arbitrary 32-byte inputs; op: 0 mul, 1 add, 2 sub, 3 invert(a)
void t_sc_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint32_t wa[8], wb[8], o[8];
    ld(wa, a, 8);
    ld(wb, b, 8);
    sc x, y, r;
    sc_to_mont(x, wa);
    sc_to_mont(y, wb);
    if (op == 0) sc_montmul(r, x, y);
    if (op == 1) sc_add(r, x, y);
    if (op == 2) sc_sub(r, x, y);
    if (op == 3) sc_invert_mont(r, x);
    sc_from_mont(o, r);
    st(out, o, 8);
}
void t_sc_from_wide(const uint8_t* in64, uint8_t* out) {
    uint32_t w[16], o[8];
    ld(w, in64, 16);
    sc r;
    sc_from_wide(r, w);
    sc_from_mont(o, r);
    st(out, o, 8);
}
void t_sc_recode(const uint8_t* in, int16_t* d) {
    uint32_t w[8];
    ld(w, in, 8);
    sc_recode_s8(d, w);
}
void t_sc_recode_w(int w, int nw, const uint8_t* in, int* d) {
    uint32_t x[8];
    ld(x, in, 8);
    sc_recode_w(w, nw, x, [&](int i, int v) { d[i] = v; });
}
void t_blake3_32(const uint8_t* in, uint8_t* out) {
    uint32_t w[8], o[8];
    ld(w, in, 8);
    blake3_hash32(o, w);
    st(out, o, 8);
}
void t_blake3_128(const uint8_t* in, uint8_t* out) {
    uint32_t w[32], o[8];
    ld(w, in, 32);
    blake3_hash128(o, w, w + 8, w + 16, w + 24);
    st(out, o, 8);
}
// wide node hashes (Blake2b-512): leaf = D(C32), parent = D(C_L || C_R || H_L64 || H_R64)
void t_blake2b_32(const uint8_t* in, uint8_t* out64) {
    uint32_t w[8], o[16];
    ld(w, in, 8);
    node_hash_leaf_w<16>(DG_BLAKE2B, o, w);
    st(out64, o, 16);
}
void t_blake2b_192(const uint8_t* in, uint8_t* out64) {
    uint32_t w[48], o[16];
    ld(w, in, 48);
    node_hash_parent_w<16>(DG_BLAKE2B, o, w, w + 8, w + 16, w + 32);
    st(out64, o, 16);
}
void t_seed_wide(const uint8_t* seed, uint32_t dom, uint64_t a, uint64_t b, uint8_t* out64) {
    uint32_t s[8], o[16];
    ld(s, seed, 8);
    seed_wide(o, s, dom, a, b);
    st(out64, o, 16);
}
// Merlin "test protocol" vector and a challenge_scalar-shaped 64-byte squeeze
void t_merlin(const char* app, int app_len, const char* label, int ll, const char* msg, int ml, const char* cl, int cll,
              uint8_t* out64) {
    Strobe s;
    merlin_init(s, app, app_len);
    merlin_append_bytes(s, label, ll, msg, ml);
    uint32_t w[16];
    merlin_challenge_wide(s, cl, cll, w);
    st(out64, w, 16);
}
// The m commitments of a proof through the block-parallel form k_rv_absorb_V uses (hash.h: absorb_block_word & co., the same
// functions the kernel calls; the 25 "lanes" are a loop here and the permutation is the one-lane one), in `phases` phases like
// dapol_range_verify_batch, and through the byte-wise Strobe.  app / extra: a transcript head of any length before the
// commitments (the stream then starts at another offset of the 166-byte block).  out: 25 state words + pos + pos_begin, twice.
void t_absorb_v(int m, const uint32_t* Vw, int phases, const char* extra, int extra_len, uint64_t* out_block, uint64_t* out_bytes) {
    Strobe s;
    merlin_init(s, LBL_APP_TRANSCRIPT);
    merlin_append_bytes(s, LBL_DOM_SEP, extra, extra_len);
    merlin_append_u64(s, LBL_M, (uint64_t)m);
    const uint32_t pos0 = s.pos, pb0 = s.pos_begin, end_abs = pos0 + RV_V_BYTES * (uint32_t)m, nfull = end_abs / STROBE_R;
    uint64_t a[25];
    for (int i = 0; i < 25; i++) a[i] = s.s[i];
    const uint32_t last_w = 8 * (uint32_t)m - 1;
    auto word = [&](uint32_t beta, uint32_t l) {
        const uint32_t w0 = absorb_first_word(absorb_window(beta, l, pos0));
        return absorb_block_word(beta, l, pos0, pb0, end_abs, Vw[w0 < last_w ? w0 : last_w], Vw[w0 + 1 < last_w ? w0 + 1 : last_w],
                                 Vw[w0 + 2 < last_w ? w0 + 2 : last_w]);
    };
    for (int k = 0; k < phases; k++) {
        const int j0 = k * (m / phases), j1 = k == phases - 1 ? m : (k + 1) * (m / phases);
        const bool last_phase = j1 >= m;
        const uint32_t beta_begin = j0 > 0 ? (RV_V_BYTES * (uint32_t)j0 + pos0) / STROBE_R : 0u;
        const uint32_t beta_end = last_phase ? nfull : (RV_V_BYTES * (uint32_t)j1 + pos0) / STROBE_R;
        for (uint32_t beta = beta_begin; beta < beta_end; beta++) {
            for (uint32_t l = 0; l < 25; l++) a[l] ^= word(beta, l);
            a[20] ^= absorb_runf_word(beta, pos0, pb0);
            keccak_f1600(a);
        }
        if (last_phase)
            for (uint32_t l = 0; l < 25; l++) a[l] ^= word(beta_end, l);
    }
    uint32_t pe, pbe;
    absorb_end_position(pos0, (uint32_t)m, pe, pbe);
    for (int i = 0; i < 25; i++) out_block[i] = a[i];
    out_block[25] = pe; out_block[26] = pbe;
    for (int j = 0; j < m; j++) merlin_append_words(s, LBL_V, Vw + 8 * j, 8);
    for (int i = 0; i < 25; i++) out_bytes[i] = s.s[i];
    out_bytes[25] = s.pos; out_bytes[26] = s.pos_begin;
}
void t_digest(int kind, const uint8_t* in, int n, uint8_t* out) {
    Digest d;
    uint32_t stack[B3_STACK_DEPTH * 8];
    dg_init_long(d, kind, stack);           // inputs of any length (BLAKE3 beyond one chunk: the chaining-value stack)
    dg_update(d, in, (uint32_t)n);
    uint32_t o[8];
    dg_final(d, o);
    st(out, o, 8);
}
// the single-chunk digest (no stack): returns 1 when the input overflowed one BLAKE3 chunk
int t_digest_short(int kind, const uint8_t* in, int n, uint8_t* out) {
    Digest d;
    dg_init(d, kind);
    dg_update(d, in, (uint32_t)n);
    uint32_t o[8];
    dg_final(d, o);
    st(out, o, 8);
    return d.overflow ? 1 : 0;
}
// decompress(a) + decompress(b) (general addition of two decoded points), compressed
void t_add_compressed(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint32_t wa[8], wb[8], o[8];
    ld(wa, a, 8); ld(wb, b, 8);
    ge_p3 p, q, r;
    ge_decompress(p, wa);
    ge_decompress(q, wb);
    ge_add(r, p, q);
    ge_compress(o, r);
    st(out, o, 8);
}
// the body of k_verify_paths (kernels_verify.h) for one entity, on the host
int t_verify_path(int height, uint64_t idx, const uint8_t* leafC, const uint8_t* leafH, const uint8_t* pC, const uint8_t* pH,
                  const uint8_t* rootC, const uint8_t* rootH) {
    uint32_t c[8], h[8], sc_[8], sh[8], hn[8], rc[8], rh[8];
    ld(c, leafC, 8); ld(h, leafH, 8); ld(rc, rootC, 8); ld(rh, rootH, 8);
    ge_p3 acc, sp;
    bool good = ge_decompress(acc, c);
    for (int k = 0; k < height; k++) {
        size_t slot = (size_t)(height - 1 - k);
        ld(sc_, pC + 32 * slot, 8);
        ld(sh, pH + 32 * slot, 8);
        good &= ge_decompress(sp, sc_);
        if ((idx >> k) & 1) blake3_hash128(hn, sc_, c, sh, h);
        else blake3_hash128(hn, c, sc_, h, sh);
        ge_p3 t;
        ge_add(t, acc, sp);
        acc = t;
        ge_compress(c, acc);
        for (int i = 0; i < 8; i++) h[i] = hn[i];
    }
    for (int i = 0; i < 8; i++) good &= (c[i] == rc[i]) & (h[i] == rh[i]);
    return good;
}
// SHAKE256 / SHA3-512 through the generic sponge
void t_sponge(int rate, int domain, const uint8_t* in, int n, uint8_t* out, int outlen) {
    Sponge sp;
    sponge_init(sp, rate);
    for (int i = 0; i < n; i++) sponge_absorb_byte(sp, in[i]);
    sponge_finish(sp, (uint8_t)domain);
    for (int i = 0; i < outlen; i++) out[i] = sponge_squeeze_byte(sp);
}
}
