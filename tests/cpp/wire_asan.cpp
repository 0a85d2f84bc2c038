// Host-only build of the wire parser (dapol_amd/csrc/host_wire.inc with DAPOL_WIRE_HOST_ONLY: the batched GPU point validation
// replaced by the same ge_decompress on the host) under AddressSanitizer + UBSan: ADVICE r2 (high).  Every output buffer is
// heap-allocated at EXACTLY the size the first (sizes-only) pass reports, so any over-read of the wire or over-write of an output
// is a sanitizer abort.  Cases: well-formed wires of both policies; the advisory's example (h = S = 32 with one 480-byte
// aggregated proof); fewer siblings than levels; every truncation of a good wire; 120,000 random mutations of good wires.
// Build + run: tests/test_wire_asan.py
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dapol_hip.h"
#include "ge.h"

using namespace dapol;

static thread_local std::string g_last_error;
static int32_t fail(int32_t code, const char* msg) { g_last_error = msg; return code; }
struct dapol_ctx { int unused; };
static size_t ctx_hash_bytes(const dapol_ctx* c) { return c->unused == 64 ? 64 : 32; }     // (64: a Blake2b context's 64-byte node hashes)

#define DAPOL_WIRE_HOST_ONLY 1
#include "wire_scope.inc"
#include "policy_plan.inc"
#include "host_wire.inc"

static int g_hb = 32;           // bytes of a node hash in the wires under test: 32, or 64 (a Blake2b context, DapolProofNode = C32 || H64)
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 7; rng_state ^= rng_state >> 9; return rng_state * 0x2545F4914F6CDD1Dull; }

// serialises an all-zero proof set (zero scalars are canonical, the zero string is the identity's encoding) through the library's
// own serialiser: a well-formed wire of the given shape
static std::vector<uint8_t> good_wire(int height, size_t k, size_t S, int policy, int agg, int n_bits) {
    const size_t es = dapol_entity_proof_size((int32_t)S, policy, agg, n_bits);
    if (es == 0) { printf("bad shape\n"); exit(2); }
    std::vector<uint8_t> blob(es, 0), C(S * 32 + 1, 0), H(S * (size_t)g_hb + 1, 7), out(dapol_proof_wire_size_d(g_hb, height, k, S, policy, agg, n_bits));
    std::vector<uint64_t> leaves(k);
    for (size_t i = 0; i < k; i++) leaves[i] = i;
    int32_t rc = dapol_proof_serialize_d(g_hb, height, k, leaves.data(), S, C.data(), H.data(), policy, agg, n_bits, blob.data(), out.data());
    if (rc) { printf("serialize failed %d\n", rc); exit(2); }
    return out;
}

// the two-pass protocol of every binding, with exact-size heap buffers; returns the code of the pass that failed (or 0)
static int32_t parse(int policy, int n_bits, const uint8_t* wire_in, size_t len) {
    uint8_t* wire = (uint8_t*)malloc(len ? len : 1);              // exact-size copy: reads past `len` are caught
    memcpy(wire, wire_in, len);
    dapol_ctx ctx{g_hb};
    int32_t h = 0, agg = 0;
    size_t k = 0, S = 0, bl = 0, cons = 0;
    int32_t rc = dapol_proof_deserialize(&ctx, policy, n_bits, wire, len, &h, &k, &S, &agg, &bl, nullptr, nullptr, nullptr, nullptr, &cons);
    if (rc == DAPOL_OK) {
        uint64_t* leaf = (uint64_t*)malloc(k * 8 ? k * 8 : 1);
        uint8_t* C = (uint8_t*)malloc(S * 32 ? S * 32 : 1);
        uint8_t* H = (uint8_t*)malloc(S * (size_t)g_hb ? S * (size_t)g_hb : 1);
        uint8_t* blob = (uint8_t*)malloc(bl ? bl : 1);
        rc = dapol_proof_deserialize(&ctx, policy, n_bits, wire, len, &h, &k, &S, &agg, &bl, leaf, C, H, blob, &cons);
        if (rc == DAPOL_OK) {
            // what the verifier's entry points will read, from (height, policy, aggregation factor) alone, must be inside what was decoded
            const size_t es = dapol_entity_proof_size((int32_t)S, policy, agg, n_bits);
            if (es != bl || cons > len || (k == 1 && S != (size_t)h)) { printf("accepted an inconsistent wire: es %zu bl %zu S %zu h %d\n", es, bl, S, h); exit(1); }
        }
        free(leaf); free(C); free(H); free(blob);
    }
    free(wire);
    return rc;
}

static void put_be(std::vector<uint8_t>& w, uint64_t x, int nb) { for (int i = nb - 1; i >= 0; i--) w.push_back((uint8_t)(i < 8 ? x >> (8 * i) : 0)); }

int main() {
    int bad = 0;
    auto expect = [&](const char* what, int32_t got, int32_t want) { if (got != want) { printf("FAIL %s: got %d want %d (%s)\n", what, got, want, g_last_error.c_str()); bad++; } };
    struct Shape { int h; size_t k, S; int policy, agg, n_bits; };
    const Shape shapes[] = {{8, 1, 8, DAPOL_POLICY_PADDING, 8, 8}, {8, 1, 8, DAPOL_POLICY_SPLITTING, 5, 8}, {32, 1, 32, DAPOL_POLICY_PADDING, 32, 64},
                            {8, 10, 13, DAPOL_POLICY_SPLITTING, 1, 8}, {6, 1, 6, DAPOL_POLICY_PADDING, 0, 16}, {24, 1, 24, DAPOL_POLICY_SPLITTING, 24, 64}};
    for (int pass = 0; pass < 2; pass++)                  // 32-byte node hashes, then 64-byte ones (two of the shapes)
    for (const Shape& sh : shapes) {
        g_hb = pass ? 64 : 32;
        if (pass && &sh - shapes >= 2 && &sh - shapes != 3) continue;
        std::vector<uint8_t> w = good_wire(sh.h, sh.k, sh.S, sh.policy, sh.agg, sh.n_bits);
        expect("good wire", parse(sh.policy, sh.n_bits, w.data(), w.size()), DAPOL_OK);
        for (size_t cut = 0; cut < w.size(); cut += (w.size() > 4000 ? 7 : 1)) {          // every truncation: an error, never a crash
            int32_t rc = parse(sh.policy, sh.n_bits, w.data(), cut);
            if (rc == DAPOL_OK) { printf("FAIL truncated wire accepted at %zu of %zu\n", cut, w.size()); bad++; break; }
        }
        for (int it = 0; it < 20000; it++) {                                                 // mutations of the framing and of random bytes
            std::vector<uint8_t> m = w;
            const int nmut = 1 + (int)(rnd() % 3);
            for (int j = 0; j < nmut; j++) {
                size_t pos = (rnd() & 1) ? rnd() % (m.size() < 64 ? m.size() : 64) : rnd() % m.size();
                if (rnd() % 4 == 0 && m.size() > 700) pos = m.size() - (size_t)(32 + g_hb) * sh.S - 40 + rnd() % 40;      // the MerkleProof header
                if (pos >= m.size()) pos = m.size() - 1;
                m[pos] = (uint8_t)rnd();
            }
            if (rnd() % 8 == 0) m.resize(rnd() % (m.size() + 1));
            (void)parse(sh.policy, sh.n_bits, m.data(), m.size());                          // (parse() exits on an accepted inconsistent wire)
        }
    }
    g_hb = 32;
    // the advisory's example: h = S = 32, no individual proofs, ONE 480-byte aggregated proof (an 8-bit one-party proof's size)
    {
        std::vector<uint8_t> w;
        put_be(w, 480, 8); w.insert(w.end(), 480, 0); put_be(w, 0, 8);
        put_be(w, 1, 8); put_be(w, 32, 8); put_be(w, 32, 2); w.insert(w.end(), 4, 0); w.insert(w.end(), 32 * 64, 0);
        expect("advisory example", parse(DAPOL_POLICY_PADDING, 8, w.data(), w.size()), DAPOL_ERR_VALUE_DECODING);
    }
    // a single-leaf proof with fewer siblings than levels
    {
        std::vector<uint8_t> w;
        put_be(w, 672, 8); w.insert(w.end(), 672, 0); put_be(w, 0, 8);                     // 8 parties of 8 bits
        put_be(w, 1, 8); put_be(w, 8, 8); put_be(w, 16, 2); w.insert(w.end(), 2, 0); w.insert(w.end(), 8 * 64, 0);
        expect("S < h", parse(DAPOL_POLICY_PADDING, 8, w.data(), w.size()), DAPOL_ERR_VALUE_DECODING);
    }
    // splitting: two aggregated proofs where the plan for (S = 3, agg = 3) wants 2 + 1 parties
    {
        std::vector<uint8_t> w;
        put_be(w, 2, 2);
        for (int i = 0; i < 2; i++) { put_be(w, 480, 8); w.insert(w.end(), 480, 0); }
        put_be(w, 0, 8);
        put_be(w, 1, 8); put_be(w, 3, 8); put_be(w, 3, 2); w.insert(w.end(), 1, 0); w.insert(w.end(), 3 * 64, 0);
        expect("splitting sizes", parse(DAPOL_POLICY_SPLITTING, 8, w.data(), w.size()), DAPOL_ERR_VALUE_DECODING);
    }
    if (bad) { printf("%d failures\n", bad); return 1; }
    printf("wire asan clean\n");
    return 0;
}
