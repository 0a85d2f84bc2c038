// Exercises include/dapol.hpp (the C++ mirror of the reference's public surface) against libdapol_hip.so.
// Without a GPU it checks the loud-failure contract (DapolError code 16, no CPU fallback) and the host-only size
// functions; with a GPU it runs Dapol::new_blank + build + root + generate_proof and prints the root for the Python
// test to compare with the C-ABI result.
#include <cstdio>
#include <cstring>
#include "dapol.hpp"

int main(int argc, char** argv) {
    using namespace dapol;
    if (dapol_range_proof_size(64, 1) != 672 || dapol_entity_proof_size(32, DAPOL_POLICY_PADDING, 32, 64) != 992) {
        std::printf("FAIL sizes\n");
        return 1;
    }
    std::shared_ptr<Context> ctx;
    try {
        ctx = std::make_shared<Context>(0, 8);
    } catch (const DapolError& e) {
        if (e.code == DAPOL_ERR_NO_DEVICE) { std::printf("NO_DEVICE %s\n", e.what()); return 0; }
        std::printf("FAIL ctx %d\n", e.code);
        return 1;
    }
    const int height = 6;
    std::vector<uint64_t> idx = {3, 9, 40}, vals = {5, 7, 11};
    std::vector<Bytes32> bl(3);
    for (int i = 0; i < 3; i++) { bl[i].fill(0); bl[i][0] = (uint8_t)(i + 1); bl[i][5] = 0x77; }
    Bytes32 seed;
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)i;
    Dapol d = Dapol::new_blank(ctx, height, height, Policy::Padding);
    d.build(idx, vals, bl, seed);
    DapolNode root = d.root_raw();
    std::printf("ROOT ");
    for (uint8_t b : root.com) std::printf("%02x", b);
    std::printf(" ");
    for (uint8_t b : root.hash) std::printf("%02x", b);
    std::printf(" %llu\n", (unsigned long long)root.get_value());
    auto proof = d.generate_proof(9, seed, 8);
    auto none = d.generate_proof(10, seed, 8);                     // no liability at leaf 10 -> None
    if (!proof || none || proof->merkle_siblings.size() != (size_t)height) { std::printf("FAIL proof\n"); return 1; }
    DapolNode a = ctx->node_new(5, bl[0]), b = ctx->node_new(7, bl[1]);
    DapolNode m = ctx->merge(a, b);
    if (m.get_value() != 12) { std::printf("FAIL merge\n"); return 1; }
    try {
        Dapol bad = Dapol::new_blank(ctx, height, height, Policy::Padding);
        bad.build({9, 3}, {1, 2}, {bl[0], bl[1]}, seed);           // unsorted input: the reference panics in smtree
        std::printf("FAIL unsorted accepted\n");
        return 1;
    } catch (const DapolError& e) {
        if (e.code != DAPOL_ERR_INVALID_ARGUMENT) { std::printf("FAIL code %d\n", e.code); return 1; }
    }
    {   // src/tests.rs:41-48: a blank Dapol grown by update() has the same root as build()
        Dapol u = Dapol::new_blank(ctx, height, height, Policy::Padding);
        u.update(idx[2], vals[2], bl[2], seed);
        u.update(idx[0], vals[0], bl[0]);
        u.update(idx[1], 1, bl[0]);
        u.update(idx[1], vals[1], bl[1]);                          // replaces
        DapolNode ur = u.root_raw();
        if (ur.com != root.com || ur.hash != root.hash || ur.get_value() != root.get_value()) { std::printf("FAIL update\n"); return 1; }
    }
    {   // src/tests.rs:50-70: one proof for a batch of leaves verifies against root and leaves
        Dapol bd = Dapol::new_blank(ctx, height, 2, Policy::Splitting);
        bd.build(idx, vals, bl, seed);
        auto bp = bd.generate_proof_batch({3, 40}, seed, 8);
        if (!bp || bp->merkle_siblings.empty()) { std::printf("FAIL batch\n"); return 1; }
        std::vector<DapolProofNode> lv = {ctx->node_new(5, bl[0]).get_proof_node(), ctx->node_new(11, bl[2]).get_proof_node()};
        if (!bp->verify_batch(*ctx, bd.root(), lv, seed)) { std::printf("FAIL batch verify\n"); return 1; }
        std::swap(lv[0], lv[1]);
        if (bp->verify_batch(*ctx, bd.root(), lv, seed)) { std::printf("FAIL batch verify accepted swapped leaves\n"); return 1; }
        if (bd.generate_proof_batch({3, 41}, seed, 8)) { std::printf("FAIL batch none\n"); return 1; }
        // src/proof/tests.rs:6-35: serialize -> deserialize -> verify_batch; and the single-leaf flavour (src/tests.rs:72-93)
        std::swap(lv[0], lv[1]);
        std::vector<uint8_t> wire = bp->serialize();
        DapolBatchProof back = DapolBatchProof::deserialize(*ctx, Policy::Splitting, 8, wire);
        if (back.leaf_indexes != bp->leaf_indexes || back.range_proofs != bp->range_proofs || back.aggregation_factor != 2 ||
            !back.verify_batch(*ctx, bd.root(), lv)) { std::printf("FAIL batch wire round trip\n"); return 1; }
        auto sp = bd.generate_proof(40, seed, 8);
        DapolProof sback = DapolProof::deserialize(*ctx, Policy::Splitting, 8, sp->serialize());
        if (sback.leaf_index != 40 || !sback.verify(*ctx, bd.root(), lv[1])) { std::printf("FAIL single wire round trip\n"); return 1; }
        if (sback.verify(*ctx, bd.root(), lv[0])) { std::printf("FAIL single verify accepted another leaf\n"); return 1; }
        wire.pop_back();
        try { DapolBatchProof::deserialize(*ctx, Policy::Splitting, 8, wire); std::printf("FAIL truncated accepted\n"); return 1; }
        catch (const DapolError& e) { if (e.code != DAPOL_ERR_BYTES_NOT_ENOUGH) { std::printf("FAIL truncated code %d\n", e.code); return 1; } }
    }
    {   // src/dapol/tests.rs:18-107 with Dapol::<blake2::Blake2s, RangeProofPadding>::new
        auto c2 = std::make_shared<Context>(0, 8, DAPOL_DIGEST_BLAKE2S);
        std::vector<Liability> liab;
        const char* iids[4] = {"a", "b", "c", "d"};
        const char* eids[4] = {"w", "x", "y", "z"};
        const uint64_t lv[4] = {3, 5, 7, 11};
        for (int i = 0; i < 4; i++) liab.push_back({liability_id_from_str(iids[i]), liability_id_from_str(eids[i]), lv[i]});
        DapolOptions opt;
        opt.audit_seed = {'t', 'e', 's', 't'};
        opt.tree_height = 4;
        opt.aggregation_factor = 2;
        opt.secret = seed;
        Dapol t = Dapol::create(c2, DAPOL_DIGEST_BLAKE2S, liab, opt, Policy::Padding);
        if (t.root_raw().get_value() != 26) { std::printf("FAIL root value\n"); return 1; }                   // tests.rs:24
        const uint64_t want[4] = {7, 12, 2, 4};                                                                // tests.rs:30-85
        for (int i = 0; i < 4; i++) {
            auto by_id = t.generate_proof_for_id(liability_id_from_str(iids[i]), seed, 8);
            auto by_ix = t.generate_proof(want[i], seed, 8);
            if (!by_id || !by_ix || by_id->leaf_index != want[i] || by_id->range_proofs != by_ix->range_proofs) { std::printf("FAIL for_id %d\n", i); return 1; }
            for (size_t s2 = 0; s2 < by_id->merkle_siblings.size(); s2++)
                if (by_id->merkle_siblings[s2].com != by_ix->merkle_siblings[s2].com || by_id->merkle_siblings[s2].hash != by_ix->merkle_siblings[s2].hash) { std::printf("FAIL path %d\n", i); return 1; }
        }
        auto ba = t.generate_proof_batch_for_ids({liability_id_from_str("a"), liability_id_from_str("b")}, seed, 8);   // tests.rs:88-107
        auto bi = t.generate_proof_batch({7, 12}, seed, 8);
        if (!ba || !bi || ba->range_proofs != bi->range_proofs || ba->merkle_siblings.size() != bi->merkle_siblings.size()) { std::printf("FAIL batch_for_ids\n"); return 1; }
        if (t.generate_proof_for_id(liability_id_from_str("zz"), seed, 8)) { std::printf("FAIL unknown id\n"); return 1; }
        try {
            liab.push_back({liability_id_from_str("a"), liability_id_from_str("q"), 1});
            Dapol::create(c2, DAPOL_DIGEST_BLAKE2S, liab, opt);
            std::printf("FAIL duplicate accepted\n");
            return 1;
        } catch (const DapolError& e) {
            if (e.code != DAPOL_ERR_DUPLICATED_INTERNAL_ID) { std::printf("FAIL dup code %d\n", e.code); return 1; }
        }
    }
    std::printf("OK proof_bytes=%zu\n", proof->range_proofs.size());
    return 0;
}
