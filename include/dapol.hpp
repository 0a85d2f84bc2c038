// dapol.hpp -- C++ host-side mirror of the reference crate's public surface for the GPU proving path
// (src/lib.rs:1-11), over the C ABI of dapol_hip.h.  Same names, argument meaning and error behaviour:
//   Result<_, DapolError>  -> throws dapol::DapolError          (src/errors.rs:6-17)
//   Option<_>              -> std::optional                      (src/dapol/mod.rs:148-190)
//   panics                 -> throws dapol::DapolError{code 8}   (src/range/padding.rs:95-98, smtree build)
// Header-only; link with -ldapol_hip.  No computation happens here.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dapol_hip.h"

namespace dapol {

using Bytes32 = std::array<uint8_t, 32>;

// utils::get_secret (src/utils.rs:19-26): 32 bytes from the OS CSPRNG.  Every seed of this API that hides something
// (padding blindings, prover nonces, verifier weights) is either passed explicitly or drawn here -- never a zero default.
inline Bytes32 get_secret() {
    Bytes32 s{};
    size_t got = 0;
    if (FILE* f = std::fopen("/dev/urandom", "rb")) {
        got = std::fread(s.data(), 1, s.size(), f);
        std::fclose(f);
    }
    if (got != s.size()) throw std::runtime_error("dapol: no OS randomness (/dev/urandom)");
    return s;
}

struct DapolError : std::runtime_error {      // src/errors.rs:6-17 (+ the ABI's extra codes)
    int32_t code;
    DapolError(int32_t c) : std::runtime_error(std::string(dapol_strerror(c)) + " [" + dapol_last_error() + "]"), code(c) {}
};
inline void check(int32_t rc) { if (rc != DAPOL_OK) throw DapolError(rc); }

// DapolProofNode<D> (src/proof/node.rs:17-22): commitment + hash, both 32 bytes (D = blake3::Hasher).
struct DapolProofNode {
    Bytes32 com{}, hash{};
    bool operator==(const DapolProofNode& o) const { return com == o.com && hash == o.hash; }   // src/proof/node.rs:24-29
};
// DapolNode<D> (src/dapol/node.rs:19-25) as it crosses the boundary: value, blinding, commitment, hash.
struct DapolNode {
    uint64_t v = 0;
    Bytes32 v_blinding{}, com{}, hash{};
    uint64_t get_value() const { return v; }
    const Bytes32& get_blinding() const { return v_blinding; }
    DapolProofNode get_proof_node() const { return {com, hash}; }      // ProofExtractable (node.rs:91-96)
    bool operator==(const DapolNode& o) const { return v == o.v; }     // node.rs:115-120: equal iff values are equal
};

class Context {                                // PedersenGens::default() + BulletproofGens::new(64, m), once
  public:
    // digest = the node hash D of Dapol<D, R>: DAPOL_DIGEST_BLAKE3 or DAPOL_DIGEST_BLAKE2S.  This mirror's node types carry 32-byte
    // hashes (Bytes32), like Dapol::new's DIGEST_SIZE check (src/dapol/mod.rs:101-103): the 64-byte DAPOL_DIGEST_BLAKE2B of the
    // new_blank + build path is served by the C ABI itself (include/dapol_hip.h), and is refused HERE before any buffer could be too small.
    // max_parties = the largest aggregation a proof may need, rounded up to a power of two: 32 serves trees up to height 32
    // (aggregation_factor <= tree_height), 64 the reference's whole range (MAX_TREE_HEIGHT = 64, src/dapol/mod.rs:26)
    explicit Context(int device = 0, int max_parties = 32, int digest = DAPOL_DIGEST_BLAKE3) { digest32(digest); check(dapol_ctx_create(device, max_parties, digest, &h_)); }
    // ... with settings (dapol_options: zeros = the library's own choices; options.struct_size is filled in here)
    Context(int device, int max_parties, int digest, dapol_options options) {
        digest32(digest);
        options.struct_size = (int32_t)sizeof(dapol_options);
        check(dapol_ctx_create_opts(device, max_parties, digest, &options, &h_));
    }
    dapol_options options() const { dapol_options o{}; check(dapol_ctx_get_options(h_, &o)); return o; }
    void set_options(dapol_options o) { o.struct_size = (int32_t)sizeof(dapol_options); check(dapol_ctx_set_options(h_, &o)); }
    ~Context() { dapol_ctx_destroy(h_); }
    static void digest32(int digest) { if (digest != DAPOL_DIGEST_BLAKE3 && digest != DAPOL_DIGEST_BLAKE2S) throw DapolError(DAPOL_ERR_INVALID_DIGEST_SIZE); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    dapol_ctx* get() const { return h_; }
    // DapolNode::new (src/dapol/node.rs:29-45)
    DapolNode node_new(uint64_t value, const Bytes32& blinding) const {
        DapolNode n;
        n.v = value;
        n.v_blinding = blinding;
        check(dapol_commit_hash_batch(h_, 1, &value, blinding.data(), n.com.data(), n.hash.data()));
        return n;
    }
    // Mergeable::merge (src/dapol/node.rs:64-80)
    DapolNode merge(const DapolNode& l, const DapolNode& r) const {
        DapolNode p;
        check(dapol_merge_batch(h_, 1, l.com.data(), l.hash.data(), &l.v, l.v_blinding.data(), r.com.data(), r.hash.data(), &r.v,
                                r.v_blinding.data(), p.com.data(), p.hash.data(), &p.v, p.v_blinding.data()));
        return p;
    }
  private:
    dapol_ctx* h_ = nullptr;
};

enum class Policy { Padding = DAPOL_POLICY_PADDING, Splitting = DAPOL_POLICY_SPLITTING };

// RangeProofPadding / RangeProofSplitting (src/range/padding.rs:20-23, splitting.rs:22-25): aggregated + individual proofs.
struct RangeProofs {
    Policy policy = Policy::Padding;
    std::vector<std::vector<uint8_t>> aggregated, individual;
};

// RangeProvable::generate_proof (src/range/mod.rs:29) for one sibling list, through the batched prover.
inline RangeProofs generate_proof(const Context& ctx, Policy policy, const std::vector<uint64_t>& secrets, const std::vector<Bytes32>& blindings,
                                  size_t aggregation_factor, const Bytes32& nonce_seed, uint64_t stream_id, int n_bits = 64) {
    if (aggregation_factor > secrets.size() || secrets.size() != blindings.size()) throw DapolError(DAPOL_ERR_INVALID_ARGUMENT);
    RangeProofs out;
    out.policy = policy;
    uint64_t slot = 0;
    auto prove = [&](size_t start, size_t count, size_t m) {
        std::vector<uint64_t> v(m, 0);
        std::vector<uint8_t> r(m * 32, 0);
        for (size_t j = 0; j < m; j++) {
            if (j < count) { v[j] = secrets[start + j]; std::memcpy(&r[32 * j], blindings[start + j].data(), 32); }
            else r[32 * j] = 1;                                            // (0, Scalar::one())  padding.rs:100-103
        }
        std::vector<uint8_t> p(dapol_range_proof_size(n_bits, (int)m));
        check(dapol_range_prove_batch(ctx.get(), n_bits, (int)m, 1, v.data(), r.data(), nonce_seed.data(), &stream_id, slot, nullptr, p.data()));
        slot += m * (2 * (uint64_t)n_bits + 4);
        return p;
    };
    auto np2 = [](size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; };
    if (policy == Policy::Padding) {
        out.aggregated.push_back(prove(0, aggregation_factor, np2(aggregation_factor)));
    } else {
        size_t base = np2(aggregation_factor), pos = 0;
        while (pos < aggregation_factor) {
            if (aggregation_factor & base) { out.aggregated.push_back(prove(pos, base, base)); pos += base; }
            base >>= 1;
        }
    }
    for (size_t i = aggregation_factor; i < secrets.size(); i++) out.individual.push_back(prove(i, 1, 1));
    return out;
}

// DapolProof::serialize / deserialize (src/proof/mod.rs:68-85): range_proof || merkle_path, shared by the two proof shapes
// below.  Deserialisation throws DapolError with DecodingError's codes (BytesNotEnough / ValueDecodingError); every sibling
// commitment is decompress-validated on the GPU (src/proof/node.rs:88-94).
inline std::vector<uint8_t> proof_serialize(int height, const std::vector<uint64_t>& leaves, const std::vector<DapolProofNode>& siblings, Policy policy,
                                            size_t aggregation_factor, int n_bits, const std::vector<uint8_t>& range_proofs) {
    size_t S = siblings.size(), n = dapol_proof_wire_size(height, leaves.size(), S, (int)policy, (int)aggregation_factor, n_bits);
    if (n == 0) throw DapolError(DAPOL_ERR_INVALID_ARGUMENT);
    std::vector<uint8_t> C(S * 32 + 1), H(S * 32 + 1), wire(n);
    for (size_t i = 0; i < S; i++) { std::memcpy(&C[i * 32], siblings[i].com.data(), 32); std::memcpy(&H[i * 32], siblings[i].hash.data(), 32); }
    check(dapol_proof_serialize(height, leaves.size(), leaves.data(), S, C.data(), H.data(), (int)policy, (int)aggregation_factor, n_bits, range_proofs.data(),
                                wire.data()));
    return wire;
}
struct DecodedProof {
    int height = 0;
    size_t aggregation_factor = 0;
    std::vector<uint64_t> leaves;
    std::vector<DapolProofNode> siblings;
    std::vector<uint8_t> range_proofs;
};
inline DecodedProof proof_deserialize(const Context& ctx, Policy policy, int n_bits, const std::vector<uint8_t>& wire) {
    int32_t h = 0, agg = 0;
    size_t k = 0, S = 0, bl = 0;
    check(dapol_proof_deserialize(ctx.get(), (int)policy, n_bits, wire.data(), wire.size(), &h, &k, &S, &agg, &bl, nullptr, nullptr, nullptr, nullptr, nullptr));
    DecodedProof d;
    d.height = h; d.aggregation_factor = (size_t)agg;
    d.leaves.resize(k); d.siblings.resize(S); d.range_proofs.resize(bl);
    std::vector<uint8_t> C(S * 32 + 1), H(S * 32 + 1);
    std::vector<uint64_t> leaf(k + 1);
    std::vector<uint8_t> blob(bl + 1);
    check(dapol_proof_deserialize(ctx.get(), (int)policy, n_bits, wire.data(), wire.size(), &h, &k, &S, &agg, &bl, leaf.data(), C.data(), H.data(), blob.data(), nullptr));
    for (size_t i = 0; i < k; i++) d.leaves[i] = leaf[i];
    for (size_t i = 0; i < S; i++) { std::memcpy(d.siblings[i].com.data(), &C[i * 32], 32); std::memcpy(d.siblings[i].hash.data(), &H[i * 32], 32); }
    std::memcpy(d.range_proofs.data(), blob.data(), bl);
    return d;
}

// DapolProof<D, R> (src/proof/mod.rs:15-22) for single leaves: Merkle siblings (root side first) + range proofs.
struct DapolProof {
    uint64_t leaf_index = 0;
    std::vector<DapolProofNode> merkle_siblings;
    std::vector<uint8_t> range_proofs;           // aggregated proofs then individual proofs, concatenated
    Policy policy = Policy::Padding;
    size_t aggregation_factor = 0;
    int n_bits = 64, height = 0;
    std::vector<uint8_t> serialize() const { return proof_serialize(height, {leaf_index}, merkle_siblings, policy, aggregation_factor, n_bits, range_proofs); }
    static DapolProof deserialize(const Context& ctx, Policy policy, int n_bits, const std::vector<uint8_t>& wire) {
        DecodedProof d = proof_deserialize(ctx, policy, n_bits, wire);
        if (d.leaves.size() != 1) throw DapolError(DAPOL_ERR_VALUE_DECODING);
        DapolProof p;
        p.leaf_index = d.leaves[0]; p.merkle_siblings = std::move(d.siblings); p.range_proofs = std::move(d.range_proofs);
        p.policy = policy; p.aggregation_factor = d.aggregation_factor; p.n_bits = n_bits; p.height = d.height;
        return p;
    }
    // DapolProof::verify (src/proof/mod.rs:41-47)
    bool verify(const Context& ctx, const DapolProofNode& root, const DapolProofNode& leaf) const {
        size_t h = merkle_siblings.size();
        std::vector<uint8_t> C(h * 32 + 1), H(h * 32 + 1);
        for (size_t i = 0; i < h; i++) { std::memcpy(&C[i * 32], merkle_siblings[i].com.data(), 32); std::memcpy(&H[i * 32], merkle_siblings[i].hash.data(), 32); }
        uint8_t ok = 0;
        // the length-checked entry point: a proof whose sibling count or range-proof bytes do not fit (height, policy, aggregation
        // factor) -- e.g. one deserialised from hostile bytes -- is simply invalid
        check(dapol_verify_entities_checked(ctx.get(), height, 1, &leaf_index, leaf.com.data(), leaf.hash.data(), h, C.data(), H.data(), root.com.data(),
                                            root.hash.data(), (int)policy, (int)aggregation_factor, n_bits, range_proofs.data(), range_proofs.size(), nullptr, &ok));
        return ok != 0;
    }
};
// DapolProof<D, R> made by generate_proof_batch: MerkleProof::new_batch(leaf indexes) + the deduplicated siblings
// (level by level from the root side, left to right) + the range proofs over exactly those siblings.
struct DapolBatchProof {
    std::vector<uint64_t> leaf_indexes;
    std::vector<DapolProofNode> merkle_siblings;
    std::vector<uint8_t> range_proofs;
    Policy policy = Policy::Padding;
    size_t aggregation_factor = 0;
    int n_bits = 64, height = 0;
    std::vector<uint8_t> serialize() const { return proof_serialize(height, leaf_indexes, merkle_siblings, policy, aggregation_factor, n_bits, range_proofs); }
    static DapolBatchProof deserialize(const Context& ctx, Policy policy, int n_bits, const std::vector<uint8_t>& wire) {
        DecodedProof d = proof_deserialize(ctx, policy, n_bits, wire);
        DapolBatchProof p;
        p.leaf_indexes = std::move(d.leaves); p.merkle_siblings = std::move(d.siblings); p.range_proofs = std::move(d.range_proofs);
        p.policy = policy; p.aggregation_factor = d.aggregation_factor; p.n_bits = n_bits; p.height = d.height;
        return p;
    }
    // DapolProof::verify_batch (src/proof/mod.rs:49-54)
    // Without a seed the library draws the verifier's batching scalars' seed from the OS (the crate uses thread_rng).
    bool verify_batch(const Context& ctx, const DapolProofNode& root, const std::vector<DapolProofNode>& leaves) const {
        return verify_batch_impl(ctx, root, leaves, nullptr);
    }
    bool verify_batch(const Context& ctx, const DapolProofNode& root, const std::vector<DapolProofNode>& leaves, const Bytes32& verify_seed) const {
        return verify_batch_impl(ctx, root, leaves, verify_seed.data());
    }
    bool verify_batch_impl(const Context& ctx, const DapolProofNode& root, const std::vector<DapolProofNode>& leaves, const uint8_t* verify_seed) const {
        if (leaves.size() != leaf_indexes.size()) return false;
        size_t k = leaves.size(), S = merkle_siblings.size();
        std::vector<uint8_t> lC(k * 32), lH(k * 32), sC(S * 32 + 1), sH(S * 32 + 1);
        for (size_t i = 0; i < k; i++) { std::memcpy(&lC[i * 32], leaves[i].com.data(), 32); std::memcpy(&lH[i * 32], leaves[i].hash.data(), 32); }
        for (size_t i = 0; i < S; i++) { std::memcpy(&sC[i * 32], merkle_siblings[i].com.data(), 32); std::memcpy(&sH[i * 32], merkle_siblings[i].hash.data(), 32); }
        uint8_t ok = 0;
        check(dapol_verify_batch_checked(ctx.get(), height, k, leaf_indexes.data(), lC.data(), lH.data(), S, sC.data(), sH.data(), root.com.data(),
                                         root.hash.data(), (int)policy, (int)aggregation_factor, n_bits, range_proofs.data(), range_proofs.size(), verify_seed, &ok));
        return ok != 0;
    }
};

// LiabilityId / Liability / DapolOptions (src/dapol/mod.rs:36-56)
using LiabilityId = std::vector<uint8_t>;
inline LiabilityId liability_id_from_str(const std::string& s) { return LiabilityId(s.begin(), s.end()); }
struct Liability {
    LiabilityId internal_id, external_id;
    uint64_t value = 0;
};
struct DapolOptions {
    std::vector<uint8_t> audit_seed;
    int tree_height = 0;
    size_t aggregation_factor = 0;
    Bytes32 secret = get_secret();    // smtree Secret: here the seed of the positional padding draws.  Random unless the caller
                                      // sets it: a known seed makes every padding blinding publicly derivable (node.rs:86-88)
};

// Dapol<D, R> (src/dapol/mod.rs:78-83)
class Dapol {
  public:
    // Dapol::new (mod.rs:100-128): validation, leaf derivation (build_leaf_nodes, on the GPU), tree build.  D = the
    // context's digest for node hashes AND leaf derivation, as in the reference.  Throws DapolError with the code of
    // TreeHeightTooBig / SparsityTooSmall / DuplicatedInternalId / FailedToMapIndex.
    static Dapol create(std::shared_ptr<Context> ctx, int digest, const std::vector<Liability>& liabilities, const DapolOptions& options,
                        Policy policy = Policy::Padding) {
        Dapol d = new_blank(std::move(ctx), options.tree_height, options.aggregation_factor, policy);
        size_t n = liabilities.size();
        std::vector<uint8_t> iid, eid;
        std::vector<uint32_t> ioff(n + 1, 0), eoff(n + 1, 0);
        std::vector<uint64_t> vals(n);
        for (size_t i = 0; i < n; i++) {
            iid.insert(iid.end(), liabilities[i].internal_id.begin(), liabilities[i].internal_id.end());
            eid.insert(eid.end(), liabilities[i].external_id.begin(), liabilities[i].external_id.end());
            ioff[i + 1] = (uint32_t)iid.size();
            eoff[i + 1] = (uint32_t)eid.size();
            vals[i] = liabilities[i].value;
        }
        iid.push_back(0); eid.push_back(0);                                 // never pass a null data pointer
        std::vector<uint64_t> idx(n), v(n), by_entity(n);
        std::vector<uint32_t> order(n);
        std::vector<Bytes32> r(n);
        check(dapol_build_leaf_nodes(d.ctx_->get(), digest, options.audit_seed.data(), options.audit_seed.size(), options.tree_height, n, iid.data(),
                                     ioff.data(), eid.data(), eoff.data(), vals.data(), idx.data(), v.data(), n ? r[0].data() : nullptr,
                                     order.data(), by_entity.data()));
        for (size_t i = 0; i < n; i++) d.id_to_idx_map_[liabilities[i].internal_id] = by_entity[i];
        d.build(idx, v, r, options.secret, true);
        return d;
    }
    // Dapol::generate_proof_for_id / generate_proof_batch_for_ids (mod.rs:148-164): None for an unknown id.
    std::optional<DapolProof> generate_proof_for_id(const LiabilityId& id, const Bytes32& nonce_seed, int n_bits = 64) const {
        auto it = id_to_idx_map_.find(id);
        if (it == id_to_idx_map_.end()) return std::nullopt;
        return generate_proof(it->second, nonce_seed, n_bits);
    }
    std::optional<DapolBatchProof> generate_proof_batch_for_ids(const std::vector<LiabilityId>& ids, const Bytes32& nonce_seed, int n_bits = 64) const {
        std::vector<uint64_t> idx;
        for (auto& id : ids) {
            auto it = id_to_idx_map_.find(id);
            if (it == id_to_idx_map_.end()) return std::nullopt;
            idx.push_back(it->second);
        }
        return generate_proof_batch(idx, nonce_seed, n_bits);
    }
    const std::map<LiabilityId, uint64_t>& id_to_idx_map() const { return id_to_idx_map_; }
    // Dapol::new_blank (mod.rs:196-204)
    static Dapol new_blank(std::shared_ptr<Context> ctx, int height, size_t aggregation_factor, Policy policy = Policy::Padding) {
        Dapol d;
        d.ctx_ = std::move(ctx);
        d.height_ = height;
        d.aggregation_factor_ = aggregation_factor;
        d.policy_ = policy;
        return d;
    }
    // Dapol::build (mod.rs:206-208): input = sorted (tree index, value, blinding); pad_seed stands in for thread_rng().
    void build(const std::vector<uint64_t>& idx, const std::vector<uint64_t>& values, const std::vector<Bytes32>& blindings,
               const Bytes32& pad_seed, bool enforce_sparsity = false) {
        if (idx.size() != values.size() || idx.size() != blindings.size()) throw DapolError(DAPOL_ERR_INVALID_ARGUMENT);
        dapol_tree* t = nullptr;
        check(dapol_tree_build(ctx_->get(), height_, idx.size(), idx.data(), values.data(), blindings.empty() ? nullptr : blindings[0].data(),
                               pad_seed.data(), enforce_sparsity ? 1 : 0, &t));
        tree_.reset(t, [](dapol_tree* p) { dapol_tree_destroy(p); });
    }
    // Dapol::update (mod.rs:211-213): inserts the liability at idx or replaces the one already there.  On a blank
    // Dapol the first update creates the tree with pad_seed; later calls ignore pad_seed (a tree has one seed).  Without a
    // seed a blank Dapol draws one from the OS (the reference draws every padding blinding from thread_rng, node.rs:87).
    void update(uint64_t idx, uint64_t value, const Bytes32& blinding) { update(idx, value, blinding, tree_ ? Bytes32{} : get_secret()); }
    void update(uint64_t idx, uint64_t value, const Bytes32& blinding, const Bytes32& pad_seed) {
        if (!tree_) return build({idx}, {value}, {blinding}, pad_seed);
        check(dapol_tree_update(tree_.get(), 1, &idx, &value, blinding.data()));
    }
    // The batched form: k updates applied in order by one level-parallel rebuild.
    void update(const std::vector<uint64_t>& idx, const std::vector<uint64_t>& values, const std::vector<Bytes32>& blindings) {
        update(idx, values, blindings, tree_ ? Bytes32{} : get_secret());
    }
    void update(const std::vector<uint64_t>& idx, const std::vector<uint64_t>& values, const std::vector<Bytes32>& blindings,
                const Bytes32& pad_seed) {
        if (idx.size() != values.size() || idx.size() != blindings.size()) throw DapolError(DAPOL_ERR_INVALID_ARGUMENT);
        if (idx.empty()) return;
        if (!tree_) {
            update(idx[0], values[0], blindings[0], pad_seed);
            if (idx.size() > 1) check(dapol_tree_update(tree_.get(), idx.size() - 1, idx.data() + 1, values.data() + 1, blindings[1].data()));
            return;
        }
        check(dapol_tree_update(tree_.get(), idx.size(), idx.data(), values.data(), blindings[0].data()));
    }
    // Dapol::root_raw / Dapol::root (mod.rs:134-141)
    DapolNode root_raw() const {
        DapolNode n;
        check(dapol_tree_root(tree_.get(), n.com.data(), n.hash.data(), &n.v, n.v_blinding.data()));
        return n;
    }
    DapolProofNode root() const { return root_raw().get_proof_node(); }
    // Dapol::generate_proof (mod.rs:167-169): None when there is no liability at the leaf.
    std::optional<DapolProof> generate_proof(uint64_t leaf_idx, const Bytes32& nonce_seed, int n_bits = 64) const {
        auto v = generate_proofs({leaf_idx}, nonce_seed, n_bits);
        if (!v) return std::nullopt;
        return std::move((*v)[0]);
    }
    // Dapol::generate_proof_batch (mod.rs:172-190): ONE proof for all the leaves (strictly increasing indexes); None
    // when there is no liability at one of them.
    std::optional<DapolBatchProof> generate_proof_batch(const std::vector<uint64_t>& leaves, const Bytes32& nonce_seed, int n_bits = 64) const {
        size_t S = 0;
        check(dapol_batch_siblings(height_, leaves.size(), leaves.data(), &S, nullptr, nullptr));
        size_t es = dapol_entity_proof_size((int)S, (int)policy_, (int)aggregation_factor_, n_bits);
        if (es == 0) throw DapolError(DAPOL_ERR_INVALID_ARGUMENT);
        std::vector<uint8_t> C(S * 32 + 1), H(S * 32 + 1);
        DapolBatchProof out;
        out.range_proofs.resize(es);
        int32_t rc = dapol_prove_batch(ctx_->get(), tree_.get(), leaves.size(), leaves.data(), (int)policy_, (int)aggregation_factor_, n_bits,
                                       nonce_seed.data(), C.data(), H.data(), out.range_proofs.data());
        if (rc == DAPOL_ERR_UNKNOWN_LEAF) return std::nullopt;
        check(rc);
        out.leaf_indexes = leaves;
        out.merkle_siblings.resize(S);
        for (size_t s = 0; s < S; s++) {
            std::memcpy(out.merkle_siblings[s].com.data(), &C[s * 32], 32);
            std::memcpy(out.merkle_siblings[s].hash.data(), &H[s * 32], 32);
        }
        out.policy = policy_; out.aggregation_factor = aggregation_factor_; out.n_bits = n_bits; out.height = height_;
        return out;
    }
    // Many single-leaf proofs in one GPU batch (the throughput path).
    std::optional<std::vector<DapolProof>> generate_proofs(const std::vector<uint64_t>& leaves, const Bytes32& nonce_seed, int n_bits = 64) const {
        size_t es = dapol_entity_proof_size(height_, (int)policy_, (int)aggregation_factor_, n_bits);
        if (es == 0) throw DapolError(DAPOL_ERR_INVALID_ARGUMENT);
        size_t b = leaves.size(), h = (size_t)height_;
        std::vector<uint8_t> C(b * h * 32), H(b * h * 32), R(b * es);
        int32_t rc = dapol_prove_entities(ctx_->get(), tree_.get(), b, leaves.data(), (int)policy_, (int)aggregation_factor_, n_bits,
                                          nonce_seed.data(), C.data(), H.data(), R.data());
        if (rc == DAPOL_ERR_UNKNOWN_LEAF) return std::nullopt;
        check(rc);
        std::vector<DapolProof> out(b);
        for (size_t e = 0; e < b; e++) {
            out[e].leaf_index = leaves[e];
            out[e].merkle_siblings.resize(h);
            for (size_t s = 0; s < h; s++) {
                std::memcpy(out[e].merkle_siblings[s].com.data(), &C[(e * h + s) * 32], 32);
                std::memcpy(out[e].merkle_siblings[s].hash.data(), &H[(e * h + s) * 32], 32);
            }
            out[e].range_proofs.assign(R.begin() + e * es, R.begin() + (e + 1) * es);
            out[e].policy = policy_; out[e].aggregation_factor = aggregation_factor_; out[e].n_bits = n_bits; out[e].height = height_;
        }
        return out;
    }
  private:
    std::map<LiabilityId, uint64_t> id_to_idx_map_;
    std::shared_ptr<Context> ctx_;
    std::shared_ptr<dapol_tree> tree_;
    int height_ = 0;
    size_t aggregation_factor_ = 0;
    Policy policy_ = Policy::Padding;
};

}  // namespace dapol
