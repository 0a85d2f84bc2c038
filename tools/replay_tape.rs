//! replay_tape.rs -- closes the "parity unpinned" items of this repository on a machine that HAS cargo.
//!
//! NOT COMPILED HERE (the build image has no Rust toolchain, no crates, no network): written against the public APIs of
//! bulletproofs 4.0.0, curve25519-dalek-ng 4.1.1, merlin 3.0.0, smtree 0.1.2, blake3 0.3.8 as the reference's Cargo.toml
//! pins them.  Drop it into the reference crate as `examples/replay_tape.rs`
//! (add `serde_json = "1"` and `hex = "0.4"` to [dev-dependencies]) and run
//!
//!     cargo run --release --example replay_tape -- <path to this repo>/tests/golden/from_reference
//!
//! It replays THIS library's randomness contract (include/dapol_hip.h, "Randomness contract") through the real crates
//! and dumps what they produce; `pytest tests/test_from_reference.py` then compares the oracle and the GPU with those
//! bytes.  Every file pins one assumption DESIGN.md section 2 lists:
//!   range_*.json  transcript labels, generator chain, party draw order, proof layout   (RangeProof::prove_multiple_with_rng)
//!   tree_*.json   node algebra, padding positions, sibling ORDER, MerkleProof wire bytes (smtree build + proofs)
//!   usize_*.json  byte order of smtree::utils::usize_to_bytes
//! If a vector disagrees, the fix is one line: dapol_amd/csrc/labels.h (labels), dapol_wire_config (wire / order).
use bulletproofs::{BulletproofGens, PedersenGens, RangeProof};
use curve25519_dalek_ng::scalar::Scalar;
use merlin::Transcript;
use rand::{CryptoRng, Error, RngCore};
use smtree::{
    index::TreeIndex,
    pad_secret::Secret,
    traits::{Mergeable, Paddable, ProofExtractable, Serializable},
    tree::SparseMerkleTree,
    utils::usize_to_bytes,
};
use std::{env, fs, path::Path};

use dapol::{DapolNode, DapolProofNode};

// ---------------------------------------------------------------------------------------------------------------------
// Seed mode: draw(seed; domain, a, b) = first 64-byte XOF block of BLAKE3-keyed(seed, LE32 domain | LE64 a | LE64 b).
fn draw(seed: &[u8; 32], domain: u32, a: u64, b: u64) -> [u8; 64] {
    let mut msg = [0u8; 20];
    msg[0..4].copy_from_slice(&domain.to_le_bytes());
    msg[4..12].copy_from_slice(&a.to_le_bytes());
    msg[12..20].copy_from_slice(&b.to_le_bytes());
    let mut h = blake3::Hasher::new_keyed(seed);
    h.update(&msg);
    let mut out = [0u8; 64];
    h.finalize_xof().fill(&mut out);
    out
}
fn first32(w: [u8; 64]) -> [u8; 32] {
    let mut k = [0u8; 32];
    k.copy_from_slice(&w[..32]);
    k
}
/// Nonce key of one proof: seed -> (6: stream, first slot) -> (7: n, m) -> chained BLAKE3 over the commitments, 31 per chunk.
fn nonce_key(seed: &[u8; 32], stream: u64, slot_base: u64, n: usize, m: usize, commitments: &[[u8; 32]]) -> [u8; 32] {
    let mut key = first32(draw(seed, 6, stream, slot_base));
    key = first32(draw(&key, 7, n as u64, m as u64));
    for group in commitments.chunks(31) {
        let mut h = blake3::Hasher::new();
        h.update(&key);
        for c in group {
            h.update(c);
        }
        key = *h.finalize().as_bytes();
    }
    key
}
/// The RNG handed to the prover: every Scalar::random(rng) is ONE fill_bytes of 64 bytes = one slot of the tape.
struct TapeRng {
    key: [u8; 32],
    stream: u64,
    slot: u64,
}
impl RngCore for TapeRng {
    fn next_u32(&mut self) -> u32 { unimplemented!("the prover only calls fill_bytes(64)") }
    fn next_u64(&mut self) -> u64 { unimplemented!("the prover only calls fill_bytes(64)") }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        assert_eq!(dest.len(), 64, "Scalar::random draws 64 bytes; anything else breaks the slot contract");
        dest.copy_from_slice(&draw(&self.key, 2, self.stream, self.slot));
        self.slot += 1;
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), Error> { self.fill_bytes(dest); Ok(()) }
}
impl CryptoRng for TapeRng {}

fn dump_range(out: &Path, n: usize, m: usize, seed: &[u8; 32], stream: u64) {
    let pc = PedersenGens::default();
    let bp = BulletproofGens::new(64, m);            // the reference always builds 64-bit generators (src/range/mod.rs:50,66)
    let values: Vec<u64> = (0..m as u64).map(|j| (j * 2654435761 + 12345) & ((1u128 << n) - 1) as u64).collect();
    let blindings: Vec<Scalar> = (0..m as u64).map(|j| Scalar::from_bytes_mod_order_wide(&draw(seed, 9, stream, j))).collect();
    let commitments: Vec<[u8; 32]> = values.iter().zip(&blindings).map(|(v, r)| pc.commit(Scalar::from(*v), *r).compress().to_bytes()).collect();
    let mut rng = TapeRng { key: nonce_key(seed, stream, 0, n, m, &commitments), stream, slot: 0 };
    let mut t = Transcript::new(&[]);              // src/range/mod.rs:51,67
    let (proof, coms) = RangeProof::prove_multiple_with_rng(&bp, &pc, &mut t, &values, &blindings, n, &mut rng).expect("prove");
    assert_eq!(rng.slot as usize, m * (2 * n + 4), "draw count differs from the slot contract");
    for (a, b) in coms.iter().zip(&commitments) { assert_eq!(&a.to_bytes(), b); }
    let j = serde_json::json!({
        "kind": "range", "n": n, "m": m, "seed": hex::encode(seed), "stream_id": stream, "values": values,
        "blindings": blindings.iter().map(|s| hex::encode(s.as_bytes())).collect::<Vec<_>>(),
        "commitments": commitments.iter().map(hex::encode).collect::<Vec<_>>(),
        "proof": hex::encode(proof.to_bytes()),
    });
    fs::write(out.join(format!("range_{}_{}.json", n, m)), serde_json::to_string_pretty(&j).unwrap()).unwrap();
}

// ---------------------------------------------------------------------------------------------------------------------
// A DapolNode whose padding is positional (what the arguments of Paddable::padding -- ignored by the reference,
// src/dapol/node.rs:86-88 -- were meant for): blinding = draw(secret; 1, level above the leaves, index in the level) mod l.
#[derive(Clone, Default, Debug)]
struct TapedNode(DapolNode<blake3::Hasher>);
thread_local! { static TREE_HEIGHT: std::cell::Cell<usize> = std::cell::Cell::new(0); }
fn index_of(idx: &TreeIndex) -> u64 {
    let (h, path) = (idx.get_height(), idx.get_path());
    let mut x = 0u64;
    for b in 0..h { x = (x << 1) | ((path[b / 8] >> (7 - b % 8)) & 1) as u64; }
    x
}
impl Mergeable for TapedNode {
    fn merge(l: &TapedNode, r: &TapedNode) -> TapedNode { TapedNode(DapolNode::merge(&l.0, &r.0)) }
}
impl Paddable for TapedNode {
    fn padding(idx: &TreeIndex, secret: &Secret) -> TapedNode {
        let mut seed = [0u8; 32];
        seed.copy_from_slice(secret.as_bytes());
        let level = TREE_HEIGHT.with(|h| h.get()) - idx.get_height();
        let r = Scalar::from_bytes_mod_order_wide(&draw(&seed, 1, level as u64, index_of(idx)));
        TapedNode(DapolNode::new(0, r))
    }
}
impl ProofExtractable for TapedNode {
    type ProofNode = DapolProofNode<blake3::Hasher>;
    fn get_proof_node(&self) -> Self::ProofNode { self.0.get_proof_node() }
}

fn dump_tree(out: &Path, height: usize, leaves: &[(u64, u64)], seed: &[u8; 32]) {
    TREE_HEIGHT.with(|h| h.set(height));
    let secret = Secret::from_bytes(seed).expect("secret");      // adapt to smtree 0.1.2's constructor if it differs
    let list: Vec<(TreeIndex, TapedNode)> = leaves.iter().map(|(i, v)| {
        let r = Scalar::from_bytes_mod_order_wide(&draw(seed, 9, *i, 0));
        (TreeIndex::from_u64(height, *i), TapedNode(DapolNode::new(*v, r)))
    }).collect();
    let mut tree = SparseMerkleTree::<TapedNode>::new(height);
    tree.build(&list, &secret);
    let root = tree.get_root();
    let mut paths = Vec::new();
    for (idx, _) in &list {
        let proof = smtree::proof::MerkleProof::<TapedNode>::generate_inclusion_proof(&tree, &[*idx]).expect("proof");
        let sibs: Vec<_> = (0..proof.get_siblings_num()).map(|k| {
            let s = proof.get_sibling_at_idx(k);
            serde_json::json!({"C": hex::encode(s.get_com().compress().as_bytes()), "H": hex::encode(s.get_hash())})
        }).collect();
        paths.push(serde_json::json!({"leaf": index_of(idx), "siblings": sibs, "merkle_wire": hex::encode(proof.serialize())}));
    }
    let batch: Vec<TreeIndex> = list.iter().take(3).map(|x| x.0).collect();
    let bproof = smtree::proof::MerkleProof::<TapedNode>::generate_inclusion_proof(&tree, &batch).expect("batch proof");
    let bsibs: Vec<_> = (0..bproof.get_siblings_num()).map(|k| hex::encode(bproof.get_sibling_at_idx(k).get_com().compress().as_bytes())).collect();
    let j = serde_json::json!({
        "kind": "tree", "height": height, "pad_seed": hex::encode(seed),
        "leaves": list.iter().zip(leaves).map(|((_, n), (i, v))| serde_json::json!({"idx": i, "v": v, "r": hex::encode(n.0.get_blinding().as_bytes())})).collect::<Vec<_>>(),
        "root_C": hex::encode(root.0.get_com().compress().as_bytes()), "root_H": hex::encode(root.0.get_hash()), "root_v": root.0.get_value(),
        "paths": paths,
        "batch": {"leaves": batch.iter().map(index_of).collect::<Vec<_>>(), "sibling_C": bsibs, "merkle_wire": hex::encode(bproof.serialize())},
    });
    fs::write(out.join(format!("tree_{}.json", height)), serde_json::to_string_pretty(&j).unwrap()).unwrap();
}

fn main() {
    let dir = env::args().nth(1).expect("usage: replay_tape <tests/golden/from_reference>");
    let out = Path::new(&dir);
    fs::create_dir_all(out).unwrap();
    let seed: [u8; 32] = core::array::from_fn(|i| i as u8);
    for (n, m) in [(8usize, 1usize), (8, 2), (16, 4), (64, 1), (64, 2), (64, 32)] {
        dump_range(out, n, m, &seed, 7);
    }
    dump_tree(out, 4, &[(2, 7), (4, 11), (7, 3), (12, 5)], &seed);
    dump_tree(out, 8, &[(1, 10), (2, 20), (77, 30), (200, 40), (201, 50), (255, 60)], &seed);
    for (value, bytes) in [(0x0102usize, 2usize), (672, 8), (1, 8)] {
        let j = serde_json::json!({"kind": "usize", "value": value, "bytes": bytes, "hex": hex::encode(usize_to_bytes(value, bytes))});
        fs::write(out.join(format!("usize_{}_{}.json", value, bytes)), j.to_string()).unwrap();
    }
}
