//! replay_tape.rs -- closes the "parity unpinned" items of this repository on a machine that HAS cargo.
//!
//! NOT COMPILED HERE (the build image has no Rust toolchain, no crates, no network): written against the public APIs of
//! bulletproofs 4.0.0, curve25519-dalek-ng 4.1.1, merlin 3.0.0, smtree 0.1.2, blake3 0.3.8 as the reference's Cargo.toml
//! pins them.  Drop it into the reference crate as `examples/replay_tape.rs`
//! (add `serde_json = "1"` and `hex = "0.4"` to [dev-dependencies]) and run
//!
//!     cargo run --release --example replay_tape -- <path to this repo>/tests/golden/from_reference
//!
//! It runs the real crates on a recorded byte stream (every Scalar::random draw is kept: the TAPE of include/dapol_hip.h's
//! randomness contract) and dumps the draws beside what the crates produced; `pytest tests/test_from_reference.py` then replays the
//! tapes through the oracle and through the GPU library's tape entry points and compares the bytes.  Nothing of this library's seed
//! derivation is re-implemented on the Rust side.  Every file pins one assumption DESIGN.md section 2 lists:
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
// TAPE mode (round 5): nothing of this library's seed derivation is re-implemented here.  The harness hands the crates an ordinary
// deterministic byte stream, RECORDS every 64-byte draw they take (one Scalar::random = one fill_bytes(64) = one tape slot) and
// dumps the draws beside what the crates made of them; the library then replays the very same draws through its tape entry
// points -- dapol_range_prove_batch(tape), dapol_prove_entities_tape, dapol_tree_build_tape -- and must produce the same bytes.
struct RecRng {
    state: u64,                 // SplitMix64: any deterministic stream will do, the tape carries the bytes themselves
    tape: Vec<[u8; 64]>,
}
impl RecRng {
    fn new(seed: u64) -> Self { RecRng { state: seed, tape: Vec::new() } }
    fn next(&mut self) -> u64 {
        self.state = self.state.wrapping_add(0x9E3779B97F4A7C15);
        let mut z = self.state;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
        z ^ (z >> 31)
    }
    fn draw(&mut self) -> [u8; 64] {
        let mut out = [0u8; 64];
        for k in 0..8 { out[8 * k..8 * k + 8].copy_from_slice(&self.next().to_le_bytes()); }
        self.tape.push(out);
        out
    }
}
impl RngCore for RecRng {
    fn next_u32(&mut self) -> u32 { unimplemented!("the prover only calls fill_bytes(64)") }
    fn next_u64(&mut self) -> u64 { unimplemented!("the prover only calls fill_bytes(64)") }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        assert_eq!(dest.len(), 64, "Scalar::random draws 64 bytes; anything else breaks the slot contract");
        let d = self.draw();
        dest.copy_from_slice(&d);
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), Error> { self.fill_bytes(dest); Ok(()) }
}
impl CryptoRng for RecRng {}

fn dump_range(out: &Path, n: usize, m: usize, seed: u64) {
    let pc = PedersenGens::default();
    let bp = BulletproofGens::new(64, m);            // the reference always builds 64-bit generators (src/range/mod.rs:50,66)
    let values: Vec<u64> = (0..m as u64).map(|j| (j * 2654435761 + 12345) & ((1u128 << n) - 1) as u64).collect();
    let mut brng = RecRng::new(seed ^ 0xB11D);       // the parties' blindings: inputs of the proof, not part of its tape
    let blindings: Vec<Scalar> = (0..m).map(|_| Scalar::from_bytes_mod_order_wide(&brng.draw())).collect();
    let commitments: Vec<[u8; 32]> = values.iter().zip(&blindings).map(|(v, r)| pc.commit(Scalar::from(*v), *r).compress().to_bytes()).collect();
    let mut rng = RecRng::new(seed);
    let mut t = Transcript::new(&[]);              // src/range/mod.rs:51,67
    let (proof, coms) = RangeProof::prove_multiple_with_rng(&bp, &pc, &mut t, &values, &blindings, n, &mut rng).expect("prove");
    assert_eq!(rng.tape.len(), m * (2 * n + 4), "draw count differs from the slot contract (include/dapol_hip.h)");
    for (a, b) in coms.iter().zip(&commitments) { assert_eq!(&a.to_bytes(), b); }
    let j = serde_json::json!({
        "kind": "range", "n": n, "m": m, "values": values,
        "blindings": blindings.iter().map(|s| hex::encode(s.as_bytes())).collect::<Vec<_>>(),
        "commitments": commitments.iter().map(hex::encode).collect::<Vec<_>>(),
        "tape": rng.tape.iter().map(hex::encode).collect::<Vec<_>>(),          // the draws, in the order the crate took them
        "proof": hex::encode(proof.to_bytes()),
    });
    fs::write(out.join(format!("range_{}_{}.json", n, m)), serde_json::to_string_pretty(&j).unwrap()).unwrap();
}

// ---------------------------------------------------------------------------------------------------------------------
// A DapolNode whose padding RECORDS its draw with its position (the arguments of Paddable::padding -- ignored by the reference,
// src/dapol/node.rs:86-88 -- say where the node stands): the library's tree tape is these draws sorted by (level above the
// leaves, index in the level), whatever order smtree created the nodes in.
#[derive(Clone, Default, Debug)]
struct TapedNode(DapolNode<blake3::Hasher>);
thread_local! {
    static TREE_HEIGHT: std::cell::Cell<usize> = std::cell::Cell::new(0);
    static PAD_RNG: std::cell::RefCell<RecRng> = std::cell::RefCell::new(RecRng::new(0));
    static PAD_DRAWS: std::cell::RefCell<Vec<(usize, u64, [u8; 64])>> = std::cell::RefCell::new(Vec::new());
}
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
    fn padding(idx: &TreeIndex, _secret: &Secret) -> TapedNode {
        let level = TREE_HEIGHT.with(|h| h.get()) - idx.get_height();
        let d = PAD_RNG.with(|r| r.borrow_mut().draw());
        PAD_DRAWS.with(|p| p.borrow_mut().push((level, index_of(idx), d)));
        TapedNode(DapolNode::new(0, Scalar::from_bytes_mod_order_wide(&d)))
    }
}
impl ProofExtractable for TapedNode {
    type ProofNode = DapolProofNode<blake3::Hasher>;
    fn get_proof_node(&self) -> Self::ProofNode { self.0.get_proof_node() }
}

fn dump_tree(out: &Path, height: usize, leaves: &[(u64, u64)], seed: u64) {
    TREE_HEIGHT.with(|h| h.set(height));
    PAD_RNG.with(|r| *r.borrow_mut() = RecRng::new(seed));
    PAD_DRAWS.with(|p| p.borrow_mut().clear());
    let secret = Secret::from_bytes(&[7u8; 32]).expect("secret");      // (unused by DAPOL's padding; adapt to smtree 0.1.2's constructor if it differs)
    let mut lrng = RecRng::new(seed ^ 0x1EAF);
    let list: Vec<(TreeIndex, TapedNode)> = leaves.iter().map(|(i, v)| {
        let r = Scalar::from_bytes_mod_order_wide(&lrng.draw());
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
    // every padding node smtree made, with its position; a position drawn twice (smtree padding a node again in a later call) keeps
    // its FIRST draw -- the node that is in the tree
    let mut pads: Vec<(usize, u64, [u8; 64])> = Vec::new();
    PAD_DRAWS.with(|p| for d in p.borrow().iter() { if !pads.iter().any(|q| q.0 == d.0 && q.1 == d.1) { pads.push(*d); } });
    pads.sort_by_key(|d| (d.0, d.1));
    let j = serde_json::json!({
        "kind": "tree", "height": height,
        "pad_draws": pads.iter().map(|d| serde_json::json!({"level": d.0, "index": d.1, "draw": hex::encode(d.2)})).collect::<Vec<_>>(),
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
    for (n, m) in [(8usize, 1usize), (8, 2), (16, 4), (64, 1), (64, 2), (64, 32)] {
        dump_range(out, n, m, 7 + (n * 100 + m) as u64);
    }
    dump_tree(out, 4, &[(2, 7), (4, 11), (7, 3), (12, 5)], 41);
    dump_tree(out, 8, &[(1, 10), (2, 20), (77, 30), (200, 40), (201, 50), (255, 60)], 42);
    for (value, bytes) in [(0x0102usize, 2usize), (672, 8), (1, 8)] {
        let j = serde_json::json!({"kind": "usize", "value": value, "bytes": bytes, "hex": hex::encode(usize_to_bytes(value, bytes))});
        fs::write(out.join(format!("usize_{}_{}.json", value, bytes)), j.to_string()).unwrap();
    }
}
