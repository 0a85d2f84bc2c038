"""Vectors dumped from the REAL reference crate by tools/replay_tape.rs (tests/golden/from_reference/*.json): the crates run on a
recorded byte stream, every Scalar::random draw is kept (the TAPE of the randomness contract), and the draws travel with what the
crates made of them.  Here the tapes are replayed through the oracle (CPU) and through libdapol_hip.so's tape entry points (GPU) and the
bytes compared.  The build image cannot produce the vectors (no Rust toolchain): while the directory holds none, the tests that need
them are skipped -- but the SAME loader and comparisons run on every pass over files of the same schema made by the Python oracle
(tests/golden/gen_from_reference_like.py; they pin the plumbing and, on the GPU, the tape entry points; nothing about the crates)."""
import glob
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC_DIR = os.path.join(HERE, "golden", "from_reference")
sys.path.insert(0, os.path.join(HERE, "golden"))


def _vectors(kind, vec_dir=VEC_DIR):
    out = []
    for f in sorted(glob.glob(os.path.join(vec_dir, "*.json"))):
        j = json.load(open(f))
        if j.get("kind") == kind:
            out.append(j)
    if not out:
        pytest.skip("no %s vectors from the reference crate yet (tools/replay_tape.rs makes them)" % kind)
    return out


def _ints(hexes):
    return [int.from_bytes(bytes.fromhex(h), "little") for h in hexes]


def _pad_tape(c):
    """tree vector -> ({(level, index): draw}, the draws in tape order = (level bottom-up, index ascending))"""
    pads = sorted(c["pad_draws"], key=lambda d: (d["level"], d["index"]))
    return {(d["level"], d["index"]): bytes.fromhex(d["draw"]) for d in pads}, b"".join(bytes.fromhex(d["draw"]) for d in pads), pads


@pytest.fixture(scope="module")
def oracle_made(tmp_path_factory):
    import gen_from_reference_like
    d = str(tmp_path_factory.mktemp("from_reference_like"))
    gen_from_reference_like.main(d, small=True)
    return d


# ----------------------------------------------------------------------------------------------------------- oracle (CPU)
def check_oracle_range(pyref, vec_dir):
    for c in _vectors("range", vec_dir):
        tape = pyref.Tape(draws=[bytes.fromhex(h) for h in c["tape"]])
        assert pyref.range_prove(c["values"], _ints(c["blindings"]), c["n"], tape).hex() == c["proof"]
        assert tape.pos == len(c["tape"]) == c["m"] * (2 * c["n"] + 4)                                       # the slot contract


def check_oracle_tree(pyref, vec_dir):
    for c in _vectors("tree", vec_dir):
        draws, _, _ = _pad_tape(c)
        leaves = [(l["idx"], pyref.node_new(l["v"], int.from_bytes(bytes.fromhex(l["r"]), "little"))) for l in c["leaves"]]
        tree = pyref.Tree(c["height"], leaves, draws)
        assert sorted((k, i) for k in range(c["height"]) for i in tree.pad[k]) == sorted(draws)              # padding POSITIONS
        root = tree.root
        assert (root.C.hex(), root.H.hex(), root.v) == (c["root_C"], c["root_H"], c["root_v"])
        for p in c["paths"]:
            sibs = tree.path_siblings(p["leaf"])
            assert [(s.C.hex(), s.H.hex()) for s in sibs] == [(s["C"], s["H"]) for s in p["siblings"]]       # sibling ORDER
            wire = pyref.merkle_proof_serialize(c["height"], [p["leaf"]], [(s.C, s.H) for s in sibs])
            assert wire.hex() == p["merkle_wire"]                                                             # MerkleProof framing
        b = c["batch"]
        pos = pyref.batch_siblings(c["height"], b["leaves"])
        assert [tree.levels[lv][ix].C.hex() for lv, ix in pos] == b["sibling_C"]                              # batched sibling order


def test_oracle_range_proofs_equal_the_crates(pyref):
    check_oracle_range(pyref, VEC_DIR)


def test_oracle_tree_equals_smtree(pyref):
    check_oracle_tree(pyref, VEC_DIR)


def test_usize_to_bytes_byte_order():
    for c in _vectors("usize"):
        assert c["value"].to_bytes(c["bytes"], "big").hex() == c["hex"], "smtree::utils::usize_to_bytes is not big-endian: set int_big_endian = 0"


def test_loader_runs_on_oracle_made_files(pyref, oracle_made):
    """The loader and the oracle-side comparisons on files of the crate vectors' schema (made by the oracle: the plumbing)."""
    check_oracle_range(pyref, oracle_made)
    check_oracle_tree(pyref, oracle_made)
    assert len(_vectors("usize", oracle_made)) == 3


# ------------------------------------------------------------------------------------------------------------------- GPU
def check_gpu_range(hip_lib, vec_dir):
    for c in _vectors("range", vec_dir):
        ctx = hip_lib.Context(0, c["m"])
        v = np.array([c["values"]], np.uint64)
        r = np.frombuffer(b"".join(bytes.fromhex(h) for h in c["blindings"]), np.uint8).reshape(1, c["m"], 32)
        tape = np.frombuffer(b"".join(bytes.fromhex(h) for h in c["tape"]), np.uint8)
        got = ctx.range_prove_batch(c["n"], c["m"], v, r, tape=tape)                                          # dapol_range_prove_batch, tape mode
        assert got[0].tobytes().hex() == c["proof"]
        C, _ = ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
        assert [x.tobytes().hex() for x in C] == c["commitments"]


def check_gpu_tree(gpu_ctx, hip_lib, vec_dir):
    for c in _vectors("tree", vec_dir):
        idx = np.array([l["idx"] for l in c["leaves"]], np.uint64)
        v = np.array([l["v"] for l in c["leaves"]], np.uint64)
        r = np.frombuffer(b"".join(bytes.fromhex(l["r"]) for l in c["leaves"]), np.uint8).reshape(-1, 32)
        _, tape, pads = _pad_tape(c)
        level, index = hip_lib.tree_padding_positions(c["height"], idx)
        assert [(int(l), int(i)) for l, i in zip(level, index)] == [(d["level"], d["index"]) for d in pads]  # padding POSITIONS, tape order
        tree = hip_lib.Tree(gpu_ctx, c["height"], idx, v, r, None, pad_tape=tape)                             # dapol_tree_build_tape
        C, H, rv, _ = tree.root()
        assert (C.hex(), H.hex(), rv) == (c["root_C"], c["root_H"], c["root_v"])
        pC, pH, _, _ = tree.paths(idx)
        for k, p in enumerate(c["paths"]):
            assert [x.tobytes().hex() for x in pC[k]] == [s["C"] for s in p["siblings"]]
            assert [x.tobytes().hex() for x in pH[k]] == [s["H"] for s in p["siblings"]]
        level, index = hip_lib.batch_siblings(c["height"], c["batch"]["leaves"])
        assert len(level) == len(c["batch"]["sibling_C"])


@pytest.mark.gpu
def test_gpu_range_proofs_equal_the_crates(hip_lib):
    check_gpu_range(hip_lib, VEC_DIR)


@pytest.mark.gpu
def test_gpu_tree_equals_smtree(gpu_ctx, hip_lib):
    check_gpu_tree(gpu_ctx, hip_lib, VEC_DIR)


@pytest.mark.gpu
def test_gpu_loader_runs_on_oracle_made_files(gpu_ctx, hip_lib, oracle_made):
    """The GPU-side comparisons on files of the crate vectors' schema: the library's TAPE entry points (dapol_range_prove_batch with a
    tape, dapol_tree_build_tape, dapol_tree_padding_positions) against the oracle, through the code path the crate's vectors will take."""
    check_gpu_range(hip_lib, oracle_made)
    check_gpu_tree(gpu_ctx, hip_lib, oracle_made)
