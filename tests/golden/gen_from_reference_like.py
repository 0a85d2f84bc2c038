#!/usr/bin/env python3
"""Writes files in the schema of tools/replay_tape.rs's output (tests/golden/from_reference/*.json) -- but made by the PYTHON ORACLE, and
marked so ("made_by": "oracle/pyref.py").  They pin nothing about the Rust crates; they exist so that the loader and the comparisons of
tests/test_from_reference.py run on every test pass (on the CPU against the oracle itself: the plumbing; on the GPU: the library's tape
entry points against the oracle, through the very code path the crate's vectors will take).
  python3 tests/golden/gen_from_reference_like.py <out dir> [--small]"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as R  # noqa: E402


class SplitMix:
    """The byte stream tools/replay_tape.rs hands the crates (any deterministic stream will do: the tape carries the bytes)."""

    def __init__(self, seed):
        self.s = seed & (2**64 - 1)

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    def draw(self):
        return b"".join(self.next().to_bytes(8, "little") for _ in range(8))


def make_range(out, n, m, seed):
    values = [((j * 2654435761 + 12345) & ((1 << n) - 1)) for j in range(m)]
    brng = SplitMix(seed ^ 0xB11D)
    blindings = [R.scalar_from_wide(brng.draw()) for _ in range(m)]
    rng = SplitMix(seed)
    draws = [rng.draw() for _ in range(m * (2 * n + 4))]
    tape = R.Tape(draws=draws)
    proof = R.range_prove(values, blindings, n, tape)
    assert tape.pos == len(draws)
    coms = [R.node_new(v, r).C for v, r in zip(values, blindings)]
    j = {"kind": "range", "made_by": "oracle/pyref.py", "n": n, "m": m, "values": values, "blindings": [R.scalar_bytes(b).hex() for b in blindings],
         "commitments": [c.hex() for c in coms], "tape": [d.hex() for d in draws], "proof": proof.hex()}
    json.dump(j, open(os.path.join(out, "range_%d_%d.json" % (n, m)), "w"))


def make_tree(out, height, leaves, seed):
    lrng = SplitMix(seed ^ 0x1EAF)
    nodes = [(i, R.node_new(v, R.scalar_from_wide(lrng.draw()))) for i, v in leaves]
    # padding positions: the structure depends on the indexes alone, so a first build (any draws) names them
    probe = R.Tree(height, list(nodes), bytes(32))
    pos = sorted((k, i) for k in range(height) for i in probe.pad[k])
    prng = SplitMix(seed)
    draws = {p: prng.draw() for p in pos}
    tree = R.Tree(height, list(nodes), draws)
    paths = []
    for i, _ in nodes:
        sibs = tree.path_siblings(i)
        paths.append({"leaf": i, "siblings": [{"C": s.C.hex(), "H": s.H.hex()} for s in sibs],
                      "merkle_wire": R.merkle_proof_serialize(height, [i], [(s.C, s.H) for s in sibs]).hex()})
    batch = [i for i, _ in nodes[:3]]
    bpos = R.batch_siblings(height, batch)
    bs = [tree.levels[lv][ix] for lv, ix in bpos]
    j = {"kind": "tree", "made_by": "oracle/pyref.py", "height": height,
         "pad_draws": [{"level": k, "index": i, "draw": draws[(k, i)].hex()} for k, i in pos],
         "leaves": [{"idx": i, "v": nd.v, "r": R.scalar_bytes(nd.r).hex()} for i, nd in nodes],
         "root_C": tree.root.C.hex(), "root_H": tree.root.H.hex(), "root_v": tree.root.v, "paths": paths,
         "batch": {"leaves": batch, "sibling_C": [s.C.hex() for s in bs], "merkle_wire": R.merkle_proof_serialize(height, batch, [(s.C, s.H) for s in bs]).hex()}}
    json.dump(j, open(os.path.join(out, "tree_%d.json" % height), "w"))


def main(out, small=False):
    os.makedirs(out, exist_ok=True)
    for n, m in ([(8, 1), (8, 2)] if small else [(8, 1), (8, 2), (16, 4), (64, 1)]):
        make_range(out, n, m, 7 + n * 100 + m)
    make_tree(out, 4, [(2, 7), (4, 11), (7, 3), (12, 5)], 41)
    if not small:
        make_tree(out, 8, [(1, 10), (2, 20), (77, 30), (200, 40), (201, 50), (255, 60)], 42)
    for value, nbytes in ((0x0102, 2), (672, 8), (1, 8)):
        json.dump({"kind": "usize", "made_by": "oracle/pyref.py", "value": value, "bytes": nbytes, "hex": value.to_bytes(nbytes, "big").hex()},
                  open(os.path.join(out, "usize_%d_%d.json" % (value, nbytes)), "w"))


if __name__ == "__main__":
    main(sys.argv[1], "--small" in sys.argv)
