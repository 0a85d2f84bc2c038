#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the Python oracle (oracle/pyref.py).

The reference crate cannot be executed in this image (no cargo/rustc, third-party crates not vendored), so
these are DERIVED golden values: computed by the big-integer restatement after it was pinned against the
external known answers (RFC 9496 A.1/A.3, Merlin + STROBE conformance vectors, BLAKE3 official vectors) and
the reference's own known answers (src/dapol/tests.rs:30-85 index KATs, :24 root value, src/range/mod.rs:18
proof length).  Run:  python3 tests/golden/gen_golden.py [--full]     (--full adds the n=64,m=32 proof, ~1 min)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as R  # noqa: E402

SEED = bytes(range(32))


def hx(b):
    return bytes(b).hex()


def wide(dom, a, b):
    return R.seed_wide(SEED, dom, a, b)


def gen_kat():
    liab = [(b"a", b"w", 3), (b"b", b"x", 5), (b"c", b"y", 7), (b"d", b"z", 11)]
    out = {"liabilities": [[i.decode(), e.decode(), v] for i, e, v in liab], "audit_seed": "test", "height": 4}
    for dg in ("blake2s", "blake3"):
        leaves, idm = R.build_leaf_nodes(liab, b"test", 4, dg)
        byidx = dict(leaves)
        ent = {"index": {k.decode(): v for k, v in idm.items()}, "leaves": []}
        for iid, _, _ in liab:
            nd = byidx[idm[iid]]
            ent["leaves"].append({"id": iid.decode(), "idx": idm[iid], "v": nd.v, "r": hx(nd.r.to_bytes(32, "little")),
                                  "C": hx(nd.C), "H": hx(nd.H)})
        a, b = byidx[idm[b"a"]], byidx[idm[b"b"]]
        mg = R.node_merge(a, b, dg)
        ent["merge_ab"] = {"v": mg.v, "r": hx(R.scalar_bytes(mg.r)), "C": hx(mg.C), "H": hx(mg.H)}
        tree, _ = R.dapol_new(liab, b"test", 4, SEED, dg)
        ent["root"] = {"v": tree.root.v, "r": hx(R.scalar_bytes(tree.root.r)), "C": hx(tree.root.C), "H": hx(tree.root.H)}
        out[dg] = ent
    out["B"] = hx(R.B_COMPRESSED)
    out["B_blinding"] = hx(R.B_BLINDING.compress())
    G, H = R.bp_gens(8, 2)
    out["gens_n8_m2"] = {"G": [hx(p.compress()) for p in G], "H": [hx(p.compress()) for p in H]}
    out["seed_wide"] = [{"dom": d, "a": a, "b": b, "out": hx(R.seed_wide(SEED, d, a, b))}
                        for d, a, b in [(1, 0, 0), (1, 3, 77), (2, 2**40 + 5, 12345), (2, 2**64 - 1, 2**64 - 1)]]
    return out


def gen_commit():
    cases = []
    vs = [0, 1, 2, 2**32 - 1, 2**63, 2**64 - 1, 3, 5, 7, 11]
    rs = [0, 1, R.L - 1, R.L, R.L + 1, 2**255 - 1, 2**252, 8 * R.L + 3 if 8 * R.L + 3 < 2**255 else 5]
    k = 0
    for v in vs:
        for r in rs[:4] if v > 3 else rs:
            cases.append((v, r))
    for i in range(40):
        w = wide(9, 1, i)
        v = int.from_bytes(w[:8], "little") if i % 2 else int.from_bytes(w[:4], "little")
        r = int.from_bytes(w[32:], "little") & (2**255 - 1) if i % 3 == 0 else R.scalar_from_wide(wide(9, 2, i))
        cases.append((v, r))
    out = []
    for v, r in cases:
        nd = R.node_new(v, r, "blake3")
        out.append({"v": v, "r": hx(r.to_bytes(32, "little")), "C": hx(nd.C), "H": hx(nd.H)})
    return out


def tree_to_json(tree):
    lv = []
    for k, level in enumerate(tree.levels):
        lv.append([{"idx": i, "v": nd.v, "r": hx(R.scalar_bytes(nd.r) if (k > 0 or i in tree.pad[k]) else nd.r.to_bytes(32, "little")),
                    "C": hx(nd.C), "H": hx(nd.H), "pad": i in tree.pad[k]} for i, nd in sorted(level.items())])
    return lv


def gen_trees():
    out = []
    specs = [(4, [2, 4, 7, 12]), (6, [0, 1, 5, 40, 63]), (5, [17]), (8, [3 * i + (i * i) % 5 for i in range(20)]),
             (10, sorted(set((i * 2654435761) % 1024 for i in range(100)))), (3, list(range(8))), (16, [i * 4096 for i in range(16)]),
             (64, [0, 1, 2**63, 2**64 - 1, 0x0123456789ABCDEF])]
    for t, (h, idxs) in enumerate(specs):
        idxs = sorted(set(idxs))
        leaves = []
        for j, i in enumerate(idxs):
            v = int.from_bytes(wide(10, t, j)[:4], "little")
            r = R.scalar_from_wide(wide(11, t, j))
            leaves.append((i, v, r))
        tree = R.Tree(h, [(i, R.node_new(v, r)) for i, v, r in leaves], SEED)
        ent = {"height": h, "pad_seed": hx(SEED), "leaves": [{"idx": i, "v": v, "r": hx(R.scalar_bytes(r))} for i, v, r in leaves],
               "node_count": tree.node_count(),
               "root": {"v": tree.root.v, "r": hx(R.scalar_bytes(tree.root.r)), "C": hx(tree.root.C), "H": hx(tree.root.H)}}
        if h <= 10:
            ent["levels"] = tree_to_json(tree)
        ent["paths"] = {str(i): [{"v": s.v, "r": hx(R.scalar_bytes(s.r)), "C": hx(s.C), "H": hx(s.H)} for s in tree.path_siblings(i)]
                        for i in idxs[:3]}
        out.append(ent)
    return out


def gen_range(full):
    out = []
    specs = [(8, 1, 1), (8, 2, 2), (16, 4, 3), (32, 2, 4), (64, 1, 5), (64, 2, 6), (64, 4, 7), (8, 8, 8)]
    if full:
        specs.append((64, 32, 9))
        specs.append((8, 64, 10))            # 64 parties (trees higher than 32): three groups of the statement-bound nonce key
    for n, m, sid in specs:
        values = [int.from_bytes(wide(12, sid, j)[:8], "little") & (2**n - 1) for j in range(m)]
        if m >= 2:
            values[0] = 0
            values[1] = 2**n - 1
        bl = [R.scalar_from_wide(wide(13, sid, j)) for j in range(m)]
        if m >= 4:
            bl[2] = 1            # the (0, Scalar::one()) padding party of src/range/padding.rs:100-103
            values[2] = 0
        proof = R.range_prove(values, bl, n, R.Tape(seed=SEED, stream_id=sid))
        Vs = [R.pedersen_commit(v, b).compress() for v, b in zip(values, bl)]
        assert len(proof) == R.range_proof_size(n, m)
        assert R.range_verify(proof, Vs, n)
        out.append({"n": n, "m": m, "stream_id": sid, "nonce_seed": hx(SEED), "values": values,
                    "blindings": [hx(R.scalar_bytes(b)) for b in bl], "commitments": [hx(V) for V in Vs], "proof": hx(proof)})
        print("range", n, m, "ok", flush=True)
    return out


def gen_dapol():
    """End to end: tree -> inclusion path -> policy proof, small height so that pure Python finishes quickly."""
    out = []
    for h, idxs, policy, agg, leaf in [(4, [2, 4, 7, 12], "padding", 4, 7), (4, [2, 4, 7, 12], "splitting", 3, 12),
                                       (5, [1, 9, 30], "padding", 2, 9), (6, [5, 6, 50], "splitting", 6, 50)]:
        leaves = [(i, R.node_new(wide(14, h, j)[0] % 64, R.scalar_from_wide(wide(15, h, j)))) for j, i in enumerate(idxs)]
        tree = R.Tree(h, leaves, SEED)
        sibs, aggregated, individual = R.dapol_prove(tree, leaf, policy, agg, SEED, n=8)
        assert R.policy_verify(policy, aggregated, individual, [s.C for s in sibs], n=8)
        lf = tree.levels[0][leaf]
        assert R.verify_path(tree.root.C, tree.root.H, lf.C, lf.H, leaf, [(s.C, s.H) for s in sibs])
        out.append({"height": h, "n_bits": 8, "policy": policy, "agg": agg, "leaf": leaf, "pad_seed": hx(SEED), "nonce_seed": hx(SEED),
                    "leaves": [{"idx": i, "v": nd.v, "r": hx(R.scalar_bytes(nd.r))} for i, nd in leaves],
                    "root": {"C": hx(tree.root.C), "H": hx(tree.root.H), "v": tree.root.v},
                    "siblings": [{"v": s.v, "r": hx(R.scalar_bytes(s.r)), "C": hx(s.C), "H": hx(s.H)} for s in sibs],
                    "aggregated": [hx(p) for p in aggregated], "individual": [hx(p) for p in individual],
                    "serialized": hx(R.policy_serialize(policy, aggregated, individual))})
        print("dapol", h, policy, agg, "ok", flush=True)
    return out


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")
    print("wrote", name, flush=True)


if __name__ == "__main__":
    full = "--full" in sys.argv
    dump("kat.json", gen_kat())
    dump("commit.json", gen_commit())
    dump("trees.json", gen_trees())
    dump("dapol.json", gen_dapol())
    dump("range.json", gen_range(full))
