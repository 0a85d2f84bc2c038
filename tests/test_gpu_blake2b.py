"""64-byte node digests: Dapol<blake2::Blake2b, R> on the new_blank + build path, the half of the reference's own test matrix that
a 32-byte-only library cannot run (src/tests.rs:100-106; Dapol::new rejects such a digest, src/dapol/mod.rs:101-103, new_blank and
build do not, :196-208).  Every call goes through the C ABI on a DAPOL_DIGEST_BLAKE2B context; hashlib.blake2b and oracle/pyref.py
(dg = "blake2b") check."""
import hashlib

import numpy as np
import pytest

from test_gpu_parity import SEED, _rand_leaves

pytestmark = pytest.mark.gpu
H2B = lambda *parts: hashlib.blake2b(b"".join(parts)).digest()          # blake2::Blake2b = Blake2b-512, no key


def _check_hash_chain(tree, height):
    """Every stored node of every level against hashlib: leaf / padding node = D(C) (node.rs:34-36, 86-88), parent =
    D(C_L || C_R || H_L || H_R) (node.rs:66-77).  -> {level: {index: (C, H)}}"""
    levels, real = {}, {}
    for k in range(height + 1):
        idx, v, r, C, H, pad = tree.level_nodes(k)
        assert H.shape[1] == 64
        levels[k] = {int(i): (c.tobytes(), h.tobytes()) for i, c, h in zip(idx, C, H)}
        real[k] = [int(i) for i, p in zip(idx, pad) if not p]
        for i, c, h, p in zip(idx, C, H, pad):
            if k == 0 or p:
                assert h.tobytes() == H2B(c.tobytes()), (k, int(i))
    for k in range(height):
        for i in real[k + 1]:                                            # (a padding node has no children)
            (cl, hl), (cr, hr) = levels[k][2 * i], levels[k][2 * i + 1]
            assert levels[k + 1][i][1] == H2B(cl, cr, hl, hr), (k + 1, i)
    return levels


def test_blake2b_tree_paths_proofs_and_wire_vs_pyref(hip_lib, pyref):
    """A small tree against the Python restatement with dg = blake2b: every node, a single-leaf proof (path hashes, range-proof
    bytes, wire bytes, both verifiers) and a batched one."""
    ctx = hip_lib.Context(0, 8, digest=hip_lib.DIGEST_BLAKE2B)
    assert ctx.hb == 64
    height, n_bits = 5, 8
    rng = np.random.default_rng(64)
    idx, v, r = _rand_leaves(rng, height, 6, vmax=20)
    tree = hip_lib.Tree(ctx, height, idx, v, r, SEED)
    leaves = [(int(i), pyref.node_new(int(vv), int.from_bytes(rr.tobytes(), "little"), "blake2b")) for i, vv, rr in zip(idx, v, r)]
    pt = pyref.Tree(height, leaves, SEED, "blake2b")
    C, H, rv, rr = tree.root()
    assert (C, H, rv) == (pt.root.C, pt.root.H, pt.root.v) and len(H) == 64
    for level in range(height + 1):
        li, lv, lr, lC, lH, pad = tree.level_nodes(level)
        assert {int(i): (c.tobytes(), h.tobytes()) for i, c, h in zip(li, lC, lH)} == {i: (n.C, n.H) for i, n in pt.levels[level].items()}
    lC, lH = ctx.commit_hash_batch(v, r)
    assert [h.tobytes() for h in lH] == [pt.levels[0][int(i)].H for i in idx]
    sC, sH, sv, sr = tree.paths(idx)
    pC, pH, proofs = tree.prove_entities(idx, hip_lib.POLICY_PADDING, 3, n_bits, SEED)
    assert pH.shape == (6, height, 64) and pH.tobytes() == sH.tobytes()
    for k, li in enumerate(idx):
        sibs, aggregated, individual = pyref.dapol_prove(pt, int(li), "padding", 3, SEED, n=n_bits)
        assert [pH[k, s].tobytes() for s in range(height)] == [x.H for x in sibs]
        assert [pC[k, s].tobytes() for s in range(height)] == [x.C for x in sibs]
        assert proofs[k].tobytes() == b"".join(aggregated) + b"".join(individual)
        wire = hip_lib.proof_serialize(height, [li], pC[k], pH[k], hip_lib.POLICY_PADDING, 3, n_bits, proofs[k].tobytes(), hash_bytes=64)
        assert wire == pyref.dapol_proof_serialize("padding", aggregated, individual, height, [int(li)], [(x.C, x.H) for x in sibs])
        assert pyref.verify_path(pt.root.C, pt.root.H, pt.levels[0][int(li)].C, pt.levels[0][int(li)].H, int(li), [(x.C, x.H) for x in sibs], "blake2b")
    assert ctx.verify_entities(height, idx, lC, lH, pC, pH, C, H, hip_lib.POLICY_PADDING, 3, n_bits, proofs, verify_seed=SEED).all()
    batch = [int(idx[1]), int(idx[4])]
    level, index, bC, bH, blob = tree.prove_batch(batch, hip_lib.POLICY_SPLITTING, 2, n_bits, SEED)
    _, sibs, aggregated, individual = pyref.dapol_prove_batch(pt, batch, "splitting", 2, SEED, n=n_bits)
    assert [h.tobytes() for h in bH] == [x.H for x in sibs] and blob == b"".join(aggregated) + b"".join(individual)
    assert ctx.verify_batch(height, batch, lC[[1, 4]], lH[[1, 4]], bC, bH, C, H, hip_lib.POLICY_SPLITTING, 2, n_bits, blob, verify_seed=SEED)
    # Mergeable::merge and Paddable::padding on records
    mC, mH = ctx.merge_batch(lC[:1], lH[:1], lC[1:2], lH[1:2])
    assert mH[0].tobytes() == H2B(lC[0].tobytes(), lC[1].tobytes(), lH[0].tobytes(), lH[1].tobytes())
    pdC, pdH, pdr = ctx.padding_nodes(SEED, [2], [5])
    assert pdH[0].tobytes() == H2B(pdC[0].tobytes()) and pdC[0].tobytes() == pyref.node_padding(SEED, 2, 5, "blake2b").C
    # a 32-byte context disagrees about every hash, hence about the proof
    ctx3 = hip_lib.Context(0, 8)
    assert not ctx3.verify_batch(height, batch, lC[[1, 4]], lH[[1, 4], :32], bC, bH[:, :32], C, H[:32], hip_lib.POLICY_SPLITTING, 2, n_bits, blob, verify_seed=SEED)


@pytest.mark.parametrize("policy", ["padding", "splitting"])
def test_blake2b_like_the_reference_integration_test(hip_lib, policy):
    """TesterDapol::<blake2::Blake2b, R>::test (src/tests.rs:25-97, :104-105): height 10, 100 random leaves, aggregation factors
    1 ..= 10, 64-bit proofs; build-built against update-built (the reference compares the roots' VALUES, :48 -- here the whole root,
    hashes included, because padding nodes are positional); batches of 10 leaves and every single leaf: generate_proof(_batch) ->
    serialize -> deserialize -> verify(_batch), on both trees."""
    pol = hip_lib.POLICY_PADDING if policy == "padding" else hip_lib.POLICY_SPLITTING
    ctx = hip_lib.Context(0, 16, digest=hip_lib.DIGEST_BLAKE2B)
    ctx3 = hip_lib.Context(0, 16)                                        # the same liabilities under BLAKE3: everything but the hashes must agree
    height, n, n_bits = 10, 100, 64
    rng = np.random.default_rng(2024 + pol)
    idx, v, r = _rand_leaves(rng, height, n)                             # smtree's generate_sorted_index_value_pairs: random u32 values (node.rs:101-105)
    lC, lH = ctx.commit_hash_batch(v, r)
    build = hip_lib.Tree(ctx, height, idx, v, r, SEED)                   # new_blank + build
    update = hip_lib.Tree(ctx, height, idx[:1], v[:1], r[:1], SEED)     # new_blank, then one `update` per leaf (:41-44), in a shuffled order
    order = rng.permutation(np.arange(1, n))
    for k in order:
        update.update(idx[k:k + 1], v[k:k + 1], r[k:k + 1])
    rC, rH, rv, rr = build.root()
    assert update.root() == (rC, rH, rv, rr) and rv == int(v.sum()) and len(rH) == 64
    levels = _check_hash_chain(build, height)
    assert _check_hash_chain(update, height) == levels
    t3 = hip_lib.Tree(ctx3, height, idx, v, r, SEED)
    assert t3.root()[0] == rC and t3.root()[1] != rH[:32]
    for k in range(height + 1):                                          # commitments, values, blindings are the digest's business nowhere
        a, b = build.level_nodes(k), t3.level_nodes(k)
        assert all(np.array_equal(a[j], b[j]) for j in (0, 1, 2, 3, 5))
    for agg in range(1, height + 1):
        # batches of 10 (:50-69): generate_proof_batch -> serialize -> deserialize -> verify_batch
        for b0 in range(0, n, 10 if agg in (1, 10) else 50):
            batch = idx[b0:b0 + 10]
            level, index, sC, sH, blob = build.prove_batch(batch, pol, agg, n_bits, SEED)
            assert sH.shape == (len(level), 64)
            assert [h.tobytes() for h in sH] == [levels[int(l)][int(i)][1] for l, i in zip(level, index)]
            wire = hip_lib.proof_serialize(height, batch, sC, sH, pol, agg, n_bits, blob, hash_bytes=64)
            assert len(wire) == hip_lib.lib().dapol_proof_wire_size_d(64, height, 10, len(level), pol, agg, n_bits)
            d = ctx.proof_deserialize(pol, n_bits, wire)
            assert d["consumed"] == len(wire) and d["aggregation_factor"] == agg and d["sib_H"].tobytes() == sH.tobytes()
            assert ctx.verify_batch(d["height"], d["leaf_idx"], lC[b0:b0 + 10], lH[b0:b0 + 10], d["sib_C"], d["sib_H"], rC, rH, pol, agg, n_bits, d["range_blob"])
            if b0 == 0:
                bad = d["sib_H"].copy()
                bad[len(bad) // 2, 40] ^= 1                              # a bit in the UPPER half of a 64-byte hash
                assert not ctx.verify_batch(d["height"], d["leaf_idx"], lC[:10], lH[:10], d["sib_C"], bad, rC, rH, pol, agg, n_bits, d["range_blob"])
        # every single leaf (:71-93), on the build-built and on the update-built tree
        pC, pH, proofs = build.prove_entities(idx, pol, agg, n_bits, SEED)
        uC, uH, uproofs = update.prove_entities(idx, pol, agg, n_bits, SEED)
        assert pH.shape == (n, height, 64) and uH.tobytes() == pH.tobytes() and uproofs.tobytes() == proofs.tobytes()
        c3, h3, p3 = t3.prove_entities(idx, pol, agg, n_bits, SEED)
        assert p3.tobytes() == proofs.tobytes() and c3.tobytes() == pC.tobytes()      # range proofs do not see the digest
        dC, dH, dR = [], [], []
        for e in range(n):
            wire = hip_lib.proof_serialize(height, idx[e:e + 1], pC[e], pH[e], pol, agg, n_bits, proofs[e].tobytes(), hash_bytes=64)
            d = ctx.proof_deserialize(pol, n_bits, wire)
            assert (d["height"], d["aggregation_factor"], d["leaf_idx"].tolist(), d["consumed"]) == (height, agg, [int(idx[e])], len(wire))
            dC.append(d["sib_C"]); dH.append(d["sib_H"]); dR.append(np.frombuffer(d["range_blob"], np.uint8))
        dC, dH, dR = np.stack(dC), np.stack(dH), np.stack(dR)
        assert ctx.verify_entities(height, idx, lC, lH, dC, dH, rC, rH, pol, agg, n_bits, dR).all()
        if agg == 1:
            with pytest.raises(hip_lib.DapolError) as e:
                ctx.proof_deserialize(pol, n_bits, wire[:-1])            # a 64-byte hash cut short
            assert e.value.code == 6
            bad = dH.copy()
            bad[7, 3, 63] ^= 0x80                                        # the last byte of one 64-byte sibling hash of entity 7
            ok = ctx.verify_entities(height, idx, lC, lH, dC, bad, rC, rH, pol, agg, n_bits, dR)
            assert ok[7] == 0 and ok.sum() == n - 1
            badl = lH.copy()
            badl[9, 33] ^= 1
            ok = ctx.verify_entities(height, idx, lC, badl, dC, dH, rC, rH, pol, agg, n_bits, dR)
            assert ok[9] == 0 and ok.sum() == n - 1
            badr = bytearray(rH)
            badr[50] ^= 4
            assert not ctx.verify_entities(height, idx, lC, lH, dC, dH, rC, bytes(badr), pol, agg, n_bits, dR).any()


def test_blake2b_is_refused_where_the_reference_refuses_it(hip_lib):
    """Dapol::new checks D::output_size() == DIGEST_SIZE (src/dapol/mod.rs:101-103): InvalidDigestSize.  The paths that exist for the
    benchmark and for multi-GPU sharding carry 32-byte hashes in their records and say so too."""
    ctx = hip_lib.Context(0, 8, digest=hip_lib.DIGEST_BLAKE2B)
    rng = np.random.default_rng(5)
    idx, v, r = _rand_leaves(rng, 8, 12)
    for make in (lambda: hip_lib.Tree(ctx, 8, idx, v, r, SEED, enforce_sparsity=True),                      # Dapol::new
                 lambda: ctx.build_leaf_nodes([(b"a", b"w", 3)], b"test", 4, hip_lib.DIGEST_BLAKE2B),        # ... and its leaf derivation
                 lambda: hip_lib.Workload(ctx, 8, idx, v, r),
                 lambda: hip_lib.Tree(ctx, 8, idx[idx < 128], v[idx < 128], r[idx < 128], SEED, shard_bits=1)):
        with pytest.raises(hip_lib.DapolError) as e:
            make()
        assert e.value.code == 3
    with pytest.raises(hip_lib.DapolError) as e:
        hip_lib.Context(0, 8, digest=3)
    assert e.value.code == 3
    tree = hip_lib.Tree(ctx, 8, idx, v, r, SEED)                         # new_blank + build: fine
    assert len(tree.root()[1]) == 64


def test_blake2b_large_tree_levelwise_build(hip_lib, pyref):
    """20,000 leaves at height 40 (indexes beyond 2^32; above 8,192 leaves the tree is built level by level, below by phases): the
    64-byte chain over the level-wise build.  The root's hash recomputed with hashlib from the two nodes below it; for sampled leaves the
    path hashes re-merged with hashlib from the leaf up (commitments from the library: the digest does not touch them) reach the root;
    the sampled inclusion proofs verify; and the tree agrees with the BLAKE3 tree of the same leaves in everything but its hashes."""
    ctx = hip_lib.Context(0, 64, digest=hip_lib.DIGEST_BLAKE2B)
    ctx3 = hip_lib.Context(0, 64)
    height, n = 40, 20000
    rng = np.random.default_rng(40)
    idx, v, r = _rand_leaves(rng, height, n)
    n = len(idx)
    tree, t3 = hip_lib.Tree(ctx, height, idx, v, r, SEED), hip_lib.Tree(ctx3, height, idx, v, r, SEED)
    rC, rH, rv, rr = tree.root()
    assert (rC, rv, rr) == (t3.root()[0], t3.root()[2], t3.root()[3]) and len(rH) == 64 and tree.node_count() == t3.node_count()
    li, lv, lr, lC, lH, pad = tree.level_nodes(height - 1)
    top = {int(i): (c.tobytes(), h.tobytes()) for i, c, h in zip(li, lC, lH)}
    assert rH == H2B(top[0][0], top[1][0], top[0][1], top[1][1])
    who = idx[:: n // 6][:6]
    pos = np.searchsorted(idx, who)
    lC, lH = ctx.commit_hash_batch(v[pos], r[pos])
    pC, pH, proofs = tree.prove_entities(who, hip_lib.POLICY_SPLITTING, height, 64, SEED)
    c3, h3, p3 = t3.prove_entities(who, hip_lib.POLICY_SPLITTING, height, 64, SEED)
    assert pC.tobytes() == c3.tobytes() and proofs.tobytes() == p3.tobytes() and pH.shape == (6, height, 64)
    # the ancestors' commitments are the BLAKE3 tree's (same points): take them from its Merkle re-merge, i.e. from pyref on the path
    for e, leaf in enumerate(who):
        C, Hh = pyref.decompress(lC[e].tobytes()), lH[e].tobytes()
        assert Hh == H2B(lC[e].tobytes())
        Cb = lC[e].tobytes()
        for k in range(height):                                          # siblings are root side first: level k's sits at height - 1 - k
            sC, sH = pC[e, height - 1 - k].tobytes(), pH[e, height - 1 - k].tobytes()
            Hh = H2B(sC, Cb, sH, Hh) if (int(leaf) >> k) & 1 else H2B(Cb, sC, Hh, sH)
            C = C + pyref.decompress(sC)
            Cb = C.compress()
        assert (Cb, Hh) == (rC, rH), e
    assert ctx.verify_entities(height, who, lC, lH, pC, pH, rC, rH, hip_lib.POLICY_SPLITTING, height, 64, proofs).all()
    bad = pH.copy()
    bad[2, 17, 60] ^= 2
    ok = ctx.verify_entities(height, who, lC, lH, pC, bad, rC, rH, hip_lib.POLICY_SPLITTING, height, 64, proofs)
    assert ok.tolist() == [1, 1, 0, 1, 1, 1]
