"""GPU parity tests proper: every call goes through the C ABI (libdapol_hip.so); the oracles only check.
Bit-exact is the bar for everything here (integer / byte work)."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
SEED = bytes(range(32))


def _arr(hexes):
    return np.array([list(bytes.fromhex(h)) for h in hexes], np.uint8)


def _ref_tree(ref, height, idx, v, r, seed=SEED, faithful=0):
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    t = ref.ref_tree_build(height, ctypes.c_size_t(len(idx)), p(idx), p(v), p(r), seed, faithful)
    assert t is not None
    return ctypes.c_void_p(t)


def _ref_root(ref, t):
    C, H, r, v = [ctypes.create_string_buffer(32) for _ in range(3)] + [ctypes.c_uint64()]
    ref.ref_tree_root(t, C, H, ctypes.byref(v), r)
    return C.raw, H.raw, v.value, r.raw


def _rand_leaves(rng, height, n, vmax=2**32):
    if height >= 63:
        idx = np.sort(np.unique(rng.integers(0, 2**63, size=n, dtype=np.uint64)))
    else:
        idx = np.sort(rng.choice(1 << height, size=n, replace=False).astype(np.uint64)) if (1 << height) <= 1 << 22 else \
            np.sort(np.unique(rng.integers(0, 1 << height, size=n, dtype=np.uint64)))
    n = len(idx)
    v = rng.integers(0, vmax, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    return idx, v, r


# ------------------------------------------------------------------------------------------------ golden vectors
def test_generators(gpu_ctx, pyref):
    kat = load_golden("kat.json")
    assert gpu_ctx.generator(0).hex() == kat["B"] and gpu_ctx.generator(1).hex() == kat["B_blinding"]
    for j in range(2):
        for i in range(8):
            assert gpu_ctx.generator(2, j, i).hex() == kat["gens_n8_m2"]["G"][j * 8 + i]
            assert gpu_ctx.generator(3, j, i).hex() == kat["gens_n8_m2"]["H"][j * 8 + i]
    G, H = pyref.bp_gens(64, 2)
    for i in (0, 9, 63):
        assert gpu_ctx.generator(2, 1, i) == G[64 + i].compress() and gpu_ctx.generator(3, 1, i) == H[64 + i].compress()


def test_commit_hash_golden(gpu_ctx):
    cm = load_golden("commit.json")       # includes v = 0 / 2^64-1 and r = 0, 1, l-1, l, l+1, 2^255-1 (unreduced)
    C, H = gpu_ctx.commit_hash_batch([c["v"] for c in cm], _arr([c["r"] for c in cm]))
    for i, c in enumerate(cm):
        assert C[i].tobytes().hex() == c["C"] and H[i].tobytes().hex() == c["H"], i


def test_trees_golden_every_node(gpu_ctx, hip_lib):
    for t in load_golden("trees.json"):
        idx = np.array([l["idx"] for l in t["leaves"]], np.uint64)
        v = np.array([l["v"] for l in t["leaves"]], np.uint64)
        tr = hip_lib.Tree(gpu_ctx, t["height"], idx, v, _arr([l["r"] for l in t["leaves"]]), bytes.fromhex(t["pad_seed"]))
        C, H, rv, rr = tr.root()
        assert (C.hex(), H.hex(), rv, rr.hex()) == (t["root"]["C"], t["root"]["H"], t["root"]["v"], t["root"]["r"])
        assert sum(tr.node_count()) == t["node_count"]
        for k, lev in enumerate(t.get("levels", [])):
            i_, v_, r_, C_, H_, p_ = tr.level_nodes(k)
            got = {int(i_[j]): (int(v_[j]), r_[j].tobytes().hex(), C_[j].tobytes().hex(), H_[j].tobytes().hex(), bool(p_[j])) for j in range(len(i_))}
            assert got == {n["idx"]: (n["v"], n["r"], n["C"], n["H"], n["pad"]) for n in lev}, (t["height"], k)
        pk = sorted(int(k) for k in t["paths"])
        pC, pH, pv, pr = tr.paths(pk)
        for a, li in enumerate(pk):
            for s, e in enumerate(t["paths"][str(li)]):
                assert (pC[a, s].tobytes().hex(), pH[a, s].tobytes().hex(), int(pv[a, s]), pr[a, s].tobytes().hex()) == (e["C"], e["H"], e["v"], e["r"])


def test_range_proofs_golden(gpu_ctx):
    for c in load_golden("range.json"):
        n, m = c["n"], c["m"]
        if m > gpu_ctx.max_parties:
            continue                                     # (the 64-party vector: tests/test_gpu_full_range.py, on a 64-party context)
        pr = gpu_ctx.range_prove_batch(n, m, np.array(c["values"], np.uint64).reshape(1, m), _arr(c["blindings"]).reshape(1, m, 32),
                                       nonce_seed=bytes.fromhex(c["nonce_seed"]), stream_id=[c["stream_id"]])
        assert pr[0].tobytes().hex() == c["proof"], (n, m)


def test_entity_proofs_golden_both_policies(gpu_ctx, hip_lib):
    for c in load_golden("dapol.json"):
        idx = np.array([l["idx"] for l in c["leaves"]], np.uint64)
        v = np.array([l["v"] for l in c["leaves"]], np.uint64)
        tr = hip_lib.Tree(gpu_ctx, c["height"], idx, v, _arr([l["r"] for l in c["leaves"]]), bytes.fromhex(c["pad_seed"]))
        pol = hip_lib.POLICY_PADDING if c["policy"] == "padding" else hip_lib.POLICY_SPLITTING
        pC, pH, out = tr.prove_entities([c["leaf"]], pol, c["agg"], c["n_bits"], bytes.fromhex(c["nonce_seed"]))
        assert out[0].tobytes().hex() == "".join(c["aggregated"]) + "".join(c["individual"])
        for s, e in enumerate(c["siblings"]):
            assert pC[0, s].tobytes().hex() == e["C"] and pH[0, s].tobytes().hex() == e["H"]


# ------------------------------------------------------------------------------------------------ vs the C oracle, seeded
@pytest.mark.parametrize("height,n", [(1, 1), (1, 2), (2, 3), (11, 2048), (12, 700), (24, 4096), (32, 2048), (64, 300)])
def test_tree_vs_oracle_random(gpu_ctx, hip_lib, ref, height, n):
    rng = np.random.default_rng(height * 1000 + n)
    idx, v, r = _rand_leaves(rng, height, n)
    r[::7, 31] |= 0x70                        # unreduced Scalar::from_bits blindings (>= l) on some leaves
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    t = _ref_tree(ref, height, idx, v, r)
    assert tr.root() == _ref_root(ref, t)
    assert sum(tr.node_count()) == ref.ref_tree_node_count(t)
    assert tr.root()[2] == int(v.sum())                          # root value = sum of liabilities (src/dapol/tests.rs:24)
    sample = idx[:: max(1, len(idx) // 16)]
    pC, pH, pv, pr = tr.paths(sample)
    for a, li in enumerate(sample):
        sC, sH, sr, sv = [ctypes.create_string_buffer(32 * height) for _ in range(3)] + [(ctypes.c_uint64 * height)()]
        assert ref.ref_tree_path(t, ctypes.c_uint64(int(li)), sC, sH, sv, sr) == 1
        assert pC[a].tobytes() == sC.raw and pH[a].tobytes() == sH.raw and pr[a].tobytes() == sr.raw and list(map(int, pv[a])) == list(sv)
    ref.ref_tree_free(t)


@pytest.mark.parametrize("n_bits,m,b", [(64, 32, 48), (64, 1, 64), (32, 16, 32), (8, 1, 130), (16, 8, 33),
                                        (64, 4, 20), (8, 32, 40), (64, 16, 10), (32, 32, 12), (16, 16, 21), (8, 16, 5), (32, 8, 9)])
def test_range_prove_vs_oracle_random(gpu_ctx, ref, n_bits, m, b):
    rng = np.random.default_rng(n_bits * 100 + m)
    v = rng.integers(0, 2**n_bits if n_bits < 64 else 2**63, size=(b, m), dtype=np.uint64)
    if n_bits == 64:
        v[0, 0] = 2**64 - 1
    v[1 % b, 0] = 0
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = rng.integers(0, 2**63, size=b, dtype=np.uint64)
    got = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid, slot_base=5)
    ps = ref.ref_range_proof_size(n_bits, m)
    out = ctypes.create_string_buffer(ps * b)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(b), p(v), p(r), SEED, p(sid), ctypes.c_uint64(5), None, 0, out) == 0
    assert got.tobytes() == out.raw
    # every proof verifies against its commitments (round trip), a tampered one does not
    C, _ = gpu_ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    c7 = bytes([7]) + bytes(31)
    for k in (0, b - 1):
        Vs = C[k * m:(k + 1) * m].tobytes()
        assert ref.ref_range_verify(n_bits, m, got[k].tobytes(), ctypes.c_size_t(ps), Vs, c7, 0) == 1
        bad = bytearray(got[k].tobytes())
        bad[ps - 1] ^= 1
        assert ref.ref_range_verify(n_bits, m, bytes(bad), ctypes.c_size_t(ps), Vs, c7, 0) == 0


def test_tape_mode_equals_seed_mode(gpu_ctx, pyref):
    n, m, b = 16, 4, 3
    slots = m * (2 * n + 4)
    sid = [11, 2**40 + 3, 0]
    rng = np.random.default_rng(5)
    v = rng.integers(0, 2**16, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    # seed mode draws from a key bound to the statement (stream, first slot, shape, value commitments): the tape replays it
    C, _ = gpu_ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    keys = [pyref.nonce_key(SEED, s, 0, n, m, [C[i * m + j].tobytes() for j in range(m)]) for i, s in enumerate(sid)]
    tape = np.frombuffer(b"".join(pyref.seed_wide(keys[i], 2, s, k) for i, s in enumerate(sid) for k in range(slots)), np.uint8)
    a = gpu_ctx.range_prove_batch(n, m, v, r, nonce_seed=SEED, stream_id=sid)
    t = gpu_ctx.range_prove_batch(n, m, v, r, tape=tape)
    assert a.tobytes() == t.tobytes()


@pytest.mark.parametrize("height,policy,agg", [(8, 0, 8), (8, 1, 5), (6, 0, 3), (5, 1, 0), (7, 0, 0), (9, 1, 9)])
def test_entity_proofs_vs_python_oracle(gpu_ctx, hip_lib, pyref, height, policy, agg):
    """Padding and splitting policies with aggregation_factor below the height (individual proofs) vs pyref, n = 8 bits."""
    rng = np.random.default_rng(height * 10 + agg)
    idx, v, r = _rand_leaves(rng, height, 5, vmax=8)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    leaves = [(int(i), pyref.node_new(int(vv), int.from_bytes(rr.tobytes(), "little"))) for i, vv, rr in zip(idx, v, r)]
    pt = pyref.Tree(height, leaves, SEED)
    name = "padding" if policy == 0 else "splitting"
    pC, pH, out = tr.prove_entities(idx[:2], policy, agg, 8, SEED)
    for k in range(2):
        sibs, aggregated, individual = pyref.dapol_prove(pt, int(idx[k]), name, agg, SEED, n=8)
        assert out[k].tobytes() == b"".join(aggregated) + b"".join(individual)
        assert pyref.policy_verify(name, aggregated, individual, [pC[k, s].tobytes() for s in range(height)], n=8)
        lf = pt.levels[0][int(idx[k])]
        assert pyref.verify_path(pt.root.C, pt.root.H, lf.C, lf.H, int(idx[k]), [(pC[k, s].tobytes(), pH[k, s].tobytes()) for s in range(height)])


# ------------------------------------------------------------------------------------------------ edge cases / errors
def test_error_behaviour(gpu_ctx, hip_lib):
    E = hip_lib.DapolError
    r2 = np.zeros((2, 32), np.uint8)
    with pytest.raises(E) as e:
        hip_lib.Tree(gpu_ctx, 4, [5, 5], [1, 2], r2, SEED)            # duplicate leaf: smtree panics
    assert e.value.code == 8
    with pytest.raises(E) as e:
        hip_lib.Tree(gpu_ctx, 4, [7, 3], [1, 2], r2, SEED)            # unsorted
    assert e.value.code == 8
    with pytest.raises(E) as e:
        hip_lib.Tree(gpu_ctx, 4, [3, 16], [1, 2], r2, SEED)           # index >= 2^height
    assert e.value.code == 8
    with pytest.raises(E) as e:
        hip_lib.Tree(gpu_ctx, 65, [3], [1], r2[:1], SEED)             # DapolError::TreeHeightTooBig
    assert e.value.code == 1
    with pytest.raises(E) as e:
        hip_lib.Tree(gpu_ctx, 2, [0, 1, 2], [1, 2, 3], np.zeros((3, 32), np.uint8), SEED, enforce_sparsity=True)   # SparsityTooSmall
    assert e.value.code == 2
    tr = hip_lib.Tree(gpu_ctx, 4, [3, 9], [1, 2], r2, SEED)
    with pytest.raises(E) as e:
        tr.paths([4])                                                 # Dapol::generate_proof -> None
    assert e.value.code == 9
    with pytest.raises(E) as e:
        tr.prove_entities([3], 0, 5, 8, SEED)                         # aggregation_factor > #siblings: reference panics
    assert e.value.code == 8
    with pytest.raises(E) as e:
        gpu_ctx.range_prove_batch(64, 64, np.zeros((1, 64), np.uint64), np.zeros((1, 64, 32), np.uint8), nonce_seed=SEED, stream_id=[0])
    assert e.value.code == 8                                          # m > max_parties (InvalidGeneratorsLength)
    with pytest.raises(E):
        gpu_ctx.range_prove_batch(12, 1, np.zeros((1, 1), np.uint64), np.zeros((1, 1, 32), np.uint8), nonce_seed=SEED, stream_id=[0])
    C, H = gpu_ctx.commit_hash_batch(np.zeros(0, np.uint64), np.zeros((0, 32), np.uint8))      # empty batch is a no-op
    assert C.shape == (0, 32)


def test_handles_may_be_destroyed_in_any_order(hip_lib):
    """A tree keeps its context alive (garbage-collected bindings finalise in arbitrary order)."""
    ctx = hip_lib.Context(0, 1)
    r = np.ones((2, 32), np.uint8)
    r[:, 31] = 0
    tr = hip_lib.Tree(ctx, 3, [1, 6], [4, 5], r, SEED)
    before = tr.root()
    ctx.close()                                            # the caller's handle goes first
    assert tr.root() == before and tr.paths([6])[0].shape == (1, 3, 32)
    tr.update([2], [9], r[:1])
    assert tr.root()[2] == 18
    tr.close()
    ctx2 = hip_lib.Context(0, 1)                           # no stale HIP error is left behind
    assert ctx2.commit_hash_batch([1], r[:1])[0].shape == (1, 32)


def test_single_leaf_and_full_level(gpu_ctx, hip_lib, ref):
    for height, idx in ((1, [1]), (3, list(range(8))), (10, [1023])):
        idx = np.array(idx, np.uint64)
        v = np.arange(1, len(idx) + 1, dtype=np.uint64)
        r = np.tile(np.arange(32, dtype=np.uint8), (len(idx), 1))
        r[:, 31] &= 0x0F
        tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
        t = _ref_tree(ref, height, idx, v, r)
        assert tr.root() == _ref_root(ref, t)
        ref.ref_tree_free(t)


def test_update_equals_build(gpu_ctx, hip_lib, ref):
    """src/tests.rs:41-48: a tree grown by update() has the root of build() over the same liabilities -- here bit for
    bit, every level, because padding nodes are keyed by position.  Also: replacing a liability, the last of several
    updates of one index winning, and the reference's one-leaf-at-a-time loop."""
    rng = np.random.default_rng(77)
    height, n = 10, 100                                     # the reference test's shape
    idx, v, r = _rand_leaves(rng, height, n)
    full = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    perm = rng.permutation(n)
    first, rest = np.sort(perm[:40]), perm[40:]             # `rest` arrives unsorted
    tr = hip_lib.Tree(gpu_ctx, height, idx[first], v[first], r[first], SEED)
    tr.update(idx[rest], v[rest], r[rest])
    assert tr.root() == full.root() and tr.node_count() == full.node_count()
    for level in range(height + 1):
        for a, b in zip(tr.level_nodes(level), full.level_nodes(level)):
            assert np.array_equal(a, b)
    one = hip_lib.Tree(gpu_ctx, height, idx[:1], v[:1], r[:1], SEED)
    for i in range(1, 12):
        one.update(idx[i:i + 1], v[i:i + 1], r[i:i + 1])
    t = _ref_tree(ref, height, idx[:12], v[:12], r[:12])
    assert one.root() == _ref_root(ref, t)
    ref.ref_tree_free(t)
    # replace: new value and blinding at an occupied index; a duplicate inside the batch -> the last one wins
    v2, r2 = v.copy(), r.copy()
    v2[5], r2[5] = 123456, r[6]
    tr.update(np.array([idx[5], idx[5]], np.uint64), np.array([999, 123456], np.uint64), np.stack([r[7], r[6]]))
    t = _ref_tree(ref, height, idx, v2, r2)
    assert tr.root() == _ref_root(ref, t)
    ref.ref_tree_free(t)
    assert tr.root()[2] == int(v2.sum())
    # errors leave the tree as it was: index beyond 2^height
    before = tr.root()
    with pytest.raises(hip_lib.DapolError):
        tr.update(np.array([1 << height], np.uint64), np.array([1], np.uint64), r[:1])
    assert tr.root() == before
    C, H, pv, pr = tr.paths(idx[:3])                        # the updated tree serves proofs
    assert C.shape == (3, height, 32)


@pytest.mark.parametrize("k", [1, 7, 64, 1500])
def test_incremental_update_equals_build_every_level(gpu_ctx, hip_lib, k):
    """dapol_tree_update on EXISTING leaves re-merges only the k root-to-leaf paths on the device (smtree's update,
    src/dapol/mod.rs:210-213); the tree must equal dapol_tree_build over the new liabilities at every level -- values,
    blindings, commitments, hashes, padding siblings -- both by the in-place path and by the rebuild it falls back to."""
    import os
    rng = np.random.default_rng(1000 + k)
    height, n = 24, 20000
    idx, v, r = _rand_leaves(rng, height, n)
    sel = rng.choice(n, size=k, replace=False)
    if k >= 7:
        sel[1] = sel[0] ^ 1 if (sel[0] ^ 1) < n else sel[1]      # neighbours in the sorted order (often siblings high up)
        sel = np.unique(sel)
    v2, r2 = v.copy(), r.copy()
    v2[sel] = rng.integers(0, 2**40, size=len(sel), dtype=np.uint64)
    r2[sel] = rng.integers(0, 256, size=(len(sel), 32), dtype=np.uint8)
    r2[sel, 31] &= 0x7F                                          # 255-bit blindings (Scalar::from_bits), some beyond l
    want = hip_lib.Tree(gpu_ctx, height, idx, v2, r2, SEED)
    order = rng.permutation(len(sel))                             # the update list arrives unsorted
    for env in ({}, {"DAPOL_UPDATE_INCREMENTAL_MAX": "0"}):
        tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
        os.environ.update(env)
        try:
            tr.update(idx[sel][order], v2[sel][order], r2[sel][order])
        finally:
            for key in env:
                os.environ.pop(key, None)
        assert tr.root() == want.root() and tr.node_count() == want.node_count(), env
        for level in range(height + 1):
            for a, b in zip(tr.level_nodes(level), want.level_nodes(level)):
                assert np.array_equal(a, b), (env, level)
        tr.close()
    want.close()


@pytest.mark.parametrize("k_new,k_old", [(1, 0), (5, 0), (64, 0), (300, 40), (3, 3)])
def test_incremental_insert_equals_build_every_level(gpu_ctx, hip_lib, k_new, k_old):
    """dapol_tree_update with NEW indexes (what the reference's update loop does, src/tests.rs:41-48: the tree grows leaf by leaf):
    the levels that gain nodes are rewritten in order on the device, the new chains and their padding siblings are made, the
    padding node the chain replaces goes away, and the existing ancestors move by a delta.  The tree must equal dapol_tree_build
    over the enlarged leaf set at EVERY level -- indexes, values, blindings, commitments, hashes, padding siblings, node counts --
    also for batches that mix new and existing indexes, and by the rebuild path (DAPOL_NO_INCREMENTAL_INSERT=1)."""
    import os
    rng = np.random.default_rng(7000 + 10 * k_new + k_old)
    height, n = 24, 20000
    idx_all, v_all, r_all = _rand_leaves(rng, height, n + k_new)
    na = len(idx_all)                                               # (_rand_leaves drops duplicate indexes)
    new_sel = rng.choice(na, size=k_new, replace=False)
    new_sel = np.unique(new_sel)
    old_mask = np.ones(na, bool)
    old_mask[new_sel] = False
    idx0, v0, r0 = idx_all[old_mask], v_all[old_mask], r_all[old_mask]
    v2, r2 = v_all.copy(), r_all.copy()
    rep = rng.choice(np.nonzero(old_mask)[0], size=k_old, replace=False) if k_old else np.zeros(0, np.int64)
    v2[rep] = rng.integers(0, 2**40, size=len(rep), dtype=np.uint64)
    r2[rep] = rng.integers(0, 256, size=(len(rep), 32), dtype=np.uint8)
    r2[rep, 31] &= 0x7F
    want = hip_lib.Tree(gpu_ctx, height, idx_all, v2, r2, SEED)
    upd = np.concatenate([new_sel, rep])
    order = rng.permutation(len(upd))
    for env in ({}, {"DAPOL_NO_INCREMENTAL_INSERT": "1"}):
        tr = hip_lib.Tree(gpu_ctx, height, idx0, v0, r0, SEED)
        os.environ.update(env)
        try:
            tr.update(idx_all[upd][order], v2[upd][order], r2[upd][order])
        finally:
            for key in env:
                os.environ.pop(key, None)
        assert tr.root() == want.root() and tr.node_count() == want.node_count(), env
        assert tr.last_update_path() == (0 if env else (3 if k_old else 2))            # random 24-bit indexes: no two new chains share a node
        for level in range(height + 1):
            for a, b in zip(tr.level_nodes(level), want.level_nodes(level)):
                assert np.array_equal(a, b), (env, level)
        C, H, pv, pr = tr.paths(idx_all[new_sel][:4])                # the grown tree serves proofs of its new leaves
        C2, H2, pv2, pr2 = want.paths(idx_all[new_sel][:4])
        assert np.array_equal(C, C2) and np.array_equal(H, H2) and np.array_equal(pv, pv2) and np.array_equal(pr, pr2)
        tr.close()
    want.close()


def _new_chains_share_a_node(old, new, height):
    """k_tree_ins_plan's rule: a new leaf's chain runs up to its first ancestor that exists; two (neighbouring) new leaves whose
    chains share a node are left to the rebuild."""
    new = np.sort(new)
    for a in range(1, len(new)):
        m = next(t for t in range(height + 1) if np.any((old >> np.uint64(t)) == (new[a] >> np.uint64(t))))
        if any((new[a] >> np.uint64(t)) == (new[a - 1] >> np.uint64(t)) for t in range(m)):
            return True
    return False


@pytest.mark.parametrize("shard_bits,prefix", [(1, 1), (3, 5)])
def test_incremental_update_of_a_shard_tree(gpu_ctx, hip_lib, shard_bits, prefix):
    """A rank's subtree (dapol_tree_build_shard: global indexes, positional padding seeds, levels 0 .. height - shard_bits) takes the
    same in-place paths: new leaves with the shard's prefix are inserted, existing ones replaced, and the tree equals a fresh shard
    build over the enlarged set at every level; a new leaf of ANOTHER shard is refused (by the rebuild's checks) and leaves the tree alone."""
    rng = np.random.default_rng(900 + shard_bits)
    height, n = 24, 6000
    idx_all, v_all, r_all = _rand_leaves(rng, height, n * (1 << shard_bits))
    mine = (idx_all >> np.uint64(height - shard_bits)) == prefix
    idx_s, v_s, r_s = idx_all[mine], v_all[mine], r_all[mine]
    ns = len(idx_s)
    new_sel = np.unique(rng.choice(ns, size=12, replace=False))
    old_mask = np.ones(ns, bool)
    old_mask[new_sel] = False
    rep = rng.choice(np.nonzero(old_mask)[0], size=25, replace=False)
    v2, r2 = v_s.copy(), r_s.copy()
    v2[rep] = rng.integers(0, 2**40, size=len(rep), dtype=np.uint64)
    r2[rep] = rng.integers(0, 256, size=(len(rep), 32), dtype=np.uint8)
    r2[rep, 31] &= 0x7F
    want = hip_lib.Tree(gpu_ctx, height, idx_s, v2, r2, SEED, shard_bits=shard_bits)
    tr = hip_lib.Tree(gpu_ctx, height, idx_s[old_mask], v_s[old_mask], r_s[old_mask], SEED, shard_bits=shard_bits)
    upd = rng.permutation(np.concatenate([new_sel, rep]))
    tr.update(idx_s[upd], v2[upd], r2[upd])
    assert tr.last_update_path() == (0 if _new_chains_share_a_node(idx_s[old_mask], idx_s[new_sel], height) else 3)
    assert tr.root() == want.root() and tr.node_count() == want.node_count()
    for level in range(height - shard_bits + 1):
        for a, b in zip(tr.level_nodes(level), want.level_nodes(level)):
            assert np.array_equal(a, b), level
    other = idx_all[~mine][:1]
    with pytest.raises(hip_lib.DapolError):
        tr.update(other, v_all[~mine][:1], r_all[~mine][:1])
    assert tr.root() == want.root() and tr.node_count() == want.node_count()
    tr.close()
    want.close()


def test_tree_grows_leaf_by_leaf_like_the_reference_test(gpu_ctx, hip_lib, ref):
    """src/tests.rs:41-48 at its own shape (height 10): a tree grown from one leaf by 60 single-leaf updates -- every one an
    incremental insert, repeated inserts ping-ponging the level buffers -- equals build() over the same liabilities at every level
    after every tenth update, and the C oracle's root at the end."""
    rng = np.random.default_rng(41)
    height, n = 10, 61
    idx, v, r = _rand_leaves(rng, height, n)
    order = rng.permutation(n)
    tr = hip_lib.Tree(gpu_ctx, height, idx[order[:1]], v[order[:1]], r[order[:1]], SEED)
    for i in range(1, n):
        e = order[i:i + 1]
        tr.update(idx[e], v[e], r[e])
        assert tr.last_update_path() == 2
        if i % 10 == 0 or i == n - 1:
            have = np.sort(order[:i + 1])
            want = hip_lib.Tree(gpu_ctx, height, idx[have], v[have], r[have], SEED)
            assert tr.root() == want.root() and tr.node_count() == want.node_count(), i
            for level in range(height + 1):
                for a, b in zip(tr.level_nodes(level), want.level_nodes(level)):
                    assert np.array_equal(a, b), (i, level)
    t = _ref_tree(ref, height, idx, v, r)
    assert tr.root() == _ref_root(ref, t)
    ref.ref_tree_free(t)


def test_incremental_update_at_2e20_leaves(gpu_ctx, hip_lib):
    """one leaf and 64 leaves replaced in the headline tree (2^20 leaves, height 32): root, value sum and the
    sampled paths equal a fresh build's; a batch that also holds a NEW index takes the rebuild and still agrees; the in-place
    path is far below the rebuild's 45 ms."""
    import time
    import bench
    height, n = 32, 1 << 20
    idx, v, r = bench.synth_inputs(n, height, 0, n)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    rng = np.random.default_rng(5)
    v2, r2 = v.copy(), r.copy()
    for k in (1, 64):
        sel = np.sort(rng.choice(n, size=k, replace=False))
        v2[sel] = rng.integers(0, 2**32, size=k, dtype=np.uint64)
        r2[sel] = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        r2[sel, 31] &= 0x0F
        tr.update(idx[sel], v2[sel], r2[sel])                      # warm (scratch allocation)
        t0 = time.perf_counter()
        tr.update(idx[sel], v2[sel], r2[sel])
        assert tr.last_update_path() == 1
        dt = time.perf_counter() - t0
        assert dt < 0.010, "incremental update of %d leaves took %.2f ms" % (k, dt * 1e3)
    want = hip_lib.Tree(gpu_ctx, height, idx, v2, r2, SEED)
    assert tr.root() == want.root() and tr.root()[2] == int(v2.sum())
    probe = np.concatenate([idx[sel][:8], idx[::n // 8][:8]])
    for a, b in zip(tr.paths(probe), want.paths(probe)):
        assert np.array_equal(a, b)
    # a new index among the updates: inserted in place (12 levels of 2^20 nodes each move up by one), same tree as building from scratch
    new_idx = np.array([idx[3] + 1, idx[10]], np.uint64)           # idx[3] + 1 is free in the strided layout
    t0 = time.perf_counter()
    tr.update(new_idx, np.array([77, 88], np.uint64), r[:2])
    assert tr.last_update_path() == 3
    assert time.perf_counter() - t0 < 0.060, "incremental insert into the 2^20-leaf tree took %.1f ms" % ((time.perf_counter() - t0) * 1e3)
    i3 = np.concatenate([idx, new_idx[:1]])
    o = np.argsort(i3, kind="stable")
    v3 = np.concatenate([v2, [77]]).astype(np.uint64)
    r3 = np.concatenate([r2, r[:1]])
    v3[10], r3[10] = 88, r[1]
    want3 = hip_lib.Tree(gpu_ctx, height, i3[o], v3[o], r3[o], SEED)
    assert tr.root() == want3.root()


def test_value_sum_wraps_like_release_rust(gpu_ctx, hip_lib, ref):
    idx = np.array([0, 1], np.uint64)
    v = np.array([2**64 - 1, 5], np.uint64)
    r = np.ones((2, 32), np.uint8)
    r[:, 31] = 0
    tr = hip_lib.Tree(gpu_ctx, 1, idx, v, r, SEED)
    assert tr.root()[2] == 4                                          # u64 wrap (node.rs:72 in release builds)
    t = _ref_tree(ref, 1, idx, v, r)
    assert tr.root() == _ref_root(ref, t)
    ref.ref_tree_free(t)


def test_merge_batch(gpu_ctx, hip_lib, pyref):
    nodes = [pyref.node_new(3 + i, 77 + 1000 * i) for i in range(6)]
    f = lambda attr, ns: np.array([list(getattr(n, attr)) for n in ns], np.uint8)
    L, R = nodes[0::2], nodes[1::2]
    C, H, v, r = gpu_ctx.merge_batch(f("C", L), f("H", L), f("C", R), f("H", R), [n.v for n in L], np.array([list(pyref.scalar_bytes(n.r)) for n in L], np.uint8),
                                     [n.v for n in R], np.array([list(pyref.scalar_bytes(n.r)) for n in R], np.uint8))
    for i in range(3):
        p = pyref.node_merge(L[i], R[i])
        assert (C[i].tobytes(), H[i].tobytes(), int(v[i]), r[i].tobytes()) == (p.C, p.H, p.v, pyref.scalar_bytes(p.r))
    C2, H2 = gpu_ctx.merge_batch(f("C", L), f("H", L), f("C", R), f("H", R))
    assert C2.tobytes() == C.tobytes() and H2.tobytes() == H.tobytes()
    bad = f("C", L)
    bad[0, 0] ^= 1                                                    # odd s -> not a canonical ristretto encoding
    with pytest.raises(hip_lib.DapolError) as e:
        gpu_ctx.merge_batch(bad, f("H", L), f("C", R), f("H", R))
    assert e.value.code == 7


# ------------------------------------------------------------------------------------------------ sharding (section 8e)
@pytest.mark.parametrize("shard_bits", [1, 2, 3])
def test_sharded_equals_single_gpu(gpu_ctx, hip_lib, shard_bits):
    """G logical shards on one device: subtree builds + merged top levels + upper siblings give the same root, paths and
    proof bytes as the single-GPU build of the whole tree."""
    from dapol_amd import sharded
    height, n_bits, g = 7, 8, 1 << shard_bits
    rng = np.random.default_rng(shard_bits)
    idx = np.sort(np.concatenate([rng.choice(1 << (height - shard_bits), size=3, replace=False).astype(np.uint64) + np.uint64(s << (height - shard_bits))
                                  for s in range(g)]))
    v = rng.integers(0, 4, size=len(idx), dtype=np.uint64)
    r = rng.integers(0, 256, size=(len(idx), 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    full = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    fC, fH, fout = full.prove_entities(idx, hip_lib.POLICY_PADDING, height, n_bits, SEED)
    shards, recs = [], []
    for s in range(g):
        sel = (idx >> np.uint64(height - shard_bits)) == s
        t = hip_lib.Tree(gpu_ctx, height, idx[sel], v[sel], r[sel], SEED, shard_bits=shard_bits)
        shards.append((sel, t))
        recs.append(sharded.pack_record(t.root()))
    records = sharded.unpack_records(np.stack(recs), g)
    for s, (sel, t) in enumerate(shards):
        root, upper = sharded.top_levels(gpu_ctx, records, s)
        assert root == full.root()
        pC, pH, out = t.prove_entities(idx[sel], hip_lib.POLICY_PADDING, height, n_bits, SEED, upper=upper)
        assert out.tobytes() == fout[sel].tobytes()
        assert pC.tobytes() == fC[sel].tobytes() and pH.tobytes() == fH[sel].tobytes()


def test_sharded_prover_falls_back_when_a_library_collective_fails(gpu_ctx, hip_lib):
    """A collective of the library's communicator that fails mid-run (DapolError out of dapol_shard_exchange /
    dapol_comm_allreduce_u64: a timeout, an asynchronous RCCL error) aborts that communicator -- never destroys it -- and the step
    completes over torch.distributed: same root as the single-GPU tree, the checksum reduced over the other transport."""
    import torch
    from dapol_amd import sharded
    height, n_bits = 7, 8
    rng = np.random.default_rng(77)
    idx = np.sort(np.concatenate([rng.choice(1 << (height - 1), size=3, replace=False).astype(np.uint64) + np.uint64(s << (height - 1)) for s in range(2)]))
    v = rng.integers(0, 4, size=len(idx), dtype=np.uint64)
    r = rng.integers(0, 256, size=(len(idx), 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    full = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    sel = (idx >> np.uint64(height - 1)) == 0
    rec1 = sharded.pack_record(hip_lib.Tree(gpu_ctx, height, idx[~sel], v[~sel], r[~sel], SEED, shard_bits=1).root())

    class Dist:                                          # a two-rank group whose other rank is replayed
        def __init__(self, other):
            self.other = other

        def all_gather(self, out, mine):
            out[0].copy_(mine)
            out[1].copy_(torch.from_numpy(rec1))

        class ReduceOp:
            MIN = "min"

        def all_reduce(self, t, op=None, group=None):
            if op == "min":                              # the agreement after a library collective: the replayed rank saw no error
                self.agreements = getattr(self, "agreements", 0) + 1
                return
            t += torch.tensor(self.other, dtype=t.dtype)

    class BadComm:
        def __init__(self, fail_exchange):
            self.fail_exchange, self.aborted, self.good = fail_exchange, False, None

        def exchange(self, root):
            if self.fail_exchange:
                raise hip_lib.DapolError(17, "ncclAllGather did not complete within the deadline")
            recs = sharded.unpack_records(np.stack([sharded.pack_record(root), rec1]), 2)
            return sharded.top_levels(gpu_ctx, recs, 0)

        def allreduce(self, words, op=None):
            raise hip_lib.DapolError(17, "ncclAllReduce: unhandled system error")

        def abort(self):
            self.aborted = True

    def run(comm, other):
        p = sharded.ShardedProver(gpu_ctx, height, idx[sel], v[sel], r[sel], rank=0, world=2, dist=Dist(other), torch=torch, comm_device="cpu")
        p.comm = comm
        return p, p.step(SEED, SEED, n_bits)

    p0, st0 = run(None, [0, 0])                          # the torch.distributed transport from the start: this rank's own checksum
    assert p0.root == full.root()
    for fail_exchange in (True, False):                  # the exchange fails / the exchange works and the final reduce fails
        bad = BadComm(fail_exchange)
        p, st = run(bad, [5, 7])
        assert bad.aborted and p.comm is None and p.comm_ranks is None
        assert "failed" in p.exchange_path and "nccl" in p.comm_error
        assert p.dist.agreements == (1 if fail_exchange else 2)          # the ranks agreed after every library collective, none after the drop
        assert p.root == full.root()
        assert st.checksum == (st0.checksum + 5 + (7 << 32)) & 0xFFFFFFFFFFFFFFFF
        st2 = p.step(SEED, SEED, n_bits)                 # and the next step stays on the fallback
        assert st2.checksum == st.checksum and p.root == full.root()


# ------------------------------------------------------------------------------------------------ workload (bench path)
def test_padding_nodes_and_empty_shard(gpu_ctx, hip_lib, pyref):
    """Paddable::padding as an entry point: equals the oracle's positional padding node and the node the tree builder
    puts at that position; an empty shard's record is exactly that node (top_levels then gives the single-GPU root)."""
    from dapol_amd.sharded import top_levels
    pos = [(0, 5), (3, 1), (7, 0), (31, 1), (63, 1)]
    C, H, r = gpu_ctx.padding_nodes(SEED, [l for l, _ in pos], [i for _, i in pos])
    for k, (level, index) in enumerate(pos):
        nd = pyref.node_padding(SEED, level, index)
        assert (C[k].tobytes(), H[k].tobytes(), r[k].tobytes()) == (nd.C, nd.H, nd.r.to_bytes(32, "little"))
    # all leaves in the left half of a height-6 tree: shard 1 of 2 is empty
    height = 6
    idx = np.array([1, 7, 20, 31], np.uint64)
    v = np.array([4, 5, 6, 7], np.uint64)
    rr = np.tile(np.arange(32, dtype=np.uint8), (4, 1))
    rr[:, 31] = 1
    whole = hip_lib.Tree(gpu_ctx, height, idx, v, rr, SEED)
    left = hip_lib.Tree(gpu_ctx, height, idx, v, rr, SEED, shard_bits=1)
    pC, pH, pr = gpu_ctx.padding_nodes(SEED, [height - 1], [1])
    lC, lH, lv, lr = left.root()
    recs = (np.stack([np.frombuffer(lC, np.uint8), pC[0]]), np.stack([np.frombuffer(lH, np.uint8), pH[0]]),
            np.array([lv, 0], np.uint64), np.stack([np.frombuffer(lr, np.uint8), pr[0]]))
    root, upper = top_levels(gpu_ctx, recs, 0)
    assert root == whole.root()
    wC, wH, out = whole.prove_entities(idx, 0, height, 8, SEED)
    sC, sH, sout = left.prove_entities(idx, 0, height, 8, SEED, upper=upper)
    assert out.tobytes() == sout.tobytes() and wC.tobytes() == sC.tobytes()


def test_workload_matches_api_and_is_deterministic(gpu_ctx, hip_lib):
    height, n = 10, 64
    stride = (1 << height) // n
    idx = (np.arange(n, dtype=np.uint64) * np.uint64(stride))
    rng = np.random.default_rng(3)
    v = rng.integers(0, 4, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    w = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    st1 = w.run(SEED, SEED, n_bits=8)
    proofs = w.proofs(0, n, st1.proof_bytes // n)
    st2 = w.run(SEED, SEED, n_bits=8)
    assert st1.checksum == st2.checksum and st1.proofs == n and st1.msm_launches > 0 and st1.msm_ms > 0
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    assert bytes(st1.root_C) == tr.root()[0] and bytes(st1.root_H) == tr.root()[1]
    _, _, out = tr.prove_entities(idx, hip_lib.POLICY_PADDING, height, 8, SEED)
    assert out.tobytes() == proofs.tobytes()
    st3 = w.run(SEED, bytes(32), n_bits=8)                            # another nonce seed -> other proof bytes
    assert st3.checksum != st1.checksum


def test_bench_layout_at_baseline_config_properties(gpu_ctx, hip_lib, ref):
    """BASELINE.json configs[1] shape (height 24, 64-bit proofs, strided leaves) at a size the oracle still finishes:
    root equals the oracle's, root value = sum, sampled proofs bit-exact and verifying."""
    height, n = 24, 1 << 12
    idx = (np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n))
    rng = np.random.default_rng(24)
    v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    w = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    st = w.run(SEED, SEED, n_bits=64)
    ps = st.proof_bytes // n
    assert ps == 992
    t = _ref_tree(ref, height, idx, v, r)
    rC, rH, rv, _ = _ref_root(ref, t)
    assert bytes(st.root_C) == rC and bytes(st.root_H) == rH and rv == int(v.sum())
    sample = np.ascontiguousarray(idx[::512])
    out = ctypes.create_string_buffer(ps * len(sample))
    assert ref.ref_prove_entities_padding(t, ctypes.c_size_t(len(sample)), sample.ctypes.data_as(ctypes.c_void_p), 64, SEED, 0, out) == 0
    for k in range(len(sample)):
        assert w.proofs(k * 512, 1, ps).tobytes() == out.raw[k * ps:(k + 1) * ps]
    ref.ref_tree_free(t)


# ------------------------------------------------------------------------------------------------ verifier (row a9)
def test_verify_golden_and_tampered(gpu_ctx):
    for c in load_golden("range.json"):
        n, m = c["n"], c["m"]
        if m > gpu_ctx.max_parties:
            continue                                     # (the 64-party vector: tests/test_gpu_full_range.py)
        proof = np.frombuffer(bytes.fromhex(c["proof"]), np.uint8)
        Vs = _arr(c["commitments"]).reshape(1, m, 32)
        cases, expect = [proof], [1]
        for off in (3, 40, 100, 130, 170, 200, 230, len(proof) - 40, len(proof) - 5):      # A, S, T2, t_x, tau, mu, L0, a, b
            bad = proof.copy()
            bad[off] ^= 4
            cases.append(bad)
            expect.append(0)
        nc = proof.copy()                      # non-canonical scalar: t_x + l
        tx = int.from_bytes(proof[128:160].tobytes(), "little") + (2**252 + 27742317777372353535851937790883648493)
        if tx < 2**256:
            nc[128:160] = np.frombuffer(tx.to_bytes(32, "little"), np.uint8)
            cases.append(nc)
            expect.append(0)
        zero_pt = proof.copy()
        zero_pt[0:32] = 0                      # identity A: validate_and_append_point fails
        cases.append(zero_pt)
        expect.append(0)
        ok = gpu_ctx.range_verify_batch(n, m, np.stack(cases), np.repeat(Vs, len(cases), axis=0), verify_seed=SEED)
        assert list(ok) == expect, (n, m, list(ok))
        badV = Vs.copy()
        badV[0, m - 1, 1] ^= 2
        assert list(gpu_ctx.range_verify_batch(n, m, proof.reshape(1, -1), badV, verify_seed=SEED)) == [0]


@pytest.mark.parametrize("n_bits,m,b", [(64, 32, 40), (64, 1, 70), (16, 4, 65)])
def test_prove_then_verify_roundtrip(gpu_ctx, ref, n_bits, m, b):
    rng = np.random.default_rng(n_bits + m)
    v = rng.integers(0, 2**n_bits if n_bits < 64 else 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    proofs = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    C, _ = gpu_ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    Vs = C.reshape(b, m, 32)
    assert gpu_ctx.range_verify_batch(n_bits, m, proofs, Vs, verify_seed=SEED).all()
    swapped = Vs.copy()
    swapped[[0, 1]] = swapped[[1, 0]]          # proofs 0 and 1 against each other's commitments
    ok = gpu_ctx.range_verify_batch(n_bits, m, proofs, swapped, verify_seed=bytes(32))
    assert list(ok[:2]) == [0, 0] and ok[2:].all()
    # GPU verdicts agree with the CPU oracle's verifier on the same bytes
    ps = proofs.shape[1]
    c7 = bytes([7]) + bytes(31)
    for k in (0, b - 1):
        assert ref.ref_range_verify(n_bits, m, proofs[k].tobytes(), ctypes.c_size_t(ps), Vs[k].tobytes(), c7, 0) == 1


def test_out_of_range_value_is_rejected(gpu_ctx, pyref):
    """A proof made for v >= 2^n (built by the oracle's prover, which does not check) must not verify."""
    bl = [pyref.scalar_from_wide(pyref.seed_wide(SEED, 9, 0, i)) for i in range(2)]
    pr = pyref.range_prove([256, 1], bl, 8, pyref.Tape(seed=SEED, stream_id=7))
    Vs = np.array([list(pyref.pedersen_commit(v, b).compress()) for v, b in zip([256, 1], bl)], np.uint8).reshape(1, 2, 32)
    assert list(gpu_ctx.range_verify_batch(8, 2, np.frombuffer(pr, np.uint8).reshape(1, -1), Vs, verify_seed=SEED)) == [0]


# ------------------------------------------------------------------------------------------------ leaf derivation (8f #1)
def test_leaf_derivation_reference_kats(gpu_ctx, hip_lib):
    """src/dapol/tests.rs:30-85: ids a,b,c,d (external w,x,y,z), seed "test", Blake2s, height 4 -> leaves 7, 12, 2, 4;
    :24: root value 26 -- through the GPU."""
    kat = load_golden("kat.json")
    liab = [(i.encode(), e.encode(), v) for i, e, v in kat["liabilities"]]
    out = gpu_ctx.build_leaf_nodes(liab, b"test", 4, hip_lib.DIGEST_BLAKE2S)
    assert list(map(int, out["idx_by_entity"])) == [7, 12, 2, 4]
    assert list(map(int, out["leaf_idx"])) == [2, 4, 7, 12] and list(map(int, out["order"])) == [2, 3, 0, 1]
    by_id = {l["id"]: l for l in kat["blake2s"]["leaves"]}
    for p, e in enumerate(out["order"]):
        exp = by_id[kat["liabilities"][int(e)][0]]
        assert out["r"][p].tobytes().hex() == exp["r"] and int(out["v"][p]) == exp["v"]
    # the same liabilities under BLAKE3 feed the tree builder and give the golden root
    out3 = gpu_ctx.build_leaf_nodes(liab, b"test", 4, hip_lib.DIGEST_BLAKE3)
    assert {kat["liabilities"][i][0]: int(x) for i, x in enumerate(out3["idx_by_entity"])} == kat["blake3"]["index"]
    tr = hip_lib.Tree(gpu_ctx, 4, out3["leaf_idx"], out3["v"], out3["r"], bytes(range(32)), enforce_sparsity=True)
    C, H, v, r = tr.root()
    assert (C.hex(), H.hex(), v) == (kat["blake3"]["root"]["C"], kat["blake3"]["root"]["H"], 26)


def test_blake2s_dapol_like_the_reference_tests(hip_lib, pyref):
    """src/dapol/tests.rs:18-107 builds Dapol::<blake2::Blake2s, RangeProofPadding>: Blake2s is the NODE hash too.
    A Blake2s context: every node of the tree, the proofs for ids a..d (leaves 7, 12, 2, 4), the batch proof for
    [a, b] and both verifiers, against the Python restatement with dg = blake2s."""
    ctx = hip_lib.Context(0, 8, digest=hip_lib.DIGEST_BLAKE2S)
    kat = load_golden("kat.json")
    liab = [(i.encode(), e.encode(), v) for i, e, v in kat["liabilities"]]
    out = ctx.build_leaf_nodes(liab, b"test", 4, hip_lib.DIGEST_BLAKE2S)
    tr = hip_lib.Tree(ctx, 4, out["leaf_idx"], out["v"], out["r"], SEED, enforce_sparsity=True)
    pt, id_map = pyref.dapol_new(liab, b"test", 4, SEED, dg="blake2s")
    assert id_map == {b"a": 7, b"b": 12, b"c": 2, b"d": 4}
    C, H, v, r = tr.root()
    assert (C, H, v) == (pt.root.C, pt.root.H, 26)                        # tests.rs:24
    for level in range(5):
        idx, vv, rr, CC, HH, pad = tr.level_nodes(level)
        got = {int(i): (c.tobytes(), h.tobytes()) for i, c, h in zip(idx, CC, HH)}
        assert got == {i: (n.C, n.H) for i, n in pt.levels[level].items()}
    leaf_idx = np.array([2, 4, 7, 12], np.uint64)
    pC, pH, proofs = tr.prove_entities(leaf_idx, 0, 2, 8, SEED)           # build_test_options(4, 2)
    for k, li in enumerate(leaf_idx):
        sibs, aggregated, individual = pyref.dapol_prove(pt, int(li), "padding", 2, SEED, n=8)
        assert [pH[k, s].tobytes() for s in range(4)] == [x.H for x in sibs]
        assert proofs[k].tobytes() == b"".join(aggregated) + b"".join(individual)
    lC, lH = ctx.commit_hash_batch(out["v"], out["r"])
    assert [h.tobytes() for h in lH] == [pt.levels[0][int(i)].H for i in leaf_idx]
    assert ctx.verify_entities(4, leaf_idx, lC, lH, pC, pH, C, H, 0, 2, 8, proofs, verify_seed=SEED).all()
    level, index, sC, sH, blob = tr.prove_batch([7, 12], 0, 2, 8, SEED)   # generate_proof_batch_for_ids([a, b])
    _, sibs, aggregated, individual = pyref.dapol_prove_batch(pt, [7, 12], "padding", 2, SEED, n=8)
    assert [h.tobytes() for h in sH] == [x.H for x in sibs] and blob == b"".join(aggregated) + b"".join(individual)
    assert ctx.verify_batch(4, [7, 12], lC[2:], lH[2:], sC, sH, C, H, 0, 2, 8, blob, verify_seed=SEED)
    # the same proof does not verify under a BLAKE3 context (different node hashes)
    ctx3 = hip_lib.Context(0, 8)
    assert not ctx3.verify_batch(4, [7, 12], lC[2:], lH[2:], sC, sH, C, H, 0, 2, 8, blob, verify_seed=SEED)
    assert not ctx3.verify_entities(4, leaf_idx, lC, lH, pC, pH, C, H, 0, 2, 8, proofs, verify_seed=SEED).any()
    mC, mH, mv, mr = ctx.merge_batch(lC[:1], lH[:1], lC[1:2], lH[1:2], out["v"][:1], out["r"][:1], out["v"][1:2], out["r"][1:2])
    n = pyref.node_merge(pt.levels[0][2], pt.levels[0][4], "blake2s")
    assert (mC[0].tobytes(), mH[0].tobytes(), int(mv[0])) == (n.C, n.H, 18)
    with pytest.raises(hip_lib.DapolError) as e:
        hip_lib.Context(0, 8, digest=3)                                   # BLAKE3, Blake2s and (new_blank + build only, tests/test_gpu_blake2b.py) Blake2b
    assert e.value.code == 3                                              # DapolError::InvalidDigestSize


@pytest.mark.parametrize("height,n,digest", [(6, 30, "blake3"), (8, 100, "blake2s"), (5, 16, "blake3"), (20, 3000, "blake3"), (7, 64, "blake2s")])
def test_leaf_derivation_vs_oracle_with_collisions(gpu_ctx, hip_lib, pyref, height, n, digest):
    """Small heights force many index collisions: the order-dependent retry semantics of shuffle_index must match exactly."""
    rng = np.random.default_rng(height * 7 + n)
    liab = []
    for i in range(n):
        iid = b"id-%06d" % i + bytes(rng.integers(0, 256, size=int(rng.integers(0, 90)), dtype=np.uint8))
        eid = bytes(rng.integers(0, 256, size=int(rng.integers(0, 70)), dtype=np.uint8))
        liab.append((iid, eid, int(rng.integers(0, 2**32))))
    seed = b"audit seed " + bytes(rng.integers(0, 256, size=40, dtype=np.uint8))
    leaves, idm = pyref.build_leaf_nodes(liab, seed, height, digest)
    first = [pyref.shuffle_index(pyref.digest(digest, pyref.digest(digest, seed, i_), b"index_seed", e_), height, set(), digest) for i_, e_, _ in liab]
    assert height > 12 or len(set(first)) < n, "test should exercise collisions"
    out = gpu_ctx.build_leaf_nodes(liab, seed, height, hip_lib.DIGEST_BLAKE3 if digest == "blake3" else hip_lib.DIGEST_BLAKE2S)
    assert [int(x) for x in out["idx_by_entity"]] == [idm[l[0]] for l in liab]
    assert [int(x) for x in out["leaf_idx"]] == [i for i, _ in leaves]
    for p, (i, nd) in enumerate(leaves):
        assert int(out["v"][p]) == nd.v and out["r"][p].tobytes() == nd.r.to_bytes(32, "little")


def test_leaf_derivation_near_the_sparsity_bound(gpu_ctx, hip_lib):
    """the densest tree Dapol::new accepts (2^height = 2 n: height 16, 32,768 liabilities -- about 11,600
    first-choice collisions, retried in input order) against an independent restatement of build_leaf_nodes' index rule
    (src/dapol/mod.rs:341-441) on hashlib's Blake2s: every index, in entity order and in sorted order, and every blinding."""
    import hashlib
    height, n = 16, 1 << 15
    seed = b"sparsity bound"
    ids = [b"id-%08d" % i for i in range(n)]
    eids = [b"ext-%d" % (i * 7) for i in range(n)]
    dg = lambda *parts: hashlib.blake2s(b"".join(parts)).digest()
    taken, want_idx, want_r, retried = set(), [], [], 0
    for iid, eid in zip(ids, eids):
        audit_id = dg(seed, iid)
        s = dg(audit_id, b"index_seed", eid)
        for attempt in range(1 << 20):
            s = dg(s)
            idx = int.from_bytes(s[:8], "big") >> (64 - height)
            if idx not in taken:
                taken.add(idx)
                break
            retried += 1
        want_idx.append(idx)
        want_r.append(int.from_bytes(dg(audit_id, b"blind_seed", eid), "little") & (2**255 - 1))
    assert retried > 5000, "the case must exercise collision resolution"
    vals = np.arange(n, dtype=np.uint64) + 1
    out = gpu_ctx.build_leaf_nodes(list(zip(ids, eids, [int(x) for x in vals])), seed, height, hip_lib.DIGEST_BLAKE2S)
    assert [int(x) for x in out["idx_by_entity"]] == want_idx
    order = np.argsort(np.array(want_idx, np.uint64), kind="stable")
    assert np.array_equal(out["leaf_idx"], np.array(want_idx, np.uint64)[order]) and np.array_equal(out["order"], order.astype(np.uint32))
    assert np.array_equal(out["v"], vals[order])
    for p in range(0, n, 97):
        assert out["r"][p].tobytes() == want_r[int(order[p])].to_bytes(32, "little")
    # the one-lane resolution the claim / settle rounds replaced gives the same arrays
    import os
    os.environ["DAPOL_LEAF_SERIAL"] = "1"
    try:
        ser = gpu_ctx.build_leaf_nodes(list(zip(ids, eids, [int(x) for x in vals])), seed, height, hip_lib.DIGEST_BLAKE2S)
    finally:
        os.environ.pop("DAPOL_LEAF_SERIAL", None)
    for key in ("leaf_idx", "v", "r", "order", "idx_by_entity"):
        assert np.array_equal(ser[key], out[key]), key


def test_leaf_derivation_at_2e20(gpu_ctx, hip_lib, pyref):
    """dapol_build_leaf_nodes at the headline size (2^20 ids `id-%08d`, BLAKE3, height 32): every index below
    2^32 and distinct, the sorted view a permutation of the entity view, values carried; 64 sampled entities' blindings and
    first-choice indexes equal the Python restatement's (at 2^20 in 2^32 slots a collision is rare: an index that differs from
    the first choice must be one whose first choice is taken by an EARLIER entity); the tree over them sums the liabilities."""
    n, height, seed = 1 << 20, 32, b"bench-audit-seed"
    ids = np.char.add("id-", np.char.zfill(np.arange(n).astype("U8"), 8)).astype("S11")
    off = (np.arange(n + 1, dtype=np.uint64) * 11).astype(np.uint32)
    vals = np.random.default_rng(4).integers(0, 1 << 32, size=n, dtype=np.uint64)
    out = gpu_ctx.build_leaf_nodes_packed(ids.tobytes(), off, ids.tobytes(), off, vals, seed, height)
    li, be = out["leaf_idx"], out["idx_by_entity"]
    assert li.max() < (1 << height) and np.all(li[1:] > li[:-1])                     # sorted, distinct
    assert np.array_equal(np.sort(be), li) and np.array_equal(be[out["order"]], li) and np.array_equal(out["v"], vals[out["order"]])
    pos_of = np.empty(n, np.int64)
    pos_of[out["order"]] = np.arange(n)
    moved = 0
    for e in list(range(0, n, n // 60)) + [n - 1]:
        iid = bytes(ids[e])
        audit_id = pyref.digest("blake3", seed, iid)
        first = pyref.shuffle_index(pyref.digest("blake3", audit_id, b"index_seed", iid), height, set(), "blake3")
        r_want = pyref.scalar_from_bits(pyref.digest("blake3", audit_id, b"blind_seed", iid)).to_bytes(32, "little")
        assert out["r"][pos_of[e]].tobytes() == r_want
        if int(be[e]) != first:
            moved += 1
            assert first in set(int(x) for x in be[:e][be[:e] == first])             # its first choice went to an earlier entity
    assert moved <= 2
    tr = hip_lib.Tree(gpu_ctx, height, li, out["v"], out["r"], SEED)
    assert tr.root()[2] == int(vals.sum())


def test_leaf_derivation_errors(gpu_ctx, hip_lib):
    E = hip_lib.DapolError
    with pytest.raises(E) as e:
        gpu_ctx.build_leaf_nodes([(b"a", b"w", 1), (b"b", b"x", 2), (b"a", b"y", 3)], b"test", 8)
    assert e.value.code == 4                                          # DapolError::DuplicatedInternalId
    with pytest.raises(E) as e:
        gpu_ctx.build_leaf_nodes([(bytes([i]), b"e", 1) for i in range(9)], b"test", 4)
    assert e.value.code == 2                                          # DapolError::SparsityTooSmall (2^4 < 2*9)
    with pytest.raises(E) as e:
        gpu_ctx.build_leaf_nodes([(b"a", b"w", 1)], b"test", 65)
    assert e.value.code == 1                                          # DapolError::TreeHeightTooBig
    for dg in (hip_lib.DIGEST_BLAKE3, hip_lib.DIGEST_BLAKE2S):         # ids beyond one BLAKE3 chunk are legal (mod.rs:347-349): no error
        out = gpu_ctx.build_leaf_nodes([(b"a" * 1100, b"w", 1)], b"test", 8, dg)      # (values: tests/test_gpu_full_range.py)
        assert len(out["leaf_idx"]) == 1


# ------------------------------------------------------------------------------------------------ DapolProof::verify (8f #3)
@pytest.mark.parametrize("height,policy,agg", [(8, 0, 8), (8, 1, 5), (6, 0, 3), (9, 1, 9), (5, 0, 0)])
def test_prove_then_verify_entities(gpu_ctx, hip_lib, height, policy, agg):
    """The reference's round trip (src/tests.rs:66-93, 108-127): every proof verifies against root and leaf; tampering fails."""
    rng = np.random.default_rng(height * 31 + agg)
    idx, v, r = _rand_leaves(rng, height, 12, vmax=8)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    rC, rH, _, _ = tr.root()
    pC, pH, proofs = tr.prove_entities(idx, policy, agg, 8, SEED)
    lC, lH = gpu_ctx.commit_hash_batch(v, r)                              # the leaves' proof nodes
    ok = gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)
    assert ok.all()
    bad_hash = pH.copy()
    bad_hash[0, height - 1, 5] ^= 1                                       # sibling hash of entity 0
    bad_range = proofs.copy()
    bad_range[1, 70] ^= 1                                                 # range proof of entity 1
    wrong_leaf = lC.copy()
    wrong_leaf[2] = lC[3]                                                 # entity 2 presented with another leaf commitment
    assert list(gpu_ctx.verify_entities(height, idx, lC, lH, pC, bad_hash, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)) == [0] + [1] * 11
    assert list(gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, rC, rH, policy, agg, 8, bad_range, verify_seed=SEED)) == [1, 0] + [1] * 10
    assert list(gpu_ctx.verify_entities(height, idx, wrong_leaf, lH, pC, pH, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)) == [1, 1, 0] + [1] * 9
    other_root = bytes([rH[0] ^ 1]) + rH[1:]
    assert not gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, rC, other_root, policy, agg, 8, proofs, verify_seed=SEED).any()
    swapped = idx.copy()
    swapped[[4, 5]] = swapped[[5, 4]]                                     # proofs presented for the wrong positions
    okp = gpu_ctx.verify_entities(height, swapped, lC, lH, pC, pH, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)
    assert okp[:4].all() and okp[6:].all()
    # calls of many proofs check a path per LANE (k_verify_paths) instead of per wavefront: the same verdicts
    import os
    os.environ["DAPOL_PATHS_LANE"] = "1"
    try:
        assert gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, rC, rH, policy, agg, 8, proofs, verify_seed=SEED).all()
        assert list(gpu_ctx.verify_entities(height, idx, lC, lH, pC, bad_hash, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)) == [0] + [1] * 11
        assert list(gpu_ctx.verify_entities(height, idx, wrong_leaf, lH, pC, pH, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)) == [1, 1, 0] + [1] * 9
        assert list(gpu_ctx.verify_entities(height, swapped, lC, lH, pC, pH, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)) == list(okp)
    finally:
        os.environ.pop("DAPOL_PATHS_LANE", None)


@pytest.mark.parametrize("height,policy,agg,pick", [(6, 0, 4, [0, 2, 3]), (7, 1, 7, [1, 4]), (5, 0, 0, [0, 1, 2, 3, 4]), (8, 1, 3, [2]), (6, 0, None, [0, 4])])
def test_batch_proofs_vs_python_oracle(gpu_ctx, hip_lib, pyref, height, policy, agg, pick):
    """Dapol::generate_proof_batch for several leaves (src/dapol/mod.rs:172-190): deduplicated siblings and one policy
    proof over them, byte for byte against the Python restatement; agg = None -> aggregate every sibling."""
    rng = np.random.default_rng(height * 7 + len(pick))
    idx, v, r = _rand_leaves(rng, height, 5, vmax=8)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    leaves = [(int(i), pyref.node_new(int(vv), int.from_bytes(rr.tobytes(), "little"))) for i, vv, rr in zip(idx, v, r)]
    pt = pyref.Tree(height, leaves, SEED)
    name = "padding" if policy == 0 else "splitting"
    sel = [int(idx[i]) for i in pick]
    pos = pyref.batch_siblings(height, sel)
    if agg is None:
        agg = len(pos)
    level, index, sC, sH, blob = tr.prove_batch(sel, policy, agg, 8, SEED)
    assert list(zip(map(int, level), map(int, index))) == pos
    _, sibs, aggregated, individual = pyref.dapol_prove_batch(pt, sel, name, agg, SEED, n=8)
    assert [c.tobytes() for c in sC] == [x.C for x in sibs] and [h.tobytes() for h in sH] == [x.H for x in sibs]
    assert blob == b"".join(aggregated) + b"".join(individual)
    lv = [(i, pt.levels[0][i].C, pt.levels[0][i].H) for i in sel]
    assert pyref.verify_batch_paths(pt.root.C, pt.root.H, height, lv, [(x.C, x.H) for x in sibs])
    assert pyref.policy_verify(name, aggregated, individual, [x.C for x in sibs], n=8)
    if len(sel) == 1:                                                      # k = 1 is exactly generate_proof
        pC, pH, out = tr.prove_entities(sel, policy, agg, 8, SEED)
        assert out[0].tobytes() == blob and pC[0].tobytes() == sC.tobytes()


@pytest.mark.parametrize("policy", [0, 1])
def test_batch_prove_then_verify(gpu_ctx, hip_lib, policy):
    """src/tests.rs:50-70: batches of 10 of the 100 leaves of a height-10 tree verify against the root; tampering fails.
    16-bit proofs: an upper sibling holds the sum of up to 100 liabilities below 200."""
    rng = np.random.default_rng(5 + policy)
    height, n = 10, 100
    idx, v, r = _rand_leaves(rng, height, n, vmax=200)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    rC, rH, _, _ = tr.root()
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    for i in (0, 3, 9):
        sl = slice(10 * i, 10 * i + 10)
        S = len(hip_lib.batch_siblings(height, idx[sl])[0])
        agg = min(S, 5 + i)
        level, index, sC, sH, blob = tr.prove_batch(idx[sl], policy, agg, 16, SEED)
        assert len(level) == S and len(blob) == hip_lib.lib().dapol_entity_proof_size(S, policy, agg, 16)
        args = (policy, agg, 16, blob)
        assert gpu_ctx.verify_batch(height, idx[sl], lC[sl], lH[sl], sC, sH, rC, rH, *args, verify_seed=SEED)
        bad = sC.copy(); bad[S // 2] = sC[(S // 2 + 1) % S]
        assert not gpu_ctx.verify_batch(height, idx[sl], lC[sl], lH[sl], bad, sH, rC, rH, *args, verify_seed=SEED)
        bad = sH.copy(); bad[0, 0] ^= 1
        assert not gpu_ctx.verify_batch(height, idx[sl], lC[sl], lH[sl], sC, bad, rC, rH, *args, verify_seed=SEED)
        bad = bytearray(blob); bad[40] ^= 1
        assert not gpu_ctx.verify_batch(height, idx[sl], lC[sl], lH[sl], sC, sH, rC, rH, policy, agg, 16, bytes(bad), verify_seed=SEED)
        wl = lC[sl].copy(); wl[1] = wl[2]
        assert not gpu_ctx.verify_batch(height, idx[sl], wl, lH[sl], sC, sH, rC, rH, *args, verify_seed=SEED)
        assert not gpu_ctx.verify_batch(height, idx[sl], lC[sl], lH[sl], sC[:-1], sH[:-1], rC, rH, *args, verify_seed=SEED)   # a sibling short
        other = idx[sl].copy(); other[-1] = idx[(10 * i + 10) % n] if i < 9 else idx[0]
        other = np.sort(other)
        if len(hip_lib.batch_siblings(height, other)[0]) == S:                # same shape, different leaf set
            assert not gpu_ctx.verify_batch(height, other, lC[sl], lH[sl], sC, sH, rC, rH, *args, verify_seed=SEED)
    E = hip_lib.DapolError
    with pytest.raises(E) as e:
        tr.prove_batch(idx[[3, 2]], policy, 1, 16, SEED)                       # unsorted leaf list
    assert e.value.code == 8
    with pytest.raises(E) as e:
        free = next(i for i in range(1 << height) if i not in set(map(int, idx)))
        tr.prove_batch(np.sort(np.array([int(idx[0]), free], np.uint64)), policy, 1, 16, SEED)   # no liability there -> None
    assert e.value.code == 9
    with pytest.raises(E) as e:
        tr.prove_batch(idx[:2], policy, 10**6, 16, SEED)                       # aggregation_factor > #siblings: reference panics
    assert e.value.code == 8
    # nonce hygiene: a batch and the single-leaf proof of its first leaf draw from different streams
    _, _, _, _, b2 = tr.prove_batch(idx[:2], policy, 0, 16, SEED)
    _, _, _, _, b1 = tr.prove_batch(idx[:1], policy, 0, 16, SEED)
    assert b2[:64] != b1[:64]


def test_full_size_roundtrip_config1(gpu_ctx, hip_lib):
    """BASELINE configs[0] shape: 2^10 entities, height 16, 64-bit proofs -- every inclusion proof verifies (encode -> verify)."""
    height, n = 16, 1 << 10
    idx = (np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n))
    rng = np.random.default_rng(16)
    v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    rC, rH, rv, _ = tr.root()
    assert rv == int(v.sum())
    pC, pH, proofs = tr.prove_entities(idx, hip_lib.POLICY_PADDING, height, 64, SEED)
    assert proofs.shape[1] == 32 * (9 + 2 * 10)                            # 928 bytes: m = 16 parties of 64 bits
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    assert gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, rC, rH, hip_lib.POLICY_PADDING, height, 64, proofs, verify_seed=SEED).all()


def _oracle_padding_proof(ref, n_bits, height, sv, sr, leaf, seed=SEED):
    """The C oracle's padding-policy proof over one entity's siblings (root side first), padded with (0, 1)."""
    m = 1
    while m < height:
        m <<= 1
    pv, pr = np.zeros(m, np.uint64), np.zeros((m, 32), np.uint8)
    pv[:height], pr[:height] = sv, sr
    pr[height:, 0] = 1
    ref.ref_range_proof_size.restype = ctypes.c_size_t
    ps = ref.ref_range_proof_size(n_bits, m)
    out = ctypes.create_string_buffer(ps)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert ref.ref_range_prove(n_bits, m, p(pv), p(pr), seed, ctypes.c_uint64(int(leaf)), ctypes.c_uint64(0), None, 0, out) == 0
    return out.raw


def test_full_size_config0_32bit(gpu_ctx, hip_lib, ref):
    """BASELINE configs[0] as written: 2^10 entities, height 16, 32-BIT range proofs -- the whole tree against the C oracle,
    sampled proofs byte for byte, and every inclusion proof through DapolProof::verify."""
    height, n, nb = 16, 1 << 10, 32
    idx = (np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n))
    rng = np.random.default_rng(1016)
    v = rng.integers(0, 2**20, size=n, dtype=np.uint64)                    # sums stay below 2^32
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    w = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    st = w.run(SEED, SEED, n_bits=nb)
    ps = st.proof_bytes // n
    assert ps == 32 * (9 + 2 * 9)                                          # m = 16 parties of 32 bits: 864 bytes
    t = _ref_tree(ref, height, idx, v, r)
    rC, rH, rv, _ = _ref_root(ref, t)
    ref.ref_tree_free(t)
    assert bytes(st.root_C) == rC and bytes(st.root_H) == rH and rv == int(v.sum())
    sample = np.ascontiguousarray(idx[::64])
    sv, sr, sC, sH = w.paths(sample, with_nodes=True)
    for k, leaf in enumerate(sample):
        assert w.proofs(k * 64, 1, ps).tobytes() == _oracle_padding_proof(ref, nb, height, sv[k], sr[k], leaf)
    aC, aH = w.paths(idx, with_nodes=True)[2:]
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    ok = gpu_ctx.verify_entities(height, idx, lC, lH, aC, aH, bytes(st.root_C), bytes(st.root_H), hip_lib.POLICY_PADDING, height, nb,
                                 w.proofs(0, n, ps), verify_seed=SEED)
    assert ok.all()


def test_full_size_config1_properties(gpu_ctx, hip_lib, ref):
    """BASELINE configs[1] at FULL size: 2^16 entities, height 24, 64-bit proofs.  Size-independent properties: root value =
    sum of liabilities, two runs give the same checksum, another nonce seed changes it, sampled proofs equal the C oracle's
    over the same siblings, and 4,096 sampled inclusion proofs pass DapolProof::verify while a tampered one fails."""
    height, n = 24, 1 << 16
    idx = (np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n))
    rng = np.random.default_rng(2416)
    v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    w = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    (rC, rH, rv, _), _ = w.build(SEED)
    assert rv == int(v.sum())
    st = w.prove(SEED, 64)
    ps = st.proof_bytes // n
    assert ps == 992 and st.proofs == n
    c1 = st.checksum
    pos = np.arange(0, n, 16)
    sample = np.ascontiguousarray(idx[pos])
    sv, sr, sC, sH = w.paths(sample, with_nodes=True)
    proofs = np.concatenate([w.proofs(int(q), 1, ps) for q in pos])
    for k in (0, 1000, 4095):
        assert proofs[k].tobytes() == _oracle_padding_proof(ref, 64, height, sv[k], sr[k], sample[k])
    lC, lH = gpu_ctx.commit_hash_batch(v[pos], r[pos])
    args = (rC, rH, hip_lib.POLICY_PADDING, height, 64)
    assert gpu_ctx.verify_entities(height, sample, lC, lH, sC, sH, *args, proofs, verify_seed=SEED).all()
    bad = proofs.copy()
    bad[7, 500] ^= 0x10
    okb = gpu_ctx.verify_entities(height, sample, lC, lH, sC, sH, *args, bad, verify_seed=SEED)
    assert okb[7] == 0 and okb.sum() == len(pos) - 1
    assert w.prove(SEED, 64).checksum == c1                                  # deterministic
    assert w.prove(bytes(32), 64).checksum != c1                             # fresh nonces, different bytes


def test_config4_shape_1024_party_verification(hip_lib):
    """BASELINE configs[4] shape: aggregated proofs over 1,024 commitments (m = 1024, n = 64) verify; one flipped bit or one
    swapped commitment is rejected, and only in the proof it belongs to."""
    ctx = hip_lib.Context(0, 1024)
    b, m = 3, 1024
    rng = np.random.default_rng(1024)
    v = rng.integers(0, 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    proofs = ctx.range_prove_batch(64, m, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    assert proofs.shape[1] == 32 * (9 + 2 * 16)                              # 1,312 bytes
    C, _ = ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    V = C.reshape(b, m, 32)
    assert ctx.range_verify_batch(64, m, proofs, V, verify_seed=SEED).all()
    bad = proofs.copy()
    bad[1, 1000] ^= 1
    assert list(ctx.range_verify_batch(64, m, bad, V, verify_seed=SEED)) == [1, 0, 1]
    V2 = V.copy()
    V2[2, [5, 900]] = V2[2, [900, 5]]
    assert list(ctx.range_verify_batch(64, m, proofs, V2, verify_seed=SEED)) == [1, 1, 0]


@pytest.mark.parametrize("n_bits,m,b", [(64, 32, 9), (64, 1, 5), (8, 2, 4), (16, 4, 70), (32, 8, 3), (64, 16, 6), (8, 32, 5)])
def test_wavefront_transcript_equals_lane_transcript(gpu_ctx, n_bits, m, b):
    """k_rv_absorb_V (one wavefront per proof absorbs the m commitments, the default for m >= 256) must leave the STROBE
    state the lane-per-proof replay leaves: every honest proof verifies on either path (a single differing transcript byte
    would change y and fail it), and tampered proofs / swapped commitments get the same verdicts -- with and without
    cross-proof batching.  The shapes put the commitment stream at different offsets of the 166-byte STROBE block."""
    import os
    rng = np.random.default_rng(7 * n_bits + m)
    v = rng.integers(0, 2**n_bits if n_bits < 64 else 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    proofs = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    C, _ = gpu_ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    V = C.reshape(b, m, 32)

    def verdicts(p, vv):
        out = []
        try:
            for wave in ("1", "0"):
                for rlc in (None, "1"):
                    os.environ["DAPOL_VERIFY_WAVE_TRANSCRIPT"] = wave
                    if rlc:
                        os.environ["DAPOL_VERIFY_NO_RLC"] = rlc
                        os.environ.pop("DAPOL_VERIFY_RLC_MIN", None)
                    else:                       # the batched check even for a handful of proofs (by default they go one by one)
                        os.environ.pop("DAPOL_VERIFY_NO_RLC", None)
                        os.environ["DAPOL_VERIFY_RLC_MIN"] = "2"
                    out.append(list(gpu_ctx.range_verify_batch(n_bits, m, p, vv, verify_seed=SEED)))
                    # calls of up to a few thousand proofs replay each transcript on a wavefront (default); larger ones on a lane
                    os.environ["DAPOL_VERIFY_LANE_TRANSCRIPT"] = "1"
                    out.append(list(gpu_ctx.range_verify_batch(n_bits, m, p, vv, verify_seed=SEED)))
                    os.environ.pop("DAPOL_VERIFY_LANE_TRANSCRIPT", None)
        finally:
            os.environ.pop("DAPOL_VERIFY_WAVE_TRANSCRIPT", None)
            os.environ.pop("DAPOL_VERIFY_NO_RLC", None)
            os.environ.pop("DAPOL_VERIFY_LANE_TRANSCRIPT", None)
            os.environ.pop("DAPOL_VERIFY_RLC_MIN", None)
        out.append(list(gpu_ctx.range_verify_batch(n_bits, m, p, vv, verify_seed=SEED)))        # the default routing
        assert all(o == out[0] for o in out), out
        return out[0]

    assert verdicts(proofs, V) == [1] * b
    bad = proofs.copy()
    bad[b - 1, 70] ^= 2                                                         # T_1
    assert verdicts(bad, V) == [1] * (b - 1) + [0]
    V2 = V.copy()
    V2[0, m - 1, 3] ^= 1                                                        # the last commitment byte stream of proof 0
    assert verdicts(proofs, V2) == [0] + [1] * (b - 1)


@pytest.mark.parametrize("n_bits,m,b", [(16, 4, 1300), (8, 32, 300)])
def test_bucket_method_gives_the_per_proof_verdicts(gpu_ctx, n_bits, m, b):
    """Large batches sum the proofs' own points by the bucket (Pippenger) method instead of per-point tables; the switch is
    lowered here so that a moderate batch takes it.  Verdicts must equal those of the table path and of the proof-by-proof
    check: all honest, a tampered proof, a swapped commitment, an undecodable point."""
    import os
    rng = np.random.default_rng(11 * n_bits + m)
    v = rng.integers(0, 2**n_bits, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    proofs = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    C, _ = gpu_ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    V = C.reshape(b, m, 32)

    def verdicts(p, vv):
        out = []
        try:
            for pip, rlc in (("12288", None), ("0", None), ("0", "1")):
                os.environ["DAPOL_VERIFY_PIPPENGER_MIN"] = pip
                if rlc:
                    os.environ["DAPOL_VERIFY_NO_RLC"] = rlc
                else:
                    os.environ.pop("DAPOL_VERIFY_NO_RLC", None)
                out.append(list(gpu_ctx.range_verify_batch(n_bits, m, p, vv, verify_seed=SEED)))
        finally:
            os.environ.pop("DAPOL_VERIFY_PIPPENGER_MIN", None)
            os.environ.pop("DAPOL_VERIFY_NO_RLC", None)
        assert all(o == out[0] for o in out)
        return out[0]

    assert verdicts(proofs, V) == [1] * b
    bad = proofs.copy()
    bad[7, 40] ^= 1                                                             # S
    want = [1] * b
    want[7] = 0
    assert verdicts(bad, V) == want
    V2 = V.copy()
    V2[b - 1, 0], V2[b - 1, 1] = V[b - 1, 1].copy(), V[b - 1, 0].copy()        # two commitments of one proof swapped
    want = [1] * b
    want[b - 1] = 0
    assert verdicts(proofs, V2) == want
    mal = proofs.copy()
    mal[3, 0:32] = 0xFF                                                         # A is not a point
    want = [1] * b
    want[3] = 0
    assert verdicts(mal, V) == want


@pytest.mark.parametrize("n_bits,m,b", [(64, 32, 50), (64, 1, 200), (16, 4, 70), (8, 2, 3)])
def test_cross_proof_batching_gives_the_per_proof_verdicts(gpu_ctx, n_bits, m, b):
    """The verifier checks a batch through one random linear combination and falls back to the per-proof check when
    it fails: the verdict vector must be the per-proof one in every case (all good, some tampered, tampered
    commitments, malformed encodings), and DAPOL_VERIFY_NO_RLC=1 -- the reference's proof-by-proof check -- must agree."""
    import os
    rng = np.random.default_rng(n_bits * 100 + m)
    v = rng.integers(0, 2**n_bits if n_bits < 64 else 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    proofs = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    C, _ = gpu_ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    V = C.reshape(b, m, 32)

    def both(p, vv):
        os.environ.pop("DAPOL_VERIFY_NO_RLC", None)
        d = gpu_ctx.range_verify_batch(n_bits, m, p, vv, verify_seed=SEED)                       # default routing
        os.environ["DAPOL_VERIFY_RLC_MIN"] = "2"                                                 # batched whatever the count
        try:
            a = gpu_ctx.range_verify_batch(n_bits, m, p, vv, verify_seed=SEED)
        finally:
            os.environ.pop("DAPOL_VERIFY_RLC_MIN", None)
        os.environ["DAPOL_VERIFY_NO_RLC"] = "1"
        try:
            c = gpu_ctx.range_verify_batch(n_bits, m, p, vv, verify_seed=SEED)
        finally:
            os.environ.pop("DAPOL_VERIFY_NO_RLC", None)
        assert list(a) == list(c) == list(d)
        return list(a)

    assert both(proofs, V) == [1] * b
    bad = proofs.copy()
    want = [1] * b
    for k, off in ((0, 5), (b // 2, 40), (b - 1, proofs.shape[1] - 3)):       # A, a scalar, the final b
        bad[k, off] ^= 4
        want[k] = 0
    assert both(bad, V) == want
    V2 = V.copy()
    V2[1, 0] = V[2, 0]                                                          # a commitment that belongs to another proof
    want = [1] * b
    want[1] = 0
    assert both(proofs, V2) == want
    mal = proofs.copy()
    mal[2, 64:96] = 0xFF                                                        # T_1 is not a canonical point encoding
    mal[0, 32 * 7:32 * 8] = 0xFF                                                # t_x >= l
    want = [1] * b
    want[0] = want[2] = 0
    assert both(mal, V) == want


@pytest.mark.parametrize("n_bits,m", [(64, 32), (64, 4), (32, 32)])
def test_every_proving_strategy_gives_the_same_bytes(gpu_ctx, n_bits, m):
    """The hybrid inner-product argument is a re-arrangement, not a different proof: no tail at all, tails of length 32 /
    64 / 128, other lanes-per-list, chunk sizes and numbers of chunks in flight (DAPOL_CHUNK below the batch size puts several chunks on
    several streams) all produce byte-identical proofs."""
    import os
    b = 37
    rng = np.random.default_rng(n_bits + m)
    v = rng.integers(0, 2**n_bits if n_bits < 64 else 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = np.arange(b, dtype=np.uint64)
    base = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes()
    for env in ({"DAPOL_NO_TAIL": "1"}, {"DAPOL_TAIL_N": "32"}, {"DAPOL_TAIL_N": "128"}, {"DAPOL_TAIL_LPL": "4"}, {"DAPOL_TAIL_LPL": "1"},
                {"DAPOL_LPL": "32"}, {"DAPOL_LPL": "4"}, {"DAPOL_LPL": "2"}, {"DAPOL_LPL": "2", "DAPOL_MSM_OCC_CAP": "3"}, {"DAPOL_CHUNK": "64"}, {"DAPOL_TAIL_N": "256", "DAPOL_LPL": "16"},
                {"DAPOL_CHUNK": "8"}, {"DAPOL_CHUNK": "5", "DAPOL_STREAMS": "4"}, {"DAPOL_CHUNK": "16", "DAPOL_STREAMS": "1"},
                {"DAPOL_CHUNK": "36", "DAPOL_STREAMS": "3"}, {"DAPOL_NO_SPLIT": "1"}, {"DAPOL_NO_SPLIT": "1", "DAPOL_TAIL_LPL": "32"},
                # the small-call (latency) arrangements; coefficient tables off (DAPOL_NO_STAB: the s-vectors folded every round) and tails too long for them
                {"DAPOL_SMALL_TAIL": "1"}, {"DAPOL_SMALL_TAIL": "1", "DAPOL_NO_SPLIT_MAT": "1"}, {"DAPOL_SMALL_SPLIT": "16"}, {"DAPOL_SMALL_SPLIT": "2"},
                {"DAPOL_NO_SMALL_HI": "1"}, {"DAPOL_NO_PAIR": "1"}, {"DAPOL_NO_STAB": "1"}, {"DAPOL_NO_STAB": "1", "DAPOL_SMALL_TAIL": "1"},
                {"DAPOL_NO_STAB": "1", "DAPOL_NO_SPLIT": "1"}, {"DAPOL_NO_STAB": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS": "1"},
                {"DAPOL_NO_SPLIT": "1", "DAPOL_TAIL_N": "256"}, {"DAPOL_NO_SPLIT": "1", "DAPOL_TAIL_N": "128", "DAPOL_GS": "1"}, {"DAPOL_FS_SHAPE": "0"}, {"DAPOL_FS_SHAPE": "1"}, {"DAPOL_FS_SHAPE": "2", "DAPOL_NO_SPLIT": "1"}, {"DAPOL_NO_QUAD": "1"}, {"DAPOL_NO_QUAD": "1", "DAPOL_SMALL_TAIL": "1"}, {"DAPOL_SMALL_SPLIT": "4"},
                # four lanes per point for all 37 proofs (the default keeps it to calls of up to 8)
                {"DAPOL_NO_FS_PARTS": "1"}, {"DAPOL_NO_SIDE_A": "1"},
                # the generator-stationary sweep of large calls (kernels_range_gs.h), forced onto this small batch
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS_TILE": "4"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_FS_SHAPE": "0"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_FS_SHAPE": "1", "DAPOL_TAIL_LPL": "8"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_FS_SHAPE": "2", "DAPOL_TAIL_LPL": "4"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS_SLICES": "1", "DAPOL_GS_MAT_CPL": "1"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS_SLICES": "2", "DAPOL_GS_MAT_CPL": "3"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS_SLICES": "4", "DAPOL_GS_TILE": "12", "DAPOL_CHUNK": "7", "DAPOL_STREAMS": "2"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS_TILE": "64", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2", "DAPOL_MSM_SERIAL": "1"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_NO_TAIL": "1"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "5", "DAPOL_TAIL_N": "32"},
                # round 4's stream layouts (two chunks in flight on CU-masked / prioritised streams) and grid-strided, capped digit producers
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2", "DAPOL_STREAM_LAYOUT": "split_xcd"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2", "DAPOL_STREAM_LAYOUT": "split_cu"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "9", "DAPOL_STREAMS": "2", "DAPOL_STREAM_LAYOUT": "msm:32"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2", "DAPOL_STREAM_LAYOUT": "prio"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2", "DAPOL_STREAM_LAYOUT": "prio_lanes"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_PRODUCER_WAVES": "64"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2", "DAPOL_PRODUCER_WAVES": "100"},
                {"DAPOL_QUAD_MAX_WAVES": "1000000"}, {"DAPOL_QUAD_MAX_WAVES": "1000000", "DAPOL_SMALL_TAIL": "1"}, {"DAPOL_QUAD_MAX_WAVES": "1000000", "DAPOL_SMALL_SPLIT": "8"}):
        os.environ.update(env)
        try:
            got = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes()
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert got == base, env


def test_options_struct_and_env_knob_gate(hip_lib):
    """VERDICT r2 hygiene: the settings an embedder may want are fields of dapol_options on the context (readable and settable
    through the C ABI); the DAPOL_* environment variables are measurement knobs that the library reads ONLY when the process has
    opted in (DAPOL_ENV_KNOBS / dapol_env_knobs).  Same bytes under every setting."""
    import os
    L = hip_lib.lib()
    ctx = hip_lib.Context(0, 8, options=hip_lib.Options(window_bits=12, high_half_rows=-1, streams=1, tail_length=32))
    o = ctx.get_options()
    assert (o.window_bits, o.high_half_rows, o.streams, o.tail_length, o.struct_size) == (12, -1, 1, 32, __import__("ctypes").sizeof(hip_lib.Options))
    dflt = hip_lib.Context(0, 8).get_options()
    assert dflt.window_bits >= 12 and dflt.streams == 0
    b, m, n_bits = 40, 8, 32
    rng = np.random.default_rng(3)
    v = rng.integers(0, 2**32, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = np.arange(b, dtype=np.uint64)
    base = ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes()
    for opt in (hip_lib.Options(generator_stationary=1, small_call_max=1, gs_tile_rows=8), hip_lib.Options(tail_length=-1, chunk_proofs=7, streams=3),
                hip_lib.Options(generator_stationary=-1, small_call_max=8), hip_lib.Options()):
        ctx.set_options(opt)
        assert ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes() == base
        got = ctx.get_options()
        assert got.window_bits == 12 and got.gs_tile_rows == opt.gs_tile_rows and got.tail_length == opt.tail_length      # creation-time fields stay
    # the fields really select the schedule: the workload's statistics count the dominant kernel's launches per bracketed MSM
    idx, vv, rr = _rand_leaves(np.random.default_rng(9), 8, 40)
    w = hip_lib.Workload(ctx, 8, idx, vv, rr)
    w.build(SEED)
    ctx.set_options(hip_lib.Options(generator_stationary=1, small_call_max=1, gs_tile_rows=8, gs_slices=1))
    st_gs = w.prove(SEED, 64)
    ctx.set_options(hip_lib.Options(generator_stationary=1, small_call_max=1, gs_tile_rows=8))        # 40 proofs: sixteen slices of each list side by side
    st_gs8 = w.prove(SEED, 64)
    ctx.set_options(hip_lib.Options(generator_stationary=-1, small_call_max=1))
    st_ps = w.prove(SEED, 64)
    assert st_gs.msm_kernels == st_gs.msm_launches * 2 * (512 // 8) and st_ps.msm_kernels == st_ps.msm_launches      # N = 8 x 64 terms per list
    assert st_gs8.msm_kernels == st_gs8.msm_launches * 2 * (512 // 8 // 16)
    assert st_gs.checksum == st_ps.checksum == st_gs8.checksum
    ctx.set_options(hip_lib.Options())
    with pytest.raises(hip_lib.DapolError):
        ctx.set_options(hip_lib.Options(gs_tile_rows=6))
    with pytest.raises(hip_lib.DapolError):
        ctx.set_options(hip_lib.Options(gs_slices=3))
    with pytest.raises(hip_lib.DapolError):
        hip_lib.Context(0, 8, options=hip_lib.Options(window_bits=30))
    # the gate: with the knobs off an absurd variable is not even read; with them on it is (and rejected)
    old = L.dapol_env_knobs(0)
    os.environ["DAPOL_WBITS"] = "99"
    try:
        hip_lib.Context(0, 8).close() if hasattr(hip_lib.Context, "close") else hip_lib.Context(0, 8)
        L.dapol_env_knobs(1)
        with pytest.raises(hip_lib.DapolError):
            hip_lib.Context(0, 8)
    finally:
        os.environ.pop("DAPOL_WBITS", None)
        L.dapol_env_knobs(old)


def test_call_size_regimes_give_the_same_bytes(gpu_ctx, hip_lib):
    """The regimes of a prove call -- latency shapes (up to 1,023 proofs of this shape), the generator-stationary sweep (from 1,024: in
    16 / 8 / 4 slices below 65,536 proofs, with the Fiat-Shamir shapes and tail lanes of its size), and the proof-stationary throughput
    shapes they replace -- at sizes where one hands over to the next: 1,100, 5,000 and 8,192 proofs of 64 bits x 32 parties, each under
    its default and forced into the other regimes; one digest per size."""
    import hashlib
    import os
    n_bits, m = 64, 32
    for b, envs in ((1100, ({}, {"DAPOL_SMALL_MAX": "8191"}, {"DAPOL_GS": "0", "DAPOL_NO_SPLIT": "1"}, {"DAPOL_GS_SLICES": "1", "DAPOL_FS_SHAPE": "0"})),
                    (5000, ({}, {"DAPOL_SMALL_MAX": "8191"}, {"DAPOL_GS": "0", "DAPOL_NO_SPLIT": "1"}, {"DAPOL_GS_SLICES": "2", "DAPOL_FS_SHAPE": "0", "DAPOL_TAIL_LPL": "2"})),
                    (8192, ({}, {"DAPOL_GS": "0"}, {"DAPOL_SMALL_MAX": "8192"}))):
        rng = np.random.default_rng(b)
        v = rng.integers(0, 2**63, size=(b, m), dtype=np.uint64)
        r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
        r[:, :, 31] &= 0x0F
        sid = np.arange(b, dtype=np.uint64)
        digests = set()
        for env in envs:
            os.environ.update(env)
            try:
                digests.add(hashlib.sha256(gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes()).hexdigest())
            finally:
                for k in env:
                    os.environ.pop(k, None)
        assert len(digests) == 1, b


# ------------------------------------------------------------------------------- soundness of the cross-proof batch check
def _forged_cancelling_pair(pyref, seed):
    """Two 8-bit, one-party proofs that are each INVALID but whose residuals cancel in a random linear combination whose
    weights depend only on (verify seed, position in the batch) -- the round-1 derivation.  Proof 0 is honest except that
    its commitment holds 300 (out of range; the bits prove 44): residual c_0 z_0^2 * 256 * B.  Proof 1 is honest except that
    it carries A + kappa*B: residual kappa * B.  kappa = -rho_0 c_0 z_0^2 256 / rho_1."""
    L = pyref.L
    w = lambda dom, b: pyref.scalar_from_wide(pyref.seed_wide(seed, dom, b, 0)) or 1
    rho0, rho1, c0 = w(5, 0), w(5, 1), w(3, 0)
    i0, i1 = {}, {}
    p0 = pyref.range_prove([44], [0x1234567], 8, pyref.Tape(seed=seed, stream_id=1), commit_values=[300], info=i0)
    kappa = (-rho0 * c0 * i0["z"] * i0["z"] * 256 * pyref.inv(rho1)) % L
    p1 = pyref.range_prove([7], [0x7654321], 8, pyref.Tape(seed=seed, stream_id=2), A_offset=kappa, info=i1)
    assert not pyref.range_verify(p0, i0["V"], 8) and not pyref.range_verify(p1, i1["V"], 8)
    proofs = np.frombuffer(p0 + p1, np.uint8).reshape(2, -1).copy()
    V = np.frombuffer(i0["V"][0] + i1["V"][0], np.uint8).reshape(2, 1, 32).copy()
    return proofs, V


@pytest.mark.gpu
def test_cancelling_pair_is_rejected(hip_lib, pyref):
    """batch weights that ignore the proof bytes let a crafted pair of invalid proofs pass the combined check
    (the round-1 library returns [1, 1] for this pair: profiles/archive/r02_forged_pair_old_vs_new.txt).  The weights are now derived
    from the digest of every proof and commitment of the batch, so the pair is rejected under the seed it was crafted for,
    under any other seed, under the library's own OS-random seed, and inside a larger batch of honest proofs."""
    import os
    ctx = hip_lib.Context(0, 1)
    proofs, V = _forged_cancelling_pair(pyref, SEED)
    for vs in (SEED, bytes(32), None):
        assert ctx.range_verify_batch(8, 1, proofs, V, verify_seed=vs).tolist() == [0, 0]         # default: a pair is checked one by one
        os.environ["DAPOL_VERIFY_RLC_MIN"] = "2"                                                  # the batched check, which the pair attacks
        try:
            assert ctx.range_verify_batch(8, 1, proofs, V, verify_seed=vs).tolist() == [0, 0]
        finally:
            os.environ.pop("DAPOL_VERIFY_RLC_MIN", None)
    rng = np.random.default_rng(11)
    b = 70
    v = rng.integers(0, 256, size=(b, 1), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, 1, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    honest = ctx.range_prove_batch(8, 1, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    C, _ = ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    allp = np.concatenate([proofs, honest])
    allV = np.concatenate([V, C.reshape(b, 1, 32)])
    os.environ["DAPOL_VERIFY_RLC_MIN"] = "2"
    try:
        for vs in (SEED, None):
            ok = ctx.range_verify_batch(8, 1, allp, allV, verify_seed=vs)
            assert ok[:2].tolist() == [0, 0] and ok[2:].all()
    finally:
        os.environ.pop("DAPOL_VERIFY_RLC_MIN", None)


@pytest.mark.gpu
def test_nonces_are_bound_to_the_statement(gpu_ctx):
    """the same seed and stream id with a different witness, a different party count or a different
    first slot must not reuse a single nonce: A and S (bytes 0..64, functions of a_blinding, s_blinding, s_L, s_R only
    through the nonce stream and the bits) and T1, T2 change; the same statement twice gives the same bytes."""
    n, m = 16, 2
    v = np.array([[5, 9]], np.uint64)
    r = np.arange(64, dtype=np.uint8).reshape(1, 2, 32) & 0x7F
    base = gpu_ctx.range_prove_batch(n, m, v, r, nonce_seed=SEED, stream_id=[77])
    again = gpu_ctx.range_prove_batch(n, m, v, r, nonce_seed=SEED, stream_id=[77])
    assert base.tobytes() == again.tobytes()
    r2 = r.copy()
    r2[0, 1, 0] ^= 1                                              # another blinding of party 1: same bits, another statement
    other = gpu_ctx.range_prove_batch(n, m, v, r2, nonce_seed=SEED, stream_id=[77])
    assert other[0, 32:64].tobytes() != base[0, 32:64].tobytes()         # S = s_bl*B~ + <s_L,G> + <s_R,H>: nonces only
    shifted = gpu_ctx.range_prove_batch(n, m, v, r, nonce_seed=SEED, stream_id=[77], slot_base=1000)
    assert shifted[0, 32:64].tobytes() != base[0, 32:64].tobytes()
    # duplicate stream ids inside one call with different witnesses: no shared S either
    v2 = np.array([[5, 9], [5, 10]], np.uint64)
    rr = np.concatenate([r, r])
    two = gpu_ctx.range_prove_batch(n, m, v2, rr, nonce_seed=SEED, stream_id=[77, 77])
    assert two[0].tobytes() == base[0].tobytes() and two[1, 32:64].tobytes() != two[0, 32:64].tobytes()


# --------------------------------------------------------------------------------------- f2: DapolProof wire format
@pytest.mark.gpu
def test_dapol_proof_serialization_round_trip(gpu_ctx, hip_lib, pyref):
    """src/proof/tests.rs:6-35 (height 8, 20 leaves, a batch of 10, Splitting, aggregation factor 1): generate_proof_batch ->
    serialize -> deserialize -> verify_batch; and src/tests.rs:50-93's single-leaf flavour (both policies).  The wire bytes
    equal the Python restatement's; decoding errors mirror DecodingError (truncation -> 6, a sibling commitment that does not
    decompress -> 7, validated on the GPU)."""
    height, n_bits = 8, 8
    rng = np.random.default_rng(20)
    idx, v, r = _rand_leaves(rng, height, 20, vmax=12)          # every sibling value (a subtree sum) stays below 2^8
    tree = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    rC, rH, _, _ = tree.root()
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    batch = idx[:10]
    level, index, sC, sH, blob = tree.prove_batch(batch, hip_lib.POLICY_SPLITTING, 1, n_bits, SEED)
    S = len(level)
    wire = hip_lib.proof_serialize(height, batch, sC, sH, hip_lib.POLICY_SPLITTING, 1, n_bits, blob)
    ps1 = hip_lib.lib().dapol_range_proof_size(n_bits, 1)
    agg_p, ind_p = [blob[:ps1]], [blob[ps1 * (1 + i):ps1 * (2 + i)] for i in range(S - 1)]
    assert wire == pyref.dapol_proof_serialize("splitting", agg_p, ind_p, height, [int(x) for x in batch],
                                               [(sC[i].tobytes(), sH[i].tobytes()) for i in range(S)])
    d = gpu_ctx.proof_deserialize(hip_lib.POLICY_SPLITTING, n_bits, wire + b"tail")
    assert d["consumed"] == len(wire) and d["height"] == height and d["aggregation_factor"] == 1
    assert d["leaf_idx"].tolist() == batch.tolist() and d["sib_C"].tobytes() == sC.tobytes() and d["sib_H"].tobytes() == sH.tobytes()
    assert d["range_blob"] == blob
    assert gpu_ctx.verify_batch(d["height"], d["leaf_idx"], lC[:10], lH[:10], d["sib_C"], d["sib_H"], rC, rH, hip_lib.POLICY_SPLITTING,
                                d["aggregation_factor"], n_bits, d["range_blob"])
    with pytest.raises(hip_lib.DapolError) as e:
        gpu_ctx.proof_deserialize(hip_lib.POLICY_SPLITTING, n_bits, wire[:-1])
    assert e.value.code == 6
    bad = bytearray(wire)
    off = len(wire) - 64 * S + 64 * 2                                     # commitment of sibling 2 := an s with no point behind it
    bad[off:off + 32] = (2).to_bytes(32, "little")
    assert pyref.decompress(bytes(bad[off:off + 32])) is None
    with pytest.raises(hip_lib.DapolError) as e:
        gpu_ctx.proof_deserialize(hip_lib.POLICY_SPLITTING, n_bits, bytes(bad))
    assert e.value.code == 7
    # single-leaf proofs, both policies, agg < height: serialize -> deserialize -> DapolProof::verify
    for pol, name, agg in ((hip_lib.POLICY_PADDING, "padding", 5), (hip_lib.POLICY_SPLITTING, "splitting", 6)):
        pC, pH, out = tree.prove_entities(idx[:4], pol, agg, n_bits, SEED)
        for e_ in range(4):
            w = hip_lib.proof_serialize(height, [idx[e_]], pC[e_], pH[e_], pol, agg, n_bits, out[e_].tobytes())
            d = gpu_ctx.proof_deserialize(pol, n_bits, w)
            assert (d["height"], d["aggregation_factor"], d["leaf_idx"].tolist()) == (height, agg, [int(idx[e_])])
            ok = gpu_ctx.verify_entities(height, d["leaf_idx"], lC[e_:e_ + 1], lH[e_:e_ + 1], d["sib_C"][None], d["sib_H"][None], rC, rH, pol,
                                         d["aggregation_factor"], n_bits, np.frombuffer(d["range_blob"], np.uint8)[None])
            assert ok.tolist() == [1]
    # DapolProofNode on its own (proof/node.rs:74-102)
    nodes = np.concatenate([sC, sH], axis=1).tobytes()
    C2, H2 = gpu_ctx.proof_nodes_deserialize(nodes, S)
    assert C2.tobytes() == sC.tobytes() and H2.tobytes() == sH.tobytes()
    with pytest.raises(hip_lib.DapolError) as e:
        gpu_ctx.proof_nodes_deserialize(nodes[:-1], S)
    assert e.value.code == 6


@pytest.mark.gpu
def test_structurally_inconsistent_wires_are_rejected(gpu_ctx, hip_lib):
    """a wire whose pieces do not fit each other must never reach the verifier's kernels, which size their reads
    from (height, policy, aggregation factor) alone.  (a) a single-leaf wire that claims more siblings than its aggregated proof
    has parties for -- h = S = 32 with the 480-byte proof of ONE 8-bit party where the padding policy needs the 32-party one;
    (b) a single-leaf wire with fewer siblings than levels; (c) an aggregated proof of the wrong size under the splitting policy.
    All three: ValueDecodingError (7) from dapol_proof_deserialize.  And the verifier entry points themselves, handed arrays of
    the wrong length (as a binding that skipped deserialize might), answer 'invalid' without reading past them."""
    n_bits = 8
    rng = np.random.default_rng(8)
    idx, v, r = _rand_leaves(rng, 8, 20, vmax=12)
    tree = hip_lib.Tree(gpu_ctx, 8, idx, v, r, SEED)
    rC, rH, _, _ = tree.root()
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    pC, pH, out = tree.prove_entities(idx[:1], hip_lib.POLICY_PADDING, 8, n_bits, SEED)
    good = hip_lib.proof_serialize(8, idx[:1], pC[0], pH[0], hip_lib.POLICY_PADDING, 8, n_bits, out[0].tobytes())
    assert gpu_ctx.proof_deserialize(hip_lib.POLICY_PADDING, n_bits, good)["height"] == 8
    one = gpu_ctx.range_prove_batch(n_bits, 1, np.array([[5]], np.uint64), r[:1].reshape(1, 1, 32), nonce_seed=SEED,
                                    stream_id=np.array([1], np.uint64))[0].tobytes()
    assert len(one) == 480
    C32, _ = gpu_ctx.commit_hash_batch(v, r)
    be = lambda x, n: int(x).to_bytes(n, "big")

    def wire(agg_proofs, n_ind, k, S, h, policy):
        w = b""
        if policy == hip_lib.POLICY_SPLITTING:
            w += be(len(agg_proofs), 2)
        for p in agg_proofs:
            w += be(len(p), 8) + p
        w += be(n_ind, 8) + b"".join(one for _ in range(n_ind))
        w += be(k, 8) + be(S, 8) + be(h, 2) + b"".join(bytes((h + 7) // 8) for _ in range(k))
        return w + b"".join(C32[i % 20].tobytes() + lH[i % 20].tobytes() for i in range(S))

    for bad, pol in ((wire([one], 0, 1, 32, 32, hip_lib.POLICY_PADDING), hip_lib.POLICY_PADDING),       # (a)
                     (wire([one], 3, 1, 4, 8, hip_lib.POLICY_PADDING), hip_lib.POLICY_PADDING),         # (b) S = 4 < h = 8
                     (wire([one, one], 0, 1, 3, 3, hip_lib.POLICY_SPLITTING), hip_lib.POLICY_SPLITTING)):   # (c) 3 = 2 + 1: the first must hold 2 parties
        with pytest.raises(hip_lib.DapolError) as e:
            gpu_ctx.proof_deserialize(pol, n_bits, bad)
        assert e.value.code == 7, e.value
    # the verifier with arrays that do not fit its arguments: invalid, not an over-read
    ok = gpu_ctx.verify_entities(32, idx[:1], lC[:1], lH[:1], np.tile(C32[:1], (32, 1))[None], np.tile(lH[:1], (32, 1))[None], rC, rH,
                                 hip_lib.POLICY_PADDING, 32, n_bits, np.frombuffer(one, np.uint8)[None])
    assert ok.tolist() == [0]
    ok = gpu_ctx.verify_entities(8, idx[:1], lC[:1], lH[:1], pC[:, :5], pH[:, :5], rC, rH, hip_lib.POLICY_PADDING, 8, n_bits, out)
    assert ok.tolist() == [0]
    assert gpu_ctx.verify_entities(8, idx[:1], lC[:1], lH[:1], pC, pH, rC, rH, hip_lib.POLICY_PADDING, 8, n_bits, out).tolist() == [1]
    level, index, sC, sH, blob = tree.prove_batch(idx[:4], hip_lib.POLICY_PADDING, 2, n_bits, SEED)
    assert gpu_ctx.verify_batch(8, idx[:4], lC[:4], lH[:4], sC, sH, rC, rH, hip_lib.POLICY_PADDING, 2, n_bits, blob)
    assert not gpu_ctx.verify_batch(8, idx[:4], lC[:4], lH[:4], sC, sH, rC, rH, hip_lib.POLICY_PADDING, 2, n_bits, blob[:-32])


@pytest.mark.gpu
def test_sibling_order_switch(gpu_ctx, hip_lib, ref):
    """dapol_wire_config.siblings_leaf_first (smtree's sibling order is not pinned by the reference repository): with the
    switch on, paths come leaf side first, the range proof's parties follow that order (bytes = the C oracle's proof over the
    reversed parties), single-leaf and batched proofs still verify; with the switch back, the default bytes return."""
    height, n_bits = 6, 8
    rng = np.random.default_rng(6)
    idx, v, r = _rand_leaves(rng, height, 9, vmax=25)            # subtree sums stay below 2^8
    tree = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    rC, rH, _, _ = tree.root()
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    C0, H0, v0, r0 = tree.paths(idx)
    _, _, out0 = tree.prove_entities(idx, hip_lib.POLICY_PADDING, height, n_bits, SEED)
    old = hip_lib.wire_config_set(siblings_leaf_first=1)
    try:
        C1, H1, v1, r1 = tree.paths(idx)
        assert C1.tobytes() == C0[:, ::-1].tobytes() and v1.tolist() == v0[:, ::-1].tolist()
        pC, pH, out1 = tree.prove_entities(idx, hip_lib.POLICY_PADDING, height, n_bits, SEED)
        for e in range(len(idx)):
            assert out1[e].tobytes() == _oracle_padding_proof(ref, n_bits, height, v1[e], r1[e], idx[e])
        assert gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, rC, rH, hip_lib.POLICY_PADDING, height, n_bits, out1).all()
        level, index, sC, sH, blob = tree.prove_batch(idx[:4], hip_lib.POLICY_SPLITTING, 2, n_bits, SEED)
        assert list(level) == sorted(level)                                   # leaf level first
        assert gpu_ctx.verify_batch(height, idx[:4], lC[:4], lH[:4], sC, sH, rC, rH, hip_lib.POLICY_SPLITTING, 2, n_bits, blob)
    finally:
        hip_lib.wire_config_restore(old)
    _, _, out2 = tree.prove_entities(idx, hip_lib.POLICY_PADDING, height, n_bits, SEED)
    assert out2.tobytes() == out0.tobytes() and out1.tobytes() != out0.tobytes()


# ------------------------------------------------------------------------- BASELINE configs as workloads
@pytest.mark.gpu
def test_config2_workload_strided_layout_vs_oracle(gpu_ctx, hip_lib, ref):
    """configs[2] shape through the bench path (Workload on the strided layout of benches/dapol.rs:160-175, height 32, 64-bit
    proofs, padding policy, aggregation_factor = height) at 2^13 entities: root vs the C oracle's tree, 8 sampled proofs byte
    for byte, their inclusion proofs through DapolProof::verify."""
    import bench
    height, n_bits, n = 32, 64, 1 << 13
    idx, v, r = bench.synth_inputs(n, height, 0, n)
    w = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    root, st = w.build(bench.PAD_SEED)
    st = w.prove(bench.NONCE_SEED, n_bits, stats=st)
    t = _ref_tree(ref, height, idx, v, r, seed=bench.PAD_SEED)
    assert _ref_root(ref, t) == root
    assert root[2] == int(v.sum())
    ref.ref_tree_free(t)
    sample = idx[:: n // 8][:8]
    sv, sr, sC, sH = w.paths(sample, with_nodes=True)
    ps = hip_lib.lib().dapol_range_proof_size(n_bits, 32)
    got = np.stack([w.proofs(int(p), 1, ps)[0] for p in np.searchsorted(idx, sample)])
    for k in range(8):
        assert got[k].tobytes() == _oracle_padding_proof(ref, n_bits, height, sv[k], sr[k], sample[k], seed=bench.NONCE_SEED)
    pos = np.searchsorted(idx, sample)
    lC, lH = gpu_ctx.commit_hash_batch(v[pos], r[pos])
    assert gpu_ctx.verify_entities(height, sample, lC, lH, sC, sH, root[0], root[1], hip_lib.POLICY_PADDING, height, n_bits, got).all()
    assert st.proofs == n and st.proof_bytes == n * 992 and st.msm_launches > 0


@pytest.mark.gpu
def test_config2_full_size_properties(gpu_ctx, hip_lib):
    """configs[2] at FULL size (2^20 entities, height 32, 64-bit proofs), one pass of the bench workload, through
    size-independent properties: root value = sum of the liabilities, node counts of the strided layout (13.63 M
    merges -> 14,680,063 real + 12,582,912 padding nodes), a second build gives the same root (determinism), and 2,048
    sampled inclusion proofs verify against that root on the GPU."""
    import bench
    height, n_bits, n = 32, 64, 1 << 20
    idx, v, r = bench.synth_inputs(n, height, 0, n)
    w = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    root, st = w.build(bench.PAD_SEED)
    assert root[2] == int(v.sum())
    root2, _ = w.build(bench.PAD_SEED)
    assert root2 == root
    st = w.prove(bench.NONCE_SEED, n_bits, stats=st)
    assert st.proofs == n and st.proof_bytes == n * 992
    sample = idx[:: n // 2048][:2048]
    pos = np.searchsorted(idx, sample)
    _, _, sC, sH = w.paths(sample, with_nodes=True)
    got = np.stack([w.proofs(int(p), 1, 992)[0] for p in pos])
    lC, lH = gpu_ctx.commit_hash_batch(v[pos], r[pos])
    assert gpu_ctx.verify_entities(height, sample, lC, lH, sC, sH, root[0], root[1], hip_lib.POLICY_PADDING, height, n_bits, got).all()
    tampered = got.copy()
    tampered[7, 40] ^= 1
    ok = gpu_ctx.verify_entities(height, sample, lC, lH, sC, sH, root[0], root[1], hip_lib.POLICY_PADDING, height, n_bits, tampered)
    assert ok[7] == 0 and ok.sum() == 2047


@pytest.mark.gpu
def test_config3_eight_shards_height32_64bit(gpu_ctx, hip_lib):
    """configs[3] shape: 8 top-level shards, height 32, 64-bit proofs, 2^13 entities (strided layout): root, paths and proof
    bytes of every shard equal the unsharded build's."""
    import bench
    from dapol_amd import sharded
    height, n_bits, n, sb = 32, 64, 1 << 13, 3
    idx, v, r = bench.synth_inputs(n, height, 0, n)
    full = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    froot, fst = full.build(bench.PAD_SEED)
    full.prove(bench.NONCE_SEED, n_bits, stats=fst)
    fproofs = full.proofs(0, n, 992)
    shards, recs = [], []
    for s in range(8):
        sel = (idx >> np.uint64(height - sb)) == s
        w = hip_lib.Workload(gpu_ctx, height, idx[sel], v[sel], r[sel], shard_bits=sb)
        root, st = w.build(bench.PAD_SEED)
        shards.append((sel, w, st))
        recs.append(sharded.pack_record(root))
    records = sharded.unpack_records(np.stack(recs), 8)
    csum = 0
    for s, (sel, w, st) in enumerate(shards):
        root, upper = sharded.top_levels(gpu_ctx, records, s)
        assert root == froot
        st = w.prove(bench.NONCE_SEED, n_bits, upper=upper, stats=st)
        assert w.proofs(0, int(sel.sum()), 992).tobytes() == fproofs[sel].tobytes()
        probe = idx[sel][[0, -1]]
        a = w.paths(probe, upper=upper, with_nodes=True)
        b = full.paths(probe, with_nodes=True)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))


@pytest.mark.gpu
def test_config3_full_size_2e22_entities_eight_shards(gpu_ctx, hip_lib):
    """BASELINE configs[3] at its FULL size -- 2^22 entities, height 32, 64-bit proofs, 8 top-level shards -- with the 8 shards run
    one after the other on the one GPU a test box has (the multi-GPU run differs only in where each shard lives and in the
    transport of the 8 x 104-byte root records).  Properties: the global root rebuilt from the 8 subtree roots equals the
    unsharded build's root and carries the sum of the liabilities; the per-shard proof checksums add up to the unsharded run's
    checksum (2^19 entities per shard is a multiple of the checksum's period, so it is additive here); 512 sampled inclusion
    proofs out of the shards verify against the global root.  No CPU leg at this size."""
    import bench
    from dapol_amd import sharded
    height, n_bits, n, sb = 32, 64, 1 << 22, 3
    G, per = 1 << sb, n >> sb
    idx, v, r = bench.synth_inputs(n, height, 0, n)
    full = hip_lib.Workload(gpu_ctx, height, idx, v, r)
    froot, fst = full.build(bench.PAD_SEED)
    assert froot[2] == int(v.sum())
    fst = full.prove(bench.NONCE_SEED, n_bits, stats=fst)
    assert fst.proofs == n
    full_checksum = int(fst.checksum)
    del full
    shards, recs = [], []
    for g in range(G):
        sl = slice(g * per, (g + 1) * per)
        assert int(idx[sl][0] >> (height - sb)) == g and int(idx[sl][-1] >> (height - sb)) == g
        w = hip_lib.Workload(gpu_ctx, height, idx[sl], v[sl], r[sl], shard_bits=sb)
        root, st = w.build(bench.PAD_SEED)
        shards.append((w, st))
        recs.append(sharded.pack_record(root))
    records = sharded.unpack_records(np.stack(recs), G)
    total, ok_all, checked = 0, 0, 0
    rng = np.random.default_rng(22)
    for g, (w, st) in enumerate(shards):
        groot, upper = sharded.top_levels(gpu_ctx, records, g)
        assert groot == froot                                         # every shard rebuilds the same global root
        st = w.prove(bench.NONCE_SEED, n_bits, upper=upper, stats=st)
        total = (total + int(st.checksum)) & 0xFFFFFFFFFFFFFFFF
        sl = slice(g * per, (g + 1) * per)
        pos = np.sort(rng.choice(per, size=64, replace=False))
        sample = idx[sl][pos]
        _, _, sC, sH = w.paths(sample, upper=upper, with_nodes=True)
        got = np.stack([w.proofs(int(p), 1, 992)[0] for p in pos])
        lC, lH = gpu_ctx.commit_hash_batch(v[sl][pos], r[sl][pos])
        ok = gpu_ctx.verify_entities(height, sample, lC, lH, sC, sH, froot[0], froot[1], hip_lib.POLICY_PADDING, height, n_bits, got)
        ok_all += int(ok.sum())
        checked += len(ok)
        w.close()
    assert total == full_checksum
    assert ok_all == checked == 512


@pytest.mark.gpu
def test_config4_1024_proofs_of_1024_parties_with_oracle_verdict(hip_lib, ref):
    """configs[4] at its full shape on one GPU: 1,024 aggregated proofs x 1,024 commitments (2^20 entities), verification
    only.  All verify; one flipped byte is found; and the C oracle's verify_multiple gives the same verdicts on the first
    proof, honest and tampered (ref_range_verify, single thread)."""
    ctx = hip_lib.Context(0, 1024)
    b, m = 1024, 1024
    rng = np.random.default_rng(4)
    v = rng.integers(0, 2**32, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    proofs = ctx.range_prove_batch(64, m, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    C, _ = ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    V = C.reshape(b, m, 32)
    assert ctx.range_verify_batch(64, m, proofs, V).all()
    bad = proofs.copy()
    bad[0, 700] ^= 4
    bad[513, 64] ^= 1
    ok = ctx.range_verify_batch(64, m, bad, V)
    assert ok[0] == 0 and ok[513] == 0 and ok.sum() == b - 2
    # Round 5: the commitments of such a batch travel in four column blocks while the transcript replay runs in four phases behind
    # them (host_verify.inc: VArrival; k_rv_absorb_V(j0, j1)).  A commitment changed in any of the blocks -- first and last byte of a
    # block included -- turns exactly its proof's verdict, and the one-copy path (DAPOL_VERIFY_NO_PIPELINE) gives the same vector.
    import os
    Vbad = V.copy()
    hits = {3: (0, 0), 100: (255, 31), 400: (256, 0), 777: (600, 17), 1000: (1023, 31), 1023: (768, 5)}
    for pi, (j, byte) in hits.items():
        Vbad[pi, j, byte] ^= 0x10
    ok_pipe = ctx.range_verify_batch(64, m, proofs, Vbad, verify_seed=SEED)
    assert sorted(np.nonzero(ok_pipe == 0)[0].tolist()) == sorted(hits) and ok_pipe.sum() == b - len(hits)
    os.environ["DAPOL_VERIFY_NO_PIPELINE"] = "1"
    try:
        ok_one = ctx.range_verify_batch(64, m, proofs, Vbad, verify_seed=SEED)
        assert ctx.range_verify_batch(64, m, proofs, V).all()
    finally:
        del os.environ["DAPOL_VERIFY_NO_PIPELINE"]
    assert ok_one.tolist() == ok_pipe.tolist()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    c32 = bytes(range(1, 33))
    V0 = np.ascontiguousarray(V[0])
    assert ref.ref_range_verify(64, m, proofs[0].tobytes(), ctypes.c_size_t(proofs.shape[1]), p(V0), c32, 0) == 1
    assert ref.ref_range_verify(64, m, bad[0].tobytes(), ctypes.c_size_t(proofs.shape[1]), p(V0), c32, 0) == 0


# ----------------------------------------------------------------------------------- RCCL exchange inside the library
@pytest.mark.gpu
def test_rccl_exchange_one_rank_and_top_levels(gpu_ctx, hip_lib):
    """dapol_comm_* / dapol_shard_exchange with the one rank a 1-GPU box has: ncclCommInitRank, ncclAllGather and
    ncclAllReduce really run (librccl.so); with G = 1 the global root is the subtree root and there are no upper siblings.
    The merge half (dapol_shard_top_levels) is checked for 8 shards against the Python-side merge of the same records and
    against the unsharded tree."""
    from dapol_amd import sharded
    height, sb = 9, 3
    rng = np.random.default_rng(8)
    idx, v, r = _rand_leaves(rng, height, 40, vmax=1000)
    full = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    comm = hip_lib.Comm(gpu_ctx, hip_lib.comm_unique_id(), 0, 1)
    root, upper, rec = comm.exchange(full.root(), with_records=True)
    assert root == full.root() and len(upper[2]) == 0
    assert rec.tobytes() == sharded.pack_record(full.root()).tobytes()
    assert comm.allreduce([5, 2**64 - 1]).tolist() == [5, 2**64 - 1]
    assert comm.allreduce([7], hip_lib.REDUCE_MIN).tolist() == [7]
    comm.close()
    # ... and created with a deadline, which its collectives then keep (completion polled on an event, the communicator's
    # asynchronous state watched meanwhile; on expiry the communicator is aborted and the call returns DAPOL_ERR_COMM)
    comm = hip_lib.Comm(gpu_ctx, hip_lib.comm_unique_id(), 0, 1, timeout_s=30.0)
    assert comm.count() == 1
    for _ in range(3):
        root, upper = comm.exchange(full.root())
        assert root == full.root() and len(upper[2]) == 0
        assert comm.allreduce([11, 2**63], hip_lib.REDUCE_SUM).tolist() == [11, 2**63]
    comm.close()
    recs = []
    for s in range(8):
        sel = (idx >> np.uint64(height - sb)) == s
        if sel.any():
            recs.append(sharded.pack_record(hip_lib.Tree(gpu_ctx, height, idx[sel], v[sel], r[sel], SEED, shard_bits=sb).root()))
        else:
            C, H, rr = gpu_ctx.padding_nodes(SEED, [height - sb], [s])
            recs.append(sharded.pack_record((C[0].tobytes(), H[0].tobytes(), 0, rr[0].tobytes())))
    records = np.stack(recs)
    for s in range(8):
        root, upper = hip_lib.shard_top_levels(gpu_ctx, records, s)
        proot, pupper = sharded.top_levels(gpu_ctx, sharded.unpack_records(records, 8), s)
        assert root == proot == full.root()
        assert all(np.asarray(a).tobytes() == np.asarray(b).tobytes() for a, b in zip(upper, pupper))
    with pytest.raises(hip_lib.DapolError) as e:
        hip_lib.Comm(gpu_ctx, bytes(128), 0, 3)
    assert e.value.code == 8


@pytest.mark.gpu
@pytest.mark.parametrize("policy,agg", [(0, None), (1, 3), (0, 0)])
def test_batch_proof_on_a_sharded_tree(gpu_ctx, hip_lib, policy, agg):
    """generate_proof_batch (src/dapol/mod.rs:172-190) over a tree split into 4 top-level shards (one of them empty): sibling
    records gathered per shard (dapol_tree_node_records), top nodes from the exchanged root records
    (dapol_shard_top_node_records), proof by dapol_prove_batch_records -- siblings and range-proof bytes equal
    dapol_prove_batch on the unsharded tree, and the batch verifies."""
    from dapol_amd import sharded
    height, sb, n_bits = 8, 2, 16
    rng = np.random.default_rng(42)
    local = 1 << (height - sb)
    idx = np.sort(np.concatenate([rng.choice(local, size=5, replace=False).astype(np.uint64) + np.uint64(s * local) for s in (0, 1, 3)]))
    v = rng.integers(0, 1000, size=len(idx), dtype=np.uint64)
    r = rng.integers(0, 256, size=(len(idx), 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    full = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    leaves = idx[[0, 3, 6, 7, 12]]                                   # leaves of shards 0, 1 and 3
    flevel, findex, fC, fH, fblob = full.prove_batch(leaves, policy, len(hip_lib.batch_siblings(height, leaves)[0]) if agg is None else agg, n_bits, SEED)
    trees, recs = {}, []
    for s in range(4):
        sel = (idx >> np.uint64(height - sb)) == s
        if sel.any():
            trees[s] = hip_lib.Tree(gpu_ctx, height, idx[sel], v[sel], r[sel], SEED, shard_bits=sb)
            recs.append(sharded.pack_record(trees[s].root()))
        else:
            C, H, rr = gpu_ctx.padding_nodes(SEED, [height - sb], [s])
            recs.append(sharded.pack_record((C[0].tobytes(), H[0].tobytes(), 0, rr[0].tobytes())))
    records = np.stack(recs)
    lookups = {s: (lambda lv, ix, t=t: hip_lib.tree_node_records(t.h, lv, ix)) for s, t in trees.items()}
    level, index, C, H, sv, sr = sharded.assemble_batch_records(height, sb, leaves, lookups, records, gpu_ctx)
    assert level.tolist() == flevel.tolist() and index.tolist() == findex.tolist()
    assert C.tobytes() == fC.tobytes() and H.tobytes() == fH.tobytes()
    a = len(level) if agg is None else agg
    blob = hip_lib.prove_batch_records(gpu_ctx, leaves, C, sv, sr, policy, a, n_bits, SEED)
    assert blob == fblob
    pos = np.searchsorted(idx, leaves)
    lC, lH = gpu_ctx.commit_hash_batch(v[pos], r[pos])
    rC, rH, _, _ = full.root()
    assert gpu_ctx.verify_batch(height, leaves, lC, lH, C, H, rC, rH, policy, a, n_bits, blob)
    # a position no shard tree holds is reported, not invented
    _, _, _, _, found = hip_lib.tree_node_records(trees[0].h, [0, 0, height - sb], [int(idx[0]), int(idx[-1]), 0])
    assert found.tolist() == [1, 0, 1]


@pytest.mark.gpu
def test_high_half_rows_do_not_change_bytes(hip_lib, ref):
    """TableView::hi_split: the materialisation of the folded generators may look every term up in two rows (P and 2^(W*8) P)
    and walk half the window steps; with and without those rows, and at two window widths, the proofs are the oracle's."""
    import os
    n_bits, m, b = 64, 32, 6
    rng = np.random.default_rng(77)
    v = rng.integers(0, 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = np.arange(b, dtype=np.uint64) + 1000
    outs = []
    for env in ({"DAPOL_TABLE_HI": "0", "DAPOL_TABLE_GB": "3"}, {"DAPOL_TABLE_HI": "1", "DAPOL_TABLE_GB": "3"}, {"DAPOL_TABLE_HI": "1", "DAPOL_WBITS": "10"}):
        os.environ.update(env)
        try:
            ctx = hip_lib.Context(0, m)
            outs.append(ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes())
            ctx.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert outs[0] == outs[1] == outs[2]
    ps = len(outs[0]) // b
    out = ctypes.create_string_buffer(ps)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert ref.ref_range_prove(n_bits, m, p(v[0]), p(r[0]), SEED, ctypes.c_uint64(int(sid[0])), ctypes.c_uint64(0), None, 0, out) == 0
    assert out.raw == outs[0][:ps]


@pytest.mark.gpu
@pytest.mark.parametrize("height,n", [(1, 1), (1, 2), (3, 8), (5, 1), (9, 300), (13, 8192), (16, 1024), (20, 5000), (32, 1500), (64, 700), (14, 8193)])
def test_phased_and_levelwise_tree_builds_agree(gpu_ctx, hip_lib, height, n):
    """Trees of at most 8,192 leaves are built by phases (structure in one block, all padding nodes at once, sums and hashes level by
    level); larger ones level by level in one merge kernel per level.  Both must leave the same tree: root, node counts and every
    Merkle path with its secrets -- full trees, a single leaf, sibling pairs, 64-bit indexes, the switch-over sizes."""
    import os
    rng = np.random.default_rng(1000 * height + n)
    if n == 1 << height:
        idx = np.arange(n, dtype=np.uint64)
        v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
        r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        r[:, 31] &= 0x0F
    else:
        idx, v, r = _rand_leaves(rng, height, n)
        if len(idx) > 3 and height >= 2:                       # make sure real sibling pairs occur
            idx[1] = idx[0] ^ np.uint64(1)
            idx = np.unique(idx)
            v, r = v[:len(idx)], r[:len(idx)]
    pick = idx if len(idx) <= 64 else np.sort(np.unique(idx[rng.integers(0, len(idx), size=64)]))
    got = []
    for env in ({}, {"DAPOL_TREE_LEVELWISE": "1"}):
        os.environ.update(env)
        try:
            tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
            levels = [[a.tobytes() for a in tr.level_nodes(k)] for k in range(height + 1)]       # every node of every level, padding included
            got.append((tr.root(), tr.node_count(), levels, [a.tobytes() for a in tr.paths(pick)]))
            tr.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    for part in range(4):
        assert got[0][part] == got[1][part], part
