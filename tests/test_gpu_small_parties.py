"""Proofs of FEW parties in batches (round 6): the individual proofs a policy leaves per sibling beyond aggregation_factor
(/root/reference/src/range/padding.rs:104-112, splitting.rs:118-123, src/range/mod.rs:48-62) are proved as ONE grouped call per run of
equal-sized sub-proofs (host_range.inc: prove_policy_device, RangeArgs::sub_k), and large batches of short lists are swept
generator-stationary with two lookups per term (k_rp_msm_gs_hi).  Bytes must not move: against the C oracle sub-proof by sub-proof
(same stream, same first slot), against the ungrouped path, and under every arrangement of the sweep."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = bytes(range(32))


def _plan(policy, height, agg):
    """(start, count, m) per sub-proof -- policy_plan.inc restated for the checker."""
    np2 = lambda x: 1 << max(0, (x - 1).bit_length())
    plan = []
    if policy == 0:
        plan.append((0, agg, np2(agg) if agg else 1))
    else:
        base, pos = np2(agg) if agg else 1, 0
        while pos < agg:
            if agg & base:
                plan.append((pos, base, base))
                pos += base
            base >>= 1
    plan += [(i, 1, 1) for i in range(agg, height)]
    return plan


def _with_env(env, fn):
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k in env:
            os.environ.pop(k, None)


def _leaves(rng, height, n):
    if height > 22:
        idx = np.sort(np.unique(rng.integers(0, 1 << height, size=n, dtype=np.uint64)))
        n = len(idx)
    else:
        idx = np.sort(rng.choice(1 << height, size=n, replace=False).astype(np.uint64))
    v = rng.integers(0, 2**40, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    return idx, v, r


@pytest.mark.parametrize("height,policy,agg,n_bits", [(12, 0, 4, 64), (12, 1, 7, 64), (10, 0, 0, 64), (11, 1, 6, 32), (9, 0, 8, 64), (10, 1, 10, 64),
                                                      (32, 0, 24, 64), (32, 1, 24, 64)])          # (the last two: bench.py's small_parties / splitting legs)
def test_grouped_policy_proofs_vs_c_oracle(gpu_ctx, hip_lib, ref, height, policy, agg, n_bits):
    """Every sub-proof of every entity against ref_range_prove_batch with the entity's stream and the sub-proof's first slot; the
    grouped call (default), the one-call-per-sub-proof path (DAPOL_NO_GROUP) and the short-list sweep forced onto this batch
    (DAPOL_GS_SMALL_MIN) give the same bytes."""
    rng = np.random.default_rng(height * 100 + agg)
    n = 24
    idx, v, r = _leaves(rng, height, n)
    n = len(idx)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    pC, pH, got = tr.prove_entities(idx, policy, agg, n_bits, SEED)
    # siblings' secrets from the C oracle's own tree (also checks the paths the prover saw)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    t = ctypes.c_void_p(ref.ref_tree_build(height, ctypes.c_size_t(n), p(idx), p(v), p(r), SEED, 0))
    plan = _plan(policy, height, agg)
    for e in range(0, n, 5):
        sC, sH = ctypes.create_string_buffer(32 * height), ctypes.create_string_buffer(32 * height)
        svv, srr = (ctypes.c_uint64 * height)(), ctypes.create_string_buffer(32 * height)
        assert ref.ref_tree_path(t, ctypes.c_uint64(int(idx[e])), sC, sH, svv, srr) == 1
        assert pC[e].tobytes() == sC.raw
        want, slot = b"", 0
        for start, count, m in plan:
            vv = np.zeros(m, np.uint64)
            rr = np.zeros((m, 32), np.uint8)
            rr[:, 0] = 1                                        # pad parties: (0, Scalar::one())
            for j in range(count):
                vv[j] = svv[start + j]
                rr[j] = np.frombuffer(srr.raw[32 * (start + j):32 * (start + j + 1)], np.uint8)
            ps = ref.ref_range_proof_size(n_bits, m)
            out = ctypes.create_string_buffer(ps)
            sid = np.array([idx[e]], np.uint64)
            assert ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(1), p(vv), p(rr), SEED, p(sid), ctypes.c_uint64(slot), None, 0, out) == 0
            want += out.raw
            slot += m * (2 * n_bits + 4)
        assert got[e].tobytes() == want, (e, "entity differs from the oracle")
    ref.ref_tree_free(t)
    for env in ({"DAPOL_NO_GROUP": "1"}, {"DAPOL_NO_LANES": "1"}, {"DAPOL_NO_LANES": "1", "DAPOL_NO_GROUP": "1"}, {"DAPOL_LANES_MAX": "2"}, {"DAPOL_GS_SMALL_MIN": "64"}, {"DAPOL_GS_SMALL_MIN": "64", "DAPOL_NO_GS_HI": "1"},
                {"DAPOL_GS_SMALL_MIN": "64", "DAPOL_CHUNK": "67"}, {"DAPOL_CHUNK": "5"}, {"DAPOL_CHUNK": "13", "DAPOL_STREAMS": "3"}):
        again = _with_env(env, lambda: tr.prove_entities(idx, policy, agg, n_bits, SEED)[2])
        assert again.tobytes() == got.tobytes(), env


@pytest.mark.parametrize("n_bits,m", [(64, 1), (64, 2), (64, 4), (64, 8), (32, 2), (8, 8), (16, 16)])
def test_short_list_sweep_gives_the_same_bytes(gpu_ctx, ref, n_bits, m):
    """dapol_range_prove_batch of proofs of 64 ... 512 generators a side: the proof-stationary default of a batch this small against the
    generator-stationary sweep (two lookups per term, plain, sliced, other tiles, ragged chunks), and the first proofs against the
    C oracle."""
    b = 150
    rng = np.random.default_rng(n_bits + 7 * m)
    v = rng.integers(0, 2**n_bits if n_bits < 64 else 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = rng.integers(0, 2**62, size=b, dtype=np.uint64)
    base = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid, slot_base=3)
    k = 6
    ps = ref.ref_range_proof_size(n_bits, m)
    out = ctypes.create_string_buffer(ps * k)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(k), p(v), p(r), SEED, p(sid), ctypes.c_uint64(3), None, 0, out) == 0
    assert base[:k].tobytes() == out.raw
    G = {"DAPOL_GS_SMALL_MIN": "64"}
    for extra in ({}, {"DAPOL_NO_GS_HI": "1"}, {"DAPOL_GS_SLICES": "2"}, {"DAPOL_GS_SLICES": "16"}, {"DAPOL_GS_SLICES": "8", "DAPOL_NO_GS_HI": "1"},   # (more slices than a short list has accumulator slots for: clamped) {"DAPOL_GS_SLICES": "1", "DAPOL_GS_TILE": "4"}, {"DAPOL_GS_TILE": "64"},
                  {"DAPOL_CHUNK": "64"}, {"DAPOL_CHUNK": "70", "DAPOL_STREAMS": "2"}, {"DAPOL_FS_SHAPE": "0"}, {"DAPOL_FS_SHAPE": "1"},
                  {"DAPOL_NO_STAB": "1"}, {"DAPOL_TAIL_N": "32"}, {"DAPOL_TAIL_N": "64", "DAPOL_NO_GS_HI": "1"}):
        env = dict(G, **extra)
        got = _with_env(env, lambda: gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid, slot_base=3))
        assert got.tobytes() == base.tobytes(), env


def test_grouped_tape_mode_equals_seed_mode(gpu_ctx, hip_lib, pyref):
    """Tape mode through a grouped plan: the draws of entity e's sub-proof j sit at the row's slots slot_base + j m (2n + 4) ... --
    replaying the seed mode's own draws must give the seed mode's bytes (dapol_prove_entities_tape, RangeArgs::sub_slots)."""
    height, agg, n_bits, n = 6, 3, 8, 7
    rng = np.random.default_rng(11)
    idx = np.sort(rng.choice(1 << height, size=n, replace=False).astype(np.uint64))
    v = rng.integers(0, 8, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    for policy in (0, 1):
        pC, pH, seed_mode = tr.prove_entities(idx, policy, agg, n_bits, SEED)
        plan = _plan(policy, height, agg)
        Bb = gpu_ctx.generator(1)
        rows = []
        for e in range(n):
            row, slot = b"", 0
            for start, count, m in plan:
                Vs = [pC[e, start + j].tobytes() if j < count else Bb for j in range(m)]
                key = pyref.nonce_key(SEED, int(idx[e]), slot, n_bits, m, Vs)
                row += b"".join(pyref.seed_wide(key, 2, int(idx[e]), slot + k) for k in range(m * (2 * n_bits + 4)))
                slot += m * (2 * n_bits + 4)
            rows.append(row)
        tape = np.frombuffer(b"".join(rows), np.uint8)
        got = tr.prove_entities(idx, policy, agg, n_bits, None, tape=tape)[2]
        assert got.tobytes() == seed_mode.tobytes(), policy
        again = _with_env({"DAPOL_NO_GROUP": "1"}, lambda: tr.prove_entities(idx, policy, agg, n_bits, None, tape=tape)[2])
        assert again.tobytes() == seed_mode.tobytes()


def test_large_batch_of_individual_proofs_round_trip(gpu_ctx, hip_lib):
    """Size-independent property at a size the oracle cannot reach: 2^11 entities at aggregation 0 on a height-16 tree = 34,816
    individual proofs through the default path (the short-list sweep, chunks of whole rounds) verify on the GPU, and a grouped call
    equals the ungrouped one on a sample."""
    height, n = 16, 1 << 11
    rng = np.random.default_rng(3)
    idx, v, r = _leaves(rng, height, n)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    pC, pH, proofs = tr.prove_entities(idx, 0, 0, 64, SEED)
    root = tr.root()
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    ok = gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, root[0], root[1], 0, 0, 64, proofs, verify_seed=SEED)
    assert ok.all()
    sel = idx[::97]
    small = tr.prove_entities(sel, 0, 0, 64, SEED)[2]                  # 22 entities: 374 proofs, proof-stationary
    assert small.tobytes() == proofs[::97].tobytes()
    bad = proofs.copy()
    bad[5, 700] ^= 1                                                   # inside the first individual proof of entity 5
    ok = gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, root[0], root[1], 0, 0, 64, bad, verify_seed=SEED)
    assert not ok[5] and ok.sum() == n - 1
    # the grouped check (one batch of n * 16 individual proofs) and one check per sub-proof give the same verdict vector, also when the bad
    # proof is the LAST sub-proof of the last entity and when two entities are bad
    bad[n - 1, -3] ^= 4
    for env in ({}, {"DAPOL_NO_GROUP": "1"}):
        ok = _with_env(env, lambda: gpu_ctx.verify_entities(height, idx, lC, lH, pC, pH, root[0], root[1], 0, 0, 64, bad, verify_seed=SEED))
        assert not ok[5] and not ok[n - 1] and ok.sum() == n - 2, env


@pytest.mark.parametrize("height,policy,agg", [(9, 0, 4), (9, 1, 7), (8, 1, 5), (7, 0, 0)])
def test_grouped_verification_equals_per_sub_proof_verification(gpu_ctx, hip_lib, height, policy, agg):
    """dapol_verify_entities over plans with runs of equal-sized sub-proofs: all valid -> all ones; one byte flipped in every region of
    one entity's blob (the aggregated part, the first, a middle and the last individual proof) -> exactly that entity, grouped or not."""
    rng = np.random.default_rng(height + 31 * agg)
    n = 40
    idx, v, r = _leaves(rng, height, n)
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    pC, pH, proofs = tr.prove_entities(idx, policy, agg, 64, SEED)
    root = tr.root()
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    args = (height, idx, lC, lH, pC, pH, root[0], root[1], policy, agg, 64)
    assert gpu_ctx.verify_entities(*args, proofs, verify_seed=SEED).all()
    es = proofs.shape[1]
    for e, off in ((3, 40), (11, es - 672 + 100), (17, es - 2 * 672 + 5), (39, es - 1)):
        bad = proofs.copy()
        bad[e, off] ^= 0x10
        for env in ({}, {"DAPOL_NO_GROUP": "1"}):
            ok = _with_env(env, lambda: gpu_ctx.verify_entities(*args, bad, verify_seed=SEED))
            assert not ok[e] and ok.sum() == n - 1, (e, off, env)


@pytest.mark.parametrize("height,policy,agg,n_bits", [(24, 1, 24, 64), (12, 1, 7, 64), (12, 0, 4, 64), (9, 1, 9, 32), (31, 1, 31, 64), (6, 0, 0, 8)])
def test_groups_of_a_small_call_on_lanes_give_the_same_bytes(gpu_ctx, hip_lib, ref, height, policy, agg, n_bits):
    """A small call whose plan has several groups (splitting at height 24 = a 16-party + an 8-party proof, the reference's own
    `prove` case, benches/dapol.rs:71-78) runs them side by side on lanes of their own (dapol_ctx::aux): the bytes are those of the
    one-after-the-other path (DAPOL_NO_LANES=1), of the ungrouped path, and of the C oracle; a second, different call on the same
    context right after is clean; the proofs verify."""
    rng = np.random.default_rng(height * 7 + agg)
    n = 40
    idx = np.sort(np.unique(rng.integers(0, 1 << height, size=n, dtype=np.uint64)))
    n = len(idx)
    v = rng.integers(0, min(2**20, (2**n_bits - 1) // n) + 1, size=n, dtype=np.uint64)      # every subtree sum stays inside n_bits
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    tr = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    root = tr.root()
    lC, lH = gpu_ctx.commit_hash_batch(v, r)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    t = ctypes.c_void_p(ref.ref_tree_build(height, ctypes.c_size_t(n), p(idx), p(v), p(r), SEED, 0))
    plan = _plan(policy, height, agg)
    for who in (idx[:1], idx[5:7], idx[3:4]):
        pC, pH, got = tr.prove_entities(who, policy, agg, n_bits, SEED)
        for env in ({"DAPOL_NO_LANES": "1"}, {"DAPOL_NO_GROUP": "1", "DAPOL_NO_LANES": "1"}, {"DAPOL_NO_GROUP": "1"}):
            again = _with_env(env, lambda: tr.prove_entities(who, policy, agg, n_bits, SEED)[2])
            assert again.tobytes() == got.tobytes(), env
        sv, sr = (ctypes.c_uint64 * height)(), ctypes.create_string_buffer(32 * height)
        sC, sH = ctypes.create_string_buffer(32 * height), ctypes.create_string_buffer(32 * height)
        assert ref.ref_tree_path(t, ctypes.c_uint64(int(who[0])), sC, sH, sv, sr) == 1
        want, slot = b"", 0
        for start, count, m in plan:
            vv = np.zeros(m, np.uint64)
            rr = np.zeros((m, 32), np.uint8)
            rr[:, 0] = 1
            for j in range(count):
                vv[j] = sv[start + j]
                rr[j] = np.frombuffer(sr.raw[32 * (start + j):32 * (start + j + 1)], np.uint8)
            ps = ref.ref_range_proof_size(n_bits, m)
            out = ctypes.create_string_buffer(ps)
            sid = np.array([who[0]], np.uint64)
            assert ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(1), p(vv), p(rr), SEED, p(sid), ctypes.c_uint64(slot), None, 0, out) == 0
            want += out.raw
            slot += m * (2 * n_bits + 4)
        assert got[0].tobytes() == want
        pos = np.searchsorted(idx, who)
        ok = gpu_ctx.verify_entities(height, who, lC[pos], lH[pos], pC, pH, root[0], root[1], policy, agg, n_bits, got, verify_seed=SEED)
        assert ok.all()
        # the verifier's groups run on lanes too: a byte flipped in the FIRST and in the LAST sub-proof of the first entity turns its verdict,
        # lanes or not, and leaves the other entity's alone
        for off in (7, got.shape[1] - 9):
            bad = got.copy()
            bad[0, off] ^= 2
            for env in ({}, {"DAPOL_NO_LANES": "1"}):
                okb = _with_env(env, lambda: gpu_ctx.verify_entities(height, who, lC[pos], lH[pos], pC, pH, root[0], root[1], policy, agg, n_bits, bad, verify_seed=SEED))
                assert not okb[0] and okb[1:].all(), (off, env)
    ref.ref_tree_free(t)
    # the batched inclusion proof of several leaves (one proof over the deduplicated siblings) takes the same path with b = 1
    leaves = idx[:3]
    a = tr.prove_batch(leaves, policy, min(agg, 4), n_bits, SEED)
    b_ = _with_env({"DAPOL_NO_LANES": "1"}, lambda: tr.prove_batch(leaves, policy, min(agg, 4), n_bits, SEED))
    assert a[-1] == b_[-1]


def test_two_different_large_batches_back_to_back_need_no_fallback(gpu_ctx, hip_lib):
    """the forked own-points branch of the batched verifier read k_rv_tables' entries without being ordered behind
    it; with the SAME proofs in every pass the retained scratch still held the right values, so nothing showed.  Two independently
    proven batches large enough for the bucket method and its side stream (2,048 proofs x 40 own points = 81,920 >= 32,768),
    verified in turn on one context under fresh seeds: every verdict 1 and NOT ONE combined check falling back to bisection
    (dapol_diag_verify_fallbacks); a tampered proof then moves the counter and turns exactly its verdict."""
    n_bits, m, b = 64, 16, 2048
    batches = []
    for k in range(2):
        rng = np.random.default_rng(900 + k)
        v = rng.integers(0, 2**63, size=(b, m), dtype=np.uint64)
        r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
        r[:, :, 31] &= 0x0F
        proofs = gpu_ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=np.arange(k * b, (k + 1) * b, dtype=np.uint64))
        C, _ = gpu_ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
        batches.append((proofs, C.reshape(b, m, 32)))
    fb = ctypes.c_uint64()
    L = hip_lib.lib()
    assert L.dapol_diag_verify_fallbacks(ctypes.byref(fb)) == 0
    before = fb.value
    for i in range(6):
        proofs, Vs = batches[i & 1]
        ok = gpu_ctx.range_verify_batch(n_bits, m, proofs, Vs, verify_seed=os.urandom(32))
        assert ok.all(), i
    L.dapol_diag_verify_fallbacks(ctypes.byref(fb))
    assert fb.value == before, "a combined check of an all-valid batch fell back to bisection"
    bad = batches[1][0].copy()
    bad[777, 40] ^= 1
    ok = gpu_ctx.range_verify_batch(n_bits, m, bad, batches[1][1], verify_seed=os.urandom(32))
    L.dapol_diag_verify_fallbacks(ctypes.byref(fb))
    assert not ok[777] and ok.sum() == b - 1 and fb.value > before
    ok = gpu_ctx.range_verify_batch(n_bits, m, batches[0][0], batches[0][1], verify_seed=os.urandom(32))      # and the context is clean afterwards
    assert ok.all()
