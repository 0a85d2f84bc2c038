"""The randomness contract's TAPE mode at the two places the boundary promised it: the padding nodes of a tree
build (src/dapol/node.rs:86-88: Scalar::random per padding node) and the nonces of dapol_prove_entities (the bulletproofs prover's
draws).  A tape holds the 64-byte draws themselves; fed with the draws seed mode derives, tape mode must give the same bytes."""
import numpy as np
import pytest

from test_gpu_parity import SEED, _rand_leaves

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("height,n", [(6, 9), (10, 100), (16, 3000), (40, 20000)])
def test_tree_build_tape_equals_seed_mode(gpu_ctx, hip_lib, pyref, height, n):
    """dapol_tree_build_tape with one draw per padding node, in (level bottom-up, index ascending) order == dapol_tree_build with the
    seed those draws come from, at every level (small trees take the phased build in seed mode and the level-wise kernel in tape mode:
    n = 9 / 100 / 3,000; 20,000 leaves are level-wise in both)."""
    rng = np.random.default_rng(height * 1000 + n)
    idx, v, r = _rand_leaves(rng, height, n)
    seeded = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    level, index = hip_lib.tree_padding_positions(height, idx)
    assert len(level) == seeded.node_count()[1]
    assert all((int(level[i]), int(index[i])) < (int(level[i + 1]), int(index[i + 1])) for i in range(len(level) - 1))       # tape order
    for k in sorted(set(level.tolist()))[:3]:                           # ... and these ARE the tree's padding positions
        li, lv, lr, lC, lH, pad = seeded.level_nodes(k)
        assert sorted(int(i) for i, p in zip(li, pad) if p) == [int(i) for l, i in zip(level, index) if l == k]
    draws = [pyref.seed_wide(SEED, 1, int(l), int(i)) for l, i in zip(level, index)]          # domain 1: padding node at (level, index)
    taped = hip_lib.Tree(gpu_ctx, height, idx, v, r, None, pad_tape=b"".join(draws))
    assert taped.root() == seeded.root() and taped.node_count() == seeded.node_count()
    for k in range(height + 1):
        a, b = seeded.level_nodes(k), taped.level_nodes(k)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), k
    # the same draws in another order: the root's commitment and blinding are sums over all nodes and do not see the order, its hash does
    other = hip_lib.Tree(gpu_ctx, height, idx, v, r, None, pad_tape=b"".join(reversed(draws)))
    assert other.root()[0] == seeded.root()[0] and other.root()[2:] == seeded.root()[2:] and (other.root()[1] != seeded.root()[1] or len(draws) < 2)
    # inclusion proofs from a tape-built tree verify like any other
    pC, pH, proofs = taped.prove_entities(idx[:3], hip_lib.POLICY_PADDING, min(height, 4), 64, SEED)
    lC, lH = gpu_ctx.commit_hash_batch(v[:3], r[:3])
    rC, rH, _, _ = taped.root()
    assert gpu_ctx.verify_entities(height, idx[:3], lC, lH, pC, pH, rC, rH, hip_lib.POLICY_PADDING, min(height, 4), 64, proofs).all()
    # a tape that is one draw short is an error, not a tree; a tape-built tree has no seed to update from
    if len(draws):
        with pytest.raises(hip_lib.DapolError) as e:
            hip_lib.Tree(gpu_ctx, height, idx, v, r, None, pad_tape=b"".join(draws[:-1]))
        assert e.value.code == 8 and "tape" in str(e.value)
    with pytest.raises(hip_lib.DapolError) as e:
        taped.update(idx[:1], v[:1], r[:1])
    assert e.value.code == 8


@pytest.mark.parametrize("height,policy,agg,n_bits", [(8, 0, 8, 16), (8, 1, 5, 16), (6, 0, 3, 8), (5, 1, 0, 8), (9, 1, 9, 64), (7, 0, 0, 8)])
def test_prove_entities_tape_equals_seed_mode(gpu_ctx, hip_lib, pyref, height, policy, agg, n_bits):
    """dapol_prove_entities_tape: one RNG stream runs through the sub-proofs of the policy (padding.rs:104-112, splitting.rs:110-123),
    so an entity's tape is its sub-proofs' slots one after the other.  Seed mode keys every sub-proof's draws by (stream = leaf index,
    first slot, shape, the parties' value commitments): replayed as a tape they give the same proofs."""
    rng = np.random.default_rng(height * 100 + agg)
    idx, v, r = _rand_leaves(rng, height, 7, vmax=(1 << (n_bits - 4)))
    tree = hip_lib.Tree(gpu_ctx, height, idx, v, r, SEED)
    who = idx[[1, 4, 6]]
    pC, pH, want = tree.prove_entities(who, policy, agg, n_bits, SEED)
    plan, pos = pyref.policy_plan("padding" if policy == 0 else "splitting", height, agg)
    subs = [(start, count, m) for start, count, m in plan] + [(s, 1, 1) for s in range(pos, height)]
    if policy == 0 and agg == 0:                                         # padding with nothing aggregated: one proof over the (0, 1) party alone
        subs = [(0, 0, 1)] + [(s, 1, 1) for s in range(height)]
    slots = hip_lib.lib().dapol_entity_tape_slots(height, policy, agg, n_bits)
    assert slots == sum(m * (2 * n_bits + 4) for _, _, m in subs)
    Bb = gpu_ctx.generator(1)                                            # commit(0, 1) = B_blinding pads the aggregated proof (padding.rs:100-103)
    tape = b""
    for e, leaf in enumerate(who):
        base = 0
        for start, count, m in subs:
            coms = [pC[e, start + j].tobytes() for j in range(count)] + [Bb] * (m - count)
            key = pyref.nonce_key(SEED, int(leaf), base, n_bits, m, coms)
            tape += b"".join(pyref.seed_wide(key, 2, int(leaf), base + k) for k in range(m * (2 * n_bits + 4)))
            base += m * (2 * n_bits + 4)
    tC, tH, got = tree.prove_entities(who, policy, agg, n_bits, None, tape=tape)
    assert got.tobytes() == want.tobytes() and tC.tobytes() == pC.tobytes() and tH.tobytes() == pH.tobytes()
    # other draws, other proofs -- that still verify
    rng2 = np.random.default_rng(1)
    rnd = rng2.integers(0, 256, size=len(tape), dtype=np.uint8).tobytes()
    _, _, other = tree.prove_entities(who, policy, agg, n_bits, None, tape=rnd)
    assert other.tobytes() != want.tobytes()
    lC, lH = gpu_ctx.commit_hash_batch(v[[1, 4, 6]], r[[1, 4, 6]])
    rC, rH, _, _ = tree.root()
    assert gpu_ctx.verify_entities(height, who, lC, lH, pC, pH, rC, rH, policy, agg, n_bits, other).all()
    with pytest.raises(hip_lib.DapolError):
        tree.prove_entities(who, policy, agg, n_bits, None, tape=tape[:-64])
