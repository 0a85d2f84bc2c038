#!/usr/bin/env python3
"""Randomised soak of the few-party paths of round 6 against the C oracle: random trees (height 4..14), both policies, any aggregation
factor in [0, height], 8 / 16 / 32 / 64-bit proofs, 1..400 entities per call, the grouped plan under random knobs (short-list sweep
forced onto the batch or not, with or without the high-half rows, ragged chunks, several chunks in flight, grouping off).  Per case:
two random entities' whole range-proof blobs byte for byte against oracle/ref_dapol.c sub-proof by sub-proof (the entity's stream, the
sub-proof's first slot), every entity verified on the GPU (grouped check), one flipped byte at a random place of one blob rejected for
exactly that entity, and the grouped verdicts equal to the per-sub-proof ones.
usage: tools/soak_small_parties.py [cases] [seed]"""
import ctypes
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dapol_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
lib, _ = bench.build_native_oracle()
ref = ctypes.CDLL(lib)
ref.ref_range_proof_size.restype = ctypes.c_size_t
ref.ref_tree_build.restype = ctypes.c_void_p
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
ctx = capi.Context(0, 16)
SEED = bench.NONCE_SEED


def plan(policy, height, agg):
    np2 = lambda x: 1 << max(0, (x - 1).bit_length())
    out = []
    if policy == 0:
        out.append((0, agg, np2(agg) if agg else 1))
    else:
        base, pos = np2(agg) if agg else 1, 0
        while pos < agg:
            if agg & base:
                out.append((pos, base, base))
                pos += base
            base >>= 1
    return out + [(i, 1, 1) for i in range(agg, height)]


KNOBS = [{}, {"DAPOL_GS_SMALL_MIN": "64"}, {"DAPOL_GS_SMALL_MIN": "64", "DAPOL_NO_GS_HI": "1"}, {"DAPOL_GS_SMALL_MIN": "64", "DAPOL_CHUNK": None},
         {"DAPOL_CHUNK": None, "DAPOL_STREAMS": "2"}, {"DAPOL_NO_GROUP": "1"}, {"DAPOL_GS_SMALL_MIN": "64", "DAPOL_GS_SLICES": "2"}]
for case in range(cases):
    height = int(rng.integers(4, 15))
    policy = int(rng.integers(0, 2))
    agg = int(rng.integers(0, min(height, 16) + 1))
    if policy == 0 and agg > 16:
        agg = 16
    n_bits = [8, 16, 32, 64][int(rng.integers(0, 4))]
    n = int(rng.integers(1, min(400 if n_bits > 8 else 200, 1 << height) + 1))
    idx = np.sort(rng.choice(1 << height, size=n, replace=False).astype(np.uint64))
    vmax = min(1 << 40, ((1 << n_bits) - 1) // n)                      # every subtree sum (a sibling's value) stays inside n_bits
    v = rng.integers(0, vmax + 1, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    env = dict(KNOBS[int(rng.integers(0, len(KNOBS)))])
    for k in list(env):
        if env[k] is None:
            env[k] = str(int(rng.integers(3, 200)))
    tree = capi.Tree(ctx, height, idx, v, r, SEED)
    os.environ.update(env)
    try:
        pC, pH, proofs = tree.prove_entities(idx, policy, agg, n_bits, SEED)
    finally:
        for k in env:
            os.environ.pop(k, None)
    t = ctypes.c_void_p(ref.ref_tree_build(height, ctypes.c_size_t(n), p(idx), p(v), p(r), SEED, 0))
    pl = plan(policy, height, agg)
    for e in rng.choice(n, size=min(2, n), replace=False):
        sC, sH = ctypes.create_string_buffer(32 * height), ctypes.create_string_buffer(32 * height)
        sv, sr = (ctypes.c_uint64 * height)(), ctypes.create_string_buffer(32 * height)
        assert ref.ref_tree_path(t, ctypes.c_uint64(int(idx[e])), sC, sH, sv, sr) == 1
        want, slot = b"", 0
        for start, count, m in pl:
            vv = np.zeros(m, np.uint64)
            rr = np.zeros((m, 32), np.uint8)
            rr[:, 0] = 1
            for j in range(count):
                vv[j] = sv[start + j]
                rr[j] = np.frombuffer(sr.raw[32 * (start + j):32 * (start + j + 1)], np.uint8)
            ps = ref.ref_range_proof_size(n_bits, m)
            out = ctypes.create_string_buffer(ps)
            sid = np.array([idx[e]], np.uint64)
            assert ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(1), p(vv), p(rr), SEED, p(sid), ctypes.c_uint64(slot), None, 0, out) == 0
            want += out.raw
            slot += m * (2 * n_bits + 4)
        assert proofs[e].tobytes() == want, (case, height, policy, agg, n_bits, n, env, int(e))
    ref.ref_tree_free(t)
    root = tree.root()
    lC, lH = ctx.commit_hash_batch(v, r)
    args = (height, idx, lC, lH, pC, pH, root[0], root[1], policy, agg, n_bits)
    ok = ctx.verify_entities(*args, proofs, verify_seed=os.urandom(32))
    assert ok.all(), (case, "valid proofs rejected", height, policy, agg, n_bits, n)
    bad = proofs.copy()
    be, bo = int(rng.integers(0, n)), int(rng.integers(0, proofs.shape[1]))
    bad[be, bo] ^= 1 << int(rng.integers(0, 8))
    okb = ctx.verify_entities(*args, bad, verify_seed=os.urandom(32))
    os.environ["DAPOL_NO_GROUP"] = "1"
    try:
        oku = ctx.verify_entities(*args, bad, verify_seed=os.urandom(32))
    finally:
        os.environ.pop("DAPOL_NO_GROUP", None)
    assert not okb[be] and okb.sum() == n - 1 and (okb == oku).all(), (case, "tamper", height, policy, agg, n_bits, n, be, bo)
    tree.close()
    print("case %d ok: height %d %s agg %d, %d-bit, %d entities, %d sub-proofs per entity, knobs %s" %
          (case, height, "padding" if policy == 0 else "splitting", agg, n_bits, n, len(pl), env), flush=True)
print("soak_small_parties: %d cases ok" % cases)
