#!/usr/bin/env python3
"""Randomised soak of the small-call (latency) paths against the C oracle: random small trees (phased build), one to three
inclusion proofs per call (quad MSM, wavefront Fiat-Shamir kernels, division-step inversions), byte comparison with
oracle/ref_dapol.c for the padding policy, verification of every proof (wavefront path check, quad own-point ladders), and a
tampered copy that must fail.  usage: tools/soak_small_calls.py [cases] [seed]"""
import ctypes
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dapol_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
lib, _ = bench.build_native_oracle()
ref = ctypes.CDLL(lib)
ref.ref_range_proof_size.restype = ctypes.c_size_t
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
ctx = capi.Context(0, 64)
SEED = bench.NONCE_SEED
done = 0
for case in range(cases):
    height = int(rng.integers(2, 41))
    n_bits = int(rng.choice([8, 16, 32, 64]))
    n = int(rng.integers(1, min(300, 1 << min(height, 20)) + 1))
    if height >= 63:
        idx = np.unique(rng.integers(0, 2**63, size=n, dtype=np.uint64))
    else:
        idx = np.unique(rng.integers(0, 1 << height, size=n, dtype=np.uint64))
    n = len(idx)
    vmax = max(2, (1 << n_bits) // (2 * n)) if n_bits < 64 else 2**40          # sums stay in range
    v = rng.integers(0, vmax, size=n, dtype=np.uint64)
    r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    r[:, 31] &= 0x0F
    tree = capi.Tree(ctx, height, idx, v, r, bench.PAD_SEED)
    rC, rH, _, _ = tree.root()
    lC, lH = ctx.commit_hash_batch(v, r)
    k = int(rng.integers(1, min(3, n) + 1))
    pick = np.sort(rng.choice(n, size=k, replace=False))
    for pol in (capi.POLICY_PADDING, capi.POLICY_SPLITTING):
        pC, pH, proofs = tree.prove_entities(idx[pick], pol, height, n_bits, SEED)
        ok = ctx.verify_entities(height, idx[pick], lC[pick], lH[pick], pC, pH, rC, rH, pol, height, n_bits, proofs)
        assert ok.all(), (case, height, n_bits, n, pol, ok)
        bad = proofs.copy()
        bad[0, int(rng.integers(0, bad.shape[1]))] ^= 1 << int(rng.integers(0, 8))
        okb = ctx.verify_entities(height, idx[pick], lC[pick], lH[pick], pC, pH, rC, rH, pol, height, n_bits, bad)
        assert not okb[0] and okb[1:].all(), (case, "tampered proof accepted", okb)
        if pol == capi.POLICY_PADDING:
            _, _, sv, sr = tree.paths(idx[pick])
            m = 1
            while m < height:
                m <<= 1
            for e in range(k):
                pv, pr = np.zeros(m, np.uint64), np.zeros((m, 32), np.uint8)
                pv[:height], pr[:height] = sv[e], sr[e]
                pr[height:, 0] = 1
                out = ctypes.create_string_buffer(ref.ref_range_proof_size(n_bits, m))
                assert ref.ref_range_prove(n_bits, m, p(pv), p(pr), SEED, ctypes.c_uint64(int(idx[pick[e]])), ctypes.c_uint64(0), None, 0, out) == 0
                assert out.raw == proofs[e].tobytes(), (case, height, n_bits, n, "proof bytes differ from the oracle's")
    if case % 4 == 0 and n >= 4:
        # a call of several proofs takes other shapes (no quad above 8 proofs, the tail argument above 32, lane pairs above 256): its
        # proofs must be the bytes the one-proof calls give, which the oracle has just checked for this tree
        kb = int(rng.integers(4, min(n, 90) + 1))
        pk = np.sort(rng.choice(n, size=kb, replace=False))
        pol = capi.POLICY_PADDING if case % 8 == 0 else capi.POLICY_SPLITTING
        pC, pH, together = tree.prove_entities(idx[pk], pol, height, n_bits, SEED)
        for e in (0, kb // 2, kb - 1):
            _, _, alone = tree.prove_entities(idx[pk[e]:pk[e] + 1] if False else idx[[pk[e]]], pol, height, n_bits, SEED)
            assert alone[0].tobytes() == together[e].tobytes(), (case, kb, e, "a proof depends on the size of the call")
        ok = ctx.verify_entities(height, idx[pk], lC[pk], lH[pk], pC, pH, rC, rH, pol, height, n_bits, together)
        assert ok.all(), (case, kb, "batch of proofs not verified")
    tree.close()
    done += 1
    if done % 10 == 0:
        print("%d cases ok" % done, flush=True)
print("soak: %d random cases, all proofs byte-identical to the oracle (padding policy), all verified, all tampered copies rejected" % done)
