#!/usr/bin/env python3
"""Randomised soak of the wavefront transcript replay (k_rv_absorb_V + k_rv_transcript<1>, kernels_verify.h) against the
lane-per-proof replay of the same proofs: random party counts m (1 .. 1,024: the commitment stream ends at every offset of the
166-byte STROBE block), bit widths and batch sizes; every honest proof must verify on both paths, with and without cross-proof
batching, and a commitment changed in its first / last byte, or a proof byte, must turn exactly its proof on both.
usage: tools/soak_verify_wave.py [cases] [seed]"""
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dapol_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
SEED = bytes(range(32))
ctx = capi.Context(0, 1024)
KNOBS = ("DAPOL_VERIFY_WAVE_TRANSCRIPT", "DAPOL_VERIFY_LANE_TRANSCRIPT", "DAPOL_VERIFY_NO_RLC", "DAPOL_VERIFY_RLC_MIN", "DAPOL_VERIFY_NO_PIPELINE")


def verdicts(n_bits, m, proofs, V):
    out = []
    try:
        for wave in ("1", "0"):
            for rlc in (False, True):
                for lane in (False, True):
                    for k in KNOBS:
                        os.environ.pop(k, None)
                    os.environ["DAPOL_VERIFY_WAVE_TRANSCRIPT"] = wave
                    if rlc:
                        os.environ["DAPOL_VERIFY_RLC_MIN"] = "2"
                    else:
                        os.environ["DAPOL_VERIFY_NO_RLC"] = "1"
                    if lane:
                        os.environ["DAPOL_VERIFY_LANE_TRANSCRIPT"] = "1"
                    out.append(tuple(int(x) for x in ctx.range_verify_batch(n_bits, m, proofs, V, verify_seed=SEED)))
    finally:
        for k in KNOBS:
            os.environ.pop(k, None)
    return out


seen = set()
for case in range(cases):
    lgm = int(rng.integers(0, 11))
    m = 1 << lgm
    n_bits = int(rng.choice([8, 16, 32, 64]))
    b = int(rng.integers(1, 7)) if m >= 256 else int(rng.integers(1, 40))
    hi = 2**n_bits if n_bits < 64 else 2**63
    v = rng.integers(0, hi, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    proofs = ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=np.arange(b, dtype=np.uint64))
    C, _ = ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
    V = C.reshape(b, m, 32)
    want = tuple([1] * b)
    for got in verdicts(n_bits, m, proofs, V):
        assert got == want, (case, n_bits, m, b, "honest proofs", got)
    # one changed byte: a commitment's first or last byte (any party), or a proof byte
    t = int(rng.integers(0, b))
    kind = int(rng.integers(0, 3))
    V2, P2 = V.copy(), proofs.copy()
    if kind < 2:
        j = int(rng.choice([0, m - 1, int(rng.integers(0, m))]))
        V2[t, j, 0 if kind == 0 else 31] ^= 1 << int(rng.integers(0, 7 if kind else 8))
    else:
        P2[t, int(rng.integers(0, P2.shape[1]))] ^= 1 << int(rng.integers(0, 8))
    want = tuple(0 if i == t else 1 for i in range(b))
    for got in verdicts(n_bits, m, P2, V2):
        assert got == want, (case, n_bits, m, b, "tampered", kind, t, got)
    seen.add((n_bits, m))
print("soak: %d random cases (%d distinct (bits, parties) shapes, m up to %d): honest proofs verify and one changed byte turns exactly its proof, "
      "on the wavefront replay and on the lane replay, one by one and batched" % (cases, len(seen), max(s[1] for s in seen)))
