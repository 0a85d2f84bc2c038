#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: verification-only throughput of aggregated Bulletproofs with m = 1024 parties of
64 bits (one proof covers 1,024 entities' commitments).  The proofs are produced by this build's own prover first
(untimed), then dapol_range_verify_batch is timed.  Usage: python tools/bench_verify.py [n_proofs] [m]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dapol_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = 64
seed = bytes(range(32))
t0 = time.perf_counter()
ctx = capi.Context(0, m)
t_ctx = time.perf_counter() - t0
rng = np.random.default_rng(5)
v = rng.integers(0, 2**32, size=(B, m), dtype=np.uint64)
r = rng.integers(0, 256, size=(B, m, 32), dtype=np.uint8)
r[:, :, 31] &= 0x0F
t0 = time.perf_counter()
proofs = ctx.range_prove_batch(n, m, v, r, nonce_seed=seed, stream_id=np.arange(B, dtype=np.uint64))
t_prove = time.perf_counter() - t0
C, _ = ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
Vs = C.reshape(B, m, 32)
ok = ctx.range_verify_batch(n, m, proofs[:2], Vs[:2], verify_seed=seed)       # warm-up
bad = proofs.copy()
bad[B // 3, 100] ^= 1                                                           # one bad proof in the whole batch
t0 = time.perf_counter()
ok_one = ctx.range_verify_batch(n, m, bad, Vs, verify_seed=seed)
t_one_bad = time.perf_counter() - t0
one_bad_found = bool(ok_one[B // 3] == 0 and ok_one.sum() == B - 1)
t_verify = 1e9
for _ in range(3):                                                              # best of three
    t0 = time.perf_counter()
    ok = ctx.range_verify_batch(n, m, proofs, Vs, verify_seed=seed)
    t_verify = min(t_verify, time.perf_counter() - t0)
bad = proofs.copy()
bad[0, 100] ^= 1
ok_bad = ctx.range_verify_batch(n, m, bad[:2], Vs[:2], verify_seed=seed)
print(json.dumps({"config": "verify-only, %d proofs x m=%d x n=%d (proof %d bytes)" % (B, m, n, proofs.shape[1]), "all_verified": bool(ok.all()),
                  "tampered_rejected": bool(ok_bad[0] == 0 and ok_bad[1] == 1), "ctx_create_s": t_ctx, "prove_s": t_prove,
                  "verify_s": t_verify, "verify_with_one_bad_proof_s": t_one_bad, "one_bad_proof_found": one_bad_found, "proofs_per_s": B / t_verify, "entities_per_s": B * m / t_verify,
                  "note": "host-inclusive (H2D of proofs + commitments); one GPU"}))
