#!/usr/bin/env python3
"""Latency of small proving calls (one user's inclusion proof on demand): dapol_prove_entities for b leaves of a 2^16-leaf,
height-32 tree, 64-bit padding-policy proofs.  Host-inclusive, median of 5.  Usage: python tools/bench_latency.py"""
import json
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dapol_amd import capi  # noqa: E402

n, height, seed = 1 << 16, 32, bytes(range(32))
rng = np.random.default_rng(3)
idx = np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n)
v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
r[:, 31] &= 0x0F
ctx = capi.Context(0, 32)
tree = capi.Tree(ctx, height, idx, v, r, seed)
rC, rH, _, _ = tree.root()
lC, lH = ctx.commit_hash_batch(v, r)
out = {}
for b in (1, 2, 4, 8, 16, 32, 64, 65, 128, 256, 512, 1024, 1025, 2048, 4096, 16384):
    sel = idx[:: n // b][:b]
    pos = np.searchsorted(idx, sel)
    tree.prove_entities(sel, capi.POLICY_PADDING, height, 64, seed)
    tp, tv = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        pC, pH, proofs = tree.prove_entities(sel, capi.POLICY_PADDING, height, 64, seed)
        tp.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ok = ctx.verify_entities(height, sel, lC[pos], lH[pos], pC, pH, rC, rH, capi.POLICY_PADDING, height, 64, proofs, verify_seed=seed)
        tv.append(time.perf_counter() - t0)
        assert ok.all()
    out[b] = {"prove_ms": 1e3 * sorted(tp)[2], "verify_ms": 1e3 * sorted(tv)[2]}
print(json.dumps(out))
