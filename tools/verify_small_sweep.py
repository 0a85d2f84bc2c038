#!/usr/bin/env python3
"""Verification latency of b inclusion proofs in one call (height 32, 64-bit, padding policy), median of 7, under the knobs given in
the environment.  usage: tools/verify_small_sweep.py b [b ...]"""
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dapol_amd import capi  # noqa: E402

n, height, seed = 1 << 12, 32, bytes(range(32))
rng = np.random.default_rng(3)
idx = np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n)
v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
r[:, 31] &= 0x0F
ctx = capi.Context(0, 32)
tree = capi.Tree(ctx, height, idx, v, r, seed)
rC, rH, _, _ = tree.root()
lC, lH = ctx.commit_hash_batch(v, r)
knobs = {k: v_ for k, v_ in os.environ.items() if k.startswith("DAPOL_") and k != "DAPOL_HIP_LIB"}
for b in [int(a) for a in sys.argv[1:]]:
    sel = idx[:: n // b][:b]
    pos = np.searchsorted(idx, sel)
    pC, pH, proofs = tree.prove_entities(sel, capi.POLICY_PADDING, height, 64, seed)
    tv = []
    for _ in range(8):
        t0 = time.perf_counter()
        ok = ctx.verify_entities(height, sel, lC[pos], lH[pos], pC, pH, rC, rH, capi.POLICY_PADDING, height, 64, proofs)
        tv.append(time.perf_counter() - t0)
        assert ok.all()
    print("verify b=%d: %.2f ms %s" % (b, 1e3 * sorted(tv[1:])[3], knobs), flush=True)
