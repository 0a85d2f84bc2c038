#!/usr/bin/env python3
"""One inclusion proof on demand (the reference's `prove` / `verify` criterion groups, benches/dapol.rs:59-141): proves and verifies
one leaf of a 1,024-leaf, height-32 tree a few times; meant to run under `rocprofv3 --kernel-trace` for tools/kernel_timeline.py."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dapol_amd import capi  # noqa: E402

height, n = 32, 1024
idx, v, r = bench.synth_inputs(n, height, 0, n)
ctx = capi.Context(0, 32)
tree = capi.Tree(ctx, height, idx, v, r, bench.PAD_SEED)
rC, rH, _, _ = tree.root()
lC, lH = ctx.commit_hash_batch(v, r)
for k in (5, 77, 300, 900):
    t0 = time.perf_counter()
    pC, pH, proofs = tree.prove_entities(idx[k:k + 1], capi.POLICY_PADDING, height, 64, bench.NONCE_SEED)
    t1 = time.perf_counter()
    ok = ctx.verify_entities(height, idx[k:k + 1], lC[k:k + 1], lH[k:k + 1], pC, pH, rC, rH, capi.POLICY_PADDING, height, 64, proofs)
    t2 = time.perf_counter()
    print("prove %.2f ms, verify %.2f ms, ok %d" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), int(ok[0])), flush=True)
