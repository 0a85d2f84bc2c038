#!/usr/bin/env python3
"""One inclusion proof on demand (the reference's `prove` / `verify` criterion groups, benches/dapol.rs:59-141): proves and verifies
one leaf of a 1,024-leaf, height-32 tree a few times and prints the best and median latency; run it under
`rocprofv3 --kernel-trace` for tools/kernel_timeline.py.  The DAPOL_* knobs of host_range.inc / host_verify.inc apply."""
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dapol_amd import capi  # noqa: E402

height, n = 32, 1024
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
idx, v, r = bench.synth_inputs(n, height, 0, n)
ctx = capi.Context(0, 32)
tree = capi.Tree(ctx, height, idx, v, r, bench.PAD_SEED)
rC, rH, _, _ = tree.root()
lC, lH = ctx.commit_hash_batch(v, r)
tp, tv = [], []
for i in range(reps):
    k = (5 + 97 * i) % n
    t0 = time.perf_counter()
    pC, pH, proofs = tree.prove_entities(idx[k:k + 1], capi.POLICY_PADDING, height, 64, bench.NONCE_SEED)
    t1 = time.perf_counter()
    ok = ctx.verify_entities(height, idx[k:k + 1], lC[k:k + 1], lH[k:k + 1], pC, pH, rC, rH, capi.POLICY_PADDING, height, 64, proofs)
    t2 = time.perf_counter()
    assert ok[0]
    tp.append(1e3 * (t1 - t0))
    tv.append(1e3 * (t2 - t1))
knobs = {k: v for k, v in os.environ.items() if k.startswith("DAPOL_") and k != "DAPOL_HIP_LIB"}
print("prove min %.2f median %.2f ms | verify min %.2f median %.2f ms | reps %d | %s" %
      (min(tp[1:] or tp), statistics.median(tp[1:] or tp), min(tv[1:] or tv), statistics.median(tv[1:] or tv), reps, knobs), flush=True)
