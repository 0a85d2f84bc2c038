#!/usr/bin/env python3
"""PCIe-inclusive rate of the proving path: the C-ABI entry point that takes and returns HOST buffers
(dapol_prove_entities through capi.Tree.prove_entities: H2D of the leaf ids, D2H of paths and proofs) beside the
device-resident workload bench.py times.  Usage: python tools/bench_pcie.py [log2_entities]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dapol_amd import capi  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n, height, seed = 1 << lg, 32, bytes(range(32))
rng = np.random.default_rng(9)
idx = np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n)
v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
r[:, 31] &= 0x0F
ctx = capi.Context(0, 32)
w = capi.Workload(ctx, height, idx, v, r)
w.build(seed)
w.prove(seed, 64)                                  # warm-up (scratch allocation)
t0 = time.perf_counter()
w.build(seed)
st = w.prove(seed, 64)
t_res = time.perf_counter() - t0
tree = capi.Tree(ctx, height, idx, v, r, seed)
tree.prove_entities(idx[:1024], capi.POLICY_PADDING, height, 64, seed)
t0 = time.perf_counter()
tree2 = capi.Tree(ctx, height, idx, v, r, seed)    # H2D of the entity arrays + build
pC, pH, proofs = tree2.prove_entities(idx, capi.POLICY_PADDING, height, 64, seed)   # D2H of 2 x 32 x 32 B of path + 992 B of proof per entity
t_host = time.perf_counter() - t0
print(json.dumps({"entities": n, "device_resident_s": t_res, "device_resident_entities_per_s": n / t_res,
                  "host_buffers_s": t_host, "host_buffers_entities_per_s": n / t_host,
                  "bytes_returned_per_entity": int(pC[0].nbytes + pH[0].nbytes + proofs[0].nbytes)}))
