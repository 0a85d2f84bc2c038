#!/usr/bin/env python3
"""DapolProof::verify throughput: builds the bench tree (2^k entities, height 32), proves every entity, then times
dapol_verify_entities (Merkle re-merge + padding-policy range verification) over all of them.  Host-inclusive.
Usage: python tools/bench_verify_entities.py [log2_entities]"""
import json
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dapol_amd import capi  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, height, seed = 1 << k, 32, bytes(range(32))
rng = np.random.default_rng(9)
idx = np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n)
v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
r[:, 31] &= 0x0F
ctx = capi.Context(0, 32)
w = capi.Workload(ctx, height, idx, v, r)
(rC, rH, rv, _), _ = w.build(seed)
st = w.prove(seed, 64)
ps = st.proof_bytes // n
proofs = w.proofs(0, n, ps)
_, _, pC, pH = w.paths(idx, with_nodes=True)
lC, lH = ctx.commit_hash_batch(v, r)
args = (height, idx, lC, lH, pC, pH, rC, rH, capi.POLICY_PADDING, height, 64)
ctx.verify_entities(*[a[:64] if isinstance(a, np.ndarray) else a for a in args], proofs[:64], verify_seed=seed)   # warm-up
out = {}
for name, env in (("batched", None), ("proof_by_proof", "1")):
    if env:
        os.environ["DAPOL_VERIFY_NO_RLC"] = env
    t0 = time.perf_counter()
    ok = ctx.verify_entities(*args, proofs, verify_seed=seed)
    out[name + "_s"] = time.perf_counter() - t0
    out[name + "_all_verified"] = bool(ok.all())
    os.environ.pop("DAPOL_VERIFY_NO_RLC", None)
bad = proofs.copy()
bad[5, 300] ^= 1
okb = ctx.verify_entities(*args, bad, verify_seed=seed)
print(json.dumps({"config": "DapolProof::verify of %d single-leaf proofs, height %d, 64-bit, padding policy" % (n, height),
                  "tampered_rejected_only": bool(okb[5] == 0 and okb.sum() == n - 1),
                  "entities_per_s_batched": n / out["batched_s"], "entities_per_s_proof_by_proof": n / out["proof_by_proof_s"], **out,
                  "note": "host-inclusive (H2D of %.0f MB of paths and proofs)" % ((pC.nbytes + pH.nbytes + proofs.nbytes) / 1e6)}))
