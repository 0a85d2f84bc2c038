import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref as R
from dapol_amd import capi
from test_gpu_parity import _rand_leaves, SEED
ctx = capi.Context(0, 32)
height, policy, agg = 8, 0, 8
rng = np.random.default_rng(height * 31 + agg)
idx, v, r = _rand_leaves(rng, height, 12, vmax=8)
tr = capi.Tree(ctx, height, idx, v, r, SEED)
rC, rH, _, _ = tr.root()
pC, pH, proofs = tr.prove_entities(idx, policy, agg, 8, SEED)
lC, lH = ctx.commit_hash_batch(v, r)
ok = ctx.verify_entities(height, idx, lC, lH, pC, pH, rC, rH, policy, agg, 8, proofs, verify_seed=SEED)
print("verify_entities", list(ok))
pathok = [R.verify_path(rC, rH, lC[k].tobytes(), lH[k].tobytes(), int(idx[k]), [(pC[k, s].tobytes(), pH[k, s].tobytes()) for s in range(height)]) for k in range(12)]
print("oracle path   ", [int(x) for x in pathok])
rok = ctx.range_verify_batch(8, 8, proofs, pC, verify_seed=SEED)
print("range batch   ", list(rok))
rok2 = [int(R.range_verify(proofs[k].tobytes(), [pC[k, s].tobytes() for s in range(8)], 8)) for k in range(12)]
print("oracle range  ", rok2)
