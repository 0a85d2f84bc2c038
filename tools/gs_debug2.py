"""Bring-up: generator-stationary against proof-stationary bytes at growing batch sizes (n = 64, m = 32)."""
import os, sys, hashlib
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
SEED = bytes(range(32))
ctx = capi.Context(0, 32)
n_bits, m = 64, 32
for b in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(b)
    v = rng.integers(0, 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = np.arange(b, dtype=np.uint64)
    out = {}
    for gs in ("0", "1"):
        print(b, "GS=" + gs, end=" ", flush=True)
        os.environ["DAPOL_GS"] = gs
        out[gs] = hashlib.sha256(ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes()).hexdigest()
        print(out[gs][:16], flush=True)
    print(b, "same" if out["0"] == out["1"] else "DIFFERENT", flush=True)
