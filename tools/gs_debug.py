"""Bring-up check of the generator-stationary MSM: each forced-GS setting against the default path's bytes, progress flushed line by line."""
import os, sys
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi

SEED = bytes(range(32))
ctx = capi.Context(0, 32)
print("ctx up", flush=True)
for n_bits, m in ((64, 4), (32, 32), (64, 32)):
    b = 37
    rng = np.random.default_rng(n_bits + m)
    v = rng.integers(0, 2**n_bits if n_bits < 64 else 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = np.arange(b, dtype=np.uint64)
    base = ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes()
    print(n_bits, m, "base ok", flush=True)
    for env in ({"DAPOL_NO_SPLIT": "1"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS_TILE": "4"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_GS_TILE": "64", "DAPOL_CHUNK": "16", "DAPOL_STREAMS": "2"},
                {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_NO_TAIL": "1"}, {"DAPOL_GS": "1", "DAPOL_NO_SPLIT": "1", "DAPOL_CHUNK": "5", "DAPOL_TAIL_N": "32"}):
        print("  ", env, end=" ", flush=True)
        os.environ.update(env)
        try:
            got = ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid).tobytes()
        finally:
            for k in env:
                os.environ.pop(k, None)
        print("same" if got == base else "DIFFERENT", flush=True)
