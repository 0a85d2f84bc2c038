"""Generator-stationary against proof-stationary MSM for calls between the small-call bound and one full chunk:
one dapol_range_prove_batch call of b proofs (n = 64 bits, m = 32 parties), best of 3, both settings.  python tools/gs_midsize_sweep.py b [b ...]"""
import os, sys, time
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
    row = {}
    for gs in ("0", "1"):
        os.environ["DAPOL_GS"] = gs
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid)
            dt = time.perf_counter() - t0
            if rep:
                best = min(best, dt)
        row[gs] = best
    print("b=%6d  proof-stationary %8.1f ms  generator-stationary %8.1f ms  (%+.1f %%)" % (b, row["0"] * 1e3, row["1"] * 1e3, 100 * (row["0"] / row["1"] - 1)), flush=True)
