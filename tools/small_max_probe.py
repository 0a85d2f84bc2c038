"""Which shape for calls of 4,096-8,000 proofs?  One dapol_range_prove_batch call (64-bit, 32 parties), best of 3: the library's defaults,
the latency shapes extended to 8,191 proofs, the generator-stationary sweep from 4,097 proofs (64-row tiles).  (profiles/archive/r04h_small_max_probe.txt
was taken with the default bound at 4,096.)"""
import os, sys, time
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
SEED = bytes(range(32)); ctx = capi.Context(0, 32); n_bits, m = 64, 32
for b in (4096, 5000, 6144, 7168, 8000):
    rng = np.random.default_rng(b)
    v = rng.integers(0, 2**63, size=(b, m), dtype=np.uint64); r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8); r[:, :, 31] &= 0x0F
    sid = np.arange(b, dtype=np.uint64); out = []
    for name, env in (("default", {}), ("small_max 8191", {"DAPOL_SMALL_MAX": "8191"}), ("gs tile 64 from 4097", {"DAPOL_GS": "1", "DAPOL_GS_TILE": "64"})):
        os.environ.update(env); best = 1e9
        for rep in range(4):
            t0 = time.perf_counter(); ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid); dt = time.perf_counter() - t0
            if rep: best = min(best, dt)
        for k in env: os.environ.pop(k, None)
        out.append("%s %7.1f ms" % (name, best * 1e3))
    print("b=%6d  " % b + "   ".join(out), flush=True)
