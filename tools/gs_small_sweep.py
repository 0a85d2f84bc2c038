"""Where should the generator-stationary sweep start?  One dapol_range_prove_batch call of b proofs (64-bit, 32 parties), best of 3,
under the library's defaults and with the sweep forced on (small-call bound lowered to 1,024) at several tile sizes.
python tools/gs_small_sweep.py b [b ...]"""
import os, sys, time
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
SEED = bytes(range(32))
ctx = capi.Context(0, 32)
n_bits, m = 64, 32
G = {"DAPOL_GS": "1", "DAPOL_SMALL_MAX": "64"}
CFGS = [("default", {}), ("latency shapes", {"DAPOL_SMALL_MAX": "8191"})] + [("gs %d slices" % k, dict(G, DAPOL_GS_SLICES=str(k))) for k in (1, 2, 4, 8, 16)]
if os.environ.get("SWEEP_FS"):
    CFGS = [("default", {})] + [("tile %d" % k, {"DAPOL_GS_TILE": str(k)}) for k in (8, 16, 32, 64)]
for b in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(b)
    v = rng.integers(0, 2**63, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = np.arange(b, dtype=np.uint64)
    out = []
    for name, env in CFGS:
        os.environ.update(env)
        best = 1e9
        try:
            for rep in range(4):
                t0 = time.perf_counter()
                ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid)
                dt = time.perf_counter() - t0
                if rep:
                    best = min(best, dt)
        finally:
            for k in env:
                os.environ.pop(k, None)
        out.append("%s %7.1f ms" % (name, best * 1e3))
    print("b=%6d  " % b + "   ".join(out), flush=True)
