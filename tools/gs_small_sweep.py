"""The schedule of mid-size calls.  One dapol_range_prove_batch call of b proofs (64-bit, 32 parties), best of 3, under the library's
defaults and with one knob forced at a time: --set=slices (default: latency shapes / whole sweep / 2..16 slices), --set=fs (Fiat-Shamir
kernel shapes), --set=tail (lanes per list of the tail MSM), --set=tile (rows per tile launch).
python tools/gs_small_sweep.py [--set=slices|fs|tail|tile|chunk] b [b ...]"""
import os, sys, time
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
SEED = bytes(range(32))
ctx = capi.Context(0, 32)
n_bits, m = 64, 32
G = {"DAPOL_GS": "1", "DAPOL_SMALL_MAX": "64"}
SETS = {
    # the latency shapes, the sweep whole, the sweep with each list in 2 .. 16 slices side by side
    "slices": [("default", {}), ("latency shapes", {"DAPOL_SMALL_MAX": "8191"})] + [("gs %d slices" % k, dict(G, DAPOL_GS_SLICES=str(k))) for k in (1, 2, 4, 8, 16)],
    "fs": [("default", {}), ("fs lane", {"DAPOL_FS_SHAPE": "0"}), ("fs pair", {"DAPOL_FS_SHAPE": "1"}), ("fs wavefront", {"DAPOL_FS_SHAPE": "2"})],
    "tail": [("default", {})] + [("tail lpl %d" % k, {"DAPOL_TAIL_LPL": str(k)}) for k in (2, 4, 8, 32)],
    "chunk": [("default", {})] + [("chunk %d" % k, {"DAPOL_CHUNK": str(k)}) for k in (16384, 21846, 32768, 43691)],
    "tile": [("default", {})] + [("tile %d" % k, {"DAPOL_GS_TILE": str(k)}) for k in (8, 16, 32, 64)],
}
args = sys.argv[1:]
which = "slices"
if args and args[0].startswith("--set="):
    which = args.pop(0)[6:]
CFGS = SETS[which]
sys.argv[1:] = args
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
