import os, sys, time
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import numpy as np
sys.path.insert(0, "/root/repo")
from dapol_amd import capi
b = int(sys.argv[1])
n, height, seed = 1 << 16, 32, bytes(range(32))
rng = np.random.default_rng(3)
idx = np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n)
v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
r[:, 31] &= 0x0F
ctx = capi.Context(0, 32)
tree = capi.Tree(ctx, height, idx, v, r, seed)
sel = idx[:: n // b][:b]
for _ in range(3):
    t0 = time.perf_counter()
    tree.prove_entities(sel, capi.POLICY_PADDING, height, 64, seed)
    print(b, 1e3 * (time.perf_counter() - t0), flush=True)
