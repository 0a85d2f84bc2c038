"""dapol_build_leaf_nodes near the sparsity bound Dapol::new allows (2^height = 2 n) and far from it: host-inclusive time, second of two
calls (profiles/archive/r04l_leaf_bound.txt: the one-lane collision resolution; r04m_leaf_bound.txt: the claim / settle rounds that replaced it)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
ctx = capi.Context(0, 8)
for lg, h in ((15, 16), (17, 18), (20, 21), (20, 24), (20, 32)):
    n = 1 << lg
    ids = np.char.add("id-", np.char.zfill(np.arange(n).astype("U8"), 8)).astype("S11")
    off = (np.arange(n + 1, dtype=np.uint64) * 11).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint64)
    packed = ids.tobytes()
    for rep in range(2):
        t0 = time.perf_counter()
        out = ctx.build_leaf_nodes_packed(packed, off, packed, off, vals, b"seed", h)
        dt = time.perf_counter() - t0
    li = out["leaf_idx"]
    print("n=2^%d height=%d: %.1f ms  distinct=%s" % (lg, h, dt * 1e3, bool(np.all(li[1:] > li[:-1]))), flush=True)
