#!/usr/bin/env python3
"""Tree-build time over a sweep of (height, entities): python tools/bench_build_sweep.py [h:n ...]  (wall ms per build, best of 5)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dapol_amd import capi  # noqa: E402

cases = [tuple(map(int, a.split(":"))) for a in sys.argv[1:]] or [(16, n) for n in (1024, 2048, 3000, 4096, 6000, 8192, 16384)] + [(32, 4096), (32, 1 << 16)]
ctx = capi.Context(0, 32)
for h, n in cases:
    idx, v, r = bench.synth_inputs(1 << (n - 1).bit_length(), h, 0, n)
    w = capi.Workload(ctx, h, idx, v, r)
    w.build(bench.PAD_SEED)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        _, st = w.build(bench.PAD_SEED)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("h=%d n=%d: best %.3f ms, median %.3f ms, device %.3f ms" % (h, n, min(ts), sorted(ts)[2], st.tree_ms), flush=True)
    w.close()
