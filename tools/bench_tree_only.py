"""The headline tree alone: python tools/bench_tree_only.py [log2_entities=20] [builds=3] -> device ms of each dapol_workload_build
(HIP events around the build).  What the tree's kernel trace / PMC passes run (tools/calls_r6d.sh)."""
import os, sys
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
import bench
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, height = 1 << lg, 32
idx, v, r = bench.synth_inputs(n, height, 0, n)
ctx = capi.Context(0, 1, options=capi.Options(high_half_rows=-1))
w = capi.Workload(ctx, height, idx, v, r)
for _ in range(reps):
    root, st = w.build(bench.PAD_SEED)
    print("tree_ms %.3f root %s" % (st.tree_ms, root[0].hex()[:16]), flush=True)
