"""One height-24 splitting-policy inclusion proof (a 16-party and an 8-party sub-proof on two lanes): python tools/lanes_timeline.py
-> under `rocprofv3 --kernel-trace`, tools/calls/calls_r6j.sh turns the trace of the LAST call into a per-queue timeline."""
import os, sys, time
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
import bench
height, n = 24, 1024
idx, v, r = bench.synth_inputs(n, height, 0, n)
ctx = capi.Context(0, 32)
tree = capi.Tree(ctx, height, idx, v, r, bench.PAD_SEED)
for k in range(4):
    t0 = time.perf_counter()
    tree.prove_entities(idx[k:k + 1], capi.POLICY_SPLITTING, height, 64, bench.NONCE_SEED)
    print("prove %.3f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
