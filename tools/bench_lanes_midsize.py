"""Mid-size calls of a plan with several groups (padding / splitting at aggregation 24 on the height-32 tree: a 32-party -- or a 16- and
an 8-party -- proof + 8 individual proofs per entity): wall ms of ONE dapol_prove_entities call of b entities, with the groups on
lanes (DAPOL_LANES_MAX large) and one after the other.  usage: python tools/bench_lanes_midsize.py b [b ...]"""
import os, sys, time
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi
import bench
height, n = 32, 1 << 14
idx, v, r = bench.synth_inputs(n, height, 0, n)
ctx = capi.Context(0, 32)
tree = capi.Tree(ctx, height, idx, v, r, bench.PAD_SEED)
for b in [int(x) for x in sys.argv[1:]]:
    for pol, name in ((capi.POLICY_PADDING, "padding"), (capi.POLICY_SPLITTING, "splitting")):
        row = []
        for env in ({"DAPOL_NO_LANES": "1"}, {"DAPOL_LANES_MAX": "100000000"}):
            os.environ.update(env)
            best, out = None, None
            for _ in range(4):
                t0 = time.perf_counter()
                out = tree.prove_entities(idx[:b], pol, 24, 64, bench.NONCE_SEED)[2]
                dt = 1e3 * (time.perf_counter() - t0)
                best = dt if best is None else min(best, dt)
            for k in env:
                os.environ.pop(k, None)
            row.append((best, hash(out.tobytes())))
        print("b=%d %s agg 24: one after the other %.2f ms | lanes %.2f ms | same bytes %s" % (b, name, row[0][0], row[1][0], row[0][1] == row[1][1]), flush=True)
