#!/usr/bin/env python3
"""Quick on-GPU parity run against oracle/pyref.py and tests/golden (used while bringing kernels up)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as R  # noqa: E402
from dapol_amd import capi  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
t0 = time.time()
ctx = capi.Context(0, 32)
print("ctx create %.2fs" % (time.time() - t0), flush=True)
kat = json.load(open(os.path.join(G, "kat.json")))
print("B ok", ctx.generator(0).hex() == kat["B"], "Bb ok", ctx.generator(1).hex() == kat["B_blinding"])
ok = all(ctx.generator(2, j, i).hex() == kat["gens_n8_m2"]["G"][j * 8 + i] and ctx.generator(3, j, i).hex() == kat["gens_n8_m2"]["H"][j * 8 + i]
         for j in range(2) for i in range(8))
print("gens ok", ok)
Gs, Hs = R.bp_gens(64, 2)
print("gens64 ok", all(ctx.generator(2, 1, i) == Gs[64 + i].compress() and ctx.generator(3, 1, i) == Hs[64 + i].compress() for i in (0, 7, 63)))
cm = json.load(open(os.path.join(G, "commit.json")))
v = np.array([c["v"] for c in cm], np.uint64)
r = np.array([list(bytes.fromhex(c["r"])) for c in cm], np.uint8)
C, H = ctx.commit_hash_batch(v, r)
bad = [i for i, c in enumerate(cm) if C[i].tobytes().hex() != c["C"] or H[i].tobytes().hex() != c["H"]]
print("commit bad", bad, flush=True)
trees = json.load(open(os.path.join(G, "trees.json")))
for t in trees:
    idx = np.array([l["idx"] for l in t["leaves"]], np.uint64)
    vv = np.array([l["v"] for l in t["leaves"]], np.uint64)
    rr = np.array([list(bytes.fromhex(l["r"])) for l in t["leaves"]], np.uint8)
    tr = capi.Tree(ctx, t["height"], idx, vv, rr, bytes.fromhex(t["pad_seed"]))
    Cr, Hr, vr, rrt = tr.root()
    ok = (Cr.hex() == t["root"]["C"] and Hr.hex() == t["root"]["H"] and vr == t["root"]["v"] and rrt.hex() == t["root"]["r"])
    nreal, npad = tr.node_count()
    okn = nreal + npad == t["node_count"]
    lv_ok = True
    if "levels" in t:
        for k, lev in enumerate(t["levels"]):
            i_, v_, r_, C_, H_, p_ = tr.level_nodes(k)
            got = {int(i_[j]): (int(v_[j]), r_[j].tobytes().hex(), C_[j].tobytes().hex(), H_[j].tobytes().hex(), bool(p_[j])) for j in range(len(i_))}
            exp = {n["idx"]: (n["v"], n["r"], n["C"], n["H"], n["pad"]) for n in lev}
            if got != exp:
                lv_ok = False
                print("  level", k, "mismatch", len(got), len(exp))
    pk = sorted(int(k) for k in t["paths"])
    pC, pH, pv, pr = tr.paths(np.array(pk, np.uint64))
    p_ok = all(pC[a, s].tobytes().hex() == t["paths"][str(li)][s]["C"] and pH[a, s].tobytes().hex() == t["paths"][str(li)][s]["H"]
               and int(pv[a, s]) == t["paths"][str(li)][s]["v"] and pr[a, s].tobytes().hex() == t["paths"][str(li)][s]["r"]
               for a, li in enumerate(pk) for s in range(t["height"]))
    print("tree h=%d n=%d root %s count %s levels %s paths %s" % (t["height"], len(idx), ok, okn, lv_ok, p_ok), flush=True)
rg = json.load(open(os.path.join(G, "range.json")))
for c in rg:
    n, m = c["n"], c["m"]
    vv = np.array(c["values"], np.uint64).reshape(1, m)
    rr = np.array([list(bytes.fromhex(b)) for b in c["blindings"]], np.uint8).reshape(1, m, 32)
    t0 = time.time()
    pr = ctx.range_prove_batch(n, m, vv, rr, nonce_seed=bytes.fromhex(c["nonce_seed"]), stream_id=[c["stream_id"]])
    got = pr[0].tobytes().hex()
    if got == c["proof"]:
        print("range n=%d m=%d OK (%.2fs)" % (n, m, time.time() - t0), flush=True)
    else:
        exp = c["proof"]
        first = next(i for i in range(0, len(exp), 64) if got[i:i + 64] != exp[i:i + 64]) // 64
        print("range n=%d m=%d MISMATCH first differing 32B element %d of %d" % (n, m, first, len(exp) // 64), flush=True)
dp = json.load(open(os.path.join(G, "dapol.json")))
for c in dp:
    idx = np.array([l["idx"] for l in c["leaves"]], np.uint64)
    vv = np.array([l["v"] for l in c["leaves"]], np.uint64)
    rr = np.array([list(bytes.fromhex(l["r"])) for l in c["leaves"]], np.uint8)
    tr = capi.Tree(ctx, c["height"], idx, vv, rr, bytes.fromhex(c["pad_seed"]))
    pol = capi.POLICY_PADDING if c["policy"] == "padding" else capi.POLICY_SPLITTING
    pC, pH, out = tr.prove_entities([c["leaf"]], pol, c["agg"], c["n_bits"], bytes.fromhex(c["nonce_seed"]))
    exp = "".join(c["aggregated"]) + "".join(c["individual"])
    print("dapol h=%d %s agg=%d: %s" % (c["height"], c["policy"], c["agg"], out[0].tobytes().hex() == exp), flush=True)
# throughput smoke: 4096 proofs of m=32,n=64
B = int(os.environ.get("QUICK_B", "4096"))
rng = np.random.default_rng(1)
vv = rng.integers(0, 2**32, size=(B, 32), dtype=np.uint64)
rr = rng.integers(0, 256, size=(B, 32, 32), dtype=np.uint8)
rr[:, :, 31] &= 0x0f
t0 = time.time()
pr = ctx.range_prove_batch(64, 32, vv, rr, nonce_seed=bytes(range(32)), stream_id=np.arange(B, dtype=np.uint64))
dt = time.time() - t0
print("prove %d x (n=64,m=32): %.2fs -> %.1f proofs/s (host-inclusive)" % (B, dt, B / dt), flush=True)
