#!/usr/bin/env python3
"""Randomised soak of dapol_tree_update: random trees (height 4-40, 1-3,000 leaves; every third one a rank's SHARD of a taller tree:
1-3 prefix bits above it, global indexes), sequences of twelve updates that mix replaced and new
leaves (also duplicates inside a batch, neighbours, batches that fall back to the rebuild); after every update root and node counts, and
after every fourth every node of every level, against a fresh dapol_tree_build of the same liabilities.  usage: tools/soak_tree_update.py [sequences] [seed]"""
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import sys
import collections

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dapol_amd import capi  # noqa: E402

seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
ctx = capi.Context(0, 8)
SEED = bytes(range(32))
paths = collections.Counter()


def rand_r(k):
    r = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
    r[:, 31] &= 0x7F if rng.integers(0, 2) else 0x0F
    return r


for s in range(seqs):
    height = int(rng.integers(4, 41))
    space = 1 << height
    sb = int(rng.integers(1, 4)) if s % 3 == 2 else 0                           # a shard tree: `height` local levels under sb prefix bits
    base = int(rng.integers(0, 1 << sb)) << height
    total_h = height + sb
    if total_h > 62:
        sb, base, total_h = 0, 0, height
    mk = lambda keys_local: capi.Tree(ctx, total_h, keys_local + np.uint64(base), np.array([cur[int(i)][0] for i in keys_local], np.uint64),
                                      np.stack([cur[int(i)][1] for i in keys_local]), SEED, shard_bits=sb)
    n = int(rng.integers(1, min(3000, space // 2) + 1))
    idx = np.sort(rng.choice(space, size=n, replace=False).astype(np.uint64)) if space <= (1 << 24) else np.unique(rng.integers(0, space, size=n, dtype=np.uint64))
    cur = {int(i): (int(rng.integers(0, 1 << 40)), rand_r(1)[0]) for i in idx}
    keys = np.array(sorted(cur), np.uint64)
    tr = mk(keys)
    for step in range(12):
        k_new, k_old = int(rng.integers(0, 21)), int(rng.integers(0, 21))
        if len(cur) + k_new > space // 2:
            k_new = 0
        new = []
        while len(new) < k_new:
            x = int(rng.integers(0, space))
            if rng.integers(0, 4) == 0 and cur:                                   # right beside an existing leaf
                x = (int(rng.choice(np.array(sorted(cur), np.uint64))) ^ 1) % space
            if x not in cur and x not in new:
                new.append(x)
        old = [int(x) for x in rng.choice(np.array(sorted(cur), np.uint64), size=min(k_old, len(cur)), replace=False)] if k_old else []
        upd = new + old
        if not upd:
            continue
        if rng.integers(0, 3) == 0:
            upd = upd + [upd[0]]                                                  # a duplicate inside the batch: the last one wins
        order = rng.permutation(len(upd))
        ui = np.array(upd, np.uint64)[order]
        uv = rng.integers(0, 1 << 40, size=len(upd), dtype=np.uint64)
        ur = rand_r(len(upd))
        tr.update(ui + np.uint64(base), uv, ur)
        paths[tr.last_update_path()] += 1
        for a, b, c in zip(ui, uv, ur):
            cur[int(a)] = (int(b), c)
        keys = np.array(sorted(cur), np.uint64)
        want = mk(keys)
        assert tr.root() == want.root() and tr.node_count() == want.node_count(), (s, step, height, len(cur))
        if step % 4 == 3:
            for level in range(height + 1):
                for a, b in zip(tr.level_nodes(level), want.level_nodes(level)):
                    assert np.array_equal(a, b), (s, step, level)
        want.close()
    tr.close()
    if (s + 1) % 10 == 0:
        print("%d sequences ok  (paths so far: rebuild %d, replaced %d, inserted %d, both %d)" % (s + 1, paths[0], paths[1], paths[2], paths[3]), flush=True)
print("soak: %d sequences of 12 updates, every tree equal to a fresh build (update paths: rebuild %d, replaced in place %d, inserted in place %d, both %d)"
      % (seqs, paths[0], paths[1], paths[2], paths[3]))
