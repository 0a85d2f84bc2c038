"""The N > 1 path on CPU: two gloo processes all-gather their subtree-root records and each merges the replicated top
level; the merge primitive is the oracle here (the product uses dapol_merge_batch on the GPU).  Checks that the
exchange + top-level logic reproduces the full tree's root and the right upper siblings."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _oracle_merge(R):
    def merge(CL, HL, CR, HR, vL, rL, vR, rR):
        n = len(vL)
        C, H, v, r = np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8), np.zeros(n, np.uint64), np.zeros((n, 32), np.uint8)
        for i in range(n):
            mk = lambda c, h, vv, rr: R.Node(int(vv), int.from_bytes(bytes(rr), "little"), R.decompress(bytes(c)), bytes(c), bytes(h))
            p = R.node_merge(mk(CL[i], HL[i], vL[i], rL[i]), mk(CR[i], HR[i], vR[i], rR[i]))
            C[i], H[i], v[i], r[i] = np.frombuffer(p.C, np.uint8), np.frombuffer(p.H, np.uint8), p.v, np.frombuffer(R.scalar_bytes(p.r), np.uint8)
        return C, H, v, r
    return merge


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import pyref as R
    from dapol_amd import sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, seed = 5, bytes(range(32))
    leaves_all = [(i, R.node_new(3 + i, 1000 + i)) for i in (1, 6, 9, 13, 18, 29)]
    full = R.Tree(H, leaves_all, seed)
    # this rank's subtree = the child of the root with prefix `rank`: a height-4 tree over the global indexes
    mine = [(i, nd) for i, nd in leaves_all if (i >> (H - 1)) == rank]
    node = full.levels[H - 1][rank]                      # what dapol_tree_build_shard would return as the subtree root
    sub_root = (node.C, node.H, node.v, R.scalar_bytes(node.r))
    assert len(mine) >= 1
    buf = sharded.exchange_records(dist, torch, sharded.pack_record(sub_root), world, "cpu")
    recs = sharded.unpack_records(buf, world)
    root, upper = sharded.top_levels(None, recs, rank, merge=_oracle_merge(R))
    ok = root == (full.root.C, full.root.H, full.root.v, R.scalar_bytes(full.root.r))
    sib = full.levels[H - 1][rank ^ 1]
    ok &= upper[0][0].tobytes() == sib.C and upper[1][0].tobytes() == sib.H and int(upper[2][0]) == sib.v
    ok &= upper[3][0].tobytes() == R.scalar_bytes(sib.r)
    # and the upper sibling is exactly the root-side-first sibling of every leaf of this rank
    for i, _ in mine:
        ok &= full.path_siblings(i)[0].C == sib.C
    cs = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(cs)
    ok &= int(cs.item()) == 3
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
