"""Multi-GPU sharding of the proving path: one process per GPU, each owning one top-level
subtree of the sparse Merkle tree.  The only exchange step is an all-gather of the G subtree-root records
(C, H, v, r: 104 bytes each); every rank then merges the log2 G replicated top levels itself and proves its own
entities.  A final all-reduce sums the per-rank proof checksums.

The exchange and the reduce are RCCL calls INSIDE libdapol_hip.so (dapol_shard_exchange / dapol_comm_allreduce_u64,
include/dapol_hip.h): this module only carries the 128-byte RCCL id from rank 0 to the others (torch.distributed
broadcast) and calls them.  The torch.distributed all-gather below remains for hosts without RCCL between the ranks
(gloo: CPU tests, two ranks sharing one GPU) and as the fallback if the library's communicator cannot be created.

Indices stay global everywhere, so padding-node seeds and nonce stream ids -- and therefore every byte -- equal
the single-GPU result (tests/test_sharded.py emulates G shards on one device and checks exactly that)."""
import time

import numpy as np

from . import capi

RECORD_BYTES = 32 + 32 + 8 + 32


def pack_record(root):
    C, H, v, r = root
    return np.frombuffer(C + H + int(v).to_bytes(8, "little") + r, np.uint8).copy()


def unpack_records(buf, g):
    buf = np.asarray(buf, np.uint8).reshape(g, RECORD_BYTES)
    C, H = buf[:, :32].copy(), buf[:, 32:64].copy()
    v = np.array([int.from_bytes(buf[i, 64:72].tobytes(), "little") for i in range(g)], np.uint64)
    r = buf[:, 72:104].copy()
    return C, H, v, r


def exchange_records(dist, torch, record, world, device):
    """All-gather of the per-rank subtree-root records, in rank order: the one layer-boundary exchange of the path."""
    mine = torch.from_numpy(np.ascontiguousarray(record, np.uint8)).to(device)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return torch.stack(allr).cpu().numpy()


def top_levels(ctx, records, rank, merge=None):
    """Merges the G subtree roots up to the global root (Mergeable::merge on the GPU, dapol_merge_batch) and returns
    (root record, upper siblings of `rank`, root side first).  G must be a power of two and every shard non-empty.
    `merge` overrides the merge primitive (CPU tests plug the oracle in; the product path always uses the GPU)."""
    merge = merge or ctx.merge_batch
    C, H, v, r = records
    g = C.shape[0]
    assert g & (g - 1) == 0
    sib = []
    pos = rank
    while g > 1:
        s = pos ^ 1
        sib.append((C[s].copy(), H[s].copy(), int(v[s]), r[s].copy()))
        C, H, v, r = merge(C[0::2], H[0::2], C[1::2], H[1::2], v[0::2], r[0::2], v[1::2], r[1::2])
        g //= 2
        pos //= 2
    sib.reverse()
    upper = (np.array([x[0] for x in sib], np.uint8).reshape(-1, 32), np.array([x[1] for x in sib], np.uint8).reshape(-1, 32),
             np.array([x[2] for x in sib], np.uint64), np.array([x[3] for x in sib], np.uint8).reshape(-1, 32))
    return (C[0].tobytes(), H[0].tobytes(), int(v[0]), r[0].tobytes()), upper


def agreement_group(dist, comm_device, torch=None):
    """The process group the ranks AGREE on (ok flags of the library's collectives) -> (group, device of its tensors).  It must not
    ride the transport that may have just failed: when the default group is RCCL (comm_device "cuda") this is a side group over
    gloo / TCP with CPU tensors; a default group that already is gloo serves as it is (None).  Collective: every rank calls it at
    the same point.  If the side group cannot be made (a torch build without gloo), the agreement rides the default group with
    device tensors -- weaker (torch's own RCCL communicator, not the library's, carries it) but never a rank deciding alone.
    Precondition: dist.new_group is itself collective -- every rank must RETURN from it, with a group or with an
    exception, for the all-reduce below to be reached by all of them; a rank that dies inside new_group leaves its peers in torch's
    own rendezvous timeout, which nothing at this level can shorten.  What is handled here is the case that it returns everywhere
    but raised on some ranks (no gloo in the build, a port it could not bind)."""
    if comm_device != "cuda":
        return None, "cpu"
    group = None
    try:
        group = dist.new_group(backend="gloo")
    except Exception:
        group = None
    # new_group can fail on ONE rank only; ranks that then agreed over different groups would not be agreeing at all.  The outcome is
    # itself agreed over the default group: the side group is used only if every rank has it.
    if torch is None:
        import torch
    flag = torch.tensor([1 if group is not None else 0], dtype=torch.int64, device=comm_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return group, "cpu"
    return None, comm_device


def all_agree(dist, torch, group, ok, device="cpu"):
    """all-reduce MIN of one flag over the agreement group: True iff every rank said ok."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def create_library_comm(ctx, rank, world, dist, torch, device, timeout_s=90.0, group=None, agree=None):
    """The RCCL communicator INSIDE libdapol_hip.so for ranks that already share a torch.distributed group: rank 0 draws the
    128-byte id, a torch.distributed broadcast carries it, and every rank creates its end NON-BLOCKING with a deadline
    (dapol_comm_create_timeout: ncclCommInitRankConfig(blocking = 0) polled with ncclCommGetAsyncError; a communicator that has
    not come up by the deadline is aborted inside the call, not abandoned).  The ranks then AGREE on the outcome (all-reduce MIN,
    over the agreement group when one is given) before anybody tears anything down: if any rank failed, the ranks that did get a
    communicator abort it (ncclCommAbort -- never a destroy that would wait for a peer that is gone) and all of them get
    (None, None, reason).  Returns (comm or None, ncclCommCount or None, error text or None)."""
    comm, ranks, err, ok = None, None, None, 1
    try:
        uid = capi.comm_unique_id() if rank == 0 else bytes(capi.COMM_ID_BYTES)
        buf = torch.from_numpy(np.frombuffer(uid, np.uint8).copy()).to(device)
        dist.broadcast(buf, src=0)
        comm = capi.Comm(ctx, buf.cpu().numpy().tobytes(), rank, world, timeout_s=timeout_s)
        ranks = comm.count()
        if ranks != world:
            raise capi.DapolError(10, "ncclCommCount says %d ranks, expected %d" % (ranks, world))
    except Exception as e:
        err, ok = repr(e), 0
    if agree is not None:
        everybody = agree(bool(ok))
    else:
        flag = torch.tensor([ok], dtype=torch.int64, device="cpu" if group is not None else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        everybody = int(flag.item()) == 1
    if everybody:
        return comm, ranks, None
    if comm is not None:
        comm.abort()
    return None, None, err or "another rank could not create its communicator"


class ShardTransport:
    """The two collectives of the sharded path -- the all-gather of the subtree-root records and the final reduce -- with their
    fallback.  The first choice is the RCCL communicator inside libdapol_hip.so (`comm`); torch.distributed carries them otherwise
    (gloo: CPU tests, two ranks sharing one GPU) and after the library's communicator has failed.

    A failing collective does NOT fail on every rank by itself: the deadlines are per-rank wall clocks, and an asynchronous RCCL
    error or a timeout can hit one rank while its peers complete the very same all-gather.  If each rank decided alone, one would
    sit in a torch all_gather while the others went on to the library's all-reduce on a communicator the first has aborted --
    mismatched collectives, i.e. the hang the deadline exists to prevent.  So after EVERY library collective the ranks all-reduce
    (MIN) an ok flag over the agreement group (gloo, CPU tensors: never the transport in question) BEFORE the next collective;
    only if somebody failed do they all abort their communicators -- never destroy: a destroy waits for peers -- and redo that
    collective over torch.distributed, together.  One small gloo all-reduce per collective, two per step (tens of microseconds
    against a step of seconds)."""

    def __init__(self, ctx, rank, world, dist, torch, comm_device="cuda", merge=None):
        self.ctx, self.rank, self.world, self.dist, self.torch, self.comm_device, self.merge = ctx, rank, world, dist, torch, comm_device, merge
        self.comm, self.comm_ranks, self.comm_error = None, None, None
        self.group, self.group_device = None, "cpu"   # agreement group (None: the default group) and the device of its flag tensors
        self.agreements = 0                       # ok-flag all-reduces done so far (diagnostics / tests)
        self.path = "none (single GPU)" if world == 1 else self._torch_path()

    def _torch_path(self, after_failure=False):
        return "torch.distributed (%s)%s" % ("RCCL" if self.comm_device == "cuda" else "gloo", ", after the library's collective failed" if after_failure else "")

    def create_comm(self, timeout_s=90.0, even_alone=False):
        """Collective: every rank calls it.  Creates the agreement group, then the library's communicator.
        even_alone: also with ONE rank (a test hook: a 1-GPU box then exercises the gloo side group beside torch's RCCL group, the
        id broadcast and the non-blocking creation exactly as N ranks would)."""
        import os
        if (self.world == 1 and not even_alone) or self.comm_device != "cuda" or os.environ.get("DAPOL_EXCHANGE", "").lower() == "torch":
            return
        self.group, self.group_device = agreement_group(self.dist, self.comm_device, self.torch)
        self.comm, self.comm_ranks, self.comm_error = create_library_comm(self.ctx, self.rank, self.world, self.dist, self.torch, self.comm_device,
                                                                          timeout_s, agree=self.agree)
        if self.comm is not None:
            self.path = "libdapol_hip.so (RCCL inside the library)"

    def agree(self, ok):
        self.agreements += 1
        return all_agree(self.dist, self.torch, self.group, ok, self.group_device)

    def drop_comm(self, err):
        """Every rank has agreed that a collective of the library's communicator failed somewhere: abort it, carry on over torch."""
        self.comm_error = repr(err) if err is not None else "the collective failed on another rank"
        try:
            self.comm.abort()
        except Exception:
            pass
        self.comm, self.comm_ranks = None, None
        self.path = self._torch_path(after_failure=True)

    def _library(self, call):
        """Runs one collective of the library's communicator and the agreement after it.  -> (True, result) when EVERY rank's call
        succeeded; (False, None) after the ranks have dropped their communicators together."""
        res, err = None, None
        try:
            res = call()
        except Exception as e:          # ANY error on this rank (a ctypes / numpy error, not only a DapolError) must reach the
            err = e                     # agreement: the peers are about to wait in it, and would sit there until gloo's own timeout
        except BaseException:
            # KeyboardInterrupt / SystemExit: this rank is LEAVING.  It must not enter another collective on its way out (it would
            # block in it until the peers arrive or the timeout fires -- ADVICE r5): abort the communicator, which the peers' next
            # library collective notices by its deadline, and go.
            try:
                self.comm.abort()
            except Exception:
                pass
            self.comm, self.comm_ranks = None, None
            raise
        agreed = self.agree(err is None)
        if err is not None and not isinstance(err, capi.DapolError):
            self.drop_comm(err)         # the peers have just learnt that this rank failed and drop theirs; this rank does not go on
            raise err
        if agreed:
            return True, res
        self.drop_comm(err)
        return False, None

    def exchange(self, root):
        """All-gather of the G subtree-root records + the replicated top levels -> (global root record, upper siblings of this rank)."""
        if self.world == 1:
            return root, None
        if self.comm is not None:
            ok, res = self._library(lambda: self.comm.exchange(root))                                 # RCCL over xGMI, in the library
            if ok:
                return res
        buf = exchange_records(self.dist, self.torch, pack_record(root), self.world, self.comm_device)
        return top_levels(self.ctx, unpack_records(buf, self.world), self.rank, merge=self.merge)

    def reduce_u64(self, word, op="sum"):
        """The final reduce of one 64-bit word: wrapping sum (proof-transcript checksums) or min (AND of 0/1 verdicts)."""
        if self.world == 1:
            return int(word)
        if self.comm is not None:
            ok, res = self._library(lambda: int(self.comm.allreduce([int(word)], capi.REDUCE_SUM if op == "sum" else capi.REDUCE_MIN)[0]))
            if ok:
                return res
        t = self.torch
        if op == "min":
            x = t.tensor([int(word)], dtype=t.int64, device=self.comm_device)
            self.dist.all_reduce(x, op=self.dist.ReduceOp.MIN)
            return int(x.item())
        cs = t.tensor([int(word) & 0xFFFFFFFF, int(word) >> 32], dtype=t.int64, device=self.comm_device)    # two 32-bit halves: no int64 overflow
        self.dist.all_reduce(cs)
        return (int(cs[0].item()) + (int(cs[1].item()) << 32)) & 0xFFFFFFFFFFFFFFFF

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


class ShardedProver:
    """bench.py's step: build this rank's subtree, exchange roots, prove this rank's entities."""

    def __init__(self, ctx, height, leaf_idx, v, r32, rank=0, world=1, dist=None, torch=None, comm_device="cuda"):
        assert world & (world - 1) == 0, "number of GPUs must be a power of two"
        self.ctx, self.height, self.rank, self.world, self.dist, self.torch = ctx, height, rank, world, dist, torch
        self.comm_device = comm_device            # "cuda": RCCL over xGMI; "cpu": gloo (tests)
        self.shard_bits = world.bit_length() - 1
        self.idx = np.ascontiguousarray(leaf_idx, np.uint64)
        # A rank without liabilities (possible with hash-derived indexes and few entities) contributes the padding node that
        # stands at its subtree's root position, Paddable::padding at (height - shard_bits, rank), and proves nothing.
        self.w = capi.Workload(ctx, height, leaf_idx, v, r32, shard_bits=self.shard_bits) if len(self.idx) else None
        self.upper = None
        self.root = None
        self.phases = None                        # host-clock milliseconds of the last step's phases + the library's collective timings
        self.transport = ShardTransport(ctx, rank, world, dist, torch, comm_device)
        self.transport.create_comm()

    # the transport's state under the names bench.py and the tests read
    comm = property(lambda self: self.transport.comm, lambda self, c: setattr(self.transport, "comm", c))
    comm_ranks = property(lambda self: self.transport.comm_ranks)
    comm_error = property(lambda self: self.transport.comm_error)

    @property
    def exchange_path(self):
        t = self.transport
        if self.world == 1:
            return t.path
        if t.comm is not None:
            return "dapol_shard_exchange (ncclAllGather inside libdapol_hip.so)"
        return t.path.replace("torch.distributed (", "torch.distributed all_gather (")

    PHASE_KEYS = ("build_ms", "exchange_ms", "prove_ms", "reduce_ms", "step_ms", "allgather_us", "top_levels_us", "allreduce_us")

    def step(self, pad_seed, nonce_seed, n_bits=64):
        """One step.  Leaves in self.phases what each part took on THIS rank's host clock (every part ends with the host waiting
        for the device, so these are device-inclusive) and, when the library's communicator carried the collectives, their device
        times from HIP events inside the library (dapol_comm_timing_get: the all-gather's time includes the wait for the slowest
        rank; the smallest value over the ranks is the collective's own latency)."""
        t0 = time.perf_counter()
        if self.w is None:
            C, H, r = self.ctx.padding_nodes(pad_seed, [self.height - self.shard_bits], [self.rank])
            root, st = (C[0].tobytes(), H[0].tobytes(), 0, r[0].tobytes()), capi.WorkloadStats()
            t1 = time.perf_counter()
            self.root, self.upper = self.transport.exchange(root)
            t2 = t3 = time.perf_counter()
        else:
            root, st = self.w.build(pad_seed)
            t1 = time.perf_counter()
            self.root, self.upper = self.transport.exchange(root)
            t2 = time.perf_counter()
            st = self.w.prove(nonce_seed, n_bits, upper=self.upper, stats=st)
            t3 = time.perf_counter()
        # final reduce of the aggregated proof transcript checksum (wrapping 64-bit sum)
        st.checksum = self.transport.reduce_u64(st.checksum, "sum")
        t4 = time.perf_counter()
        ph = {"build_ms": 1e3 * (t1 - t0), "exchange_ms": 1e3 * (t2 - t1), "prove_ms": 1e3 * (t3 - t2), "reduce_ms": 1e3 * (t4 - t3),
              "step_ms": 1e3 * (t4 - t0), "allgather_us": float("nan"), "top_levels_us": float("nan"), "allreduce_us": float("nan")}
        if self.transport.comm is not None:
            tm = self.transport.comm.timing()
            ph.update(allgather_us=tm.last_allgather_us, top_levels_us=tm.last_top_levels_us, allreduce_us=tm.last_allreduce_us)
        self.phases = ph
        return st

    def sample_paths(self, leaf_ids, pad_seed, with_nodes=False):
        if self.w is None:
            raise capi.DapolError(9, "this rank's shard holds no liabilities: nothing to sample")
        return self.w.paths(leaf_ids, upper=self.upper, with_nodes=with_nodes)

    def sample_proofs(self, leaf_ids, proof_size):
        """Proofs of the given leaves from the last step (leaf_ids must be leaves of this rank, any order)."""
        if self.w is None:
            raise capi.DapolError(9, "this rank's shard holds no liabilities: nothing to sample")
        pos = np.searchsorted(self.idx, np.ascontiguousarray(leaf_ids, np.uint64))
        out = np.zeros((len(pos), proof_size), np.uint8)
        for k, p in enumerate(pos):
            out[k] = self.w.proofs(int(p), 1, proof_size)[0]
        return out


def assemble_batch_records(height, shard_bits, leaves, shard_lookups, records, ctx):
    """Sibling records of a batched proof over a sharded tree, in dapol_batch_siblings order.
    shard_lookups[g](level, index) -> (C, H, v, r, found) for the positions inside shard g (dapol_tree_node_records on that
    rank's tree; on a real multi-GPU run the rows come back through an all-gather); records = the exchanged [G][104]
    subtree-root records, used for the positions at and above the shard roots (dapol_shard_top_node_records)."""
    level, index = capi.batch_siblings(height, leaves)
    S = len(level)
    C, H, r = (np.zeros((S, 32), np.uint8) for _ in range(3))
    v = np.zeros(S, np.uint64)
    local_h = height - shard_bits
    top = np.nonzero(level >= local_h)[0]
    if len(top):
        tC, tH, tv, tr = capi.shard_top_node_records(ctx, records, level[top] - local_h, index[top])
        C[top], H[top], v[top], r[top] = tC, tH, tv, tr
    low = np.nonzero(level < local_h)[0]
    owner = (index[low] >> (local_h - level[low]).astype(np.uint64)).astype(np.int64)
    for g in sorted(set(owner.tolist())):
        sel = low[owner == g]
        gC, gH, gv, gr, found = shard_lookups[g](level[sel], index[sel])
        if not found.all():
            raise capi.DapolError(9, "shard %d does not hold a sibling of the batch" % g)
        C[sel], H[sel], v[sel], r[sel] = gC, gH, gv, gr
    return level, index, C, H, v, r
