"""bench.py's N > 1 flow on a single-GPU box: two ranks share the GPU, the exchange goes over gloo instead of RCCL (RCCL
refuses two ranks on one device) -- everything else (subtree per rank, all-gather of the root records, replicated top level,
upper siblings, checksum all-reduce, max-over-ranks timing, rank 0's parity and verification legs) is the real path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu(hip_lib):
    import socket
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, DAPOL_BENCH_BACKEND="gloo", DAPOL_TABLE_GB="3", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2-entities", "9", "--weak", "--height", "16", "--steps", "2", "--warmup", "1", "--cpu-budget-s", "3"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["entities_total"] == 1024
    assert line["cpu_baseline"]["kind"] == "port" and "rank 0" in line["cpu_baseline"]["note_multi_gpu"]      # a scaling line is self-contained
    assert line["roofline"]["whole_step"]["achieved"] > 0
    assert line["parity"]["bit_exact"] and line["parity"]["proofs_compared"] > 0
    assert line["parity"]["inclusion_proofs_verified_on_gpu"] == line["parity"]["inclusion_proofs_checked"] == 512
    # the same 1,024 entities on one rank give the same aggregate checksum (the reduce of the per-rank transcripts)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log2-entities", "10", "--height", "16", "--no-cpu-baseline", "--no-secondary"], cwd=ROOT,
                         env=dict(os.environ, DAPOL_TABLE_GB="3"), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert single["parity"]["inclusion_proofs_verified_on_gpu"] == single["parity"]["inclusion_proofs_checked"]
    print("checksums", line["checksum"], single["checksum"])      # (each rank's sum also folds in ITS subtree root: not comparable bit for bit)


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_spawns_its_ranks(hip_lib):
    """`python3 bench.py --gpus 2` with no launcher (WORLD_SIZE unset) must not be a SystemExit: the parent --
    which has not touched the GPU -- starts the two ranks under torch.distributed.run itself and relays rank 0's line.  N > 1
    defaults to the metric's configuration: STRONG scaling, --log2-entities names the TOTAL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DAPOL_BENCH_BACKEND="gloo", DAPOL_TABLE_GB="3")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2-entities", "10", "--height", "16", "--steps", "2", "--warmup", "1", "--cpu-budget-s", "3"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    # the headline right after the timed region, then the complete line (the driver takes the last); everything else went to stderr
    assert len(lines) == 2 and json.loads(lines[0])["complete"] is False and json.loads(lines[0])["cpu_baseline"] is None
    assert json.loads(lines[0])["value"] == json.loads(lines[1])["value"]
    line = json.loads(lines[1])
    assert line["complete"] is True
    mg = line["multi_gpu"]                                           # an N > 1 line explains itself: per-rank phases, the two collectives
    assert mg["ranks"] == 2 and mg["steps"] == 2 and len(mg["per_rank_ms"]["step"]) == 2
    assert mg["step_ms"]["max"] >= mg["step_ms"]["min"] > 0 and mg["prove_ms"]["imbalance"] >= 0
    assert mg["exchange"]["host_ms"]["max_over_ranks_mean_over_steps"] > 0 and mg["reduce"]["host_ms"]["max_over_ranks_mean_over_steps"] > 0
    assert mg["exchange"]["allgather_device_us"] is None             # (gloo here: the library's communicator, whose HIP events these are, needs RCCL)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    assert line["config"]["entities_total"] == 1024 and line["config"]["entities_per_gpu"] == 512
    assert "strong scaling" in line["config"]["workload"] and "gloo" in line["config"]["exchange"]
    assert line["parity"]["bit_exact"] and line["parity"]["inclusion_proofs_verified_on_gpu"] == line["parity"]["inclusion_proofs_checked"]
    assert line["secondary"] is None and line["cpu_baseline"]["value"] > 0   # the secondary legs are N = 1 only; the CPU baseline is in every line


@pytest.mark.gpu
def test_bench_verify_mode_two_ranks(hip_lib):
    """BASELINE configs[4] on N > 1 GPUs: replicas of the verifier, the proofs divided over the ranks, the verdicts AND-ed by an
    all-reduce MIN inside the timed region (gloo here: two ranks share the one GPU; RCCL's path is the same call with one rank in
    test_rccl_exchange_one_rank_and_top_levels)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DAPOL_BENCH_BACKEND="gloo", DAPOL_TABLE_GB="3")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "verify", "--gpus", "2", "--verify-proofs", "16", "--verify-parties", "32", "--steps", "5",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["unit"] == "commitments/s"
    assert line["all_verified"] and line["one_bad_proof_turns_the_job_verdict"]
    assert "8 per GPU" in line["config"]["workload"] and "MIN" in line["config"]["verdict_reduce"]


_ONE_RANK_RCCL = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from dapol_amd import capi, sharded
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, NCCL_SOCKET_IFNAME="lo")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))      # torch's own RCCL communicator, as bench.py makes it
ctx = capi.Context(0, 8)
tr = sharded.ShardTransport(ctx, 0, 1, dist, torch, "cuda")
tr.create_comm(timeout_s=60.0, even_alone=True)
assert tr.comm is not None and tr.comm_ranks == 1, tr.comm_error
assert tr.group is not None and tr.group_device == "cpu"                  # the side gloo group exists beside the RCCL default group
assert tr.agreements == 1                                                 # the ranks agreed on the creation's outcome over it
rng = np.random.default_rng(1)
idx = np.sort(rng.choice(256, size=9, replace=False)).astype(np.uint64)
v = rng.integers(0, 9, size=9, dtype=np.uint64)
r = rng.integers(0, 256, size=(9, 32), dtype=np.uint8); r[:, 31] &= 0x0F
root = capi.Tree(ctx, 8, idx, v, r, bytes(32)).root()
ok, res = tr._library(lambda: tr.comm.exchange(root))                     # a library collective + the agreement after it
assert ok and res[0] == root and tr.agreements == 2
ok, res = tr._library(lambda: int(tr.comm.allreduce([41], capi.REDUCE_SUM)[0]))
assert ok and res == 41 and tr.agreements == 3
# a failing collective: every rank (here: the one) drops the communicator after the agreement, the torch transport takes over
def boom():
    raise capi.DapolError(19, "ncclAllReduce did not complete within 60000 ms")
ok, res = tr._library(boom)
assert not ok and tr.comm is None and "after the library's collective failed" in tr.path and tr.agreements == 4
t = torch.tensor([7], dtype=torch.int64, device="cuda"); dist.all_reduce(t); assert int(t.item()) == 7    # torch's RCCL group still works
dist.destroy_process_group()
print("ONE-RANK-RCCL-OK")
"""


@pytest.mark.gpu
def test_library_communicator_and_agreement_group_beside_torch_rccl(hip_lib):
    """What `bench.py --gpus N` sets up on a multi-GPU node, with the one rank a 1-GPU box has: torch.distributed over RCCL as the
    default group, the side gloo group the ranks agree over, the 128-byte id carried by a torch broadcast, the library's own
    communicator created non-blocking, its collectives each followed by the agreement, and the agreed drop to torch's transport."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", _ONE_RANK_RCCL % {"root": ROOT, "port": port}], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ONE-RANK-RCCL-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


@pytest.mark.gpu
def test_preflight_one_rccl_rank_and_two_gloo_ranks(hip_lib):
    """`bench.py --gpus N --preflight`: < 30 s, no proving.  With the one rank a 1-GPU box has, over RCCL: the
    library's communicator comes up non-blocking, ncclCommCount == 1, 100 rounds of the step's two collectives through the transport
    (library call + agreement), the HIP-event timings are there.  With two ranks sharing the GPU over gloo: the same flow through
    torch.distributed, every rank's global root equal and equal to the torch path's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "DAPOL_BENCH_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preflight"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    d = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["n_gpus"] == 1 and d["backend"] == "nccl" and all(d["checks"].values())
    assert d["checks"]["library_communicator_up"] and d["checks"]["nccl_comm_count_equals_n"] and d["rccl_ranks_in_library_communicator"] == 1
    t = d["timings"]
    assert t["exchange_host"]["iters"] == 100 and t["library_device_us"]["exchanges"] == 100 and t["library_device_us"]["reduces"] == 100
    assert 0 < t["library_device_us"]["allgather_mean"] < 5000 and 0 < t["library_device_us"]["allreduce_mean"] < 5000
    assert d["wall_s_since_process_start"] < 120                           # (the first `import torch` of a fresh box is most of it)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight", "--preflight-iters", "20"], cwd=ROOT,
                         env=dict(env, DAPOL_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stderr[-2000:]
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["ok"] and d2["n_gpus"] == 2 and all(d2["checks"].values()) and d2["checks"]["library_root_equals_torch_path_root"]
    assert "gloo" in d2["exchange_path"] and d2["timings"]["exchange_host_torch"]["iters"] == 4
