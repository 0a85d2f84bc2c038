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
           port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2-entities", "9", "--height", "16", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["entities_total"] == 1024 and line["cpu_baseline"] is None
    assert line["parity"]["bit_exact"] and line["parity"]["proofs_compared"] > 0
    assert line["parity"]["inclusion_proofs_verified_on_gpu"] == line["parity"]["inclusion_proofs_checked"] == 512
    # the same 1,024 entities on one rank give the same aggregate checksum (the reduce of the per-rank transcripts)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log2-entities", "10", "--height", "16", "--no-cpu-baseline"], cwd=ROOT,
                         env=dict(os.environ, DAPOL_TABLE_GB="3"), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert single["parity"]["inclusion_proofs_verified_on_gpu"] == single["parity"]["inclusion_proofs_checked"]
    print("checksums", line["checksum"], single["checksum"])
