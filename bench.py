#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: entities/sec for DAPOL+ tree build + 64-bit range-proof
generation (one padding-policy inclusion proof per entity, aggregation_factor = height; benches/dapol.rs:155).

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A step = one pass of the hot path over the whole synthetic entity set, inputs already resident in HBM:
tree build (commit + hash + merge, padding nodes made on the fly) followed by one aggregated Bulletproof per
entity.  N = 1 workload: BASELINE.json configs[2] -- 2^20 entities, height 32, 64-bit proofs (the configuration
the metric is quoted on).  N > 1: weak scaling, 2^20 entities per GPU, each GPU owning one top-level subtree; the
only exchange is an all-gather of the N subtree-root records and an all-reduce of the proof checksum.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).  The CPU oracle (oracle/) is used
only for the `cpu_baseline` leg and a byte-for-byte spot check of sampled proofs -- never in the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PAD_SEED = bytes((i * 7 + 1) & 0xFF for i in range(32))
NONCE_SEED = bytes((i * 13 + 5) & 0xFF for i in range(32))


def algorithmic_bytes(height, n_bits):
    """compulsory bytes per entity (each input read once, each output written once)."""
    lgn = 0
    while (1 << lgn) < n_bits * _np2(height):
        lgn += 1
    proof = 32 * (9 + 2 * lgn)
    prove = height * (8 + 32) + height * 64 + height * 64 + proof + 16       # secrets + (C,H) read + path written + proof + framing
    return prove


def _np2(x):
    p = 1
    while p < x:
        p <<= 1
    return p


def synth_inputs(n_total, height, first, count):
    """benches/dapol.rs:160-175 + src/dapol/node.rs:101-105: strided leaves, value = random u32, random blinding."""
    rng = np.random.Generator(np.random.PCG64(0xD4901))
    v_all = rng.integers(0, 1 << 32, size=n_total, dtype=np.uint64)
    r_all = rng.integers(0, 256, size=(n_total, 32), dtype=np.uint8)
    r_all[:, 31] &= 0x0F                                         # < 2^252 < l: canonical scalars
    stride = (1 << height) // n_total
    idx = (np.arange(first, first + count, dtype=np.uint64) * np.uint64(stride))
    return idx, v_all[first:first + count].copy(), r_all[first:first + count].copy()


def cpu_baseline(ref, height, n_bits, idx, v, r, sample_paths, budget_s=20.0):
    """Times the C restatement (kind 'port') on this host: tree build on a bounded slice + proofs for sampled entities."""
    cores = os.cpu_count() or 1
    ref.ref_tree_build.restype = ctypes.c_void_p
    ref.ref_range_proof_size.restype = ctypes.c_size_t
    nt = min(len(idx), 4096)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    t0 = time.perf_counter()
    t = ctypes.c_void_p(ref.ref_tree_build(height, ctypes.c_size_t(nt), p(idx[:nt]), p(v[:nt]), p(r[:nt]), PAD_SEED, 0))
    tree_s = time.perf_counter() - t0
    ref.ref_tree_free(t)
    sv, sr, leaf_ids = sample_paths
    m = _np2(height)
    ps = ref.ref_range_proof_size(n_bits, m)
    # parties: siblings root side first, padded with (0, Scalar::one())
    def parties(k):
        pv = np.zeros(m, np.uint64)
        pr = np.zeros((m, 32), np.uint8)
        pv[:height] = sv[k]
        pr[:height] = sr[k]
        pr[height:, 0] = 1
        return pv, pr
    # one proof first to size the sample
    pv, pr = parties(0)
    out1 = ctypes.create_string_buffer(ps)
    t0 = time.perf_counter()
    ref.ref_range_prove(n_bits, m, p(pv), p(pr), NONCE_SEED, ctypes.c_uint64(int(leaf_ids[0])), ctypes.c_uint64(0), None, 0, out1)
    one = time.perf_counter() - t0
    # pilot batch (one proof per core) to learn the parallel rate on this host, then size the sample to ~budget_s
    npilot = min(len(leaf_ids), cores)
    PV = np.zeros((npilot, m), np.uint64)
    PR = np.zeros((npilot, m, 32), np.uint8)
    for k in range(npilot):
        PV[k], PR[k] = parties(k)
    sid = np.ascontiguousarray(leaf_ids[:npilot], dtype=np.uint64)
    outp = ctypes.create_string_buffer(ps * npilot)
    t0 = time.perf_counter()
    ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(npilot), p(PV), p(PR), NONCE_SEED, p(sid), ctypes.c_uint64(0), None, 0, outp)
    pilot = time.perf_counter() - t0
    ns = int(max(npilot, min(len(leaf_ids), npilot * budget_s / max(pilot, 1e-3))))
    PV = np.zeros((ns, m), np.uint64)
    PR = np.zeros((ns, m, 32), np.uint8)
    for k in range(ns):
        PV[k], PR[k] = parties(k)
    out = ctypes.create_string_buffer(ps * ns)
    sid = np.ascontiguousarray(leaf_ids[:ns], dtype=np.uint64)
    t0 = time.perf_counter()
    ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(ns), p(PV), p(PR), NONCE_SEED, p(sid), ctypes.c_uint64(0), None, 0, out)
    prove_s = time.perf_counter() - t0
    per_entity = tree_s / nt + prove_s / ns
    return {"value": 1.0 / per_entity, "unit": "entities/s", "cores": cores, "kind": "port",
            "sample": "C restatement (oracle/ref_dapol.c, OpenMP over entities): tree build of %d entities at height %d (%.2f s) + %d "
                      "padding-policy proofs m=%d n=%d (%.2f s, %.3f s/proof single-thread); reference Rust not runnable (no toolchain)"
                      % (nt, height, tree_s, ns, m, n_bits, prove_s, one),
            "single_thread_proof_s": one}, out.raw, ns, ps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-entities", type=int, default=20, help="entities per GPU = 2^this (default: BASELINE configs[2])")
    ap.add_argument("--height", type=int, default=32)
    ap.add_argument("--n-bits", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        # One rank per GPU over RCCL.  DAPOL_BENCH_BACKEND=gloo is a test hook: it lets two ranks share one GPU (RCCL refuses
        # duplicate devices), so the whole N > 1 flow can be exercised on a single-GPU box.
        backend = os.environ.get("DAPOL_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    from __graft_entry__ import build, ORACLE_LIB
    if rank == 0:
        build()
    if dist is not None:
        dist.barrier()
    from dapol_amd import capi
    from dapol_amd.sharded import ShardedProver

    n_per_gpu = 1 << args.log2_entities
    n_total = n_per_gpu * world
    height, n_bits = args.height, args.n_bits
    idx, v, r = synth_inputs(n_total, height, rank * n_per_gpu, n_per_gpu)
    ctx = capi.Context(local_rank, _np2(height))
    prover = ShardedProver(ctx, height, idx, v, r, rank, world, dist, torch,
                           comm_device="cuda" if os.environ.get("DAPOL_BENCH_BACKEND", "nccl") == "nccl" else "cpu")

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    stats = None
    for _ in range(args.warmup):
        stats = prover.step(PAD_SEED, NONCE_SEED, n_bits)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = prover.step(PAD_SEED, NONCE_SEED, n_bits)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=prover.comm_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        if dist is not None:
            dist.barrier()                     # leave together with rank 0 (which still runs its parity / verification legs)
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = n_total * args.steps / elapsed
    ab = algorithmic_bytes(height, n_bits)
    msm_avg_ms = stats.msm_ms / max(1, stats.msm_launches)
    ach = (stats.proofs * ab / 1e9) / (stats.msm_ms / 1e3) if stats.msm_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                "kernel": "k_rp_msm", "launches": int(stats.msm_launches), "avg_launch_ms": msm_avg_ms,
                "algorithmic_bytes_per_entity": ab,
                "note": "achieved/frac use the ALGORITHMIC bytes of SURVEY 8d (6,384 B per entity); the kernel itself is integer-VALU "
                        "bound (255-bit modular multiply-adds) at the socket power cap and deliberately spends HBM bandwidth on wide "
                        "(17-bit) window tables: see traffic (measured HBM bytes per launch), valu_roof and DESIGN.md sections 5 and 8"}
    # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/pmc_summary.py),
    # measured on full 73,728-proof launches of this kernel; only quoted when this run's launches have that size too.
    pmc = os.path.join(ROOT, "profiles", "msm_pmc.json")
    if os.path.exists(pmc) and n_per_gpu >= 60000 and height == 32 and n_bits == 64:
        try:
            pj = json.load(open(pmc))
            roofline["traffic"] = pj.get("hbm_bytes_per_launch")
            roofline["traffic_GBps"] = pj.get("hbm_GBps")
            roofline["traffic_frac_of_peak"] = pj.get("hbm_GBps", 0.0) / PEAK_HBM_GBS
            roofline["valu_cycles_per_inst"] = pj.get("cycles_per_valu_inst_per_simd")
            # the binding roof: integer-VALU issue.  The same point additions in a register-only loop (tools/ubench_madd.hip,
            # profiles/r01b_ubench_madd.txt) issue at 4,350 cycles / 1,075 instructions = 4.05 cycles per wave instruction.
            if pj.get("cycles_per_valu_inst_per_simd"):
                roofline["valu_issue_frac"] = 4.05 / pj["cycles_per_valu_inst_per_simd"]
        except Exception:
            pass
    # the binding roof in its own units (register-only cost of the kernel's point additions, clocks, socket power)
    vr = os.path.join(ROOT, "profiles", "valu_roof.json")
    if roofline.get("traffic") is not None and os.path.exists(vr):
        try:
            roofline["valu_roof"] = {k: v for k, v in json.load(open(vr)).items() if k != "source"}
        except Exception:
            pass
    cpu = None
    parity = None
    if not args.no_cpu_baseline and os.path.exists(ORACLE_LIB):
        ref = ctypes.CDLL(ORACLE_LIB)
        ns_max = 4096
        sample_ids = idx[:: max(1, n_per_gpu // ns_max)][:ns_max]
        sv, sr = prover.sample_paths(sample_ids, PAD_SEED)
        # The timed CPU baseline is reported at N = 1 only; at N > 1 the same code runs with a 2-second budget, purely as
        # the parity check of rank 0's shard (the sharded tree must give the bytes the oracle gives for those siblings).
        cpu, cpu_bytes, ns, ps = cpu_baseline(ref, height, n_bits, idx, v, r, (sv, sr, sample_ids), budget_s=20.0 if world == 1 else 2.0)
        gpu_bytes = prover.sample_proofs(sample_ids[:ns], ps)
        parity = {"proofs_compared": ns, "bit_exact": bool(gpu_bytes.tobytes() == cpu_bytes)}
        if world > 1:
            cpu = None
    # encode -> verify round trip at full size: sampled inclusion proofs of the timed run through DapolProof::verify on the GPU
    nv = min(2048, n_per_gpu)
    vids = np.ascontiguousarray(idx[:: max(1, n_per_gpu // nv)][:nv])
    vpos = np.searchsorted(idx, vids)
    _, _, vC, vH = prover.sample_paths(vids, PAD_SEED, with_nodes=True)
    lC, lH = ctx.commit_hash_batch(v[vpos], r[vpos])
    vproofs = prover.sample_proofs(vids, int(stats.proof_bytes // max(1, stats.proofs)))
    rC, rH = prover.root[0], prover.root[1]
    okv = ctx.verify_entities(height, vids, lC, lH, vC, vH, rC, rH, capi.POLICY_PADDING, height, n_bits, vproofs, verify_seed=PAD_SEED)
    parity = dict(parity or {}, inclusion_proofs_verified_on_gpu=int(okv.sum()), inclusion_proofs_checked=int(len(okv)))
    line = {
        "metric": "entities/sec (tree build + 64-bit range-proof gen), 2^20 leaves, 1/2/4/8 GPU",
        "value": value, "unit": "entities/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32 limbs (255-bit modular integers)", "data": "synthetic",
        "config": {"workload": "2^%d entities%s, height=%d, %d-bit range proofs, padding policy, aggregation_factor=height, BLAKE3 node hash"
                               % (args.log2_entities, " per GPU" if world > 1 else "", height, n_bits),
                   "entities_total": n_total, "proof_bytes": int(stats.proof_bytes // max(1, stats.proofs)),
                   "sharding": "none" if world == 1 else "top-level subtrees, all-gather of %d subtree roots" % world},
        "phases_ms": {"tree_build": stats.tree_ms, "prove": stats.prove_ms},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        "checksum": "%016x" % stats.checksum,
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
