#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: entities/sec for DAPOL+ tree build + 64-bit range-proof
generation (one padding-policy inclusion proof per entity, aggregation_factor = height; benches/dapol.rs:59-91,149-175).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--budget-s S]
      N > 1: one rank per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this process IS a rank; without one
      the process -- which has not touched the GPU -- starts `python -m torch.distributed.run --nproc-per-node N ... bench.py`
      as a child, relays rank 0's JSON line and exits with the child's code.
  python bench.py --gpus N --preflight   < 30 s, no proving: RCCL communicator inside the library up (ncclCommCount == N), the two
                                    collectives of a step timed over 100 rounds, every rank's global root equal, link topology to stderr
  python bench.py --mode build      the reference's `build` criterion group (benches/dapol.rs:24-57) on the GPU, and the
                                    incremental dapol_tree_update on the headline tree
  python bench.py --mode verify [--gpus N]   BASELINE configs[4]: verification-only, aggregated proofs of 1,024 parties
  python bench.py --mode criterion  the reference's three criterion groups as it defines them (benches/dapol.rs:24-141):
                                    build, prove (ONE inclusion proof per iteration), verify; N = 1,024, heights 16 / 24 / 32

A step = one pass of the hot path over the whole synthetic entity set, inputs already resident in HBM:
tree build (commit + hash + merge, padding nodes made on the fly) followed by one aggregated Bulletproof per
entity.  The workload is the one the metric names -- BASELINE.json configs[2]: 2^20 entities IN TOTAL, height 32, 64-bit
proofs -- at every N: N > 1 is STRONG scaling by default (each GPU owns one top-level subtree holding 2^20 / N entities);
--log2-entities-total 22 gives configs[3] (2^22 over 8); --weak keeps 2^--log2-entities per GPU instead.  The only
exchange is an all-gather of the N subtree-root records and an all-reduce of the proof checksum.

Wall budget.  One step of configs[2] takes ~19 s, so the driver's `--steps 20 --warmup 5` (25 steps) does not fit its 600 s
limit with everything else the line carries.  The GPU loop therefore has a wall budget (--budget-s, default 570 s counted from
process start, including imports / build / context creation and a reserve for the legs after the timed region): after the first warm-up step the number of timed steps is clamped to what fits
(never below 3) and extra warm-up steps are dropped.  The line reports the steps actually run (`steps`, `warmup`) and
what was asked for (`steps_requested`, `warmup_requested`).  A heartbeat goes to stderr after every step.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).  Everything in `roofline` outside
`from_profiles` is measured by THIS run (HIP events on the launch streams inside the library).  The CPU oracle
(oracle/) is used only for the `cpu_baseline` leg and a byte-for-byte spot check of sampled proofs -- never in the
timed region.
"""
import time

T_PROC0 = time.time()

import argparse   # noqa: E402
import ctypes     # noqa: E402
import hashlib    # noqa: E402
import json       # noqa: E402
import os         # noqa: E402
import subprocess # noqa: E402
import sys        # noqa: E402

import numpy as np  # noqa: E402

if os.environ.get("DAPOL_BENCH_T0"):            # a rank started by spawn_ranks(): the wall budget counts from the parent's start
    try:
        T_PROC0 = float(os.environ["DAPOL_BENCH_T0"])
    except ValueError:
        pass
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the CPU-baseline leg's OpenMP team must not spin beside the GPU launches

PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PAD_SEED = bytes((i * 7 + 1) & 0xFF for i in range(32))
NONCE_SEED = bytes((i * 13 + 5) & 0xFF for i in range(32))
METRIC = "entities/sec (tree build + 64-bit range-proof gen), 2^20 leaves, 1/2/4/8 GPU"


def log(msg):
    print("[bench %6.1fs] %s" % (time.time() - T_PROC0, msg), file=sys.stderr, flush=True)


def _np2(x):
    p = 1
    while p < x:
        p <<= 1
    return p


def proof_bytes(height, n_bits):
    lgn = 0
    while (1 << lgn) < n_bits * _np2(height):
        lgn += 1
    return 32 * (9 + 2 * lgn)


def algorithmic_bytes(height, n_bits, log2_n):
    """compulsory bytes per entity (each input read once, each output written once).
    prove: sibling secrets (8 + 32) + (C, H) read + path written, per level, + proof + 16 B framing  (6,384 B at H = 32);
    tree (bench layout): 48 B of leaf input + 104 B per node written, (2N - 1 + 2cN) / N nodes per entity, c = H - log2 N
    (2,752 B at 2^20 x H = 32)."""
    prove = height * (8 + 32) + height * 64 + height * 64 + proof_bytes(height, n_bits) + 16
    c = max(0, height - log2_n)
    tree = 48 + int(round((2.0 + 2.0 * c) * 104))
    return prove, tree


def synth_inputs(n_total, height, first, count, n_bits=64):
    """benches/dapol.rs:160-175 + src/dapol/node.rs:101-105: strided leaves, value = random u32, random blinding.  (With range proofs
    narrower than 64 bits the values shrink so that every subtree sum -- what the proofs are about -- stays inside the range.)"""
    rng = np.random.Generator(np.random.PCG64(0xD4901))
    v_all = rng.integers(0, 1 << 32, size=n_total, dtype=np.uint64)
    if n_bits < 64:
        v_all %= np.uint64(max(1, min(1 << 32, (1 << n_bits) // max(1, n_total))))
    r_all = rng.integers(0, 256, size=(n_total, 32), dtype=np.uint8)
    r_all[:, 31] &= 0x0F                                         # < 2^252 < l: canonical scalars
    stride = (1 << height) // n_total
    idx = (np.arange(first, first + count, dtype=np.uint64) * np.uint64(stride))
    return idx, v_all[first:first + count].copy(), r_all[first:first + count].copy()


def usable_cores():
    """Threads this process may actually use: scheduler affinity capped by the cgroup CPU quota (the GPU boxes expose
    every hardware thread of the host but grant a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
            break
        except Exception:
            continue
    return n


CPU_NATIVE_CFLAGS = "-O3 -march=native -std=gnu11 -fPIC -fopenmp"


def build_native_oracle():
    """The timed CPU baseline is compiled ON THIS HOST with -O3 -march=native (the checker build that travels with the
    repo is -O2, generic ISA).  Returns (path, cflags) or (None, reason)."""
    odir = os.path.join(ROOT, "oracle")
    try:
        cpu = open("/proc/cpuinfo").read()
        model = [l for l in cpu.splitlines() if l.startswith("model name")][:1]
        flags = [l for l in cpu.splitlines() if l.startswith("flags")][:1]
        tag = hashlib.sha256(("".join(model + flags)).encode()).hexdigest()[:12]
    except Exception:
        tag = "host"
    out = os.path.join(odir, "_build", "libdapol_ref_native_%s.so" % tag)
    srcs = [os.path.join(odir, f) for f in ("ref_math.c", "ref_dapol.c")]
    if os.path.exists(out) and all(os.path.getmtime(s) <= os.path.getmtime(out) for s in srcs):
        return out, CPU_NATIVE_CFLAGS
    os.makedirs(os.path.dirname(out), exist_ok=True)
    try:
        subprocess.run(["gcc"] + CPU_NATIVE_CFLAGS.split() + ["-shared"] + srcs + ["-o", out], check=True, cwd=odir,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=120)
        return out, CPU_NATIVE_CFLAGS
    except Exception as e:          # fall back to the generic checker build, and say so in the line
        return None, "native build failed (%s)" % type(e).__name__


def cpu_baseline(generic_lib, height, n_bits, idx, v, r, sample_paths, budget_s=10.0):
    """Times the C restatement (kind 'port') on this host's cores, bounded to ~budget_s of wall time:
      * tuned    = generators / PedersenGens derived once (what any sane CPU user would run);
      * faithful = the reference's own per-call costs switched on: BulletproofGens::new + PedersenGens::default per
        proof (src/range/mod.rs:49-50,65-66), children re-compressed in every merge (src/dapol/node.rs:67-68);
      each single-threaded (the reference is single-threaded) and on all usable cores (OpenMP over entities).
    `value` = tuned, all cores: the strongest CPU number, so that the GPU/CPU ratio is not flattered."""
    cores = usable_cores()
    native, cflags = build_native_oracle()
    ref = ctypes.CDLL(native or generic_lib)
    ref.ref_tree_build.restype = ctypes.c_void_p
    ref.ref_range_proof_size.restype = ctypes.c_size_t
    have_threads = hasattr(ref, "ref_set_threads")
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    sv, sr, leaf_ids = sample_paths
    m = _np2(height)
    ps = ref.ref_range_proof_size(n_bits, m)

    def parties(k0, cnt):           # siblings root side first, padded with (0, Scalar::one())
        PV = np.zeros((cnt, m), np.uint64)
        PR = np.zeros((cnt, m, 32), np.uint8)
        PV[:, :height] = sv[k0:k0 + cnt]
        PR[:, :height] = sr[k0:k0 + cnt]
        PR[:, height:, 0] = 1
        return PV, PR

    def prove(cnt, faithful, threads):
        cnt = max(1, min(cnt, len(leaf_ids)))
        if have_threads:
            ref.ref_set_threads(threads)
        PV, PR = parties(0, cnt)
        sid = np.ascontiguousarray(leaf_ids[:cnt], dtype=np.uint64)
        out = ctypes.create_string_buffer(ps * cnt)
        t0 = time.perf_counter()
        ref.ref_range_prove_batch(n_bits, m, ctypes.c_size_t(cnt), p(PV), p(PR), NONCE_SEED, p(sid), ctypes.c_uint64(0), None, faithful, out)
        return time.perf_counter() - t0, out.raw, cnt

    def tree(nt, faithful, threads):
        if have_threads:
            ref.ref_set_threads(threads)
        t0 = time.perf_counter()
        t = ctypes.c_void_p(ref.ref_tree_build(height, ctypes.c_size_t(nt), p(idx[:nt]), p(v[:nt]), p(r[:nt]), PAD_SEED, faithful))
        dt = time.perf_counter() - t0
        ref.ref_tree_free(t)
        return dt

    t_begin = time.perf_counter()
    prove(1, 0, 1)                                   # untimed: fills the tuned variant's generator cache
    one_t, _, _ = prove(1, 0, 1)                     # single thread, tuned
    one_f, _, _ = prove(1, 1, 1)                     # single thread, faithful
    nt1 = min(len(idx), 256)
    tree1_t, tree1_f = tree(nt1, 0, 1) / nt1, tree(nt1, 1, 1) / nt1
    # all cores: pilot of one proof per thread, then a sample sized to the remaining budget (60 % tuned, 40 % faithful)
    pilot_s, _, npilot = prove(cores, 0, cores)
    left = max(1.0, budget_s - (time.perf_counter() - t_begin))
    n_t = int(max(npilot, npilot * 0.5 * left / max(pilot_s, 1e-3)))
    prove_t, bytes_t, n_t = prove(n_t, 0, cores)
    n_f = int(max(cores, n_t * 0.35 * one_t / max(one_f, 1e-3)))
    prove_f, _, n_f = prove(n_f, 1, cores)
    ntm = min(len(idx), 4096)
    treem_t, treem_f = tree(ntm, 0, cores) / ntm, tree(ntm, 1, cores) / ntm
    if have_threads:
        ref.ref_set_threads(cores)
    val = 1.0 / (treem_t + prove_t / n_t)
    res = {"value": val, "unit": "entities/s", "cores": cores, "kind": "port",
           "cflags": cflags if native else "-O2 generic (%s)" % cflags,
           "variant": "tuned (generators derived once), all usable cores",
           "faithful_value": 1.0 / (treem_f + prove_f / n_f),
           "single_thread_value": 1.0 / (tree1_t + one_t),
           "single_thread_faithful_value": 1.0 / (tree1_f + one_f),
           "single_thread_proof_s": one_t, "single_thread_faithful_proof_s": one_f,
           "host_hw_threads": os.cpu_count(), "cpu_leg_wall_s": time.perf_counter() - t_begin,
           "sample": "C restatement of the reference path (oracle/ref_dapol.c, OpenMP over entities; NOT the Rust crate: no Rust "
                     "toolchain in this image): tree build of %d entities at height %d + %d tuned / %d faithful padding-policy "
                     "proofs m=%d n=%d on %d threads (%.2f s / %.2f s); 1 proof each single-threaded"
                     % (ntm, height, n_t, n_f, m, n_bits, cores, prove_t, prove_f)}
    return res, bytes_t, n_t, ps


def plan_steps(steps_req, warm_req, t_step, seconds_left):
    """The wall budget: (further warm-up steps, timed steps) that fit `seconds_left` after the first warm-up step took t_step
    seconds.  Timed steps are never fewer than min(3, requested); further warm-ups come only out of slack."""
    if t_step is None or t_step <= 0:
        return 0, steps_req
    fit = int(seconds_left / (t_step * 1.03))
    extra_warm = min(max(0, warm_req - 1), max(0, fit - steps_req))
    steps = min(steps_req, max(min(3, steps_req), fit - extra_warm))
    return extra_warm, steps


def plan_workload(args, world):
    """(entities in total, per GPU, log2 of the total, "strong" | "weak").  The metric is quoted on 2^20 leaves at 1 / 2 / 4 / 8
    GPUs, so the TOTAL is what --log2-entities (default 20) or --log2-entities-total names and N > 1 divides it (strong
    scaling); --weak keeps 2^--log2-entities per GPU.  At N = 1 the two are the same workload; the label is the one of the SERIES
    the line belongs to, so that the lines of N = 1, 2, 4, 8 agree: "strong" unless --weak."""
    if world & (world - 1):
        raise SystemExit("the number of GPUs must be a power of two (each rank owns one top-level subtree)")
    if args.weak and args.log2_entities_total is None:
        n_per_gpu = 1 << args.log2_entities
        return n_per_gpu * world, n_per_gpu, args.log2_entities + (world.bit_length() - 1), "weak"
    lg_total = args.log2_entities_total if args.log2_entities_total is not None else args.log2_entities
    n_total = 1 << lg_total
    if n_total % world or n_total // world < 1:
        raise SystemExit("2^%d entities cannot be divided over %d GPUs" % (lg_total, world))
    return n_total, n_total // world, lg_total, "strong"


def kernel_src_sha():
    """Identity of the kernel sources a profile was taken on (profiles/*.json carry it; bench.py quotes a profile only
    when it matches the build that is running)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dapol_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith(".") or not os.path.isfile(os.path.join(d, f)):
            continue                                  # (sources only: a stray cache directory is not part of the build)
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def from_profiles(sha, full_size):
    """Numbers that THIS run cannot measure (PMC counters need rocprofv3 around the process): quoted from tracked
    files under profiles/ with the file name and the kernel-source hash they were measured on; `current_build` says
    whether that is the build running now."""
    out = {}
    for key, fn in (("msm_pmc", "msm_pmc.json"), ("valu_roof", "valu_roof.json")):
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            continue
        try:
            pj = json.load(open(path))
        except Exception:
            continue
        pj = {k: v for k, v in pj.items() if k != "source"}
        pj["file"] = "profiles/" + fn
        pj["current_build"] = bool(pj.get("kernel_src_sha") == sha)
        out[key] = pj
    out["applies_to_this_workload"] = bool(full_size)
    return out


def gather_phase_rows(dist, torch, phase_rows, world, device):
    """Every rank's [step][key] matrix of phase times -> [rank][step][key] on every rank (one all-gather, outside the timed region)."""
    if dist is None or world == 1:
        return np.asarray(phase_rows, np.float64)[None]
    mine = torch.tensor(phase_rows, dtype=torch.float64, device=device)
    allp = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    return torch.stack(allp).cpu().numpy()


def rank_decomposition(per_rank, keys, world, exchange_path):
    """per_rank[rank][step][key] (ShardedProver.PHASE_KEYS + the device times of the two phases) -> the `multi_gpu` object of the
    line: per rank the mean over the timed steps of every phase; over the ranks min / max / imbalance of the build, the proving and
    the whole step; and the two collectives -- host milliseconds of the calls (they include waiting for the slowest rank to arrive:
    a rank that finishes its build early waits in the all-gather) and, when the library's communicator carried them, the device
    microseconds from HIP events around ncclAllGather / the top-level merge / ncclAllReduce.  The all-gather's own latency is its
    MINIMUM over the ranks (the last rank to arrive waits for nobody); max - min is the arrival skew."""
    import warnings
    a = np.asarray(per_rank, np.float64)                     # [R][S][K]
    with warnings.catch_warnings():                          # (a column of NaNs -- no library communicator -- is a None in the line, not a warning)
        warnings.simplefilter("ignore", category=RuntimeWarning)
        mean = np.nanmean(a, axis=1) if a.shape[1] else np.zeros((a.shape[0], len(keys)))
    col = {k: mean[:, i] for i, k in enumerate(keys)}

    def spread(x):
        x = np.asarray(x, np.float64)
        if np.all(np.isnan(x)):
            return None
        lo, hi = float(np.nanmin(x)), float(np.nanmax(x))
        return {"min": lo, "max": hi, "mean": float(np.nanmean(x)), "imbalance": (hi / lo - 1.0) if lo > 0 else None}

    def per_step_stats(key):
        i = keys.index(key)
        x = a[:, :, i]
        if np.all(np.isnan(x)):
            return None
        return {"min_over_ranks_mean_over_steps": float(np.nanmean(np.nanmin(x, axis=0))),
                "max_over_ranks_mean_over_steps": float(np.nanmean(np.nanmax(x, axis=0))),
                "max_over_ranks_and_steps": float(np.nanmax(x))}

    out = {"ranks": int(a.shape[0]), "steps": int(a.shape[1]),
           "per_rank_ms": {"build": [float(x) for x in col["build_ms"]], "exchange": [float(x) for x in col["exchange_ms"]],
                           "prove": [float(x) for x in col["prove_ms"]], "reduce": [float(x) for x in col["reduce_ms"]],
                           "step": [float(x) for x in col["step_ms"]],
                           "tree_device": [float(x) for x in col["tree_device_ms"]], "prove_device": [float(x) for x in col["prove_device_ms"]]},
           "build_ms": spread(col["build_ms"]), "prove_ms": spread(col["prove_ms"]), "step_ms": spread(col["step_ms"]),
           "exchange": {"path": exchange_path, "host_ms": per_step_stats("exchange_ms"), "allgather_device_us": per_step_stats("allgather_us"),
                        "top_levels_device_us": per_step_stats("top_levels_us"),
                        "note": "host_ms = dapol_shard_exchange (or the torch all_gather + top-level merge) on the rank's host clock, incl. the wait "
                                "for the slowest rank's build; *_device_us = HIP events inside the library around ncclAllGather (min over ranks = "
                                "its own latency, max - min = arrival skew) and around the replicated merge of the top log2 N levels"},
           "reduce": {"host_ms": per_step_stats("reduce_ms"), "allreduce_device_us": per_step_stats("allreduce_us"),
                      "note": "the final all-reduce of the proof-transcript checksum (dapol_comm_allreduce_u64); host_ms includes the wait for the "
                              "slowest rank's proving"}}
    if world == 1:
        out["note"] = "single GPU: no exchange, no reduce; the phases are this rank's"
    return out


def showtopo_to_stderr(notes):
    """`rocm-smi --showtopo` to stderr (a child process, never an exec), to be called BEFORE this process touches the GPU: what a
    multi-GPU run that hangs or falls back needs to have left behind.  Diagnostics, not a check: the links are what they are."""
    try:
        topo = subprocess.run(["rocm-smi", "--showtopo"], capture_output=True, text=True, timeout=25)
        print(topo.stdout[-6000:], file=sys.stderr, flush=True)
        notes["rocm_smi_showtopo_rc"] = topo.returncode
    except Exception as e:
        notes["rocm_smi_showtopo"] = repr(e)


def inline_preflight(tr, root, rank, world, dist, torch, comm_device, ctx, iters=10, allow_fallback=False, merge=None):
    """The collective half of --preflight INSIDE the driver's own `--gpus N` command (the driver never passes
    --preflight): right after the contexts and the transport exist, `iters` rounds of the step's exchange and of its final reduce
    through the very transport the timed loop will drive, every rank's global root equal, the reduce's sum right -- and, over RCCL,
    the library's communicator up with ncclCommCount == N (a run that would spend its 25 steps on the torch.distributed fallback is
    a failed run unless --allow-exchange-fallback says otherwise).  -> (the `multi_gpu.preflight` object, ok on EVERY rank)."""
    t0 = time.perf_counter()
    lib_comm = tr.comm is not None
    checks = {}
    allow_fallback = allow_fallback or os.environ.get("DAPOL_EXCHANGE", "").lower() == "torch"      # (an explicit choice of the torch transport)
    if comm_device == "cuda" and not allow_fallback:
        checks["library_communicator_up"] = lib_comm
        checks["nccl_comm_count_equals_n"] = bool(lib_comm and tr.comm_ranks == world)
    pc_checks, res, groot, ok = preflight_collectives(tr, root, rank, world, dist, torch, comm_device, iters, ctx, lib_comm, merge=merge,
                                                      local_ok=all(bool(c) for c in checks.values()))
    checks.update(pc_checks)
    return {"ok": bool(ok), "iters": iters, "checks": checks, "timings": res, "exchange_path": tr.path, "rccl_ranks_in_library_communicator": tr.comm_ranks,
            "comm_error": tr.comm_error, "global_root_C": groot[0].hex(), "seconds": time.perf_counter() - t0}, bool(ok)


def inline_preflight_reduce(tr, rank, world, dist, torch, comm_device, iters=10, allow_fallback=False):
    """--mode verify at N > 1: the replicas only share the AND of their verdicts, so the check before the first pass is the
    communicator (up, ncclCommCount == N over RCCL) and `iters` rounds of that reduce through the transport, agreed by all ranks."""
    t0 = time.perf_counter()
    lib_comm = tr.comm is not None
    checks = {}
    allow_fallback = allow_fallback or os.environ.get("DAPOL_EXCHANGE", "").lower() == "torch"
    if comm_device == "cuda" and not allow_fallback:
        checks["library_communicator_up"] = lib_comm
        checks["nccl_comm_count_equals_n"] = bool(lib_comm and tr.comm_ranks == world)
    ts, good = [], True
    for _ in range(iters):
        t1 = time.perf_counter()
        good &= tr.reduce_u64(rank + 1, "sum") == world * (world + 1) // 2
        ts.append(1e6 * (time.perf_counter() - t1))
    checks["reduce_sum_correct"] = bool(good)
    checks["reduce_min_is_an_and"] = tr.reduce_u64(0 if rank == world - 1 else 1, "min") == 0 and tr.reduce_u64(1, "min") == 1
    checks["transport_still_on_the_library_communicator"] = (tr.comm is not None) == lib_comm
    flag = torch.tensor([1 if all(bool(c) for c in checks.values()) else 0], dtype=torch.int64, device=comm_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = int(flag.item()) == 1
    ts.sort()
    return {"ok": ok, "iters": iters, "checks": checks, "reduce_host_us": {"median": ts[len(ts) // 2], "min": ts[0], "max": ts[-1]},
            "reduce_path": tr.path, "rccl_ranks_in_library_communicator": tr.comm_ranks, "comm_error": tr.comm_error,
            "seconds": time.perf_counter() - t0}, ok


def init_dist(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (a launcher started a different number of ranks)" % (args.gpus, world))
    if world > 1 and rank == 0:
        showtopo_to_stderr({})                     # before `import torch` / any HIP call of this process
    import torch
    dist = None
    backend = os.environ.get("DAPOL_BENCH_BACKEND", "nccl")
    if world > 1:
        # one node: RCCL's bootstrap sockets (torch's communicator and the library's own) stay on loopback, like MASTER_ADDR
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        import torch.distributed as dist
        # One rank per GPU over RCCL.  DAPOL_BENCH_BACKEND=gloo is a test hook: it lets two ranks share one GPU (RCCL refuses
        # duplicate devices), so the whole N > 1 flow can be exercised on a single-GPU box.
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world, torch, dist, backend


def mode_prove(args):
    rank, local_rank, world, torch, dist, backend = init_dist(args)
    log("rank %d/%d: torch imported" % (rank, world))
    from __graft_entry__ import build, ORACLE_LIB
    if rank == 0:
        build()
    if dist is not None:
        dist.barrier()
    from dapol_amd import capi
    from dapol_amd.sharded import ShardedProver

    n_total, n_per_gpu, lg_total, scaling = plan_workload(args, world)
    height, n_bits = args.height, args.n_bits
    idx, v, r = synth_inputs(n_total, height, rank * n_per_gpu, n_per_gpu, n_bits)
    opts = capi.Options(profile=capi.PROFILE_HOST) if args.profile == "host" else None
    ctx = capi.Context(local_rank, _np2(height), options=opts)
    log("context ready (generators + window tables; profile %s, %d-bit windows)" % (args.profile, ctx.get_options().window_bits))
    comm_device = "cuda" if backend == "nccl" else "cpu"
    prover = ShardedProver(ctx, height, idx, v, r, rank, world, dist, torch, comm_device=comm_device)
    preflight = None
    if world > 1 and not args.no_inline_preflight:
        # this rank's REAL subtree root (one untimed build, ~50 ms) through 10 rounds of the step's two collectives: < 3 s
        if prover.w is not None:
            sub_root, _ = prover.w.build(PAD_SEED)
        else:
            pC, pH, pr = ctx.padding_nodes(PAD_SEED, [height - prover.shard_bits], [rank])
            sub_root = (pC[0].tobytes(), pH[0].tobytes(), 0, pr[0].tobytes())
        preflight, pf_ok = inline_preflight(prover.transport, sub_root, rank, world, dist, torch, comm_device, ctx, allow_fallback=args.allow_exchange_fallback)
        if rank == 0:
            log("inline preflight %s in %.2f s (%s; ncclCommCount %s)" % ("ok" if pf_ok else "FAILED", preflight["seconds"], preflight["exchange_path"], preflight["rccl_ranks_in_library_communicator"]))
        if not pf_ok:
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": None, "unit": "entities/s", "n_gpus": world, "preflight_failed": True,
                                  "multi_gpu": {"preflight": preflight}, "complete": True,
                                  "note": "the multi-GPU transport failed its checks before the first step: nothing was timed (pass "
                                          "--allow-exchange-fallback to time the torch.distributed fallback anyway)",
                                  "wall_s_since_process_start": time.time() - T_PROC0}), flush=True)
            if prover.comm is not None:
                prover.comm.abort()
                prover.comm = None
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def agree_min(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    # ---- warm-up: one step always; a second one (when asked for) is the warm step the timed loop is sized from -- the first one is
    # cold (allocations, code objects: ~3 s more) and would under-count what fits; further ones only if they fit
    deadline = T_PROC0 + args.budget_s
    # what follows the timed region, as measured on the driver's run of round 3 (BENCH_r03: 34 s in all): CPU baseline 10 s + parity /
    # verification legs ~9 s + the N = 1 secondary legs ~15 s + ~22 s for the full-size host-buffer leg (which falls back to a 2^16 sample
    # when the budget is short); 15 % on top
    post_reserve = 1.15 * ((args.cpu_budget_s + 4.0 if not args.no_cpu_baseline else 0.0) + 9.0 + (43.0 if (world == 1 and not args.no_secondary) else 0.0))
    warm_req, steps_req = args.warmup, args.steps
    warm_done = 0
    t_step = None
    for _ in range(min(2, warm_req)):
        sync()
        t0 = time.perf_counter()
        prover.step(PAD_SEED, NONCE_SEED, n_bits)
        sync()
        t_step = time.perf_counter() - t0
        warm_done += 1
        if rank == 0:
            log("warm-up step %d: %.2f s" % (warm_done, t_step))
    extra_warm, steps = plan_steps(steps_req, warm_req - (warm_done - 1), t_step, deadline - time.time() - post_reserve)
    if steps < steps_req and world == 1 and not args.no_secondary:
        # short of time (a slow first `import torch`, a slow box): the requested steps come before the secondary legs, which are
        # then run only as far as the budget still allows after the timed region
        lean_reserve = 1.15 * ((args.cpu_budget_s + 4.0 if not args.no_cpu_baseline else 0.0) + 9.0)
        _, steps = plan_steps(steps_req, 1, t_step, deadline - time.time() - lean_reserve)
        extra_warm = 0
    extra_warm, steps = agree_min(extra_warm), agree_min(steps)
    for _ in range(extra_warm):
        prover.step(PAD_SEED, NONCE_SEED, n_bits)
        warm_done += 1
    if rank == 0 and (steps != steps_req or warm_done != warm_req):
        log("wall budget %.0f s: running %d timed steps (asked %d) after %d warm-up (asked %d)" % (args.budget_s, steps, steps_req, warm_done, warm_req))

    # ---- timed region: EXACTLY `steps` steps between barrier + synchronize on both sides
    acc = {"tree_ms": 0.0, "prove_ms": 0.0, "msm_ms": 0.0, "msm_launches": 0, "proofs": 0, "mat_ms": 0.0, "mat_launches": 0, "msm_kernels": 0,
           "mat_kernels": 0, "msm_all_ms": 0.0, "msm_span_ms": 0.0}
    stats = None
    phase_rows = []                                # per timed step, this rank: ShardedProver.PHASE_KEYS (host ms / device us)
    sync()
    t0 = time.perf_counter()
    for s in range(steps):
        stats = prover.step(PAD_SEED, NONCE_SEED, n_bits)
        for k in acc:
            acc[k] += getattr(stats, k)
        phase_rows.append([prover.phases[k] for k in prover.PHASE_KEYS] + [stats.tree_ms, stats.prove_ms])
        if rank == 0:
            log("step %d/%d done (%.1f s since the timed region began; tree %.1f ms, proofs %.0f ms)" %
                (s + 1, steps, time.perf_counter() - t0, stats.tree_ms, stats.prove_ms))
    sync()
    elapsed = time.perf_counter() - t0
    free_b, total_b = torch.cuda.mem_get_info()            # what the context, the tree and the prover's scratch hold after the timed steps
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # every rank's per-step phase times -> rank 0 (after the timed region): what an N > 1 line needs to explain itself
    per_rank = gather_phase_rows(dist, torch, phase_rows, world, comm_device)
    # The library's communicator has done its work: every rank finalises it HERE, together (they have just left the same
    # all-reduce), not one by one at interpreter exit while the others may already be gone.
    exchange_path, comm_ranks, comm_error = prover.exchange_path, prover.comm_ranks, prover.comm_error
    if prover.comm is not None:
        prover.comm.close()
        prover.comm = None
    if rank != 0:
        if dist is not None:
            dist.barrier()                     # leave together with rank 0 (which still runs its parity / verification legs)
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / steps
    value = n_total * steps / elapsed
    ab_prove, ab_tree = algorithmic_bytes(height, n_bits, lg_total)
    # Dominant kernel: the fixed-base MSM of the plain rounds (S commitment + never-fold rounds).  Large calls run it
    # generator-stationary -- k_rp_msm_gs, a few hundred tile launches per MSM -- smaller ones proof-stationary (k_rp_msm<0, .>).
    # The library brackets every MSM with HIP events on the stream it runs on (MsmTiming) and counts the kernel launches inside
    # the brackets; the materialisation of the folded generators (k_rp_mat_gs / k_rp_msm<1, .>) is bracketed separately.
    # The library reports UNIONS of the bracket intervals (time during which a plain MSM / a materialisation / either was running),
    # which with several chunks in flight (DAPOL_STREAMS > 1) is less than the sum of the bracket lengths.
    plain_s, mat_s = acc["msm_ms"] / 1e3, acc["mat_ms"] / 1e3
    msm_s = acc["msm_all_ms"] / 1e3 if acc["msm_all_ms"] > 0 else plain_s + mat_s
    launches = int(acc["msm_kernels"]) or int(acc["msm_launches"])
    gs = int(acc["msm_kernels"]) > int(acc["msm_launches"])
    # Two clocks per launch.  `avg_launch_ms` is the SPAN of a launch -- bracket lengths summed / launches inside: what
    # `rocprofv3 --kernel-trace --stats` of this command reports as the kernel's average.  `avg_launch_ms_exclusive` divides the time
    # during which such a launch was running at all (the union of the brackets) by the launches.  With ONE chunk in flight (the
    # default since round 4) brackets never overlap and the two are the same number; with DAPOL_STREAMS=2 the launches of the two
    # chunks overlap pairwise, the span is about twice the exclusive time, and the exclusive time is an upper bound of the chip-wide
    # cost of one launch (the other chunk's kernels also run inside those intervals).  `achieved` uses the exclusive time.
    avg_launch_span_ms = (acc["msm_span_ms"] if acc["msm_span_ms"] > 0 else acc["msm_ms"]) / max(1, launches)
    avg_launch_ms = acc["msm_ms"] / max(1, launches)
    # algorithmic bytes per launch (6,384 B of compulsory traffic per entity on the prove path, spread evenly over the
    # time of the fixed-base MSM kernels) = proofs x 6,384 B x (share of that time spent in this kernel) / its launches
    bytes_per_launch = acc["proofs"] * ab_prove * (plain_s / (plain_s + mat_s) if plain_s + mat_s > 0 else 0.0) / max(1, launches)
    ach = (bytes_per_launch / 1e9) / (avg_launch_ms / 1e3) if avg_launch_ms > 0 else 0.0
    sha = kernel_src_sha()
    full_size = n_per_gpu >= 65536 and height == 32 and n_bits == 64
    prof = from_profiles(sha, full_size)
    traffic = None
    valu = None
    pmc = prof.get("msm_pmc")
    if pmc and pmc.get("current_build") and full_size and gs == ("k_rp_msm_gs" in str(pmc.get("kernel"))):
        traffic = pmc.get("hbm_bytes_per_launch")                      # PMC passes of THIS build (tools/pmc_only.sh), per launch of the same kernel
        roof = prof.get("valu_roof", {})
        cyc, reg = pmc.get("cycles_per_valu_inst_per_simd"), roof.get("register_only_cycles_per_valu_inst")
        valu = {"cycles_per_valu_inst_per_simd": cyc, "register_only_cycles_per_valu_inst": reg,
                "valu_issue_frac": (reg / cyc) if cyc and reg else None,
                "clock_ghz": pmc.get("effective_clock_GHz"), "clock_frac_of_2p4": (pmc.get("effective_clock_GHz") or 0) / 2.4,
                "valu_instructions_per_launch": pmc.get("SQ_INSTS_VALU"), "l2_hit_rate": pmc.get("l2_hit_rate"),
                "l2_read_miss_latency_cycles": pmc.get("l2_read_miss_latency_cycles")}
    roofline = {
        "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": traffic,
        "traffic_over_algorithmic": (traffic / bytes_per_launch) if traffic and bytes_per_launch else None,
        "binding_resource": "integer VALU issue (no MFMA: 255-bit modular arithmetic); see `valu`",
        "valu": valu,
        "kernel": "k_rp_msm_gs" if gs else "k_rp_msm", "launches": launches, "avg_launch_ms": avg_launch_span_ms,
        "avg_launch_ms_exclusive": avg_launch_ms, "launches_overlapping": (avg_launch_span_ms / avg_launch_ms) if avg_launch_ms > 0 else None,
        "kernel_time_s": plain_s,
        "msm_brackets": int(acc["msm_launches"]), "avg_msm_ms": acc["msm_ms"] / max(1, int(acc["msm_launches"])),
        "materialisation": {"kernel": "k_rp_mat_gs" if gs else "k_rp_msm<1, 32>", "launches": int(acc["mat_kernels"]), "kernel_time_s": mat_s},
        "algorithmic_bytes_per_entity": ab_prove, "algorithmic_bytes_per_launch": bytes_per_launch,
        "whole_step": {"algorithmic_bytes_per_entity": ab_prove + ab_tree,
                       "achieved": (n_per_gpu * steps * (ab_prove + ab_tree) / 1e9) / elapsed,
                       "frac": (n_per_gpu * steps * (ab_prove + ab_tree) / 1e9) / elapsed / PEAK_HBM_GBS,
                       "note": "tree + proof bytes (SURVEY 8d: 9,136 B at 2^20 x H=32) over the whole step time, per GPU"},
        "kernel_share_of_step": plain_s / elapsed if elapsed > 0 else None,
        "msm_share_of_step": msm_s / elapsed if elapsed > 0 else None,
        "note": "achieved = ALGORITHMIC bytes per launch of the dominant kernel (SURVEY 8d, prove path: 6,384 B per entity at H=32, spread over the "
                "fixed-base MSM time; this kernel's share / its launches) / avg_launch_ms_exclusive (HIP events around every MSM on the stream "
                "it runs on; union of those intervals / tile launches inside).  avg_launch_ms is the per-launch SPAN a kernel trace shows: the "
                "same number with one chunk in flight (the default), larger with DAPOL_STREAMS=2, whose launches overlap pairwise.  By construction ~1e-5 of the HBM peak: the path does ~1e7 modular "
                "multiplications per 9 KB of compulsory traffic; what binds is integer-VALU issue (`valu`, from the PMC pass of this "
                "build).  `traffic` = L2<->fabric bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are counted by "
                "that counter: for the generator-stationary kernel most of them ARE Infinity-Cache hits, DESIGN.md section 5); it and "
                "`valu` are only set from a PMC pass of the build that is running; older profiles are under from_profiles with their hash.",
        "kernel_src_sha": sha, "from_profiles": prof}
    multi_gpu = rank_decomposition(per_rank, list(prover.PHASE_KEYS) + ["tree_device_ms", "prove_device_ms"], world, exchange_path)
    if preflight is not None and isinstance(multi_gpu, dict):
        multi_gpu["preflight"] = preflight

    def make_line(cpu, parity, secondary, complete):
        return {
            "metric": METRIC,
            "value": value, "unit": "entities/s", "n_gpus": world, "steps": steps, "warmup": warm_done,
            "steps_requested": steps_req, "warmup_requested": warm_req,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "int32 limbs (255-bit modular integers)", "data": "synthetic",
            "config": {"workload": "2^%d entities%s, height=%d, %d-bit range proofs, padding policy, aggregation_factor=height, BLAKE3 node hash"
                                   % (lg_total if scaling == "strong" or world == 1 else args.log2_entities,
                                      " in total (strong scaling: 2^%d per GPU)" % (lg_total - (world.bit_length() - 1)) if scaling == "strong" and world > 1
                                      else (" per GPU (weak scaling)" if world > 1 else ""), height, n_bits),
                       "entities_total": n_total, "entities_per_gpu": n_per_gpu, "proof_bytes": int(stats.proof_bytes // max(1, stats.proofs)),
                       "sharding": "none" if world == 1 else "top-level subtrees, all-gather of %d subtree roots" % world,
                       "exchange": exchange_path, "rccl_ranks_in_library_communicator": comm_ranks,
                       "exchange_fallback_reason": comm_error,
                       "profile": args.profile, "window_bits": int(ctx.get_options().window_bits), "high_half_rows": bool(ctx.get_options().high_half_rows > 0),
                       "device_memory_in_use_gb": (total_b - free_b) / 1e9},
            "phases_ms": {"tree_build": acc["tree_ms"] / steps, "prove": acc["prove_ms"] / steps},
            "multi_gpu": multi_gpu,
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "secondary": secondary,
            "checksum": "%016x" % stats.checksum,
            "complete": complete,
            "wall_s_since_process_start": time.time() - T_PROC0,
        }

    # The headline goes out NOW, as soon as the timed region has ended: the legs below (CPU baseline, parity, secondary: ~60 s) come
    # after it, and a run killed at the driver's limit must not lose the line.  The same line is printed again at the end with
    # `cpu_baseline`, `parity` and `secondary` filled in ("complete": true); whoever reads the output takes the LAST line.
    print(json.dumps(make_line(None, None, None, False)), flush=True)
    cpu = None
    parity = None
    if not args.no_cpu_baseline and os.path.exists(ORACLE_LIB) and prover.w is not None:
        log("CPU baseline + parity legs")
        ns_max = 4096
        sample_ids = idx[:: max(1, n_per_gpu // ns_max)][:ns_max]
        sv, sr = prover.sample_paths(sample_ids, PAD_SEED)
        # Rank 0, the same bounded leg at every N (the other ranks wait in the closing barrier), so that a line of the scaling series
        # is self-contained; it is also the parity check of rank 0's shard (the sharded tree must give the oracle's bytes).
        cpu, cpu_bytes, ns, ps = cpu_baseline(ORACLE_LIB, height, n_bits, idx, v, r, (sv, sr, sample_ids), budget_s=args.cpu_budget_s)
        gpu_bytes = prover.sample_proofs(sample_ids[:ns], ps)
        parity = {"proofs_compared": ns, "bit_exact": bool(gpu_bytes.tobytes() == cpu_bytes)}
        if world > 1 and cpu is not None:
            cpu["note_multi_gpu"] = "timed on rank 0's host process while ranks 1..%d idle in a barrier; the sample is of rank 0's shard" % (world - 1)
    # encode -> verify round trip at full size: sampled inclusion proofs of the timed run through DapolProof::verify on the GPU
    if prover.w is not None:
        nv = min(2048, n_per_gpu)
        vids = np.ascontiguousarray(idx[:: max(1, n_per_gpu // nv)][:nv])
        vpos = np.searchsorted(idx, vids)
        _, _, vC, vH = prover.sample_paths(vids, PAD_SEED, with_nodes=True)
        lC, lH = ctx.commit_hash_batch(v[vpos], r[vpos])
        vproofs = prover.sample_proofs(vids, int(stats.proof_bytes // max(1, stats.proofs)))
        rC, rH = prover.root[0], prover.root[1]
        okv = ctx.verify_entities(height, vids, lC, lH, vC, vH, rC, rH, capi.POLICY_PADDING, height, n_bits, vproofs, verify_seed=os.urandom(32))
        parity = dict(parity or {}, inclusion_proofs_verified_on_gpu=int(okv.sum()), inclusion_proofs_checked=int(len(okv)))
    secondary = None
    if world == 1 and not args.no_secondary and prover.w is not None and (deadline + 15.0 - time.time()) < 20.0:
        secondary = {"skipped": "wall budget: %.0f s left of %.0f" % (deadline + 15.0 - time.time(), args.budget_s)}
    elif world == 1 and not args.no_secondary and prover.w is not None:
        log("secondary legs (splitting policy, API layout, host-buffer entry points)")
        try:
            secondary = secondary_legs(ctx, capi, prover, height, n_bits, idx, v, r, n_per_gpu, host_leg_deadline=deadline - 30.0, t_step_hint=ms_per_step / 1e3)      # (the full-size host-buffer leg only if it ends 60 s before the driver's 600 s)
        except Exception as e:                                   # the headline line must not die with a secondary leg
            secondary = {"error": repr(e)}
    print(json.dumps(make_line(cpu, parity, secondary, True)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def secondary_legs(ctx, capi, prover, height, n_bits, idx, v, r, n_per_gpu, host_leg_deadline=None, t_step_hint=None):
    """SURVEY 8(d)'s secondary measurements, each on a BOUNDED sample of the same workload (the sample is named in the entry):
      splitting    the reference bench's other policy (benches/dapol.rs:71-78).  At aggregation_factor = height = 32 the splitting
                   plan is ONE 32-party proof -- the padding policy's proof, byte for byte -- so that is checked, and the two
                   policies are timed where they differ: aggregation factor 24 (splitting: 16 + 8 parties; padding: 32) plus 8
                   individual proofs per entity;
      api_layout   Dapol::new's way in (src/dapol/mod.rs:323-441): dapol_build_leaf_nodes over 2^20 ids `id-%08d` (BLAKE3 index
                   derivation, collision resolution, sort) -> tree build -> proofs of a sample;
      host_buffers the PCIe-inclusive rate: tree from host arrays + dapol_prove_entities returning paths and proofs to the host."""
    import torch
    out = {}
    sync = torch.cuda.synchronize
    w = prover.w
    ns = min(n_per_gpu, 1 << 16)
    ps = proof_bytes(height, n_bits)
    pad32 = w.proofs(0, min(64, ns), ps).copy()                                     # of the timed run (padding, aggregation 32)
    sync(); t0 = time.perf_counter()
    st = w.prove(NONCE_SEED, n_bits, first=0, count=ns, upper=prover.upper, policy=capi.POLICY_SPLITTING, aggregation_factor=height)
    sync(); dt = time.perf_counter() - t0
    same = bool(np.array_equal(w.proofs(0, min(64, ns), ps), pad32))
    out["splitting"] = {"sample": "first 2^%d entities of the workload" % (ns.bit_length() - 1),
                        "agg32_entities_per_s": ns / dt, "agg32_bytes_equal_padding": same}
    n24 = min(n_per_gpu, 1 << 14)
    a24 = min(24, max(1, height - 8))                           # (24 on the headline's height-32 tree)
    for pol, name in ((capi.POLICY_SPLITTING, "splitting"), (capi.POLICY_PADDING, "padding")):
        w.prove(NONCE_SEED, n_bits, first=0, count=min(256, n24), upper=prover.upper, policy=pol, aggregation_factor=a24)      # warm the shapes
        sync(); t0 = time.perf_counter()
        st = w.prove(NONCE_SEED, n_bits, first=0, count=n24, upper=prover.upper, policy=pol, aggregation_factor=a24)
        sync(); dt = time.perf_counter() - t0
        out["splitting"]["agg24_%s_entities_per_s" % name] = n24 / dt
        out["splitting"]["agg24_%s_proof_bytes" % name] = int(st.proof_bytes // max(1, st.proofs))
    out["splitting"]["agg24_sample"] = "first 2^%d entities, aggregation factor %d on the height-%d tree" % (n24.bit_length() - 1, a24, height)
    out["small_parties"] = small_parties_leg(ctx, capi, w, prover.upper, height, n_bits, n24, a24)
    # leave the workload as the timed run left it (sampled proofs are read from it afterwards)
    w.prove(NONCE_SEED, n_bits, upper=prover.upper)
    # ---- API layout
    n_ids = min(1 << 20, 1 << max(1, height - 1))               # (2^20 on the headline; never beyond Dapol::new's sparsity rule 2^height >= 2n)
    ids = np.char.add("id-", np.char.zfill(np.arange(n_ids).astype("U8"), 8)).astype("S11")
    packed = ids.tobytes()
    off = (np.arange(n_ids + 1, dtype=np.uint64) * 11).astype(np.uint32)
    vals = np.random.default_rng(3).integers(0, 1 << 32, size=n_ids, dtype=np.uint64)
    t0 = time.perf_counter()
    lf = ctx.build_leaf_nodes_packed(packed, off, packed, off, vals, b"bench-audit-seed", height)
    t_leaf = time.perf_counter() - t0
    wa = capi.Workload(ctx, height, lf["leaf_idx"], lf["v"], lf["r"])
    sync(); t0 = time.perf_counter()
    root, sta = wa.build(PAD_SEED)
    sync(); t_build = time.perf_counter() - t0
    na = min(1 << 16, n_ids)
    sync(); t0 = time.perf_counter()
    sta = wa.prove(NONCE_SEED, n_bits, first=0, count=na, stats=sta)
    sync(); t_prove = time.perf_counter() - t0
    sample = lf["leaf_idx"][:na:max(1, na // 256)][:256]
    pos = np.searchsorted(lf["leaf_idx"], sample)
    _, _, sC, sH = wa.paths(sample, with_nodes=True)
    got = np.stack([wa.proofs(int(p), 1, ps)[0] for p in pos])
    lC, lH = ctx.commit_hash_batch(lf["v"][pos], lf["r"][pos])
    okv = ctx.verify_entities(height, sample, lC, lH, sC, sH, root[0], root[1], capi.POLICY_PADDING, height, n_bits, got)
    out["api_layout"] = {"ids": "2^%d ids id-%%08d, BLAKE3 index derivation, height %d" % (n_ids.bit_length() - 1, height), "build_leaf_nodes_s": t_leaf,
                         "build_leaf_nodes_entities_per_s": n_ids / t_leaf, "host_inclusive": True,
                         "indexes_distinct": bool(len(np.unique(lf["leaf_idx"])) == n_ids),
                         "root_value_equals_sum": bool(root[2] == int(vals.sum())), "tree_build_ms": t_build * 1e3,
                         "prove_sample": "first 2^%d leaves in index order" % (na.bit_length() - 1), "prove_entities_per_s": na / t_prove,
                         "sampled_proofs_verified": int(okv.sum()), "sampled_proofs_checked": int(len(okv))}
    wa.close()
    # ---- host buffers (PCIe-inclusive): the WHOLE workload once when the wall budget allows (~22 s at 2^20), else a 2^16 sample
    full = host_leg_deadline is None or (host_leg_deadline - time.time()) >= 1.6 * (t_step_hint or 20.0) + 6.0
    nh = n_per_gpu if full else min(n_per_gpu, 1 << 16)
    tree = capi.Tree(ctx, height, idx[:1024], v[:1024], r[:1024], PAD_SEED)
    tree.prove_entities(idx[:1024], capi.POLICY_PADDING, height, n_bits, NONCE_SEED)      # warm the call path
    tree.close()
    t0 = time.perf_counter()
    tree = capi.Tree(ctx, height, idx[:nh], v[:nh], r[:nh], PAD_SEED)                      # H2D of the entity arrays + build
    t1 = time.perf_counter()
    pC, pH, proofs = tree.prove_entities(idx[:nh], capi.POLICY_PADDING, height, n_bits, NONCE_SEED)   # D2H: 2 x 32 x 32 B of path + 992 B of proof per entity
    t_host = time.perf_counter() - t0
    t_build_host = t1 - t0
    tree.close()
    bytes_in = int(idx[:nh].nbytes + v[:nh].nbytes + r[:nh].nbytes)
    bytes_out = int(pC.nbytes + pH.nbytes + proofs.nbytes)
    del pC, pH, proofs
    # what the copies alone cost: the same byte counts between PAGEABLE host memory (what a caller's arrays are) and the device
    hin = torch.empty(bytes_in, dtype=torch.uint8)
    sync(); t0 = time.perf_counter(); din = hin.cuda(); sync(); h2d_ms = 1e3 * (time.perf_counter() - t0)
    dout = torch.empty(bytes_out, dtype=torch.uint8, device="cuda")
    sync(); t0 = time.perf_counter(); hout = dout.cpu(); sync(); d2h_ms = 1e3 * (time.perf_counter() - t0)
    del din, dout, hin, hout
    out["host_buffers"] = {"sample": ("the whole workload: 2^%d entities" if nh == n_per_gpu else "2^%d entities") % (nh.bit_length() - 1) +
                                     " through dapol_tree_build + dapol_prove_entities (host arrays in, paths and proofs back)",
                           "full_workload": bool(nh == n_per_gpu), "entities_per_s": nh / t_host, "seconds": t_host, "tree_build_incl_h2d_ms": 1e3 * t_build_host,
                           "bytes_in": bytes_in, "bytes_returned": bytes_out, "bytes_returned_per_entity": bytes_out // nh,
                           "h2d_ms_same_bytes_pageable": h2d_ms, "d2h_ms_same_bytes_pageable": d2h_ms,
                           "note": "PCIe-inclusive; never the headline `value` (inputs resident in HBM).  h2d / d2h: plain copies of the same byte "
                                   "counts between pageable host memory and the device, timed on their own"}
    return out


# register-only rate of table additions on one MI355X: 256 CUs x 4 SIMDs x 64 lanes per 3,902 cycles (tools/ubench_madd.hip) at 2.4 GHz
ADDITION_RATE = 256 * 4 * 64 / 3902.0 * 2.4e9


def small_parties_leg(ctx, capi, w, upper, height, n_bits, n_ent, agg_hi, log2_proofs=17):
    """Proofs of FEW parties in large batches: what the policies leave per sibling beyond aggregation_factor
    (src/range/padding.rs:104-112, splitting.rs:118-123, src/range/mod.rs:48-62).  Two kinds of rows:
      policy   entities/s of the device-resident workload at aggregation factors `agg_hi` (24 on the headline tree) and 8, padding
               policy, on the first `n_ent` entities (library HIP events around the proving pipeline);
      batch    proofs/s of dapol_range_prove_batch over 2^17 proofs of (64 bits, m parties), m = 1, 2, 4, 8 -- device milliseconds
               of the proving pipeline (dapol_diag_range_prove_ms: inputs in HBM, proofs not yet copied back) -- each with its
               fraction of the register-only table-addition rate when the proof is charged its never-fold additions
               ((1 + lg N) MSMs of 2 N terms x 15 windows)."""
    res = {"policy": {}, "batch": {}, "addition_rate_per_s": ADDITION_RATE}
    for agg in sorted({agg_hi, min(8, height)}, reverse=True):
        w.prove(NONCE_SEED, n_bits, first=0, count=min(256, n_ent), upper=upper, policy=capi.POLICY_PADDING, aggregation_factor=agg)
        st = w.prove(NONCE_SEED, n_bits, first=0, count=n_ent, upper=upper, policy=capi.POLICY_PADDING, aggregation_factor=agg)
        res["policy"]["padding_agg%d" % agg] = {"entities": n_ent, "device_ms": st.prove_ms, "entities_per_s": n_ent / st.prove_ms * 1e3,
                                                "individual_proofs_per_entity": height - agg, "proof_bytes_per_entity": int(st.proof_bytes // max(1, st.proofs))}
    b = 1 << log2_proofs
    rng = np.random.default_rng(17)
    ms_dev = ctypes.c_double()
    sid = np.arange(b, dtype=np.uint64)
    for m in (1, 2, 4, 8):
        vv = rng.integers(0, 2**63, size=b * m, dtype=np.uint64)
        rr = rng.integers(0, 256, size=(b * m, 32), dtype=np.uint8)
        rr[:, 31] &= 0x0F
        ctx.range_prove_batch(n_bits, m, vv[:256 * m], rr[:256 * m], nonce_seed=NONCE_SEED, stream_id=sid[:256])         # warm the shapes
        proofs = ctx.range_prove_batch(n_bits, m, vv, rr, nonce_seed=NONCE_SEED, stream_id=sid)
        capi.lib().dapol_diag_range_prove_ms(ctypes.byref(ms_dev))
        N = n_bits * m
        adds = (1 + (N.bit_length() - 1)) * 2 * N * 15
        C, _ = ctx.commit_hash_batch(vv[:64 * m], rr[:64 * m])
        ok = ctx.range_verify_batch(n_bits, m, proofs[:64], C.reshape(64, m, 32), verify_seed=os.urandom(32))
        res["batch"]["64x%d" % m] = {"proofs": b, "device_ms": ms_dev.value, "proofs_per_s": b / ms_dev.value * 1e3,
                                     "table_additions_per_proof": adds, "frac_of_addition_rate": b / ms_dev.value * 1e3 * adds / ADDITION_RATE,
                                     "first_64_verified_on_gpu": int(ok.sum())}
        del proofs, vv, rr
    return res


def preflight_collectives(tr, root, rank, world, dist, torch, comm_device, iters, ctx, lib_comm, merge=None, local_ok=True):
    """The collective half of --preflight over a ShardTransport `tr` that is already set up: `iters` rounds of the step's exchange
    (all-gather of the subtree roots + replicated top levels) and of its final reduce, through the transport exactly as the timed
    loop drives it (library communicator + agreement after every collective, or torch.distributed); the same exchange over
    torch.distributed for comparison; every rank's global root equal; the reduce's sum right.  -> (checks, timings, global root,
    ok on EVERY rank).  `merge` overrides the merge primitive of the torch path (CPU tests plug the oracle in)."""
    from dapol_amd import capi
    from dapol_amd.sharded import exchange_records, top_levels, pack_record, unpack_records
    checks, res = {}, {}

    def timed(fn, n):
        ts = []
        out = None
        for _ in range(n):
            t0 = time.perf_counter()
            out = fn()
            ts.append(1e6 * (time.perf_counter() - t0))
        ts.sort()
        return out, {"iters": n, "median_us": ts[len(ts) // 2], "min_us": ts[0], "p90_us": ts[int(0.9 * (len(ts) - 1))], "max_us": ts[-1]}

    if world > 1:
        do_exchange, do_reduce = (lambda: tr.exchange(root)), (lambda: tr.reduce_u64(rank + 1, "sum"))
    else:      # ONE rank: the transport short-cuts (nothing to exchange); call the library's collectives + agreement as N ranks would
        do_exchange = lambda: (tr._library(lambda: tr.comm.exchange(root))[1] if tr.comm is not None else (root, None))
        do_reduce = lambda: (tr._library(lambda: int(tr.comm.allreduce([rank + 1], capi.REDUCE_SUM)[0]))[1] if tr.comm is not None else rank + 1)
    dist.barrier()
    (groot, upper), res["exchange_host"] = timed(do_exchange, iters)
    red, res["reduce_host"] = timed(do_reduce, iters)
    checks["reduce_sum_correct"] = red == world * (world + 1) // 2
    checks["transport_still_on_the_library_communicator"] = (tr.comm is not None) == lib_comm
    checks["upper_siblings_one_per_shard_bit"] = (upper is None and world == 1) or (upper is not None and len(upper[0]) == world.bit_length() - 1)
    if tr.comm is not None:
        tm = tr.comm.timing()
        res["library_device_us"] = {"allgather_mean": tm.sum_allgather_us / max(1, tm.exchanges), "top_levels_mean": tm.sum_top_levels_us / max(1, tm.exchanges),
                                    "allreduce_mean": tm.sum_allreduce_us / max(1, tm.reduces), "exchanges": int(tm.exchanges), "reduces": int(tm.reduces)}
        res["library_host_us"] = {"exchange_mean": tm.sum_exchange_host_us / max(1, tm.exchanges), "reduce_mean": tm.sum_reduce_host_us / max(1, tm.reduces)}
    if world > 1:          # the same exchange over torch.distributed (the fallback transport): comparison + cross-check of the root
        def torch_exchange():
            buf = exchange_records(dist, torch, pack_record(root), world, comm_device)
            return top_levels(ctx, unpack_records(buf, world), rank, merge=merge)
        (troot, _), res["exchange_host_torch"] = timed(torch_exchange, max(1, iters // 5))
        checks["library_root_equals_torch_path_root"] = troot == groot
    dig = np.frombuffer(hashlib.sha256(groot[0] + groot[1] + int(groot[2]).to_bytes(8, "little") + groot[3]).digest()[:8], np.int64).copy()
    mine = torch.from_numpy(dig).to(comm_device)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    checks["global_root_equal_on_every_rank"] = all(int(a.item()) == int(mine.item()) for a in allr)
    flag = torch.tensor([1 if (local_ok and all(bool(c) for c in checks.values())) else 0], dtype=torch.int64, device=comm_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return checks, res, groot, int(flag.item()) == 1


def mode_preflight(args):
    """`bench.py --gpus N --preflight`: < 30 s, no proving.  What a multi-GPU run needs before it is worth 10 minutes of a node, and
    what a FAILED one needs to leave behind: the link topology (`rocm-smi --showtopo`, to stderr, before this process touches the GPU),
    torch.distributed up, the library's own RCCL communicator up (non-blocking creation with a deadline) with ncclCommCount == N, a
    small shard tree per rank, then --preflight-iters rounds of the step's two collectives through the SAME transport object the
    timed loop uses (dapol_shard_exchange = ncclAllGather + replicated top levels, dapol_comm_allreduce_u64 = ncclAllReduce, each
    followed by the ranks' agreement), timed on the host and by the library's HIP events; the same over torch.distributed for
    comparison; every rank's global root equal; the reduce giving N (N + 1) / 2.  One JSON line ("metric": "preflight ..."),
    exit code 0 iff every check passed.  Works with ONE rank too (a 1-GPU box exercises every call site)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    checks, notes = {}, {}
    if rank == 0:
        try:      # before anything here initialises the GPU; a child process, never an exec
            topo = subprocess.run(["rocm-smi", "--showtopo"], capture_output=True, text=True, timeout=25)
            print(topo.stdout[-6000:], file=sys.stderr, flush=True)
            notes["rocm_smi_showtopo_rc"] = topo.returncode          # diagnostics, not a check: the links are what they are
        except Exception as e:
            notes["rocm_smi_showtopo"] = repr(e)
    import torch
    import torch.distributed as dist
    backend = os.environ.get("DAPOL_BENCH_BACKEND", "nccl")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    if "MASTER_PORT" not in os.environ:            # one rank without a launcher
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    t0 = time.perf_counter()
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    comm_device = "cuda" if backend == "nccl" else "cpu"
    x = torch.ones(1, dtype=torch.int64, device=comm_device)
    dist.all_reduce(x)
    checks["torch_distributed_up"] = int(x.item()) == world
    t_torch = time.perf_counter() - t0
    log("rank %d/%d: torch.distributed (%s) up in %.1f s" % (rank, world, backend, t_torch))
    from __graft_entry__ import build
    if rank == 0:
        build()
    dist.barrier()
    from dapol_amd import capi
    from dapol_amd.sharded import ShardTransport, exchange_records, top_levels, pack_record, unpack_records
    t0 = time.perf_counter()
    # tree nodes only need the rows of B and B_blinding: the narrowest windows, one party -- a context in a fraction of a second
    ctx = capi.Context(local_rank, 1, options=capi.Options(window_bits=8, high_half_rows=-1))
    t_ctx = time.perf_counter() - t0
    tr = ShardTransport(ctx, rank, world, dist, torch, comm_device)
    t0 = time.perf_counter()
    tr.create_comm(timeout_s=args.preflight_timeout_s, even_alone=True)
    t_comm = time.perf_counter() - t0
    lib_comm = tr.comm is not None
    if comm_device == "cuda":
        checks["library_communicator_up"] = lib_comm
        checks["nccl_comm_count_equals_n"] = bool(lib_comm and tr.comm_ranks == world)
    else:
        notes["library_communicator"] = "not created: the ranks share torch.distributed over %s (a test hook; RCCL needs one GPU per rank)" % backend
    log("rank %d: library communicator %s in %.1f s (%s)" % (rank, "up" if lib_comm else "NOT up", t_comm, tr.comm_error or "ncclCommCount %s" % tr.comm_ranks))
    height, n = 32, 1 << 10
    bits = world.bit_length() - 1
    idx, v, r = synth_inputs(n * world, height, rank * n, n)
    tree = capi.Tree(ctx, height, idx, v, r, PAD_SEED, shard_bits=bits)
    root = tree.root()

    pc_checks, res, groot, ok = preflight_collectives(tr, root, rank, world, dist, torch, comm_device, max(1, args.preflight_iters), ctx, lib_comm, local_ok=all(bool(c) for c in checks.values()))
    checks.update(pc_checks)
    path, comm_ranks, comm_err = tr.path, tr.comm_ranks, tr.comm_error
    tree.close()
    tr.close()
    if rank == 0:
        print(json.dumps({"metric": "preflight of the multi-GPU path (no proving)", "ok": ok, "n_gpus": world, "backend": backend,
                          "checks": checks, "notes": notes, "exchange_path": path, "rccl_ranks_in_library_communicator": comm_ranks,
                          "comm_error": comm_err, "timings": res,
                          "setup_s": {"torch_distributed": t_torch, "context": t_ctx, "library_communicator": t_comm},
                          "shard_tree": "2^10 leaves per rank, height %d, %d shard bit(s)" % (height, bits),
                          "global_root_C": groot[0].hex(), "wall_s_since_process_start": time.time() - T_PROC0}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def spawn_ranks(args):
    """`--gpus N` without a launcher: this process has not touched the GPU (no torch import, no HIP call), so it may start the
    ranks itself -- N fresh children under torch.distributed.run, never an exec -- relay rank 0's JSON line, and leave with
    the launcher's exit code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DAPOL_BENCH_T0"] = repr(T_PROC0)                     # the ranks count the wall budget from THIS process's start
    log("no launcher (WORLD_SIZE unset): starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:8]) + " ..."))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    got_json = False
    for ln in p.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            got_json = True
            print(ln, flush=True)                   # at once: rank 0 prints the headline right after the timed region and the complete
        else:                                       # line at the end; a run cut short in between must not lose the first
            print(ln, file=sys.stderr, flush=True)
    rc = p.wait()
    if not got_json and rc == 0:
        rc = 1
    sys.exit(rc)


def mode_build(args):
    """The reference's `build` criterion group (benches/dapol.rs:24-57): Dapol::new_blank + build over pre-made leaf nodes,
    strided leaves, h in {16, 32} x N in {1024, 2048, 4096}; plus the headline tree (2^20 x 32).  Inputs resident in HBM."""
    import torch
    from __graft_entry__ import build, ORACLE_LIB
    build()
    from dapol_amd import capi
    ctx = capi.Context(0, 32)
    ref = ctypes.CDLL(ORACLE_LIB) if os.path.exists(ORACLE_LIB) and not args.no_cpu_baseline else None
    if ref is not None:
        ref.ref_tree_build.restype = ctypes.c_void_p
    rows = []
    cases = [(h, n) for h in (16, 32) for n in (1024, 2048, 4096)] + [(args.height, 1 << args.log2_entities)]
    inputs = {}
    for h, n in cases:                                       # every GPU measurement first: the CPU leg's OpenMP team would otherwise
        idx, v, r = synth_inputs(n, h, 0, n)                 # compete with the launching thread for the box's CPU share
        inputs[(h, n)] = (idx, v, r)
        w = capi.Workload(ctx, h, idx, v, r)
        root, st = w.build(PAD_SEED)                       # warm-up
        torch.cuda.synchronize()
        reps = max(3, min(args.steps if args.steps > 3 else 10, 50))
        t0 = time.perf_counter()
        dev_ms = 0.0
        for _ in range(reps):
            root, st = w.build(PAD_SEED)
            dev_ms += st.tree_ms
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        rows.append({"height": h, "entities": n, "wall_ms": wall * 1e3, "device_ms": dev_ms / reps, "entities_per_s": n / wall,
                     "root_value": root[2], "_root": root})
        w.close()
        log("build h=%d n=%d: %.3f ms" % (h, n, wall * 1e3))
    for row in rows:
        root = row.pop("_root")
        h, n = row["height"], row["entities"]
        if ref is not None and n <= 4096:
            idx, v, r = inputs[(h, n)]
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            if hasattr(ref, "ref_set_threads"):
                ref.ref_set_threads(1)                       # the reference is single-threaded
            t0 = time.perf_counter()
            t = ctypes.c_void_p(ref.ref_tree_build(h, ctypes.c_size_t(n), p(idx), p(v), p(r), PAD_SEED, 1))
            row["cpu_faithful_1thread_ms"] = (time.perf_counter() - t0) * 1e3
            oC, oH, orr, ov = [ctypes.create_string_buffer(32) for _ in range(3)] + [ctypes.c_uint64()]
            ref.ref_tree_root(t, oC, oH, ctypes.byref(ov), orr)
            ref.ref_tree_free(t)
            row["root_bit_exact_vs_oracle"] = bool((oC.raw, oH.raw, ov.value, orr.raw) == root)
    # dapol_tree_update on the headline tree: k existing liabilities replaced, the k root-to-leaf paths re-merged in place on the
    # device (smtree's update re-merges one path, src/dapol/mod.rs:210-213); beside it the rebuild it replaces for such updates.
    update_rows = []
    h, n = args.height, 1 << args.log2_entities
    idx, v, r = inputs[(h, n)]
    tree = capi.Tree(ctx, h, idx, v, r, PAD_SEED)
    rng = np.random.default_rng(11)
    for k in (1, 64, 4096):
        if k > n // 8:
            continue
        sel = np.sort(rng.choice(n, size=k, replace=False))
        nv = rng.integers(0, 1 << 32, size=k, dtype=np.uint64)
        nr = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        nr[:, 31] &= 0x0F
        tree.update(idx[sel], nv, nr)                                  # warm-up (scratch allocation)
        ts = []
        for _ in range(11):
            t0 = time.perf_counter()
            tree.update(idx[sel], nv, nr)
            ts.append(time.perf_counter() - t0)
        row = {"leaves_replaced": k, "incremental_ms": 1e3 * sorted(ts)[len(ts) // 2]}
        if k == 1:
            ctx.set_options(capi.Options(update_incremental_max=-1))       # the rebuild every update took before round 3
            try:
                t0 = time.perf_counter()
                tree.update(idx[sel], nv, nr)
                row["rebuild_ms"] = 1e3 * (time.perf_counter() - t0)
            finally:
                ctx.set_options(capi.Options())
        update_rows.append(row)
        log("update k=%d: %.3f ms" % (k, row["incremental_ms"]))
    # ... and NEW liabilities (the reference's update loop grows its tree this way, src/tests.rs:41-48): the levels that gain nodes
    # are rewritten in order on the device (12 levels of 2^20 nodes in the strided layout), nothing is recomputed for nodes that move
    stride = int(idx[1] - idx[0]) if n > 1 else 2
    insert_rows = []
    free = 1
    for k in (1, 64):
        if stride < 4:
            break
        ts = []
        for rep in range(6):
            pick = np.sort(rng.choice(n, size=k, replace=False))
            new_idx = idx[pick] + np.uint64(free)                       # free slots of the strided layout; other slots every repetition
            free += 1
            if free >= stride:
                break
            nv = rng.integers(0, 1 << 32, size=k, dtype=np.uint64)
            nr = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
            nr[:, 31] &= 0x0F
            t0 = time.perf_counter()
            tree.update(new_idx, nv, nr)
            if rep:                                                     # (the first repetition allocates the levels' second buffers)
                ts.append(time.perf_counter() - t0)
            assert tree.last_update_path() == 2
        if ts:
            insert_rows.append({"leaves_inserted": k, "incremental_ms": 1e3 * sorted(ts)[len(ts) // 2]})
            log("insert k=%d: %.3f ms" % (k, insert_rows[-1]["incremental_ms"]))
    tree.close()
    big = rows[-1]
    ab_prove, ab_tree = algorithmic_bytes(args.height, args.n_bits, args.log2_entities)
    ach = big["entities"] * ab_tree / 1e9 / (big["device_ms"] / 1e3)
    print(json.dumps({"metric": "tree build (Dapol::new_blank + build), entities/s", "value": big["entities_per_s"], "unit": "entities/s",
                      "n_gpus": 1, "higher_is_better": True, "data": "synthetic", "dtype": "int32 limbs (255-bit modular integers)",
                      "config": {"workload": "benches/dapol.rs:24-57 build group + 2^%d x height %d" % (args.log2_entities, args.height)},
                      "cases": rows, "update": {"tree": "2^%d leaves x height %d, host-inclusive (H2D of the k records)" % (args.log2_entities, args.height),
                                               "cases": update_rows, "inserts": insert_rows},
                      "roofline": {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                                   "kernel": "k_tree_pad_level + k_tree_merge<1> (+ scan / flags)", "algorithmic_bytes_per_entity": ab_tree}}), flush=True)


def mode_verify(args):
    """BASELINE configs[4]: verification-only throughput of aggregated Bulletproofs with m = 1,024 parties of 64 bits (the
    reference's `verify` group is benches/dapol.rs:93-141), on N GPUs.  Verification does not shard a tree: the ranks are
    REPLICAS of the verifier, each owning a slice of the proofs -- --verify-proofs in total, divided over the ranks (strong
    scaling; --weak: that many per rank) -- and the only exchange is the AND of the verdicts: dapol_comm_allreduce_u64(MIN) over
    RCCL inside the library (torch.distributed MIN under the gloo test hook), once per step, inside the timed region.
    The proofs are made by this build's prover first (untimed); then dapol_range_verify_batch over host buffers is timed
    (host-inclusive: H2D of proofs + commitments).  value = commitments verified by all ranks / max-over-ranks time."""
    rank, local_rank, world, torch, dist, backend = init_dist(args)
    from __graft_entry__ import build, ORACLE_LIB
    if rank == 0:
        build()
    if dist is not None:
        dist.barrier()
    from dapol_amd import capi
    from dapol_amd.sharded import ShardTransport
    B_total, m, n = args.verify_proofs, args.verify_parties, 64
    if args.weak:
        B, B_total = B_total, B_total * world
    else:
        if B_total % world:
            raise SystemExit("--verify-proofs must be a multiple of the number of GPUs")
        B = B_total // world
    scaling = "weak" if args.weak else "strong"
    ctx = capi.Context(local_rank, m)
    comm_device = "cuda" if backend == "nccl" else "cpu"
    # the verdicts' AND: dapol_comm_allreduce_u64 MIN inside the library, with the agreed fallback to torch.distributed (ShardTransport)
    transport = ShardTransport(ctx, rank, world, dist, torch, comm_device)
    transport.create_comm()

    preflight = None
    if world > 1 and not args.no_inline_preflight:
        preflight, pf_ok = inline_preflight_reduce(transport, rank, world, dist, torch, comm_device, allow_fallback=args.allow_exchange_fallback)
        if rank == 0:
            log("inline preflight %s in %.2f s (%s)" % ("ok" if pf_ok else "FAILED", preflight["seconds"], preflight["reduce_path"]))
        if not pf_ok:
            if rank == 0:
                print(json.dumps({"metric": "verification-only throughput, aggregated Bulletproofs (m=%d), commitments/s" % m, "value": None,
                                  "unit": "commitments/s", "n_gpus": world, "preflight_failed": True, "multi_gpu": {"preflight": preflight},
                                  "wall_s_since_process_start": time.time() - T_PROC0}), flush=True)
            if transport.comm is not None:
                transport.comm.abort()
                transport.comm = None
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)

    def reduce_path():
        if world == 1:
            return "none (single GPU)"
        if transport.comm is not None:
            return "dapol_comm_allreduce_u64 MIN (ncclAllReduce inside libdapol_hip.so)"
        return transport.path.replace("torch.distributed (", "torch.distributed all_reduce MIN (")

    def verdict_and(all_ok):
        return transport.reduce_u64(all_ok, "min")

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # TWO independently proven batches, verified in turn, every pass under a fresh seed: with one batch and one
    # seed every pass finds the previous pass's per-proof tables still in the retained scratch, and a kernel that read them before
    # they were rewritten (round 5's fork of the own points before k_rv_tables) went unnoticed.  A pass that falls back to bisection
    # on these all-valid batches fails the run (dapol_diag_verify_fallbacks).
    batches = []
    for k in range(2):
        rng = np.random.default_rng(5 + rank + 1000 * k)
        v = rng.integers(0, 2**32, size=(B, m), dtype=np.uint64)
        r = rng.integers(0, 256, size=(B, m, 32), dtype=np.uint8)
        r[:, :, 31] &= 0x0F
        log("rank %d: proving batch %d of %d x m=%d (untimed setup)" % (rank, k, B, m))
        pk = ctx.range_prove_batch(n, m, v, r, nonce_seed=NONCE_SEED, stream_id=np.arange((2 * rank + k) * B, (2 * rank + k + 1) * B, dtype=np.uint64))
        C, _ = ctx.commit_hash_batch(v.reshape(-1), r.reshape(-1, 32))
        batches.append((pk, np.ascontiguousarray(C.reshape(B, m, 32))))
    proofs, Vs = batches[0]
    fb = ctypes.c_uint64()
    capi.lib().dapol_diag_verify_fallbacks(ctypes.byref(fb))
    fb0 = int(fb.value)
    for w_ in range(max(2, args.warmup)):
        pk, Vk = batches[w_ & 1]
        ok = ctx.range_verify_batch(n, m, pk, Vk, verify_seed=os.urandom(32))
        verdict_and(int(ok.all()))
    steps = max(args.steps, 5)
    sync()
    t0 = time.perf_counter()
    all_and = 1
    for s_ in range(steps):
        pk, Vk = batches[s_ & 1]
        ok = ctx.range_verify_batch(n, m, pk, Vk, verify_seed=os.urandom(32))
        all_and &= verdict_and(int(ok.all()))
    sync()
    elapsed = time.perf_counter() - t0
    capi.lib().dapol_diag_verify_fallbacks(ctypes.byref(fb))
    fallbacks = int(fb.value) - fb0
    if fallbacks and all_and == 1:
        raise SystemExit("rank %d: %d combined check(s) fell back to bisection on all-valid batches -- stale scratch or a race between passes" % (rank, fallbacks))
    seed = os.urandom(32)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    dt = elapsed / steps
    # the same passes from PAGE-LOCKED host buffers (what an embedder gets by pinning its proof / commitment arrays: the column blocks
    # then arrive by DMA instead of through the runtime's staging copies) -- reported beside the headline, never as it
    dt_pinned = None
    if world == 1 and not args.no_pinned_leg:
        pinned = [(torch.from_numpy(pk).pin_memory(), torch.from_numpy(Vk).pin_memory()) for pk, Vk in batches]
        okp = ctx.range_verify_batch(n, m, pinned[0][0].numpy(), pinned[0][1].numpy(), verify_seed=seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in range(steps):
            pp, pv = pinned[s_ & 1]
            okp = ctx.range_verify_batch(n, m, pp.numpy(), pv.numpy(), verify_seed=os.urandom(32))
            if not okp.all():
                raise SystemExit("pinned-buffer pass disagrees with the pageable one")
        torch.cuda.synchronize()
        dt_pinned = (time.perf_counter() - t0) / steps
        del pinned
    # one bad proof on ONE rank must turn the job's verdict
    and_bad, lb = 0, 1
    if not args.no_bad_proof_leg:
        bad = proofs.copy()
        if rank == world - 1:
            bad[B // 3, 100] ^= 1
        ok_bad = ctx.range_verify_batch(n, m, bad, Vs, verify_seed=seed)
        and_bad = verdict_and(int(ok_bad.all()))
        local_bad_found = bool(ok_bad[B // 3] == 0 and ok_bad.sum() == B - 1) if rank == world - 1 else bool(ok_bad.all())
        lb = verdict_and(int(local_bad_found))
    verdict_reduce, comm_ranks, comm_err = reduce_path(), transport.comm_ranks, transport.comm_error
    transport.close()                  # every rank, together: they have just left the same reduce
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    cpu = None
    if not args.no_cpu_baseline and os.path.exists(ORACLE_LIB) and world == 1:
        ref = ctypes.CDLL(ORACLE_LIB)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        c32 = bytes(range(1, 33))
        t0 = time.perf_counter()
        okc = ref.ref_range_verify(n, m, proofs[0].tobytes(), ctypes.c_size_t(proofs.shape[1]), p(np.ascontiguousarray(Vs[0])), c32, 0)
        t1 = time.perf_counter() - t0
        cpu = {"value": m / t1, "unit": "commitments/s", "cores": 1, "kind": "port", "sample": "one m=%d proof, tuned, single thread: %.2f s; verdict %d" % (m, t1, okc)}
    ab = proofs.shape[1] + m * 32
    # HBM traffic of one verification pass: from the PMC passes of tools/profile_verify.sh, quoted only when they were taken on the
    # kernel sources that are running and on this very shape (1,024 proofs of 1,024 parties on one GPU)
    traffic, vpmc = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "verify_pmc.json")) as f:
            vp = json.load(f)
        cur = vp.get("kernel_src_sha") == kernel_src_sha()
        vpmc = {"file": "profiles/verify_pmc.json", "current_build": bool(cur), "dominant_kernel": vp.get("dominant_kernel"),
                "dominant_share_of_kernel_time": vp.get("dominant_share_of_kernel_time"), "per_pass": vp.get("per_pass")}
        if cur and world == 1 and B == 1024 and m == 1024:
            traffic = (vp["per_pass"]["hbm_read_MB"] + vp["per_pass"]["hbm_write_MB"]) * 1e6
    except (OSError, ValueError, KeyError):
        pass
    print(json.dumps({"metric": "verification-only throughput, aggregated Bulletproofs (m=%d), commitments/s" % m, "value": B_total * m / dt,
                      "unit": "commitments/s", "n_gpus": world, "steps": steps, "warmup": max(2, args.warmup), "ms_per_step": dt * 1e3,
                      "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "data": "synthetic",
                      "dtype": "int32 limbs (255-bit modular integers)",
                      "config": {"workload": "%d proofs x m=%d x n=64 (proof %d bytes) in total, %d per GPU, host-inclusive, replicas of the verifier"
                                             % (B_total, m, proofs.shape[1], B),
                                 "verdict_reduce": verdict_reduce, "rccl_ranks_in_library_communicator": comm_ranks, "reduce_fallback_reason": comm_err},
                      "ms_per_step_pinned_host_buffers": (dt_pinned * 1e3 if dt_pinned else None),
                      "multi_gpu": ({"preflight": preflight} if preflight is not None else None),
                      "all_verified": bool(all_and == 1), "batches_alternated": 2, "fresh_verify_seed_per_pass": True,
                      "combined_check_fallbacks_in_timed_region": fallbacks, "one_bad_proof_turns_the_job_verdict": (None if args.no_bad_proof_leg else bool(and_bad == 0 and lb == 1)),
                      "roofline": {"bound": "hbm", "achieved": B * ab / 1e9 / dt, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": B * ab / 1e9 / dt / PEAK_HBM_GBS, "traffic": traffic, "algorithmic_bytes_per_proof": ab,
                                   "algorithmic_bytes_per_pass": B * ab, "traffic_over_algorithmic": (traffic / (B * ab)) if traffic else None,
                                   "from_profiles": vpmc,
                                   "note": "per GPU and per PASS (one verification of this rank's proofs; the step is a chain of ~40 launches, none of which "
                                           "dominates): proof + commitment bytes / step time.  traffic = FETCH_SIZE x 2 + WRITE_SIZE summed over the verifier's "
                                           "kernels of one pass (tools/profile_verify.sh), set only from PMC passes of the running build.  The floor of a pass is "
                                           "the serial transcript replay: 253 + 56 dependent Keccak-f per proof at ~5 us on a wavefront, ~1.6 ms whatever the batch size"},
                      "cpu_baseline": cpu}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _plan(policy, n, agg):
    """(start, count, parties) of the aggregated sub-proofs + the individual ones (padding.rs:88-118, splitting.rs:100-129)."""
    out = []
    if policy == 0:
        out.append((0, agg, _np2(agg)))
    else:
        base, pos = _np2(agg), 0
        while pos < agg:
            if agg & base:
                out.append((pos, base, base))
                pos += base
            base >>= 1
    return out + [(i, 1, 1) for i in range(agg, n)]


def mode_criterion(args):
    """benches/dapol.rs as the reference defines it: group `build` (:24-57), group `prove` (:59-91: Dapol::generate_proof of one
    random leaf per iteration, N = 1,024, heights 16 / 24 / 32, RangeProofSplitting and RangeProofPadding, aggregation_factor =
    height) and group `verify` (:93-141: DapolProof::verify of such a proof).  These are LATENCY measurements of single calls
    (host-inclusive: the proof comes back to the host), median of 10 like criterion's sample_size(10).  CPU beside each: the C
    restatement, single-threaded like the reference, faithful (per-call generators)."""
    import torch
    from __graft_entry__ import build, ORACLE_LIB
    build()
    from dapol_amd import capi
    ctx = capi.Context(0, 32)
    ref = None
    if not args.no_cpu_baseline and os.path.exists(ORACLE_LIB):
        native, _ = build_native_oracle()
        ref = ctypes.CDLL(native or ORACLE_LIB)
        ref.ref_range_proof_size.restype = ctypes.c_size_t
        if hasattr(ref, "ref_set_threads"):
            ref.ref_set_threads(1)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(0xD4)
    n, n_bits, rows = 1024, 64, []
    ctx32, ctx64 = ctx, None
    for height in (16, 24, 32, 64):
        if height > 32:
            # MAX_TREE_HEIGHT (src/dapol/mod.rs:26): aggregation_factor = height = 64 parties, 4,096 generators a side.  A context of
            # its own: 2 x 4,096 table rows -- at the default 40 GB budget these are 16-bit windows (16 of them instead of 15).
            ctx32.close()
            ctx = ctx64 = capi.Context(0, 64)
        idx, v, r = synth_inputs(n, height, 0, n)
        tree = capi.Tree(ctx, height, idx, v, r, PAD_SEED)
        rC, rH, _, _ = tree.root()
        lC, lH = ctx.commit_hash_batch(v, r)
        for pol, name in ((capi.POLICY_SPLITTING, "splitting"), (capi.POLICY_PADDING, "padding")):
            picks = rng.integers(0, n, size=11)
            tp, tv = [], []
            for it, k in enumerate(picks):
                leaf = idx[k:k + 1]
                t0 = time.perf_counter()
                pC, pH, proofs = tree.prove_entities(leaf, pol, height, n_bits, NONCE_SEED)
                t1 = time.perf_counter()
                ok = ctx.verify_entities(height, leaf, lC[k:k + 1], lH[k:k + 1], pC, pH, rC, rH, pol, height, n_bits, proofs)
                t2 = time.perf_counter()
                assert ok.all()
                if it:                                              # the first iteration warms the call path up
                    tp.append(t1 - t0)
                    tv.append(t2 - t1)
            row = {"height": height, "policy": name, "prove_ms": 1e3 * sorted(tp)[len(tp) // 2], "verify_ms": 1e3 * sorted(tv)[len(tv) // 2],
                   "proof_bytes": int(proofs.shape[1]), "parties": _np2(height), "window_bits": int(ctx.get_options().window_bits)}
            if ref is not None:
                _, _, sv, sr = tree.paths(idx[picks[0]:picks[0] + 1])
                t0 = time.perf_counter()
                slot, made = 0, []
                for start, cnt, m in _plan(pol, height, height):
                    pv, pr = np.zeros(m, np.uint64), np.zeros((m, 32), np.uint8)
                    pv[:cnt], pr[:cnt] = sv[0, start:start + cnt], sr[0, start:start + cnt]
                    pr[cnt:, 0] = 1
                    out = ctypes.create_string_buffer(ref.ref_range_proof_size(n_bits, m))
                    ref.ref_range_prove(n_bits, m, p(pv), p(pr), NONCE_SEED, ctypes.c_uint64(int(idx[picks[0]])), ctypes.c_uint64(slot), None, 1, out)
                    slot += m * (2 * n_bits + 4)
                    made.append((m, pv, pr, out))
                row["cpu_prove_1thread_faithful_ms"] = 1e3 * (time.perf_counter() - t0)
                # the range proofs of the same inclusion proof checked by the restatement's verifier (the Merkle re-merge, ~1 % of it,
                # is not included); commitments computed outside the timed part, as a verifier receives them
                coms = []
                for m, pv, pr, out in made:
                    C, H = np.zeros((m, 32), np.uint8), np.zeros((m, 32), np.uint8)
                    ref.ref_commit_hash(ctypes.c_size_t(m), p(pv), p(pr), p(C), p(H))
                    coms.append(C)
                t0 = time.perf_counter()
                good = all(ref.ref_range_verify(n_bits, m, out, ctypes.c_size_t(len(out.raw)), p(C), NONCE_SEED, 1) == 1
                           for (m, pv, pr, out), C in zip(made, coms))
                row["cpu_verify_1thread_faithful_ms"] = 1e3 * (time.perf_counter() - t0)
                row["cpu_verdict"] = bool(good)
            rows.append(row)
            log("prove/%s/%d: %.2f ms, verify: %.2f ms" % (name, height, row["prove_ms"], row["verify_ms"]))
        tree.close()
    print(json.dumps({"metric": "criterion groups of the reference (benches/dapol.rs:59-141): latency of ONE generate_proof / verify, ms",
                      "unit": "ms", "higher_is_better": False, "n_gpus": 1, "data": "synthetic", "dtype": "int32 limbs (255-bit modular integers)",
                      "config": {"workload": "N = 1,024 strided leaves, heights 16 / 24 / 32 (the reference's) and 64 (its MAX_TREE_HEIGHT), 64-bit proofs, aggregation_factor = height, BLAKE3"},
                      "value": [r_["prove_ms"] for r_ in rows if r_["height"] == 32 and r_["policy"] == "padding"][0],
                      "prove_verify": rows, "build": "python bench.py --mode build"}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("prove", "build", "verify", "criterion"), default="prove")
    ap.add_argument("--budget-s", type=float, default=570.0,
                    help="wall budget of the whole process, counted from its start; the timed steps are clamped to fit (>= 3)")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0, help="wall budget of the CPU-baseline leg")
    ap.add_argument("--no-pinned-leg", action="store_true", help="--mode verify: skip the extra passes from page-locked host buffers (profiling: keeps the kernel counts per pass)")
    ap.add_argument("--no-bad-proof-leg", action="store_true", help="--mode verify: skip the untimed pass with one bad proof (profiling: its per-proof re-check is not part of a pass)")
    ap.add_argument("--log2-entities", type=int, default=20,
                    help="entities IN TOTAL = 2^this (default 20: BASELINE configs[2], the metric's workload at every N); per GPU with --weak")
    ap.add_argument("--log2-entities-total", type=int, default=None, help="the same, spelled out (configs[3]: 22 with --gpus 8)")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling, 2^--log2-entities per GPU (default: strong, the total is divided)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (splitting policy, API layout, host buffers)")
    ap.add_argument("--no-inline-preflight", action="store_true", help="N > 1: skip the collective checks that run before the first step")
    ap.add_argument("--allow-exchange-fallback", action="store_true",
                    help="N > 1 over RCCL: time the run even if the library's communicator did not come up (torch.distributed carries the exchange)")
    ap.add_argument("--height", type=int, default=32)
    ap.add_argument("--n-bits", type=int, default=64)
    ap.add_argument("--verify-proofs", type=int, default=1024)
    ap.add_argument("--verify-parties", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile", choices=("bench", "host"), default="bench",
                    help="dapol_options.profile: bench = the library's throughput defaults (17-bit windows + high-half rows, two-round chunks: ~175 GB "
                         "in use); host = sized for an embedder that shares the GPU (16-bit windows, no high-half rows, one-round chunks)")
    ap.add_argument("--preflight", action="store_true",
                    help="< 30 s, no proving: the library's RCCL communicator up with ncclCommCount == N, the step's two collectives timed, every rank's root equal, rocm-smi --showtopo to stderr")
    ap.add_argument("--preflight-iters", type=int, default=100)
    ap.add_argument("--preflight-timeout-s", type=float, default=30.0, help="deadline of the communicator's creation and of each of its collectives")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps must be >= 1 and --warmup >= 0")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.mode in ("prove", "verify"):
        return spawn_ranks(args)                  # (never returns)
    if args.preflight:
        return mode_preflight(args)
    if args.mode == "build":
        return mode_build(args)
    if args.mode == "verify":
        return mode_verify(args)
    if args.mode == "criterion":
        return mode_criterion(args)
    return mode_prove(args)


if __name__ == "__main__":
    main()
