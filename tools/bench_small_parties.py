"""Proofs of FEW parties in large batches: what the policies leave per entity when aggregation_factor < height
(src/range/padding.rs:104-112, splitting.rs:118-123) -- one individual 64-bit proof per sibling beyond the factor.

  python tools/bench_small_parties.py [--entities 14] [--proofs 17] [--only policy|batch]

policy legs: a device-resident workload (height 32, 2^entities entities), padding / splitting at several aggregation factors; the
             time is the library's own HIP-event bracket around the proving pipeline (dapol_workload_stats.prove_ms).
batch legs:  dapol_range_prove_batch of 2^proofs proofs of (64 bits, m parties), m = 1, 2, 4, 8; the time is the device bracket
             dapol_diag_range_prove_ms reports (inputs already copied, outputs not yet), the wall time beside it.
Every line carries the proofs' checksum / a digest, so A/B runs (DAPOL_NO_GROUP=1, DAPOL_GS_SMALL_MIN=...) can be compared for bytes.
"""
import argparse, ctypes, hashlib, json, os, sys, time
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dapol_amd import capi

# register-only rate of table additions on one MI355X (256 CUs x 4 SIMDs x 64 lanes / 3,902 cycles x 2.4 GHz)
ADD_RATE = 256 * 4 * 64 / 3902 * 2.4e9


def adds_per_proof(n, m, nwin=15):
    """Table additions a proof of m parties needs in the never-fold form (7 + ... MSMs of 2 n m terms, nwin windows each)."""
    N = n * m
    lg = N.bit_length() - 1
    return (1 + lg) * 2 * N * nwin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=14)
    ap.add_argument("--proofs", type=int, default=17)
    ap.add_argument("--only", choices=("policy", "batch", "verify"), default=None)
    ap.add_argument("--verify-entities", type=int, default=13)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--aggs", default="32,24,8,0")
    ap.add_argument("--ms", default="1,2,4,8")
    args = ap.parse_args()
    seed = bytes(range(32))
    height, n_bits = 32, 64
    ctx = capi.Context(0, 32)
    rng = np.random.default_rng(5)
    out = {}
    if args.only == "verify":
        # DapolProof::verify (src/proof/mod.rs:41-47) of entities whose range proofs are mostly individual ones: host-inclusive wall time
        n = 1 << args.verify_entities
        idx = np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n)
        v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
        r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        r[:, 31] &= 0x0F
        tree = capi.Tree(ctx, height, idx, v, r, seed)
        rC, rH, _, _ = tree.root()
        lC, lH = ctx.commit_hash_batch(v, r)
        fb = ctypes.c_uint64()
        for agg in [int(a) for a in args.aggs.split(",")]:
            pC, pH, proofs = tree.prove_entities(idx, capi.POLICY_PADDING, agg, n_bits, seed)
            vargs = (height, idx, lC, lH, pC, pH, rC, rH, capi.POLICY_PADDING, agg, n_bits)
            ctx.verify_entities(*[a[:64] if isinstance(a, np.ndarray) else a for a in vargs], proofs[:64], verify_seed=seed)
            best = None
            for _ in range(args.reps):
                t0 = time.perf_counter()
                ok = ctx.verify_entities(*vargs, proofs, verify_seed=seed)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            capi.lib().dapol_diag_verify_fallbacks(ctypes.byref(fb))
            row = {"entities": n, "verify_s": best, "entities_per_s": n / best, "all_verified": bool(ok.all()), "fallbacks_so_far": int(fb.value),
                   "mb_in": (pC.nbytes + pH.nbytes + proofs.nbytes) / 1e6}
            out["verify_padding_agg%d" % agg] = row
            print("verify", agg, json.dumps(row), flush=True)
        print(json.dumps(out), flush=True)
        return
    if args.only != "batch":
        n = 1 << args.entities
        idx = np.arange(n, dtype=np.uint64) * np.uint64((1 << height) // n)
        v = rng.integers(0, 2**32, size=n, dtype=np.uint64)
        r = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        r[:, 31] &= 0x0F
        w = capi.Workload(ctx, height, idx, v, r)
        w.build(seed)
        for pol, pname in ((capi.POLICY_PADDING, "padding"), (capi.POLICY_SPLITTING, "splitting")):
            for agg in [int(a) for a in args.aggs.split(",")]:
                if pol == capi.POLICY_SPLITTING and agg in (32, 0):
                    continue
                w.prove(seed, n_bits, first=0, count=min(n, 512), policy=pol, aggregation_factor=agg)      # warm the shapes
                best, st = None, None
                for _ in range(args.reps):
                    st = w.prove(seed, n_bits, first=0, count=n, policy=pol, aggregation_factor=agg)
                    best = st.prove_ms if best is None else min(best, st.prove_ms)
                singles = height - agg + (1 if (agg == 0 or (pol == capi.POLICY_SPLITTING and agg & 1)) else 0)
                row = {"entities": n, "prove_ms": best, "entities_per_s": n / best * 1e3, "proof_bytes_per_entity": int(st.proof_bytes // n),
                       "individual_proofs_per_entity": singles, "checksum": "%016x" % st.checksum}
                if agg == 0:
                    row["single_proofs_per_s"] = n * singles / best * 1e3
                    row["frac_of_addition_rate"] = row["single_proofs_per_s"] * adds_per_proof(64, 1) / ADD_RATE
                out["%s_agg%d" % (pname, agg)] = row
                print(pname, agg, json.dumps(row), flush=True)
        w.close()
    if args.only != "policy":
        b = 1 << args.proofs
        ms_dev = ctypes.c_double()
        for m in [int(x) for x in args.ms.split(",")]:
            tot = b * m
            v = rng.integers(0, 2**63, size=tot, dtype=np.uint64)
            r = rng.integers(0, 256, size=(tot, 32), dtype=np.uint8)
            r[:, 31] &= 0x0F
            sid = np.arange(b, dtype=np.uint64)
            ctx.range_prove_batch(n_bits, m, v[:256 * m], r[:256 * m], nonce_seed=seed, stream_id=sid[:256])
            best, wall, proofs = None, None, None
            for _ in range(args.reps):
                t0 = time.perf_counter()
                proofs = ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=seed, stream_id=sid)
                dt = 1e3 * (time.perf_counter() - t0)
                capi.lib().dapol_diag_range_prove_ms(ctypes.byref(ms_dev))
                best = ms_dev.value if best is None else min(best, ms_dev.value)
                wall = dt if wall is None else min(wall, dt)
            row = {"proofs": b, "m": m, "device_ms": best, "wall_ms": wall, "proofs_per_s": b / best * 1e3,
                   "frac_of_addition_rate": b / best * 1e3 * adds_per_proof(64, m) / ADD_RATE,
                   "sha256_16": hashlib.sha256(proofs.tobytes()).hexdigest()[:16]}
            out["batch_64_%d" % m] = row
            print("batch", m, json.dumps(row), flush=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
