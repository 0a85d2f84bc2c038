#!/usr/bin/env python3
"""Randomised soak of the generator-stationary prover against the C oracle: random (bits, parties) shapes with at least 1,024 generators per
side, random batch sizes between 1,024 and 40,000 proofs (sliced sweeps below 65,536) (one or several chunks, ragged last chunks), random values and blindings; six
random proofs of every batch compared byte for byte with oracle/ref_dapol.c, all six verified on the GPU, one tampered copy rejected.
usage: tools/soak_gs.py [cases] [seed]"""
import ctypes
import os
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")     # the library reads its DAPOL_* knobs only in a process that opts in
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dapol_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
lib, _ = bench.build_native_oracle()
ref = ctypes.CDLL(lib)
ref.ref_range_proof_size.restype = ctypes.c_size_t
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
ctx = capi.Context(0, 64)
SEED = bench.NONCE_SEED
for case in range(cases):
    n_bits, m = [(64, 32), (64, 16), (32, 32), (32, 64), (16, 64), (64, 64)][int(rng.integers(0, 6))]
    b = int(rng.integers(1024, 40001)) if n_bits * m <= 2048 else int(rng.integers(1024, 14001))
    if rng.integers(0, 4) == 0:
        os.environ["DAPOL_CHUNK"] = str(int(rng.integers(3000, 20000)))         # several (ragged) chunks ...
    os.environ["DAPOL_STREAMS"] = str(int(rng.integers(1, 3)))                  # ... one in flight (the default since round 4) or two
    vmax = (1 << n_bits) if n_bits < 64 else (1 << 63)
    v = rng.integers(0, vmax, size=(b, m), dtype=np.uint64)
    r = rng.integers(0, 256, size=(b, m, 32), dtype=np.uint8)
    r[:, :, 31] &= 0x0F
    sid = rng.integers(0, 1 << 40, size=b, dtype=np.uint64)
    try:
        proofs = ctx.range_prove_batch(n_bits, m, v, r, nonce_seed=SEED, stream_id=sid)
    finally:
        os.environ.pop("DAPOL_CHUNK", None)
        os.environ.pop("DAPOL_STREAMS", None)
    ps = ref.ref_range_proof_size(n_bits, m)
    assert proofs.shape == (b, ps)
    pick = rng.choice(b, size=6, replace=False)
    for k in pick:
        out = ctypes.create_string_buffer(ps)
        ref.ref_range_prove(n_bits, m, p(np.ascontiguousarray(v[k])), p(np.ascontiguousarray(r[k])), SEED, ctypes.c_uint64(int(sid[k])), ctypes.c_uint64(0), None, 0, out)
        assert out.raw == proofs[k].tobytes(), (case, n_bits, m, b, int(k))
    C, _ = ctx.commit_hash_batch(v[pick].reshape(-1), r[pick].reshape(-1, 32))
    Vs = C.reshape(6, m, 32)
    sub = np.ascontiguousarray(proofs[pick])
    assert ctx.range_verify_batch(n_bits, m, sub, Vs).all()
    bad = sub.copy()
    bad[2, int(rng.integers(0, ps))] ^= 1 << int(rng.integers(0, 8))
    ok = ctx.range_verify_batch(n_bits, m, bad, Vs)
    assert ok[2] == 0 and ok.sum() == 5
    print("case %d: n=%d m=%d b=%d ok" % (case, n_bits, m, b), flush=True)
print("soak: %d random generator-stationary batches, sampled proofs byte-identical to the oracle, verified, tampered copies rejected" % cases)
