"""The C-ABI library: builds for gfx950 without a GPU, loads, exports every symbol include/dapol_hip.h declares, and the
pure host logic (sizes, policy planning, argument validation) behaves like the reference.  No compute calls."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dapol_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dapol_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(hip_lib):
    lib = hip_lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(hip_lib.EXPORTED_SYMBOLS) == syms, "capi.py binds a different set than the header declares"


def test_sizes(hip_lib, pyref):
    lib = hip_lib.lib()
    assert [lib.dapol_range_proof_size(64, m) for m in (1, 32, 1024)] == [672, 992, 1312]    # src/range/mod.rs:18
    assert lib.dapol_range_proof_size(8, 2) == 544
    for bad in ((7, 1), (64, 3), (64, 0), (128, 1)):
        assert lib.dapol_range_proof_size(*bad) == 0
    for height in (1, 4, 5, 16, 24, 32):
        for agg in range(0, height + 1):
            for pol, name in ((0, "padding"), (1, "splitting")):
                plan, pos = pyref.policy_plan(name, height, agg)
                exp = sum(pyref.range_proof_size(64, mm) for _, _, mm in plan) + 672 * (height - pos)
                if name == "padding" and agg == 0:
                    exp = pyref.range_proof_size(64, 1) + 672 * height
                assert lib.dapol_entity_proof_size(height, pol, agg, 64) == exp, (height, agg, name)
        assert lib.dapol_entity_proof_size(height, 0, height + 1, 64) == 0      # reference: index out of bounds panic
        assert lib.dapol_entity_proof_size(height, 2, 1, 64) == 0


def test_error_strings_and_no_device_behaviour(hip_lib):
    lib = hip_lib.lib()
    assert b"height" in lib.dapol_strerror(1) and b"duplicated" in lib.dapol_strerror(4).lower()
    h = ctypes.c_void_p()
    rc = lib.dapol_ctx_create(0, 3, 0, ctypes.byref(h))
    assert rc == 8                                   # max_parties must be a power of two
    rc = lib.dapol_ctx_create(0, 32, 3, ctypes.byref(h))
    assert rc == 3                                   # BLAKE3 / Blake2s / Blake2b (0, 1, 2) and nothing else (DapolError::InvalidDigestSize)
    import torch
    if not torch.cuda.is_available():
        rc = lib.dapol_ctx_create(0, 32, 0, ctypes.byref(h))
        assert rc == 16 and not h.value              # fails loudly: no CPU fallback
        assert b"no usable HIP device" in lib.dapol_last_error()


def test_wire_format_matches_reference_layout(hip_lib, pyref):
    """Serializable for RangeProofPadding / RangeProofSplitting (src/range/padding.rs:38-69, splitting.rs:36-84) against the
    golden serialisations made by the oracle; decoding errors mirror DecodingError."""
    import pytest
    from conftest import load_golden
    for c in load_golden("dapol.json"):
        pol = 0 if c["policy"] == "padding" else 1
        blob = bytes.fromhex("".join(c["aggregated"]) + "".join(c["individual"]))
        wire = hip_lib.range_proofs_serialize(c["height"], pol, c["agg"], c["n_bits"], blob)
        assert wire.hex() == c["serialized"]
        agg, ind, used = hip_lib.range_proofs_deserialize(pol, c["n_bits"], wire + b"trailing")
        assert [a.hex() for a in agg] == c["aggregated"] and [i.hex() for i in ind] == c["individual"] and used == len(wire)
        with pytest.raises(hip_lib.DapolError) as e:
            hip_lib.range_proofs_deserialize(pol, c["n_bits"], wire[:-3])
        assert e.value.code == 6                                      # DecodingError::BytesNotEnough
        bad = bytearray(wire)
        off = (2 if pol else 0) + 8 + 32 * 4                          # t_x of the first aggregated proof := 2^256 - 1 (not canonical)
        bad[off:off + 32] = b"\\xff" * 32
        with pytest.raises(hip_lib.DapolError) as e:
            hip_lib.range_proofs_deserialize(pol, c["n_bits"], bytes(bad))
        assert e.value.code == 7                                      # DecodingError::ValueDecodingError
    lib = hip_lib.lib()
    assert lib.dapol_range_proofs_wire_size(32, 0, 32, 64) == 8 + 992 + 8
    assert lib.dapol_range_proofs_wire_size(32, 1, 24, 64) == 2 + (8 + 928) + (8 + 864) + 8 + 8 * 672


def test_batch_sibling_plan_is_host_only_and_matches_oracle(hip_lib, pyref):
    """Positions of the siblings of a batched Merkle proof (generate_proof_batch, src/dapol/mod.rs:172-190): index
    arithmetic only, so it runs without a GPU; checked against the Python restatement, heights up to 64."""
    import numpy as np
    import pytest
    rng = np.random.default_rng(11)
    for height, k in [(1, 1), (1, 2), (3, 8), (6, 4), (10, 10), (16, 40), (33, 7), (64, 9), (64, 1)]:
        hi = (1 << height) if height < 64 else (1 << 64)
        leaves = sorted({int(x) % hi for x in rng.integers(0, 2**63, size=k, dtype=np.uint64) * 2 + rng.integers(0, 2, size=k, dtype=np.uint64)})
        if height <= 3:
            leaves = list(range(min(k, 1 << height)))
        level, index = hip_lib.batch_siblings(height, leaves)
        assert list(zip(map(int, level), map(int, index))) == pyref.batch_siblings(height, leaves)
        assert len(level) <= len(leaves) * height
    assert len(hip_lib.batch_siblings(5, [7])[0]) == 5                       # one leaf: its whole path
    for bad in ([2, 1], [3, 3], [32], []):
        with pytest.raises(hip_lib.DapolError) as e:
            hip_lib.batch_siblings(5, bad)
        assert e.value.code == 8


def test_dapol_proof_wire_layout_host_only(hip_lib, pyref):
    """DapolProof::serialize (src/proof/mod.rs:68-73) = R::serialize() || MerkleProof::serialize(): the C-ABI serialiser (host
    only) against the Python restatement, for single-leaf and batched shapes, under the default dapol_wire_config and with
    every switch moved (byte order, field widths, path width)."""
    import numpy as np
    rng = np.random.default_rng(3)
    lib = hip_lib.lib()
    for c in [x for x in load_golden_cases() if x["policy"] in ("padding", "splitting")][:6]:
        pol = 0 if c["policy"] == "padding" else 1
        height, n_bits, agg = c["height"], c["n_bits"], c["agg"]
        blob = bytes.fromhex("".join(c["aggregated"]) + "".join(c["individual"]))
        sC = rng.integers(0, 256, size=(height, 32), dtype=np.uint8)           # serialisation does not look inside the nodes
        sH = rng.integers(0, 256, size=(height, 32), dtype=np.uint8)
        sibs = [(sC[i].tobytes(), sH[i].tobytes()) for i in range(height)]
        agg_p = [bytes.fromhex(a) for a in c["aggregated"]]
        ind_p = [bytes.fromhex(a) for a in c["individual"]]
        for kw, cfg in [({}, {}),
                        (dict(big_endian=False), dict(int_big_endian=0)),
                        (dict(batch_num_bytes=2, sibling_num_bytes=4, tree_height_bytes=1), dict(batch_num_bytes=2, sibling_num_bytes=4, tree_height_bytes=1)),
                        (dict(path_bytes_full=True), dict(path_bytes_full=1))]:
            old = hip_lib.wire_config_set(**cfg)
            try:
                wire = hip_lib.proof_serialize(height, [c["leaf"]], sC, sH, pol, agg, n_bits, blob)
                assert len(wire) == lib.dapol_proof_wire_size(height, 1, height, pol, agg, n_bits)
            finally:
                hip_lib.wire_config_restore(old)
            # (the range part keeps its own field widths, src/range/mod.rs:19-21; only the byte order is shared)
            expect_range = pyref.policy_serialize(c["policy"], agg_p, ind_p)
            if kw.get("big_endian", True):                                    # (pyref.policy_serialize is the big-endian restatement)
                assert wire[:len(expect_range)] == expect_range
            merkle = pyref.merkle_proof_serialize(height, [c["leaf"]], sibs, **kw)
            assert wire[len(wire) - len(merkle):] == merkle, (kw, wire[-len(merkle):][:24].hex(), merkle[:24].hex())
            assert len(wire) == len(expect_range) + len(merkle)
    # shape checks without any golden case: k = 3 leaves, 5 siblings, height 10 -> 2-byte paths
    old = hip_lib.wire_config_get()
    assert (old.int_big_endian, old.batch_num_bytes, old.sibling_num_bytes, old.tree_height_bytes, old.path_bytes_full, old.siblings_leaf_first) == (1, 8, 8, 2, 0, 0)
    m = pyref.merkle_proof_serialize(10, [1, 513, 1023], [(bytes([i]) * 32, bytes([i + 1]) * 32) for i in range(5)])
    assert len(m) == 8 + 8 + 2 + 3 * 2 + 5 * 64
    assert m[:18] == (3).to_bytes(8, "big") + (5).to_bytes(8, "big") + (10).to_bytes(2, "big")
    assert m[18:24] == bytes([0x00, 0x40, 0x80, 0x40, 0xff, 0xc0])                  # 0000000001 | 1000000001 | 1111111111, left-aligned


def load_golden_cases():
    from conftest import load_golden
    return load_golden("dapol.json")


def test_bench_wall_budget_plan():
    """bench.py's step plan under the wall budget (the driver's --steps 20 --warmup 5 never fit 600 s)."""
    import bench
    assert bench.plan_steps(20, 5, 22.0, 450 - 30 - 45) == (0, 16)            # driver's call: clamp, no further warm-up
    assert bench.plan_steps(20, 5, 22.0, 30.0) == (0, 3)                      # never fewer than three timed steps
    assert bench.plan_steps(3, 1, 22.0, 400.0) == (0, 3)                      # the default call is untouched
    assert bench.plan_steps(2, 3, 1.0, 400.0) == (2, 2)                       # cheap steps: everything asked for
    assert bench.plan_steps(1, 1, 500.0, 10.0) == (0, 1)
    assert bench.plan_steps(5, 0, None, 100.0) == (0, 5)                      # no warm-up: nothing measured to plan with
    prove, tree = bench.algorithmic_bytes(32, 64, 20)
    assert (prove, tree, prove + tree) == (6384, 2752, 9136)
    assert bench.algorithmic_bytes(24, 64, 16) == (5040, 1920)


def test_sc_invert_vartime_matches_fermat_ladder(tmp_path):
    """The divstep inversion used for public challenges (sc.h) equals the Fermat ladder on edge values and 30,000 random scalars
    (host build of the very header the kernels compile)."""
    exe = str(tmp_path / "sc_inv")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "dapol_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "sc_invert_vartime_test.cpp"),
                    "-o", exe], check=True)
    r = subprocess.run([exe, "30000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "30000 tested, 0 bad" in r.stdout
