"""The C oracle (oracle/ref_dapol.c, the CPU baseline) against the golden vectors made by the Python oracle.  CPU only."""
import ctypes

from conftest import load_golden


def buf(n):
    return ctypes.create_string_buffer(n)


def test_generators_and_commitments(ref):
    kat = load_golden("kat.json")
    o = buf(32)
    ref.ref_generator(0, 0, 0, o)
    assert o.raw.hex() == kat["B"]
    ref.ref_generator(1, 0, 0, o)
    assert o.raw.hex() == kat["B_blinding"]
    for w in kat["seed_wide"]:
        o64 = buf(64)
        ref.ref_seed_wide(o64, bytes(range(32)), w["dom"], ctypes.c_uint64(w["a"]), ctypes.c_uint64(w["b"]))
        assert o64.raw.hex() == w["out"]
    cm = load_golden("commit.json")
    n = len(cm)
    v = (ctypes.c_uint64 * n)(*[c["v"] for c in cm])
    r = b"".join(bytes.fromhex(c["r"]) for c in cm)
    C, H = buf(32 * n), buf(32 * n)
    ref.ref_commit_hash(ctypes.c_size_t(n), v, r, C, H)
    for i, c in enumerate(cm):
        assert C.raw[32 * i:32 * i + 32].hex() == c["C"] and H.raw[32 * i:32 * i + 32].hex() == c["H"]


def test_trees(ref):
    for t in load_golden("trees.json"):
        n = len(t["leaves"])
        idx = (ctypes.c_uint64 * n)(*[l["idx"] for l in t["leaves"]])
        v = (ctypes.c_uint64 * n)(*[l["v"] for l in t["leaves"]])
        r = b"".join(bytes.fromhex(l["r"]) for l in t["leaves"])
        for faithful in (0, 1):
            tr = ctypes.c_void_p(ref.ref_tree_build(t["height"], ctypes.c_size_t(n), idx, v, r, bytes.fromhex(t["pad_seed"]), faithful))
            C, Hh, rr, vv = buf(32), buf(32), buf(32), ctypes.c_uint64()
            ref.ref_tree_root(tr, C, Hh, ctypes.byref(vv), rr)
            assert (C.raw.hex(), Hh.raw.hex(), vv.value, rr.raw.hex()) == (t["root"]["C"], t["root"]["H"], t["root"]["v"], t["root"]["r"])
            assert ref.ref_tree_node_count(tr) == t["node_count"]
            h = t["height"]
            for li, sibs in t["paths"].items():
                sC, sH, sr, sv = buf(32 * h), buf(32 * h), buf(32 * h), (ctypes.c_uint64 * h)()
                assert ref.ref_tree_path(tr, ctypes.c_uint64(int(li)), sC, sH, sv, sr) == 1
                for s in range(h):
                    assert (sC.raw[32 * s:32 * s + 32].hex(), sH.raw[32 * s:32 * s + 32].hex(), sv[s], sr.raw[32 * s:32 * s + 32].hex()) == \
                        (sibs[s]["C"], sibs[s]["H"], sibs[s]["v"], sibs[s]["r"])
            assert ref.ref_tree_path(tr, ctypes.c_uint64(2**63 + 12345), buf(32 * h), buf(32 * h), (ctypes.c_uint64 * h)(), buf(32 * h)) == 0 or h == 64
            ref.ref_tree_free(tr)
    # unsorted / duplicate / out-of-range leaves are rejected (smtree panics)
    idx = (ctypes.c_uint64 * 2)(5, 5)
    v = (ctypes.c_uint64 * 2)(1, 2)
    assert ref.ref_tree_build(4, ctypes.c_size_t(2), idx, v, bytes(64), bytes(32), 0) is None
    idx = (ctypes.c_uint64 * 2)(3, 16)
    assert ref.ref_tree_build(4, ctypes.c_size_t(2), idx, v, bytes(64), bytes(32), 0) is None


def test_range_proofs_and_verify(ref):
    c7 = bytes([7]) + bytes(31)
    for c in load_golden("range.json"):
        n, m = c["n"], c["m"]
        ps = ref.ref_range_proof_size(n, m)
        v = (ctypes.c_uint64 * m)(*c["values"])
        r = b"".join(bytes.fromhex(b) for b in c["blindings"])
        Vs = b"".join(bytes.fromhex(x) for x in c["commitments"])
        for faithful in ((0, 1) if n * m <= 256 else (0,)):
            out = buf(ps)
            assert ref.ref_range_prove(n, m, v, r, bytes.fromhex(c["nonce_seed"]), ctypes.c_uint64(c["stream_id"]), ctypes.c_uint64(0), None,
                                       faithful, out) == 0
            assert out.raw.hex() == c["proof"]
            assert ref.ref_range_verify(n, m, out, ctypes.c_size_t(ps), Vs, c7, faithful) == 1
        bad = bytearray(out.raw)
        bad[70] ^= 1
        assert ref.ref_range_verify(n, m, bytes(bad), ctypes.c_size_t(ps), Vs, c7, 0) == 0
        assert ref.ref_range_verify(n, m, out, ctypes.c_size_t(ps - 32), Vs, c7, 0) == 0
        badV = bytearray(Vs)
        badV[0] ^= 2
        assert ref.ref_range_verify(n, m, out, ctypes.c_size_t(ps), bytes(badV), c7, 0) == 0


def test_tape_mode_equals_seed_mode(ref, pyref):
    n, m, seed, sid = 8, 4, bytes(range(32)), 42
    slots = m * (2 * n + 4)
    v = (ctypes.c_uint64 * m)(1, 2, 3, 255)
    r = b"".join(pyref.scalar_bytes(pyref.scalar_from_wide(pyref.seed_wide(seed, 5, 0, j))) for j in range(m))
    # seed mode draws from a key bound to the statement (stream, first slot, shape, value commitments): the tape replays it
    Vs = [pyref.pedersen_commit(v[j], int.from_bytes(r[32 * j:32 * j + 32], "little")).compress() for j in range(m)]
    key = pyref.nonce_key(seed, sid, 0, n, m, Vs)
    tape = b"".join(pyref.seed_wide(key, 2, sid, s) for s in range(slots))
    ps = ref.ref_range_proof_size(n, m)
    a, b = buf(ps), buf(ps)
    ref.ref_range_prove(n, m, v, r, seed, ctypes.c_uint64(sid), ctypes.c_uint64(0), None, 0, a)
    ref.ref_range_prove(n, m, v, r, None, ctypes.c_uint64(0), ctypes.c_uint64(0), tape, 0, b)
    assert a.raw == b.raw
    assert a.raw == pyref.range_prove([1, 2, 3, 255], [int.from_bytes(r[32 * j:32 * j + 32], "little") for j in range(m)], n,
                                      pyref.Tape(draws=[tape[64 * s:64 * s + 64] for s in range(slots)]))


def test_entity_proofs_padding_policy(ref, pyref):
    for c in load_golden("dapol.json"):
        if c["policy"] != "padding" or c["agg"] != c["height"]:
            continue
        n = len(c["leaves"])
        idx = (ctypes.c_uint64 * n)(*[l["idx"] for l in c["leaves"]])
        v = (ctypes.c_uint64 * n)(*[l["v"] for l in c["leaves"]])
        r = b"".join(bytes.fromhex(l["r"]) for l in c["leaves"])
        tr = ctypes.c_void_p(ref.ref_tree_build(c["height"], ctypes.c_size_t(n), idx, v, r, bytes.fromhex(c["pad_seed"]), 0))
        leaf = (ctypes.c_uint64 * 1)(c["leaf"])
        out = buf(len(c["aggregated"][0]) // 2)
        assert ref.ref_prove_entities_padding(tr, ctypes.c_size_t(1), leaf, c["n_bits"], bytes.fromhex(c["nonce_seed"]), 0, out) == 0
        assert out.raw.hex() == c["aggregated"][0]
        ref.ref_tree_free(tr)
