"""Pins the Python oracle (oracle/pyref.py) to external known answers and to the reference's own known answers.
CPU only."""
import hashlib

from conftest import load_golden

RFC9496_MULTIPLES = """0000000000000000000000000000000000000000000000000000000000000000
e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76
6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919
94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259
da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57
e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e
f64746d3c92b13050ed8d80236a7f0007c3b3f962f5ba793d19a601ebb1df403
44f53520926ec81fbd5a387845beb7df85a96a24ece18738bdcfa6a7822a176d
903293d8f2287ebe10e2374dc1a53e0bc887e592699f02d077d5263cdd55601c
02622ace8f7303a31cafc63f8fc48fdc16e1c8c8d234b2f0d6685282a9076031
20706fd788b2720a1ed2a5dad4952b01f413bcf0e7564de8cdc816689e2db95f
bce83f8ba5dd2fa572864c24ba1810f9522bc6004afe95877ac73241cafdab42
e4549ee16b9aa03099ca208c67adafcafa4c3f3e4e5303de6026e3ca8ff84460
aa52e000df2e16f55fb1032fc33bc42742dad6bd5a8fc0be0167436c5948501f
46376b80f409b29dc2b5f6f0c52591990896e5716f41477cd30085ab7f10301e
e0c418f7c8d9c4cdd7395b93ea124f3ad99021bb681dfc3302a9d99a2e53e64e""".split()


def test_rfc9496_basepoint_multiples(pyref):
    for i, h in enumerate(RFC9496_MULTIPLES):
        assert (i * pyref.BASEPOINT).compress().hex() == h
        d = pyref.decompress(bytes.fromhex(h))
        assert d is not None and d.compress().hex() == h
    assert (pyref.L * pyref.BASEPOINT).compress() == bytes(32)


def test_rfc9496_constants_and_bad_encodings(pyref):
    R = pyref
    assert R.SQRT_AD_MINUS_ONE ** 2 % R.P == (-R.D - 1) % R.P
    assert R.INVSQRT_A_MINUS_D ** 2 * (-1 - R.D) % R.P == 1
    assert R.ONE_MINUS_D_SQ == (1 - R.D * R.D) % R.P and R.D_MINUS_ONE_SQ == (R.D - 1) ** 2 % R.P
    for bad in (R.P.to_bytes(32, "little"), (1).to_bytes(32, "little"), b"\xff" * 32, (R.P - 1).to_bytes(32, "little"),
                bytes(31) + b"\x80"):
        assert R.decompress(bad) is None


def test_rfc9496_hash_to_group_and_pedersen(pyref):
    lab = b"Ristretto is traditionally a short shot of espresso coffee"
    assert pyref.from_uniform_bytes(hashlib.sha512(lab).digest()).compress().hex() == \
        "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"
    # bulletproofs PedersenGens::default().B_blinding
    assert pyref.B_BLINDING.compress().hex() == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"


def test_merlin_and_strobe_vectors(pyref):
    t = pyref.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    s = pyref.Strobe128(b"Conformance Test Protocol")           # merlin src/strobe.rs test_conformance
    s.meta_ad(b"ms", False)
    s.meta_ad(b"g", True)
    s.ad(bytes([99]) * 1024, False)
    s.meta_ad(b"prf", False)
    p1 = s.prf(32, False)
    assert p1.hex() == "b48e645ca17c667fd5206ba57a6a228d72d8e1903814d3f17f622996d7cfefb0"
    s.meta_ad(b"key", False)
    s.key(p1, False)
    s.meta_ad(b"prf", False)
    assert s.prf(32, False).hex() == "07e45cce8078cee259e3e375bb85d75610e2d1e1201c5f645045a194edd49ff8"
    st = bytearray(200)
    st[0] ^= 0x06
    st[135] ^= 0x80
    pyref.keccak_f1600(st)
    assert bytes(st[:32]) == hashlib.sha3_256(b"").digest()


def test_blake3_official_vectors(pyref):
    assert pyref.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert pyref.blake3(b"abc").hex() == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"
    inp = bytes(i % 251 for i in range(1024))
    assert pyref.blake3(inp[:128]).hex() == "f17e570564b26578c33bb7f44643f539624b05df1a76c81f30acd548c44b45ef"
    assert pyref.blake3(inp).hex() == "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7"


def test_blake3_beyond_one_chunk(pyref):
    """Tree mode (liability ids of any length, src/dapol/mod.rs:347-349): official input pattern at the official lengths and around
    the chunk / subtree boundaries, against hashes made with the BLAKE3 team's own C code (tests/golden/gen_blake3_kat.py)."""
    from conftest import load_golden
    for vec in load_golden("blake3_long.json")["vectors"]:
        assert pyref.blake3(bytes(i % 251 for i in range(vec["len"]))).hex() == vec["hash"], vec["len"]
    # the 1,025-byte official vector, from the published test_vectors.json (independent of the generator above)
    assert pyref.blake3(bytes(i % 251 for i in range(1025))).hex() == "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444"


def test_reference_index_kats(pyref):
    """src/dapol/tests.rs:30-85 (a->7, b->12, c->2, d->4), :24 (root value 26): Blake2s, seed "test", height 4."""
    liab = [(b"a", b"w", 3), (b"b", b"x", 5), (b"c", b"y", 7), (b"d", b"z", 11)]
    tree, idm = pyref.dapol_new(liab, b"test", 4, bytes(32), "blake2s")
    assert idm == {b"a": 7, b"b": 12, b"c": 2, b"d": 4}
    assert tree.root.v == 26
    kat = load_golden("kat.json")
    assert kat["blake2s"]["index"] == {"a": 7, "b": 12, "c": 2, "d": 4}


def test_reference_errors(pyref):
    import pytest
    liab = [(b"a", b"w", 3), (b"a", b"x", 5)]
    with pytest.raises(pyref.DapolError, match="DuplicatedInternalId"):
        pyref.dapol_new(liab, b"test", 4, bytes(32), "blake3")
    with pytest.raises(pyref.DapolError, match="TreeHeightTooBig"):
        pyref.dapol_new(liab[:1], b"test", 65, bytes(32), "blake3")
    with pytest.raises(pyref.DapolError, match="SparsityTooSmall"):
        pyref.dapol_new([(bytes([i]), b"e", 1) for i in range(9)], b"test", 4, bytes(32), "blake3")
    with pytest.raises(pyref.DapolError, match="InvalidDigestSize"):
        pyref.dapol_new(liab[:1], b"test", 4, bytes(32), "blake2b")


def test_proof_sizes_and_roundtrip(pyref):
    """SINGLE_PROOF_BYTE_NUM = 672 (src/range/mod.rs:18) and the prove -> verify round trip with tamper / out-of-range."""
    R = pyref
    assert [R.range_proof_size(*a) for a in ((64, 1), (64, 32), (64, 1024), (8, 2))] == [672, 992, 1312, 544]
    seed = bytes(range(32))
    bl = [R.scalar_from_wide(R.seed_wide(seed, 9, 0, i)) for i in range(2)]
    pr = R.range_prove([5, 255], bl, 8, R.Tape(seed=seed, stream_id=7))
    Vs = [R.pedersen_commit(v, b).compress() for v, b in zip([5, 255], bl)]
    assert len(pr) == 544 and R.range_verify(pr, Vs, 8)
    bad = bytearray(pr)
    bad[40] ^= 1
    assert not R.range_verify(bytes(bad), Vs, 8)
    pr2 = R.range_prove([256, 1], bl, 8, R.Tape(seed=seed, stream_id=7))
    assert not R.range_verify(pr2, [R.pedersen_commit(256, bl[0]).compress(), R.pedersen_commit(1, bl[1]).compress()], 8)
    assert not R.range_verify(pr[:-32], Vs, 8)


def test_policy_serialization_roundtrip(pyref):
    for c in load_golden("dapol.json"):
        agg, ind = [bytes.fromhex(x) for x in c["aggregated"]], [bytes.fromhex(x) for x in c["individual"]]
        ser = pyref.policy_serialize(c["policy"], agg, ind)
        assert ser.hex() == c["serialized"]
        single = pyref.range_proof_size(c["n_bits"], 1)
        a2, i2, end = pyref.policy_deserialize(c["policy"], ser, single_size=single)
        assert (a2, i2, end) == (agg, ind, len(ser))
        assert pyref.policy_verify(c["policy"], a2, i2, [bytes.fromhex(s["C"]) for s in c["siblings"]], n=c["n_bits"])
        import pytest
        with pytest.raises(ValueError):
            pyref.policy_deserialize(c["policy"], ser[:-5], single_size=single)


def test_commitment_phase_of_the_transcript_is_a_closed_form_stream(pyref):
    """k_rv_absorb_V (kernels_verify.h) absorbs the m commitments of a range-proof transcript block by block, every lane
    computing its own bytes of a 166-byte STROBE block from the stream offset alone.  This restates that closed form in
    Python -- record j contributes [pos_begin, M|A, 'V', LE32(32), pos_begin', A, 32 bytes], the two position bytes being
    (q - d) + 1 when the previous begin_op (d = 34 or 7 bytes earlier) lies in the same block and 0 after a run_f -- and
    checks state, pos and pos_begin against merlin's byte-by-byte Strobe for heads that leave the stream at every offset."""
    R = pyref
    rate = R.STROBE_R
    import random
    rnd = random.Random(11)
    seen = set()
    for m in (1, 2, 4, 8, 32, 64):
        for pad in range(0, 170, 7):
            t = R.Transcript(b"")
            t.append_message(b"dom-sep", b"rangeproof v1")
            t.append_message(b"x", bytes(pad))                    # moves the start of the stream through the block
            t.append_message(b"m", m.to_bytes(8, "little"))
            V = [bytes(rnd.randrange(256) for _ in range(32)) for _ in range(m)]
            ref = t.strobe.clone()
            tt = R.Transcript.__new__(R.Transcript)
            tt.strobe = ref
            for v in V:
                tt.append_message(b"V", v)
            st = t.strobe.clone()
            pos0, pb0 = st.pos, st.pos_begin
            seen.add(pos0)
            total, end_abs = 41 * m, pos0 + 41 * m
            nfull = end_abs // rate
            state = bytearray(st.state)

            def stream_byte(k, q):
                j, tb = divmod(k, 41)
                if tb >= 9:
                    return V[j][tb - 9]
                if tb == 0:
                    return pb0 if j == 0 else (q - 33 if q >= 34 else 0)
                if tb == 7:
                    return q - 6 if q >= 7 else 0
                return {1: 0x12, 2: ord("V"), 3: 32, 8: 0x02}.get(tb, 0)

            for beta in range(nfull + 1):
                for q in range(rate):
                    at = beta * rate + q
                    if pos0 <= at < end_abs:
                        state[q] ^= stream_byte(at - pos0, q)
                if beta < nfull:
                    k_end = beta * rate + rate - 1 - pos0
                    je, te = divmod(k_end, 41)
                    kb = 41 * je + (7 if te >= 7 else 0)
                    ab = pos0 + kb
                    pbe = ab % rate + 1 if ab >= beta * rate else (pb0 if beta == 0 else 0)
                    state[rate] ^= pbe
                    state[rate + 1] ^= 0x04 ^ 0x80
                    R.keccak_f1600(state)
            kb = 41 * (m - 1) + 7
            pos_begin = (pos0 + kb) % rate + 1 if (pos0 + kb) // rate == nfull else 0
            assert bytes(state) == bytes(ref.state) and end_abs % rate == ref.pos and pos_begin == ref.pos_begin, (m, pad)
    assert len(seen) > 20
