"""The PRODUCT limb arithmetic (dapol_amd/csrc/{fe,ge,sc,hash}.h compiled for the host) against the big-integer
oracle.  CPU only -- the same headers are what the gfx950 kernels inline."""
import ctypes
import hashlib
import random


def buf(n):
    return ctypes.create_string_buffer(n)


def test_field_ops(host_shim, pyref):
    P = pyref.P
    rnd = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, 2**255 - 20, 2**254, 19, 2**26 - 1, 2**255 - 19 - 2**26, 2**29 - 1, 2**29, 2**232 - 1, 2**232,
            2**255 - 19 - 2**29, sum((2**29 - 1) << (29 * i) for i in range(8)) + ((2**23 - 1) << 232) - 19, 2**255 - 1 - 19, 2**58 - 1, 2**261 % P]

    def op(k, a, b):
        o = buf(32)
        host_shim.t_fe_op(k, a.to_bytes(32, "little"), b.to_bytes(32, "little"), o)
        return int.from_bytes(o.raw, "little")
    for it in range(1500):
        a = rnd.choice(edge) if it < 100 else rnd.randrange(P)
        b = rnd.choice(edge) if it < 100 and it % 2 else rnd.randrange(P)
        assert op(0, a, b) == a * b % P
        assert op(1, a, b) == a * a % P
        assert op(3, a, b) == (2 * a + b) * b % P          # loose f operand
        assert op(4, a, b) == (a - b) ** 2 % P             # tight difference squared
        assert op(5, a, b) == (-(a - 2 * b)) % P           # carry of a negative loose value
        assert op(6, a, b) == (a + b) % P                  # fe_tobytes on uncarried two-term sums / differences
        assert op(7, a, b) == (a + b) % P
        assert op(8, a, b) == (-a - b) % P
        if it < 100 and a % P:
            assert op(2, a, b) == pow(a, P - 2, P)


def test_group_law_and_codec(host_shim, pyref):
    R = pyref
    rnd = random.Random(2)
    for it in range(24):
        k = [0, 1, 2, R.L - 1, R.L, 2**255 - 1, 8 * R.L][it] if it < 7 else rnd.randrange(2**255)
        o = buf(32)
        host_shim.t_basemul((k % 2**256).to_bytes(32, "little"), o)
        assert o.raw == (k * R.BASEPOINT).compress()
    Bb = R.B_BLINDING.compress()
    for it in range(10):
        k = rnd.randrange(2**255)
        o = buf(32)
        host_shim.t_maddmul(Bb, k.to_bytes(32, "little"), it & 1, o)
        e = k * R.B_BLINDING
        assert o.raw == (-e if it & 1 else e).compress()
    for it in range(40):
        c = (rnd.randrange(R.L) * R.BASEPOINT).compress() if it < 20 else bytes(rnd.randrange(256) for _ in range(31)) + bytes([rnd.randrange(128)])
        o = buf(32)
        ok = host_shim.t_decompress(c, o)
        ref = R.decompress(c)
        assert bool(ok) == (ref is not None)
        if ok:
            assert o.raw == c
    for c in (bytes(32), R.P.to_bytes(32, "little"), (1).to_bytes(32, "little"), b"\xff" * 32):
        assert bool(host_shim.t_decompress(c, buf(32))) == (R.decompress(c) is not None)
    for it in range(20):
        u = bytes(rnd.randrange(256) for _ in range(64))
        o = buf(32)
        host_shim.t_from_uniform(u, o)
        assert o.raw == R.from_uniform_bytes(u).compress()


def test_add_of_decompressed_points(host_shim, pyref):
    """Regression: decompress returns |x|; the general addition must stay within the limb bounds for such inputs
    (Mergeable::merge on compressed records, Merkle path re-merge)."""
    R = pyref
    rnd = random.Random(11)
    pts = [rnd.randrange(R.L) * R.BASEPOINT for _ in range(40)]
    for i in range(400):
        a, b = rnd.choice(pts), rnd.choice(pts)
        o = buf(32)
        host_shim.t_add_compressed(a.compress(), b.compress(), o)
        assert o.raw == (a + b).compress(), i


def test_scalar_field(host_shim, pyref):
    L = pyref.L
    rnd = random.Random(3)
    edge = [(2**256 - 1, L - 1), (L - 1, L - 1), (0, L - 1), (2**256 - 1, 0), (2**256 - 1, 1), (L, L - 1), (2**255, 2**252), (2**29 - 1, 2**29 - 1),
            (2**232, 2**232), ((1 << 256) - (1 << 232), L - 2)]
    for it in range(600):
        a = rnd.randrange(2**256) if it % 3 else rnd.randrange(L)
        b = rnd.randrange(L)
        if it < len(edge):
            a, b = edge[it]
        for k, f in ((0, a * b % L), (1, (a + b) % L), (2, (a - b) % L)):
            o = buf(32)
            host_shim.t_sc_op(k, (a % L if k else a).to_bytes(32, "little"), b.to_bytes(32, "little"), o)
            exp = f if k == 0 else ((a % L + b) % L if k == 1 else (a % L - b) % L)
            assert int.from_bytes(o.raw, "little") == exp
        if it < 30 and a % L:
            o = buf(32)
            host_shim.t_sc_op(3, a.to_bytes(32, "little"), b.to_bytes(32, "little"), o)
            assert int.from_bytes(o.raw, "little") == pow(a, L - 2, L)
        w = [bytes(64), b"\xff" * 64, L.to_bytes(64, "little"), (L * L).to_bytes(64, "little")][it] if it < 4 else bytes(rnd.randrange(256) for _ in range(64))
        o = buf(32)
        host_shim.t_sc_from_wide(w, o)
        assert int.from_bytes(o.raw, "little") == int.from_bytes(w, "little") % L
        x = [2**255 - 1, 0, 2**255 - 2**247][it] if it < 3 else rnd.randrange(2**255)
        d = (ctypes.c_int16 * 32)()
        host_shim.t_sc_recode(x.to_bytes(32, "little"), d)
        assert sum(int(d[i]) * 256**i for i in range(32)) == x and all(abs(int(v)) <= 128 for v in d)
        for w in range(8, 17):
            nw = 255 // w + 1                                      # any 255-bit integer (unreduced leaf blindings)
            dw = (ctypes.c_int * nw)()
            host_shim.t_sc_recode_w(w, nw, x.to_bytes(32, "little"), dw)
            assert sum(int(dw[i]) << (w * i) for i in range(nw)) == x and all(abs(int(v)) <= (1 << (w - 1)) for v in dw)
            xc = x % L
            nwc = 253 // w + 1                                     # canonical scalars (digit matrices)
            dc = (ctypes.c_int * nwc)()
            host_shim.t_sc_recode_w(w, nwc, xc.to_bytes(32, "little"), dc)
            assert sum(int(dc[i]) << (w * i) for i in range(nwc)) == xc and all(-32768 <= int(v) <= 32767 for v in dc)


def test_hashes_and_transcript(host_shim, pyref):
    R = pyref
    rnd = random.Random(4)
    for it in range(30):
        m = bytes(rnd.randrange(256) for _ in range(128))
        o = buf(32)
        host_shim.t_blake3_32(m[:32], o)
        assert o.raw == R.blake3(m[:32])
        host_shim.t_blake3_128(m, o)
        assert o.raw == R.blake3(m)
        o = buf(64)                                                  # the 64-byte node hashes of Dapol<blake2::Blake2b, _> (src/tests.rs:100-101)
        w = m + m[:64]
        host_shim.t_blake2b_32(w[:32], o)
        assert o.raw == hashlib.blake2b(w[:32]).digest()
        host_shim.t_blake2b_192(w, o)
        assert o.raw == hashlib.blake2b(w).digest()
        a, b = rnd.randrange(2**64), rnd.randrange(2**64)
        host_shim.t_seed_wide(m[:32], it, ctypes.c_uint64(a), ctypes.c_uint64(b), o)
        assert o.raw == R.seed_wide(m[:32], it, a, b)
        n = rnd.randrange(0, 300)
        o = buf(200)
        host_shim.t_sponge(136, 0x1F, m * 3, n, o, 200)
        assert o.raw == hashlib.shake_256((m * 3)[:n]).digest(200)
        o = buf(64)
        host_shim.t_sponge(72, 0x06, m * 3, n, o, 64)
        assert o.raw == hashlib.sha3_512((m * 3)[:n]).digest()
    for n in list(range(0, 70)) + [127, 128, 129, 191, 192, 193, 500, 1000, 1024]:
        m = bytes(rnd.randrange(256) for _ in range(n))
        o = buf(32)
        host_shim.t_digest(1, m, n, o)
        assert o.raw == hashlib.blake2s(m).digest(), n
        host_shim.t_digest(0, m, n, o)
        assert o.raw == R.blake3(m), n
    # BLAKE3 beyond one chunk (dg_init_long: chunk chaining values + parent nodes) against vectors made by the BLAKE3 team's C code
    # (tests/golden/blake3_long.json); the stack-less digest flags such inputs instead of hashing them wrongly
    from conftest import load_golden
    for vec in load_golden("blake3_long.json")["vectors"]:
        n = vec["len"]
        m = bytes(i % 251 for i in range(n))
        o = buf(32)
        host_shim.t_digest(0, m, n, o)
        assert o.raw.hex() == vec["hash"], n
        assert host_shim.t_digest_short(0, m, n, o) == (1 if n > 1024 else 0), n
        host_shim.t_digest(1, m, n, o)
        assert o.raw == hashlib.blake2s(m).digest(), n
    o = buf(64)
    host_shim.t_merlin(b"test protocol", 13, b"some label", 10, b"some data", 9, b"challenge", 9, o)
    t = R.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert o.raw == t.challenge_bytes(b"challenge", 64)
    host_shim.t_merlin(b"", 0, b"dom-sep", 7, b"rangeproof v1", 13, b"y", 1, o)
    t = R.Transcript(b"")
    t.append_message(b"dom-sep", b"rangeproof v1")
    assert o.raw == t.challenge_bytes(b"y", 64)


def test_commitment_stream_in_blocks_equals_bytewise_strobe(host_shim):
    """k_rv_absorb_V's closed form of the STROBE stream of m commitments (hash.h: absorb_window / absorb_block_word /
    absorb_runf_word / absorb_end_position -- the functions the kernel itself calls, compiled for the host) leaves the state and
    the two positions the byte-wise Strobe leaves, for every party count, for transcript heads of every length (the stream then
    starts at every offset of the 166-byte block) and in 1, 2 or 4 phases like dapol_range_verify_batch's column blocks."""
    import ctypes
    import numpy as np
    rng = np.random.default_rng(11)
    u64 = ctypes.POINTER(ctypes.c_uint64)
    cases = 0
    for m in (1, 2, 3, 4, 5, 8, 16, 31, 32, 64, 128, 256, 1024):
        V = rng.integers(0, 2**32, size=8 * m, dtype=np.uint32)
        for extra_len in list(range(0, 170, 7)) + [165, 166, 167]:
            extra = bytes(rng.integers(0, 256, size=extra_len, dtype=np.uint8))
            for phases in (1, 2, 4):
                if m % phases:
                    continue
                a, b = np.zeros(27, np.uint64), np.zeros(27, np.uint64)
                host_shim.t_absorb_v(m, V.ctypes.data_as(ctypes.c_void_p), phases, extra, extra_len, a.ctypes.data_as(u64), b.ctypes.data_as(u64))
                assert (a == b).all(), (m, extra_len, phases, np.nonzero(a != b)[0])
                cases += 1
    assert cases > 500
