"""CPU restatement (pure Python, big integers) of the MystenLabs/dapol proving path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (dapol_amd/, bench.py's timed region) may import
this module; only tests/, __graft_entry__.smoke() and the golden-vector generator do.

It restates, for small inputs, exactly what the reference crate computes, following these reference
lines (all paths relative to /root/reference):

  * node algebra ............ src/dapol/node.rs:29-45 (new), :64-80 (merge), :86-88 (padding)
  * tree driver ............. src/dapol/mod.rs:100-128 (new), :172-190 (generate_proof_batch), :196-208
  * leaf derivation ......... src/dapol/mod.rs:323-399 (build_leaf_nodes), :408-441 (shuffle_index)
  * range proof wrappers .... src/range/mod.rs:48-78 (prove), :83-119 (verify), :124-161 (deserialize)
  * padding policy .......... src/range/padding.rs:38-69, :88-118, :169-196
  * splitting policy ........ src/range/splitting.rs:36-84, :100-129, :181-210
  * proof node / proof ...... src/proof/node.rs:56-102, src/proof/mod.rs:41-95

The arithmetic lives in crates that are NOT vendored under /root/reference (Cargo.toml:12-29):
curve25519-dalek-ng 4.1.1 (ristretto255 = RFC 9496), bulletproofs 4.0.0, merlin 3.0.0 (STROBE-128 over
Keccak-f[1600]), blake3 0.3.8, blake2 0.9, smtree 0.1.2.  Their published algorithms are restated here from
the specifications; see DESIGN.md "Oracle" for what pins each piece (RFC 9496 vectors, Merlin / STROBE
conformance vectors, BLAKE3 official vectors, hashlib cross-checks, the reference's own index KATs).

Randomness: the reference draws from thread_rng() (node.rs:87, bulletproofs prover).  Here every draw is an
explicit input -- a "tape" of 64-byte wide scalars consumed in the crate's draw order (see Tape below).
"""
import hashlib
import struct

# ----------------------------------------------------------------------------------------------------------
# Field GF(2^255-19) and ristretto255 (RFC 9496 section 4)
# ----------------------------------------------------------------------------------------------------------
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)
# RFC 9496 section 4.1 constants (checked algebraically in tests/test_oracle_kat.py)
SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
INVSQRT_A_MINUS_D = 54469307008909316920995813868745141605393597292927456921205312896311721017578
ONE_MINUS_D_SQ = 1159843021668779879193775521855586647937357759715417654439879720876111806838
D_MINUS_ONE_SQ = 40440834346308536858101042469323190826248399146238708352240133220865137265952


def is_neg(x):
    return (x % P) & 1


def ct_abs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 4.2 SQRT_RATIO_M1."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u * SQRT_M1) % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    r = ct_abs(r)
    return (correct or flipped), r


class Point:
    """Extended twisted-Edwards coordinates (X:Y:Z:T), a = -1."""
    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % P, Y % P, Z % P, T % P

    def __add__(self, o):
        A = (self.Y - self.X) * (o.Y - o.X) % P
        B = (self.Y + self.X) * (o.Y + o.X) % P
        C = self.T * 2 * D % P * o.T % P
        Dd = self.Z * 2 * o.Z % P
        E, F, G, H = B - A, Dd - C, Dd + C, B + A
        return Point(E * F, G * H, F * G, E * H)

    def __neg__(self):
        return Point(-self.X, self.Y, self.Z, -self.T)

    def __sub__(self, o):
        return self + (-o)

    def double(self):
        return self + self

    def __rmul__(self, k):
        k %= L * 8  # scalars may be unreduced (Scalar::from_bits); the group has order 8*L
        acc, base = IDENTITY, self
        while k:
            if k & 1:
                acc = acc + base
            base = base.double()
            k >>= 1
        return acc

    def __eq__(self, o):
        # ristretto equality: X1*Y2 == Y1*X2 or Y1*Y2 == X1*X2
        return (self.X * o.Y - self.Y * o.X) % P == 0 or (self.Y * o.Y - self.X * o.X) % P == 0

    def compress(self):
        """RFC 9496 4.3.2 Encode."""
        x0, y0, z0, t0 = self.X, self.Y, self.Z, self.T
        u1 = (z0 + y0) * (z0 - y0) % P
        u2 = x0 * y0 % P
        _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
        den1 = invsqrt * u1 % P
        den2 = invsqrt * u2 % P
        z_inv = den1 * den2 % P * t0 % P
        ix0 = x0 * SQRT_M1 % P
        iy0 = y0 * SQRT_M1 % P
        ench = den1 * INVSQRT_A_MINUS_D % P
        rotate = is_neg(t0 * z_inv)
        x, y, den_inv = (iy0, ix0, ench) if rotate else (x0, y0, den2)
        if is_neg(x * z_inv):
            y = (-y) % P
        s = ct_abs(den_inv * (z0 - y))
        return s.to_bytes(32, "little")


IDENTITY = Point(0, 1, 1, 0)
_by = 4 * pow(5, P - 2, P) % P
_bx = ct_abs(sqrt_ratio_m1((_by * _by - 1) % P, (D * _by * _by + 1) % P)[1])  # even root = Ed25519 base x
BASEPOINT = Point(_bx, _by, 1, _bx * _by)


def decompress(b):
    """RFC 9496 4.3.1 Decode.  Returns None for a non-canonical / invalid encoding."""
    if len(b) != 32:
        return None
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = ct_abs(2 * s * den_x)
    y = u1 * den_y % P
    t = x * y % P
    if (not was_square) or is_neg(t) or y == 0:
        return None
    return Point(x, y, 1, t)


def elligator_map(t):
    """RFC 9496 4.3.4 MAP."""
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (-ct_abs(s * t)) % P
    if not was_square:
        s = s_prime
    c = (P - 1) if was_square else r
    N = (c * (r - 1) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = N * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return Point(w0 * w3, w2 * w1, w1 * w3, w0 * w2)


def from_uniform_bytes(b64):
    """RistrettoPoint::from_uniform_bytes: two Elligator maps of the 255-bit-masked halves, added."""
    assert len(b64) == 64
    t1 = (int.from_bytes(b64[:32], "little") & (2**255 - 1)) % P
    t2 = (int.from_bytes(b64[32:], "little") & (2**255 - 1)) % P
    return elligator_map(t1) + elligator_map(t2)


# ----------------------------------------------------------------------------------------------------------
# Scalars (curve25519-dalek Scalar semantics)
# ----------------------------------------------------------------------------------------------------------
def scalar_from_bits(b32):
    """Scalar::from_bits: clear bit 255, NO reduction (src/dapol/mod.rs:385)."""
    return int.from_bytes(b32, "little") & (2**255 - 1)


def scalar_from_wide(b64):
    """Scalar::from_bytes_mod_order_wide == what Scalar::random(rng) does with 64 rng bytes."""
    return int.from_bytes(b64, "little") % L


def scalar_bytes(x):
    return (x % L).to_bytes(32, "little")


def scalar_from_canonical(b32):
    x = int.from_bytes(b32, "little")
    return x if x < L else None


def inv(x):
    return pow(x % L, L - 2, L)


# ----------------------------------------------------------------------------------------------------------
# Keccak-f[1600], STROBE-128 and Merlin (merlin 3.0.0: src/strobe.rs, src/transcript.rs)
# ----------------------------------------------------------------------------------------------------------
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
       0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
       0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
       0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(state: bytearray):
    A = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        Dd = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ Dd[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = A[x][y].to_bytes(8, "little")


STROBE_R = 166
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def clone(self):
        c = object.__new__(Strobe128)
        c.state, c.pos, c.pos_begin, c.cur_flags = bytearray(self.state), self.pos, self.pos_begin, self.cur_flags
        return c

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos, self.pos_begin = 0, 0

    def _absorb(self, data):
        for byte in data:
            self.state[self.pos] ^= byte
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _overwrite(self, data):
        for byte in data:
            self.state[self.pos] = byte
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & FLAG_T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if (flags & (FLAG_C | FLAG_K)) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)

    def key(self, data, more):
        self._begin_op(FLAG_A | FLAG_C, more)
        self._overwrite(data)


class Transcript:
    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def clone(self):
        c = object.__new__(Transcript)
        c.strobe = self.strobe.clone()
        return c

    def append_message(self, label, message):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", len(message)), True)
        self.strobe.ad(message, False)

    def append_u64(self, label, x):
        self.append_message(label, struct.pack("<Q", x))

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", n), True)
        return self.strobe.prf(n, False)

    # bulletproofs 4.0.0 src/transcript.rs (TranscriptProtocol)
    def rangeproof_domain_sep(self, n, m):
        self.append_message(b"dom-sep", b"rangeproof v1")
        self.append_u64(b"n", n)
        self.append_u64(b"m", m)

    def innerproduct_domain_sep(self, n):
        self.append_message(b"dom-sep", b"ipp v1")
        self.append_u64(b"n", n)

    def append_scalar(self, label, s):
        self.append_message(label, scalar_bytes(s))

    def append_point(self, label, comp):
        self.append_message(label, comp)

    def validate_and_append_point(self, label, comp):
        if comp == bytes(32):
            return False
        self.append_message(label, comp)
        return True

    def challenge_scalar(self, label):
        return scalar_from_wide(self.challenge_bytes(label, 64))


# ----------------------------------------------------------------------------------------------------------
# BLAKE3 (single chunk, <= 1024 bytes of input; unkeyed hash and keyed XOF)
# ----------------------------------------------------------------------------------------------------------
_B3_IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_B3_PERM = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
B3_CHUNK_START, B3_CHUNK_END, B3_PARENT, B3_ROOT, B3_KEYED_HASH = 1, 2, 4, 8, 16
_M32 = 0xFFFFFFFF


def _ror32(x, n):
    return ((x >> n) | (x << (32 - n))) & _M32


def _b3_g(s, a, b, c, d, mx, my):
    s[a] = (s[a] + s[b] + mx) & _M32
    s[d] = _ror32(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & _M32
    s[b] = _ror32(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b] + my) & _M32
    s[d] = _ror32(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & _M32
    s[b] = _ror32(s[b] ^ s[c], 7)


def blake3_compress(cv, block_words, counter, block_len, flags):
    s = list(cv) + _B3_IV[:4] + [counter & _M32, (counter >> 32) & _M32, block_len, flags]
    m = list(block_words)
    for r in range(7):
        _b3_g(s, 0, 4, 8, 12, m[0], m[1])
        _b3_g(s, 1, 5, 9, 13, m[2], m[3])
        _b3_g(s, 2, 6, 10, 14, m[4], m[5])
        _b3_g(s, 3, 7, 11, 15, m[6], m[7])
        _b3_g(s, 0, 5, 10, 15, m[8], m[9])
        _b3_g(s, 1, 6, 11, 12, m[10], m[11])
        _b3_g(s, 2, 7, 8, 13, m[12], m[13])
        _b3_g(s, 3, 4, 9, 14, m[14], m[15])
        if r < 6:
            m = [m[_B3_PERM[i]] for i in range(16)]
    for i in range(8):
        s[i] ^= s[i + 8]
        s[i + 8] ^= cv[i]
    return s


def _blake3_chunk(chunk: bytes, key_words, base_flags, counter):
    """One chunk (<= 1024 bytes) -> (input chaining value, last block words, last block length, flags) of its final compression."""
    blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
    cv = list(key_words)
    for i, blk in enumerate(blocks):
        flags = base_flags | (B3_CHUNK_START if i == 0 else 0)
        words = list(struct.unpack("<16I", blk.ljust(64, b"\0")))
        if i < len(blocks) - 1:
            cv = blake3_compress(cv, words, counter, 64, flags)[:8]
        else:
            return cv, words, counter, len(blk), flags | B3_CHUNK_END


def _blake3_hash(data: bytes, key_words, base_flags, out_len):
    """BLAKE3 of any length (the published tree mode: 1024-byte chunks, the left subtree of a node holds the largest power of two
    of chunks that leaves the right one non-empty; the root compression carries ROOT and, for longer outputs, the block counter)."""
    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]

    def node(lo, hi):                                   # -> the final compression's inputs of the subtree over chunks[lo:hi]
        if hi - lo == 1:
            return _blake3_chunk(chunks[lo], key_words, base_flags, lo)
        split = 1 << ((hi - lo - 1).bit_length() - 1)
        left = blake3_compress(*node(lo, lo + split))[:8]
        right = blake3_compress(*node(lo + split, hi))[:8]
        return list(key_words), left + right, 0, 64, base_flags | B3_PARENT

    cv, words, counter, blen, flags = node(0, len(chunks))
    out, ctr = b"", 0
    while len(out) < out_len:
        # the root's counter field is the output block counter (a one-chunk input has chunk counter 0 anyway)
        out += struct.pack("<16I", *blake3_compress(cv, words, ctr, blen, flags | B3_ROOT))
        ctr += 1
    return out[:out_len]


def _blake3_single_chunk(data: bytes, key_words, base_flags, out_len):
    assert len(data) <= 1024, "single-chunk BLAKE3 only (node hashes are 32 or 128 bytes)"
    return _blake3_hash(data, key_words, base_flags, out_len)


def blake3(data: bytes, out_len=32):
    return _blake3_hash(data, _B3_IV, 0, out_len)


def blake3_keyed(key32: bytes, data: bytes, out_len=32):
    return _blake3_single_chunk(data, struct.unpack("<8I", key32), B3_KEYED_HASH, out_len)


class _Blake3Hasher:
    digest_size = 32

    def __init__(self):
        self.buf = b""

    def update(self, d):
        self.buf += bytes(d)

    def digest(self):
        return blake3(self.buf)


DIGESTS = {
    "blake3": (_Blake3Hasher, 32),
    "blake2s": (hashlib.blake2s, 32),
    "blake2b": (hashlib.blake2b, 64),
}


def digest(name, *parts):
    h = DIGESTS[name][0]()
    for p in parts:
        h.update(p)
    return h.digest()


# ----------------------------------------------------------------------------------------------------------
# Randomness tapes (the determinism contract; DESIGN.md "Randomness")
# ----------------------------------------------------------------------------------------------------------
DOMAIN_PAD, DOMAIN_NONCE = 1, 2


def seed_wide(seed32: bytes, domain: int, a: int, b: int) -> bytes:
    """64 pseudo-random bytes for draw (domain, a, b): first XOF block of BLAKE3-keyed(seed, LE32 dom|LE64 a|LE64 b)."""
    return blake3_keyed(seed32, struct.pack("<IQQ", domain, a, b), 64)


def nonce_key(seed32: bytes, stream_id: int, slot_base: int, n: int, m: int, commitments) -> bytes:
    """Seed mode: key of one proof's nonce stream, bound to the statement it blinds (stream, first slot, shape and the
    parties' value commitments, 31 per BLAKE3 chunk) -- include/dapol_hip.h "Randomness contract".  The crate draws from
    thread_rng; a deterministic stream that ignored the statement would reuse nonces when a leaf is proved again after
    its siblings changed."""
    key = seed_wide(seed32, 6, stream_id, slot_base)[:32]
    key = seed_wide(key, 7, n, m)[:32]
    for j0 in range(0, m, 31):
        key = blake3(key + b"".join(commitments[j0:j0 + 31]))
    return key


class Tape:
    """Sequence of wide scalars (Scalar::random draws) in the order the crate draws them."""

    def __init__(self, draws=None, seed=None, domain=DOMAIN_NONCE, stream_id=0):
        self.draws, self.seed, self.domain, self.stream_id, self.pos = draws, seed, domain, stream_id, 0
        self.key = None

    def rekey(self, n, m, commitments):
        """Called by the prover once the parties' commitments exist and before its first draw."""
        if self.draws is None:
            self.key = nonce_key(self.seed, self.stream_id, self.pos, n, m, commitments)

    def next_wide(self) -> bytes:
        if self.draws is not None:
            w = self.draws[self.pos]
        else:
            w = seed_wide(self.key if self.key is not None else self.seed, self.domain, self.stream_id, self.pos)
        self.pos += 1
        return w

    def scalar(self) -> int:
        return scalar_from_wide(self.next_wide())


def pad_blinding(pad_seed: bytes, height_from_leaves: int, index: int) -> int:
    """Blinding of the padding node at (level above leaves, index in that level): positional, order-free.
    Stands in for Scalar::random(thread_rng()) at src/dapol/node.rs:87.  TAPE mode: pad_seed is a dict
    {(level, index): the node's 64-byte draw} (dapol_tree_build_tape; what a harness around the real crate records)."""
    if isinstance(pad_seed, dict):
        return scalar_from_wide(pad_seed[(height_from_leaves, index)])
    return scalar_from_wide(seed_wide(pad_seed, DOMAIN_PAD, height_from_leaves, index))


# ----------------------------------------------------------------------------------------------------------
# Generators (bulletproofs 4.0.0 src/generators.rs)
# ----------------------------------------------------------------------------------------------------------
B_COMPRESSED = BASEPOINT.compress()
B_BLINDING = from_uniform_bytes(hashlib.sha3_512(B_COMPRESSED).digest())  # PedersenGens::default().B_blinding


def pedersen_commit(v, r):
    """PedersenGens::commit(v, r) = v*B + r*B_blinding."""
    return v * BASEPOINT + r * B_BLINDING


_gens_cache = {}


def bp_gens(n, m):
    """BulletproofGens::new(n, m): returns (G, H), each a list of n*m points, party-major."""
    key = (n, m)
    if key not in _gens_cache:
        G, H = [], []
        for j in range(m):
            for label, dst in ((b"G", G), (b"H", H)):
                xof = hashlib.shake_256(b"GeneratorsChain" + label + struct.pack("<I", j)).digest(64 * n)
                for i in range(n):
                    dst.append(from_uniform_bytes(xof[64 * i:64 * i + 64]))
        _gens_cache[key] = (G, H)
    return _gens_cache[key]


# ----------------------------------------------------------------------------------------------------------
# Aggregated range proof (bulletproofs 4.0.0 src/range_proof/{mod,dealer,party}.rs, src/inner_product_proof.rs)
# ----------------------------------------------------------------------------------------------------------
def _msm(scalars, points):
    acc = IDENTITY
    for s, pt in zip(scalars, points):
        acc = acc + (s % L) * pt
    return acc


def _ip(a, b):
    return sum(x * y for x, y in zip(a, b)) % L


def range_prove(values, blindings, n, tape: Tape, label=b"", commit_values=None, A_offset=0, info=None):
    """RangeProof::prove_multiple_with_rng with Transcript::new(label) (src/range/mod.rs:48-78 uses label=[]).
    Draw order per party j: a_blinding, s_blinding, s_L[0..n), s_R[0..n); then per party: t1_blinding, t2_blinding.
    Adversarial hooks for the soundness tests (never used by the honest paths): commit_values = the values the V_j commit
    to while the bits still come from `values` (an out-of-range statement with an otherwise honest proof); A_offset = kappa
    puts A + kappa*B into the transcript and the proof; info (a dict) receives the challenges."""
    m = len(values)
    assert n in (8, 16, 32, 64) and m & (m - 1) == 0 and m > 0 and len(blindings) == m
    G, H = bp_gens(n, m)
    tr = Transcript(label)
    tr.rangeproof_domain_sep(n, m)
    # Party::new + assign_position_with_rng
    parties = []
    Vs = [pedersen_commit(v, vb).compress() for v, vb in zip(commit_values if commit_values is not None else values, blindings)]
    tape.rekey(n, m, Vs)
    for j, (v, vb) in enumerate(zip(values, blindings)):
        V = Vs[j]
        a_bl = tape.scalar()
        A = a_bl * B_BLINDING
        for i in range(n):
            A = A + (G[j * n + i] if (v >> i) & 1 else -H[j * n + i])
        s_bl = tape.scalar()
        s_L = [tape.scalar() for _ in range(n)]
        s_R = [tape.scalar() for _ in range(n)]
        S = s_bl * B_BLINDING + _msm(s_L, G[j * n:(j + 1) * n]) + _msm(s_R, H[j * n:(j + 1) * n])
        parties.append(dict(v=v, vb=vb, V=V, a_bl=a_bl, s_bl=s_bl, s_L=s_L, s_R=s_R, A=A, S=S))
    # Dealer::receive_bit_commitments
    for p in parties:
        tr.append_point(b"V", p["V"])
    A = IDENTITY
    S = IDENTITY
    for p in parties:
        A = A + p["A"]
        S = S + p["S"]
    if A_offset:
        A = A + (A_offset % L) * BASEPOINT
    A_c, S_c = A.compress(), S.compress()
    tr.append_point(b"A", A_c)
    tr.append_point(b"S", S_c)
    y = tr.challenge_scalar(b"y")
    z = tr.challenge_scalar(b"z")
    if info is not None:
        info.update(y=y, z=z, V=Vs)
    # Party::apply_challenge_with_rng
    T1 = IDENTITY
    T2 = IDENTITY
    for j, p in enumerate(parties):
        offset_y = pow(y, j * n, L)
        offset_zz = z * z % L * pow(z, j, L) % L
        l0, l1, r0, r1 = [], [], [], []
        exp_y, exp_2 = offset_y, 1
        for i in range(n):
            a_L = (p["v"] >> i) & 1
            a_R = (a_L - 1) % L
            l0.append((a_L - z) % L)
            l1.append(p["s_L"][i])
            r0.append((exp_y * (a_R + z) + offset_zz * exp_2) % L)
            r1.append(exp_y * p["s_R"][i] % L)
            exp_y = exp_y * y % L
            exp_2 = exp_2 * 2 % L
        t0 = _ip(l0, r0)
        t2 = _ip(l1, r1)
        t1 = (_ip([a + b for a, b in zip(l0, l1)], [a + b for a, b in zip(r0, r1)]) - t0 - t2) % L
        p.update(l0=l0, l1=l1, r0=r0, r1=r1, t0=t0, t1=t1, t2=t2, offset_zz=offset_zz)
        p["t1_bl"] = tape.scalar()
        p["t2_bl"] = tape.scalar()
        T1 = T1 + pedersen_commit(t1, p["t1_bl"])
        T2 = T2 + pedersen_commit(t2, p["t2_bl"])
    T1_c, T2_c = T1.compress(), T2.compress()
    tr.append_point(b"T_1", T1_c)
    tr.append_point(b"T_2", T2_c)
    x = tr.challenge_scalar(b"x")
    assert x != 0
    # Party::apply_challenge + Dealer::assemble_shares
    t_x = tau_x = mu = 0
    l_vec, r_vec = [], []
    for p in parties:
        t_x += p["t0"] + p["t1"] * x + p["t2"] * x * x
        tau_x += p["offset_zz"] * p["vb"] + p["t1_bl"] * x + p["t2_bl"] * x * x
        mu += p["a_bl"] + p["s_bl"] * x
        l_vec += [(a + b * x) % L for a, b in zip(p["l0"], p["l1"])]
        r_vec += [(a + b * x) % L for a, b in zip(p["r0"], p["r1"])]
    t_x %= L
    tau_x %= L
    mu %= L
    tr.append_scalar(b"t_x", t_x)
    tr.append_scalar(b"t_x_blinding", tau_x)
    tr.append_scalar(b"e_blinding", mu)
    w = tr.challenge_scalar(b"w")
    Q = w * BASEPOINT
    y_inv = inv(y)
    H_factors = [pow(y_inv, i, L) for i in range(n * m)]
    ipp = _ipp_create(tr, Q, H_factors, list(G), list(H), l_vec, r_vec)
    return A_c + S_c + T1_c + T2_c + scalar_bytes(t_x) + scalar_bytes(tau_x) + scalar_bytes(mu) + ipp


def _ipp_create(tr, Q, H_factors, G, H, a, b):
    """InnerProductProof::create (G_factors all one)."""
    nn = len(G)
    tr.innerproduct_domain_sep(nn)
    H = [f * h for f, h in zip(H_factors, H)]  # first-round H' = y^-i H_i, folded in eagerly (same group elements)
    out = b""
    while nn != 1:
        nn //= 2
        a_L, a_R, b_L, b_R = a[:nn], a[nn:], b[:nn], b[nn:]
        G_L, G_R, H_L, H_R = G[:nn], G[nn:], H[:nn], H[nn:]
        c_L, c_R = _ip(a_L, b_R), _ip(a_R, b_L)
        Lp = (_msm(a_L, G_R) + _msm(b_R, H_L) + c_L * Q).compress()
        Rp = (_msm(a_R, G_L) + _msm(b_L, H_R) + c_R * Q).compress()
        out += Lp + Rp
        tr.append_point(b"L", Lp)
        tr.append_point(b"R", Rp)
        u = tr.challenge_scalar(b"u")
        u_inv = inv(u)
        a = [(x * u + u_inv * yv) % L for x, yv in zip(a_L, a_R)]
        b = [(x * u_inv + u * yv) % L for x, yv in zip(b_L, b_R)]
        G = [u_inv * gl + u * gr for gl, gr in zip(G_L, G_R)]
        H = [u * hl + u_inv * hr for hl, hr in zip(H_L, H_R)]
    return out + scalar_bytes(a[0]) + scalar_bytes(b[0])


def range_proof_size(n, m):
    return 32 * (9 + 2 * (n * m).bit_length() - 2)


def range_verify(proof: bytes, commitments, n, c=None, label=b""):
    """RangeProof::from_bytes + verify_multiple (src/range/mod.rs:83-119).  `c` = the verifier's random
    batching scalar (thread_rng in the crate); any non-zero value gives the same verdict for honest proofs."""
    m = len(commitments)
    if len(proof) % 32 or len(proof) < 7 * 32:
        return False
    nel = len(proof) // 32 - 7
    if nel < 2 or (nel - 2) % 2:
        return False
    lg = (nel - 2) // 2
    if lg >= 32 or (1 << lg) != n * m or n not in (8, 16, 32, 64) or m & (m - 1) or m == 0:
        return False
    A_c, S_c, T1_c, T2_c = (proof[32 * i:32 * i + 32] for i in range(4))
    t_x, tau_x, mu = (scalar_from_canonical(proof[32 * i:32 * i + 32]) for i in range(4, 7))
    Ls = [proof[32 * (7 + 2 * k):32 * (8 + 2 * k)] for k in range(lg)]
    Rs = [proof[32 * (8 + 2 * k):32 * (9 + 2 * k)] for k in range(lg)]
    a = scalar_from_canonical(proof[-64:-32])
    b = scalar_from_canonical(proof[-32:])
    if None in (t_x, tau_x, mu, a, b):
        return False
    G, H = bp_gens(n, m)
    tr = Transcript(label)
    tr.rangeproof_domain_sep(n, m)
    for V in commitments:
        tr.append_point(b"V", V)
    if not (tr.validate_and_append_point(b"A", A_c) and tr.validate_and_append_point(b"S", S_c)):
        return False
    y = tr.challenge_scalar(b"y")
    z = tr.challenge_scalar(b"z")
    zz = z * z % L
    if not (tr.validate_and_append_point(b"T_1", T1_c) and tr.validate_and_append_point(b"T_2", T2_c)):
        return False
    x = tr.challenge_scalar(b"x")
    tr.append_scalar(b"t_x", t_x)
    tr.append_scalar(b"t_x_blinding", tau_x)
    tr.append_scalar(b"e_blinding", mu)
    w = tr.challenge_scalar(b"w")
    if c is None:
        c = scalar_from_wide(hashlib.sha512(proof).digest()) or 1
    # InnerProductProof::verification_scalars
    tr.innerproduct_domain_sep(n * m)
    us = []
    for Lp, Rp in zip(Ls, Rs):
        if not (tr.validate_and_append_point(b"L", Lp) and tr.validate_and_append_point(b"R", Rp)):
            return False
        us.append(tr.challenge_scalar(b"u"))
    us_inv = [inv(u) for u in us]
    allinv = 1
    for ui in us_inv:
        allinv = allinv * ui % L
    u_sq = [u * u % L for u in us]
    u_inv_sq = [u * u % L for u in us_inv]
    nm = n * m
    s = [allinv]
    for i in range(1, nm):
        lg_i = i.bit_length() - 1
        s.append(s[i - (1 << lg_i)] * u_sq[(lg - 1) - lg_i] % L)
    y_inv = inv(y)
    pts = [decompress(pc) for pc in (A_c, S_c, T1_c, T2_c)] + [decompress(pc) for pc in Ls] + \
          [decompress(pc) for pc in Rs] + [decompress(V) for V in commitments]
    if any(pt is None for pt in pts):
        return False
    sum_y = sum(pow(y, i, L) for i in range(nm)) % L
    sum_2 = (2**n - 1) % L
    sum_z = sum(pow(z, j, L) for j in range(m)) % L
    delta = ((z - zz) * sum_y - z * zz % L * sum_2 % L * sum_z) % L
    scal = [1, x, c * x % L, c * x % L * x % L] + u_sq + u_inv_sq
    scal += [(pow(z, j, L) * c % L * zz) % L for j in range(m)]
    acc = _msm(scal, pts)
    acc = acc + ((-mu - c * tau_x) % L) * B_BLINDING + ((w * (t_x - a * b) + c * (delta - t_x)) % L) * BASEPOINT
    g_s = [(-z - a * s[i]) % L for i in range(nm)]
    h_s = []
    exp_y_inv = 1
    for i in range(nm):
        z_and_2 = pow(z, i // n, L) * pow(2, i % n, L) % L
        h_s.append((z + exp_y_inv * (zz * z_and_2 - b * s[nm - 1 - i])) % L)
        exp_y_inv = exp_y_inv * y_inv % L
    acc = acc + _msm(g_s, G) + _msm(h_s, H)
    return acc == IDENTITY


# ----------------------------------------------------------------------------------------------------------
# Range-proof policies (src/range/padding.rs, src/range/splitting.rs)
# ----------------------------------------------------------------------------------------------------------
BIT_SIZE = 64            # src/range/mod.rs:16
SINGLE_PROOF_BYTE_NUM = 672   # :18
PROOF_SIZE_BYTE_NUM, AGGREGATED_NUM_BYTE_NUM, INDIVIDUAL_NUM_BYTE_NUM = 8, 2, 8   # :19-21


def next_pow2(x):
    return 1 if x <= 1 else 1 << (x - 1).bit_length()


def policy_plan(policy, n_siblings, agg):
    """[(start, count, padded_m)] of aggregated proofs, then the start of the individual part."""
    if agg > n_siblings:
        raise IndexError("aggregation_factor > number of siblings (reference panics: padding.rs:95-98)")
    plan = []
    if policy == "padding":
        plan.append((0, agg, next_pow2(agg)))          # padding.rs:94-105
        pos = agg
    else:
        base, pos = next_pow2(agg), 0                  # splitting.rs:106-117
        while pos < agg:
            if agg & base:
                plan.append((pos, base, base))
                pos += base
            base >>= 1
    return plan, pos


def policy_prove(policy, values, blindings, agg, tape, n=BIT_SIZE):
    """R::generate_proof -> (aggregated proofs, individual proofs).  One RNG stream across all sub-proofs."""
    plan, pos = policy_plan(policy, len(values), agg)
    aggregated = []
    for start, cnt, mm in plan:
        vs = list(values[start:start + cnt]) + [0] * (mm - cnt)
        bs = list(blindings[start:start + cnt]) + [1] * (mm - cnt)   # padding.rs:100-103: (0, Scalar::one())
        aggregated.append(range_prove(vs, bs, n, tape))
    individual = [range_prove([values[i]], [blindings[i]], n, tape) for i in range(pos, len(values))]
    return aggregated, individual


def policy_verify(policy, aggregated, individual, commitments, n=BIT_SIZE):
    """R::verify (padding.rs:169-196, splitting.rs:181-210)."""
    agg = len(commitments) - len(individual)
    plan, pos = policy_plan(policy, len(commitments), agg)
    if len(plan) != len(aggregated):
        return False
    com_padding = B_BLINDING.compress()                 # commit(0, 1), padding.rs:176-177
    for (start, cnt, mm), pr in zip(plan, aggregated):
        cs = list(commitments[start:start + cnt]) + [com_padding] * (mm - cnt)
        if not range_verify(pr, cs, n):
            return False
    return all(range_verify(pr, [commitments[pos + i]], n) for i, pr in enumerate(individual))


def _be(x, nbytes):
    return x.to_bytes(nbytes, "big")     # smtree::utils::usize_to_bytes: believed big-endian (UNPINNED)


def policy_serialize(policy, aggregated, individual):
    out = b""
    if policy == "padding":               # padding.rs:38-54
        out += _be(len(aggregated[0]), PROOF_SIZE_BYTE_NUM) + aggregated[0]
    else:                                 # splitting.rs:36-59
        out += _be(len(aggregated), AGGREGATED_NUM_BYTE_NUM)
        for pr in aggregated:
            out += _be(len(pr), PROOF_SIZE_BYTE_NUM) + pr
    out += _be(len(individual), INDIVIDUAL_NUM_BYTE_NUM)
    for pr in individual:
        out += pr
    return out


# smtree 0.1.2 MerkleProof / TreeIndex framing [3P-memory; every width a keyword so a test can move it]:
#   batch_num || sibling_num || tree_height || path_1 .. path_k || (Com || Hash)_1 .. _S          (UNPINNED)
WIRE_DEFAULT = dict(big_endian=True, batch_num_bytes=8, sibling_num_bytes=8, tree_height_bytes=2, path_bytes_full=False)


def _wi(x, nbytes, cfg):
    return x.to_bytes(nbytes, "big" if cfg["big_endian"] else "little")


def merkle_proof_serialize(height, leaf_idxs, siblings, **kw):
    """MerkleProof<DapolNode>::serialize: siblings = [(C32, H32)] (DapolProofNode::serialize = Com || Hash, proof/node.rs:74-79);
    a path = the index bits MSB first, left-aligned (TreeIndex.pos), ceil(height / 8) bytes."""
    cfg = dict(WIRE_DEFAULT, **kw)
    out = _wi(len(leaf_idxs), cfg["batch_num_bytes"], cfg) + _wi(len(siblings), cfg["sibling_num_bytes"], cfg)
    out += _wi(height, cfg["tree_height_bytes"], cfg)
    pb = 32 if cfg["path_bytes_full"] else (height + 7) // 8
    for x in leaf_idxs:
        bits = "".join("1" if (x >> (height - 1 - b)) & 1 else "0" for b in range(height)).ljust(8 * pb, "0")
        out += bytes(int(bits[8 * i:8 * i + 8], 2) for i in range(pb))
    for C, Hh in siblings:
        out += bytes(C) + bytes(Hh)
    return out


def dapol_proof_serialize(policy, aggregated, individual, height, leaf_idxs, siblings, **kw):
    """DapolProof::serialize (src/proof/mod.rs:68-73): range_proof || merkle_path."""
    return policy_serialize(policy, aggregated, individual) + merkle_proof_serialize(height, leaf_idxs, siblings, **kw)


def policy_deserialize(policy, data, begin=0, single_size=SINGLE_PROOF_BYTE_NUM):
    """Returns (aggregated, individual, new_begin) or raises ValueError (DecodingError).  single_size is 672 in the
    reference (BIT_SIZE = 64 is hard-coded, src/range/mod.rs:16-18); other bit sizes are a build extension."""
    def take_int(nb):
        nonlocal begin
        if len(data) - begin < nb:
            raise ValueError("BytesNotEnough")
        v = int.from_bytes(data[begin:begin + nb], "big")
        begin += nb
        return v

    def take(nb):
        nonlocal begin
        if len(data) - begin < nb:
            raise ValueError("BytesNotEnough")
        v = data[begin:begin + nb]
        begin += nb
        return v

    aggregated = []
    n_agg = 1 if policy == "padding" else take_int(AGGREGATED_NUM_BYTE_NUM)
    for _ in range(n_agg):
        aggregated.append(take(take_int(PROOF_SIZE_BYTE_NUM)))
    individual = [take(single_size) for _ in range(take_int(INDIVIDUAL_NUM_BYTE_NUM))]
    return aggregated, individual, begin


# ----------------------------------------------------------------------------------------------------------
# DAPOL node algebra and the sparse Merkle layout (src/dapol/node.rs, smtree 0.1.2 build)
# ----------------------------------------------------------------------------------------------------------
class Node:
    __slots__ = ("v", "r", "com", "C", "H")

    def __init__(self, v, r, com, C, H):
        self.v, self.r, self.com, self.C, self.H = v, r, com, C, H


def node_new(v, r, dg="blake3"):
    """DapolNode::new (node.rs:29-45).  r may be an unreduced Scalar::from_bits value."""
    com = pedersen_commit(v, r)
    C = com.compress()
    return Node(v, r, com, C, digest(dg, C))


def node_merge(l, r, dg="blake3"):
    """Mergeable::merge (node.rs:64-80).  u64 value sum wraps like release-mode Rust."""
    com = l.com + r.com
    return Node((l.v + r.v) & (2**64 - 1), (l.r + r.r) % L, com, com.compress(), digest(dg, l.C, r.C, l.H, r.H))


def node_padding(pad_seed, height_from_leaves, index, dg="blake3"):
    """Paddable::padding (node.rs:86-88): new(0, random) with the random draw made positional."""
    return node_new(0, pad_blinding(pad_seed, height_from_leaves, index), dg)


class Tree:
    """levels[k] = {index_at_level: Node}, k = 0 (leaves) .. height (root).  pad[k] = set of padding indices."""

    def __init__(self, height, leaves, pad_seed, dg="blake3"):
        idxs = [i for i, _ in leaves]
        assert idxs == sorted(set(idxs)), "leaves must be sorted and unique (smtree panics otherwise)"
        assert all(0 <= i < (1 << height) for i in idxs)
        self.height, self.dg = height, dg
        self.levels = [dict(leaves)]
        self.pad = [set()]
        for k in range(height):
            cur, nxt = self.levels[k], {}
            for i in sorted(cur):
                if i in self.pad[k] or (i >> 1) in nxt:
                    continue
                sib = i ^ 1
                if sib not in cur:
                    cur[sib] = node_padding(pad_seed, k, sib, dg)
                    self.pad[k].add(sib)
                nxt[i >> 1] = node_merge(cur[i & ~1], cur[i | 1], dg)
            self.levels.append(nxt)
            self.pad.append(set())
        self.root = self.levels[height][0] if leaves else None

    def path_siblings(self, leaf_idx):
        """Siblings along the path of one leaf, root side first (smtree get_merkle_path_ref order; UNPINNED)."""
        if leaf_idx not in self.levels[0] or leaf_idx in self.pad[0]:
            return None
        sibs = [self.levels[k][(leaf_idx >> k) ^ 1] for k in range(self.height)]
        return sibs[::-1]

    def node_count(self):
        return sum(len(l) for l in self.levels)


def verify_path(root_C, root_H, leaf_C, leaf_H, leaf_idx, siblings, dg="blake3"):
    """MerkleProof::verify restated with DapolProofNode::merge (src/proof/node.rs:56-69).
    siblings: [(C, H)] root side first."""
    pt = decompress(leaf_C)
    if pt is None:
        return False
    C, Hh = leaf_C, leaf_H
    for k, (sC, sH) in enumerate(reversed(siblings)):
        sp = decompress(sC)
        if sp is None:
            return False
        if (leaf_idx >> k) & 1:
            Hh = digest(dg, sC, C, sH, Hh)
        else:
            Hh = digest(dg, C, sC, Hh, sH)
        pt = pt + sp
        C = pt.compress()
    return C == root_C and Hh == root_H


# ----------------------------------------------------------------------------------------------------------
# Leaf derivation (src/dapol/mod.rs:323-441)
# ----------------------------------------------------------------------------------------------------------
MAX_TREE_HEIGHT, MIN_SPARSITY, DIGEST_SIZE, MAX_INDEX_RETRIES = 64, 2, 32, 128   # mod.rs:26-29


class DapolError(Exception):
    pass


def shuffle_index(index_seed, height, taken, dg, max_tries=MAX_INDEX_RETRIES):
    for _ in range(max_tries):
        index_seed = digest(dg, index_seed)
        idx = int.from_bytes(index_seed[:8], "big") >> (64 - height)
        if idx not in taken:
            taken.add(idx)
            return idx
    return None


def build_leaf_nodes(liabilities, audit_seed, height, dg="blake3", max_tries=MAX_INDEX_RETRIES):
    """liabilities: [(internal_id bytes, external_id bytes, value)] -> (sorted [(idx, Node)], {iid: idx}).
    max_tries: MAX_INDEX_RETRIES (mod.rs:29); lowered only by tests that want to reach FailedToMapIndex (mod.rs:369-370), which
    the sparsity rule of Dapol::new keeps below probability 2^-128 per entity at 128 tries.  The exception carries the input
    position and the audit id of the liability the reference's loop stops at (`FailedToMapIndex(audit_id, MAX_INDEX_RETRIES)`)."""
    id_map, taken, out = {}, set(), []
    for pos, (iid, eid, value) in enumerate(liabilities):
        if iid in id_map:
            raise DapolError("DuplicatedInternalId")
        audit_id = digest(dg, audit_seed, iid)
        index_seed = digest(dg, audit_id, b"index_seed", eid)
        idx = shuffle_index(index_seed, height, taken, dg, max_tries)
        if idx is None:
            raise DapolError("FailedToMapIndex", pos, audit_id)
        blind = scalar_from_bits(digest(dg, audit_id, b"blind_seed", eid))
        id_map[iid] = idx
        out.append((idx, node_new(value, blind, dg)))
    out.sort(key=lambda t: t[0])
    return out, id_map


def dapol_new(liabilities, audit_seed, height, pad_seed, dg="blake3"):
    """Dapol::new (mod.rs:100-128) validation + build."""
    if DIGESTS[dg][1] != DIGEST_SIZE:
        raise DapolError("InvalidDigestSize")
    if height > MAX_TREE_HEIGHT:
        raise DapolError("TreeHeightTooBig")
    if 2**height < len(liabilities) * MIN_SPARSITY:
        raise DapolError("SparsityTooSmall")
    leaves, id_map = build_leaf_nodes(liabilities, audit_seed, height, dg)
    return Tree(height, leaves, pad_seed, dg), id_map


def batch_siblings(height, leaf_idxs):
    """Positions [(level, index)] of the siblings of a batched Merkle proof (smtree MerkleProof::new_batch): level by
    level from the root side, left to right within a level (UNPINNED -- smtree's order is not fixed by the reference
    repository; for one leaf this is path_siblings' order).  level 0 = leaves."""
    assert list(leaf_idxs) == sorted(set(leaf_idxs)) and leaf_idxs
    out = []
    for level in range(height - 1, -1, -1):
        anc = sorted({i >> level for i in leaf_idxs})
        out += [(level, x ^ 1) for x in anc if (x ^ 1) not in anc]
    return out


def batch_nonce_seed(nonce_seed, leaf_idxs):
    """Seed of a batch's nonce stream: the caller's seed chained through the leaf list (domain 4)."""
    s = nonce_seed
    for i, leaf in enumerate(leaf_idxs):
        s = seed_wide(s, 4, leaf, i)[:32]
    return s


def dapol_prove_batch(tree, leaf_idxs, policy, agg, nonce_seed, n=BIT_SIZE):
    """Dapol::generate_proof_batch (mod.rs:172-190): (sibling positions, siblings, aggregated, individual)."""
    if any(i not in tree.levels[0] or i in tree.pad[0] for i in leaf_idxs):
        return None
    pos = batch_siblings(tree.height, leaf_idxs)
    sibs = [tree.levels[level][index] for level, index in pos]
    seed = nonce_seed if len(leaf_idxs) == 1 else batch_nonce_seed(nonce_seed, leaf_idxs)
    tape = Tape(seed=seed, domain=DOMAIN_NONCE, stream_id=leaf_idxs[0])
    aggregated, individual = policy_prove(policy, [s.v for s in sibs], [s.r for s in sibs], agg, tape, n)
    return pos, sibs, aggregated, individual


def verify_batch_paths(root_C, root_H, height, leaves, siblings, dg="blake3"):
    """MerkleProof::verify_batch restated.  leaves: [(idx, C, H)] sorted; siblings: [(C, H)] in batch_siblings order."""
    pos = batch_siblings(height, [i for i, _, _ in leaves])
    if len(pos) != len(siblings):
        return False
    cur = {}
    for i, C, Hh in leaves:
        pt = decompress(C)
        if pt is None:
            return False
        cur[i] = (pt, C, Hh)
    sib_at = {}
    for (level, index), (C, Hh) in zip(pos, siblings):
        pt = decompress(C)
        if pt is None:
            return False
        sib_at[(level, index)] = (pt, C, Hh)
    for level in range(height):
        nxt = {}
        for x in sorted(cur):
            if (x >> 1) in nxt:
                continue
            other = cur.get(x ^ 1) or sib_at[(level, x ^ 1)]
            l, r = (cur[x], other) if not x & 1 else (other, cur[x])
            pt = l[0] + r[0]
            nxt[x >> 1] = (pt, pt.compress(), digest(dg, l[1], r[1], l[2], r[2]))
        cur = nxt
    return cur[0][1] == root_C and cur[0][2] == root_H


def dapol_prove(tree, leaf_idx, policy, agg, nonce_seed, n=BIT_SIZE):
    """Dapol::generate_proof (mod.rs:167-190) for one leaf: (siblings, aggregated, individual)."""
    sibs = tree.path_siblings(leaf_idx)
    if sibs is None:
        return None
    tape = Tape(seed=nonce_seed, domain=DOMAIN_NONCE, stream_id=leaf_idx)
    aggregated, individual = policy_prove(policy, [s.v for s in sibs], [s.r for s in sibs], agg, tape, n)
    return sibs, aggregated, individual
