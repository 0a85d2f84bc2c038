"""One more independent pin for the oracle's ARITHMETIC: OpenSSL's Ed25519 (libcrypto.so.3 of this image, reached
through ctypes; CPU only) against oracle/pyref.py's field / group / scalar code.

  * 1,000 random seeds: EVP_PKEY_get_raw_public_key == Edwards encoding of clamp(SHA-512(seed)[:32]) * B computed with pyref.Point --
    full-width fixed-base multiplications against a mature implementation (RFC 9496's vectors hold sixteen small multiples);
  * 200 signatures MADE with pyref's arithmetic (R = r B, S = r + k a mod l with r, k reduced from 64 bytes -- scalar_from_wide, the
    reduction every nonce of the prover goes through, src/range/mod.rs:48-62 via Scalar::random) and VERIFIED by OpenSSL, which
    recomputes S B - k A with its own double-scalar multiplication; a flipped bit is rejected.

It pins arithmetic, not labels or framing: DESIGN.md section 2's "parity unpinned" statement stands."""
import ctypes
import ctypes.util
import hashlib

import numpy as np
import pytest

EVP_PKEY_ED25519 = 1087        # NID_ED25519


def _crypto():
    name = ctypes.util.find_library("crypto")
    if not name:
        pytest.skip("no libcrypto on this machine")
    lib = ctypes.CDLL(name)
    for fn in ("EVP_PKEY_new_raw_private_key", "EVP_PKEY_new_raw_public_key", "EVP_MD_CTX_new"):
        getattr(lib, fn).restype = ctypes.c_void_p
    lib.EVP_PKEY_new_raw_private_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.EVP_PKEY_new_raw_public_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.EVP_PKEY_get_raw_public_key.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.EVP_PKEY_free.argtypes = [ctypes.c_void_p]
    lib.EVP_MD_CTX_free.argtypes = [ctypes.c_void_p]
    lib.EVP_DigestVerifyInit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.EVP_DigestVerify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    return lib


def _edwards_bytes(R, pt):
    """RFC 8032 5.1.2: y with the sign of x in bit 255."""
    zi = pow(pt.Z, R.P - 2, R.P)
    x, y = pt.X * zi % R.P, pt.Y * zi % R.P
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def _clamped(seed):
    h = hashlib.sha512(seed).digest()
    a = int.from_bytes(h[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    return a, h[32:]


def test_public_keys_of_1000_seeds(pyref):
    lib, R = _crypto(), pyref
    rng = np.random.default_rng(2026)
    n = ctypes.c_size_t()
    for i in range(1000):
        seed = rng.integers(0, 256, size=32, dtype=np.uint8).tobytes()
        if i == 0:
            seed = bytes(32)
        if i == 1:
            seed = b"\xff" * 32
        pk = lib.EVP_PKEY_new_raw_private_key(EVP_PKEY_ED25519, None, seed, 32)
        assert pk
        out = ctypes.create_string_buffer(32)
        n.value = 32
        assert lib.EVP_PKEY_get_raw_public_key(pk, out, ctypes.byref(n)) == 1 and n.value == 32
        lib.EVP_PKEY_free(pk)
        a, _ = _clamped(seed)
        assert _edwards_bytes(R, a * R.BASEPOINT) == out.raw, i


def test_signatures_made_with_the_oracles_arithmetic_verify_in_openssl(pyref):
    lib, R = _crypto(), pyref
    rng = np.random.default_rng(7)
    for i in range(200):
        seed = rng.integers(0, 256, size=32, dtype=np.uint8).tobytes()
        msg = rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8).tobytes()
        a, prefix = _clamped(seed)
        A = _edwards_bytes(R, a * R.BASEPOINT)
        r = R.scalar_from_wide(hashlib.sha512(prefix + msg).digest())          # 512 bits reduced mod l
        Rb = _edwards_bytes(R, r * R.BASEPOINT)
        k = R.scalar_from_wide(hashlib.sha512(Rb + A + msg).digest())
        sig = Rb + R.scalar_bytes(r + k * a)
        pk = lib.EVP_PKEY_new_raw_public_key(EVP_PKEY_ED25519, None, A, 32)
        assert pk
        for tamper in (False, True):
            s = bytearray(sig)
            if tamper:
                s[int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
            md = lib.EVP_MD_CTX_new()
            assert lib.EVP_DigestVerifyInit(md, None, None, None, pk) == 1
            ok = lib.EVP_DigestVerify(md, bytes(s), 64, msg, len(msg))
            lib.EVP_MD_CTX_free(md)
            assert (ok == 1) == (not tamper), (i, tamper)
        lib.EVP_PKEY_free(pk)
