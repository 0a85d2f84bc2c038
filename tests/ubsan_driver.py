"""Runs the host-compiled device arithmetic (tests/host_shim.cpp built with UBSan: signed overflow + shifts, no recovery)
over the operations the kernels use; any limb-bound violation in fe.h / ge.h / sc.h aborts this process.
Invoked by tests/test_sanitizers.py:  python tests/ubsan_driver.py <path to the instrumented .so>"""
import ctypes
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import pyref as R  # noqa: E402

lib = ctypes.CDLL(sys.argv[1])
rnd = random.Random(5)
P = R.P


def buf(n):
    return ctypes.create_string_buffer(n)


pts = [rnd.randrange(R.L) * R.BASEPOINT for _ in range(20)]
for i in range(300):
    a, b = rnd.choice(pts), rnd.choice(pts)
    o = buf(32)
    lib.t_add_compressed(a.compress(), b.compress(), o)
    assert o.raw == (a + b).compress()
for it in range(8):
    k = rnd.randrange(2**255)
    o = buf(32)
    lib.t_basemul(k.to_bytes(32, "little"), o)
    assert o.raw == (k * R.BASEPOINT).compress()
    u = bytes(rnd.randrange(256) for _ in range(64))
    lib.t_from_uniform(u, o)
    assert o.raw == R.from_uniform_bytes(u).compress()
    lib.t_maddmul(R.B_BLINDING.compress(), k.to_bytes(32, "little"), it & 1, o)
    lib.t_decompress(bytes(rnd.randrange(256) for _ in range(31)) + b"\x00", o)
edge = [0, 1, P - 1, 2**255 - 20, 2**254]
for it in range(300):
    a = rnd.choice(edge) if it < 20 else rnd.randrange(P)
    b = rnd.randrange(P)
    for op in range(9):
        if op == 2 and a == 0:
            continue
        lib.t_fe_op(op, a.to_bytes(32, "little"), b.to_bytes(32, "little"), buf(32))
for it in range(100):
    a, b = rnd.randrange(2**256), rnd.randrange(R.L)
    for op in range(3):
        lib.t_sc_op(op, a.to_bytes(32, "little"), b.to_bytes(32, "little"), buf(32))
    for w in range(8, 17):
        nw = 255 // w + 1
        lib.t_sc_recode_w(w, nw, (a >> 1).to_bytes(32, "little"), (ctypes.c_int * nw)())
m = bytes(rnd.randrange(256) for _ in range(300))
for n in (0, 1, 63, 64, 65, 128, 300):
    lib.t_digest(0, m, n, buf(32))
    lib.t_digest(1, m, n, buf(32))
lib.t_merlin(b"test protocol", 13, b"some label", 10, b"some data", 9, b"challenge", 9, buf(64))
print("ubsan clean")
