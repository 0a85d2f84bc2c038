"""Known answers for BLAKE3 inputs longer than one 1024-byte chunk (the tree mode liability ids of any length go through,
/root/reference/src/dapol/mod.rs:347-349, 358-360), made with an implementation that is NOT this repository's: the BLAKE3
team's own C code as vendored into LLVM (llvm_blake3_hasher_* exported by libLLVM-15.so of this image).  Inputs follow the
official test-vector pattern, byte i = i mod 251, at the official lengths plus a few around the chunk / subtree boundaries.
Run here (the GPU box and the tests only read the JSON):  python tests/golden/gen_blake3_kat.py"""
import ctypes
import json
import os

LENGTHS = [0, 1, 2, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 3073, 4096, 4097, 5000, 5120, 5121, 6144, 6145,
           7168, 7169, 8192, 8193, 16384, 31744, 65536, 102400]


def main():
    L = ctypes.CDLL("/usr/lib/x86_64-linux-gnu/libLLVM-15.so")
    out = []
    for n in LENGTHS:
        data = bytes(i % 251 for i in range(n))
        st = ctypes.create_string_buffer(4096)            # sizeof(llvm_blake3_hasher) = 1,912
        L.llvm_blake3_hasher_init(st)
        L.llvm_blake3_hasher_update(st, data, ctypes.c_size_t(n))
        h = ctypes.create_string_buffer(32)
        L.llvm_blake3_hasher_finalize(st, h, ctypes.c_size_t(32))
        out.append({"len": n, "hash": h.raw.hex()})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "blake3_long.json"), "w") as f:
        json.dump({"pattern": "byte i = i mod 251", "source": "llvm_blake3_hasher (libLLVM-15.so, BLAKE3 team's C implementation)", "vectors": out}, f, indent=1)


if __name__ == "__main__":
    main()
