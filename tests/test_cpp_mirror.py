"""include/dapol.hpp (C++ host-side mirror of the reference's public types) compiled against libdapol_hip.so."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _build(hip_lib):
    hip_lib.lib()
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "dapol_hpp_smoke")
    libdir = os.path.join(ROOT, "dapol_amd")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "dapol_hpp_smoke.cpp"),
                    "-L", libdir, "-ldapol_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu(hip_lib):
    import torch
    exe = _build(hip_lib)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    if not torch.cuda.is_available():
        assert r.stdout.startswith("NO_DEVICE")


@pytest.mark.gpu
def test_cpp_mirror_matches_c_abi(hip_lib, gpu_ctx):
    exe = _build(hip_lib)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK proof_bytes=" in r.stdout, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("ROOT ")][0].split()
    bl = np.zeros((3, 32), np.uint8)
    for i in range(3):
        bl[i, 0], bl[i, 5] = i + 1, 0x77
    tr = hip_lib.Tree(gpu_ctx, 6, [3, 9, 40], [5, 7, 11], bl, bytes(range(32)))
    C, H, v, _ = tr.root()
    assert (line[1], line[2], int(line[3])) == (C.hex(), H.hex(), v)
