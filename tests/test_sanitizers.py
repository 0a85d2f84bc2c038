"""Sanitizer run of the product arithmetic on the CPU build (GPU sanitizers are not available on this pool): UBSan's
signed-integer-overflow check is what turns a violated limb bound (e.g. 19*g not fitting int32) into a hard failure."""
import os
import subprocess
import sys

from conftest import ROOT


def test_ubsan_host_build_of_device_arithmetic():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "host_shim_ubsan.so")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=signed-integer-overflow,shift", "-fno-sanitize-recover=all",
                    os.path.join(ROOT, "tests", "host_shim.cpp"), "-o", so], check=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ubsan_driver.py"), so], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ubsan clean" in r.stdout, r.stderr[-2000:]
