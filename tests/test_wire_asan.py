"""the wire parser under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on this pool).
tests/cpp/wire_asan.cpp is a host-only build of dapol_amd/csrc/host_wire.inc; see its header for the cases."""
import os
import subprocess

from conftest import ROOT


def test_wire_parser_under_asan():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "wire_asan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                    "-I", os.path.join(ROOT, "dapol_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "wire_asan.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "wire asan clean" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
