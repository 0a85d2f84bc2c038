import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The library reads its DAPOL_* measurement knobs only in a process that has opted in (include/dapol_hip.h, dapol_options): the
# tests do -- they drive every strategy through those knobs and compare bytes -- and so do the child processes they start.
os.environ.setdefault("DAPOL_ENV_KNOBS", "1")
os.environ.setdefault("DAPOL_TEST_HOOKS", "1")        # the fault-injection / limit-override knobs need this second opt-in (dapol_hip.hip: test_knob)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pyref():
    import pyref as R
    return R


@pytest.fixture(scope="session")
def ref():
    """The C oracle (oracle/ref_dapol.c), built on demand.  Test infrastructure only."""
    from __graft_entry__ import build_oracle
    lib = ctypes.CDLL(build_oracle())
    lib.ref_tree_build.restype = ctypes.c_void_p
    lib.ref_tree_node_count.restype = ctypes.c_uint64
    lib.ref_range_proof_size.restype = ctypes.c_size_t
    return lib


@pytest.fixture(scope="session")
def host_shim():
    """Product device headers compiled for the host (tests/host_shim.cpp) -- arithmetic unit tests without a GPU."""
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "host_shim.so")
    src = os.path.join(ROOT, "tests", "host_shim.cpp")
    deps = [src] + [os.path.join(ROOT, "dapol_amd", "csrc", f) for f in ("fe.h", "ge.h", "sc.h", "hash.h", "consts.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", so], check=True)
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def hip_lib():
    """libdapol_hip.so built for gfx950 (cross-compiles without a GPU)."""
    from __graft_entry__ import build_hip
    build_hip()
    from dapol_amd import capi
    return capi


@pytest.fixture(scope="session")
def gpu_ctx(hip_lib):
    return hip_lib.Context(0, 32)
