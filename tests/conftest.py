"""Shared fixtures.

`oracle` = the CPU restatement (oracle/, test infrastructure) bound through the
same ctypes ABI as the product; built on demand with oracle/Makefile (g++ only).
`device` = the HIP engine (yalla_amd/libyalla_models.so); tests that use it are
marked `gpu` and run only on the MI355X box.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "liboracle_models.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_models.cpp", "yalla_host.hpp")]
    srcs += [os.path.join(ROOT, "yalla_amd", "csrc", f)
             for f in ("model_functors.h", "models_harness.inc")]
    srcs.append(os.path.join(ROOT, "include", "slab_logic.inc"))
    stale = (not os.path.exists(ORACLE_LIB) or
             any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs))
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return ORACLE_LIB


@pytest.fixture(scope="session")
def oracle():
    from yalla_amd import _ffi
    lib = _ffi.bind(build_oracle())
    assert lib.ya_models_is_device() == 0
    return lib


@pytest.fixture(scope="session")
def device():
    from yalla_amd import _ffi
    return _ffi.device_lib()  # raises if the HIP extension is missing
