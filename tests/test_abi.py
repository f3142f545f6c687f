"""The C-ABI libraries load without a GPU and export every symbol their headers
declare (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ya_[A-Za-z0-9_]+)\s*\(", text)))


def built(path):
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    return path


def test_core_library_exports_every_declared_symbol():
    names = declared_functions("yalla_hip.h")
    assert len(names) >= 18
    lib = ctypes.CDLL(built(os.path.join(ROOT, "yalla_amd", "libyalla_hip.so")), mode=ctypes.RTLD_LOCAL)
    for name in names:
        assert hasattr(lib, name), name
    lib.ya_abi_version.restype = ctypes.c_int
    assert lib.ya_abi_version() == 10
    lib.ya_reduce_workspace_bytes.restype = ctypes.c_size_t
    assert lib.ya_reduce_workspace_bytes(3) == 1024 * 3 * 4


def test_models_library_exports_every_declared_symbol():
    names = declared_functions("yalla_models.h")
    lib = ctypes.CDLL(built(os.path.join(ROOT, "yalla_amd", "libyalla_models.so")), mode=ctypes.RTLD_LOCAL)
    for name in names:
        assert hasattr(lib, name), name
    from yalla_amd import _ffi
    assert set(names) == set(_ffi.MODELS_ABI), "ctypes table and header disagree"
    assert set(declared_functions("yalla_hip.h")) == set(_ffi.CORE_ABI)


def test_oracle_and_device_expose_the_same_models(oracle):
    from yalla_amd import _ffi, models
    dev = _ffi.bind(built(_ffi.DEVICE_LIB))
    assert dev.ya_models_is_device() == 1
    assert models(dev) == models(oracle)


def test_only_the_c_abi_is_exported():
    """-fvisibility=hidden: no C++ engine symbol may leak (they would interpose
    between the oracle and the device library when both are loaded)."""
    import subprocess
    for lib in ("yalla_amd/libyalla_hip.so", "yalla_amd/libyalla_models.so"):
        out = subprocess.run(["nm", "-D", "--defined-only", built(os.path.join(ROOT, lib))],
                             capture_output=True, text=True, check=True).stdout
        for line in out.splitlines():
            sym = line.split()[-1]
            kind = line.split()[-2]
            if sym.startswith("ya_") or sym.startswith("__hip") or kind in ("V", "D", "B", "R"):
                continue  # ABI, HIP fatbin registration, kernel stubs / data
            assert "harness" not in sym and "Solution" not in sym, sym


def strip_comments(name, text):
    if name.endswith(".py"):
        text = re.sub(r'"""(.|\n)*?"""', "", text)
        return re.sub(r"#.*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    return re.sub(r"(?m)^\s*#(?!\s*(include|if|ifdef|ifndef|define|else|endif|pragma|undef)).*", "", text)


def test_product_never_references_the_oracle():
    """Nothing under include/ or yalla_amd/ may include, link, load or call
    anything of oracle/ (comments aside; YA_ORACLE is the macro the oracle build
    defines when it compiles the shared model source)."""
    offenders = []
    for base in ("include", "yalla_amd"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc", ".o")):
                    continue
                code = strip_comments(f, open(os.path.join(dirpath, f), errors="ignore").read())
                code = code.replace("YA_ORACLE", "")
                if re.search(r"oracle|yalla_host", code, flags=re.I):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_grid_sizes_beyond_binary32_exactness_are_refused():
    """Cube ids are the reference's binary32 expression (solvers.cuh:357-360): exact only
    up to 256^3 cubes.  ya_grid_create refuses larger grids before it allocates anything
    (so this runs without a GPU) instead of silently merging neighbouring cubes."""
    lib = ctypes.CDLL(built(os.path.join(ROOT, "yalla_amd", "libyalla_hip.so")), mode=ctypes.RTLD_LOCAL)
    lib.ya_grid_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    handle = ctypes.c_void_p()
    for gs in (257, 300, 380, 1290):
        assert lib.ya_grid_create(1000, gs, ctypes.byref(handle)) == 1  # hipErrorInvalidValue
        assert not handle.value
    header = open(os.path.join(ROOT, "include", "yalla_hip.h")).read()
    assert "#define YA_MAX_GRID_SIZE 256" in header
