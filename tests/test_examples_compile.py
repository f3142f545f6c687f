"""The reference's model files (examples/*.cu), UNCHANGED, must compile against
this repo's headers for gfx950 (north_star: "every examples/*.cu model compiles
against it unchanged").  Runs only where the reference checkout exists (this
container).  A scratch tree of symlinks makes `#include "../include/x.cuh"`
resolve to this repo's include/; nothing is copied.

All 23 compile.  polarization.cu passes a whole Po_cell where the reference's own polarity.cuh wants a
Polarity (the file predates that struct); include/polarity.cuh keeps the older spelling as an
overload, pinned by the known answer in the reference's tests/test_polarity.cu:20-34.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/examples"
BASELINE_CONFIGS = ["springs", "sorting", "branching", "passive_growth"]
OTHERS = ["apical_constriction", "bending", "epithelia_double_polarity", "epithelium", "gradient",
          "growth_w_wall", "intercalation", "lineage_tracing", "migration", "random_walk",
          "sorting_prot", "turing", "turing_w_noise", "wnt", "write_vtk_w_mask",
          "intercalation_w_gradient", "model_features_sequential_addition", "teapot", "polarization"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="no reference checkout here")


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    base = tmp_path_factory.mktemp("compat")
    (base / "examples").mkdir()
    os.symlink(os.path.join(ROOT, "include"), base / "include")
    for f in os.listdir(REF):
        os.symlink(os.path.join(REF, f), base / "examples" / f)
    return base


def compile_model(tree, name):
    cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O1",
           "-Wno-error=parentheses", "-w",
           "-include", os.path.join(ROOT, "include", "compat", "cuda_names.h"),
           "-I" + os.path.join(ROOT, "include", "compat"),
           "-c", name + ".cu", "-o", str(tree / (name + ".o"))]
    proc = subprocess.run(cmd, cwd=tree / "examples", capture_output=True, text=True, timeout=900)
    return name, proc.returncode, proc.stderr[-3000:]


def compile_all(tree, names):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=6) as pool:   # hipcc runs as child processes
        results = list(pool.map(lambda n: compile_model(tree, n), names))
    failed = [(n, err) for n, rc, err in results if rc != 0]
    assert not failed, failed


def test_baseline_config_examples_compile_unchanged(tree):
    compile_all(tree, BASELINE_CONFIGS)


@pytest.mark.slow
def test_other_examples_compile_unchanged(tree):
    compile_all(tree, OTHERS)
