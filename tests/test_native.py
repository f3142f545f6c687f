"""Native test programs written against the header API (tests/native/*.cu, built by
tests/native/Makefile -- __graft_entry__.build() does it -- and run here on the GPU):
behaviour the Python harness has no model for."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def run(name, marker, args=(), cwd=None):
    exe = os.path.join(NATIVE, name)
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", NATIVE, name], check=True, capture_output=True)
    proc = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300, cwd=cwd)
    assert proc.returncode == 0 and marker in proc.stdout, proc.stdout[-2000:] + proc.stderr[-2000:]


@pytest.mark.gpu
def test_wall_forces_known_answers():
    """wall_forces / link_wall_forces / xy_wall_relu_force (reference links.cuh:142-228):
    force on cells inside the wall's range, opposite force averaged on the wall node, repeat
    calls, links added first, float4 points."""
    run("test_walls", "ALL WALL TESTS PASSED")


@pytest.mark.gpu
def test_initial_condition_shapes():
    """regular_hexagon / regular_rectangle lattices, seeded random_sphere / disk / cuboid:
    spacing, bounds, reproducibility, n_0, cell count (reference inits.cuh:14-76,157-247)."""
    run("test_shapes", "ALL SHAPE TESTS PASSED")


@pytest.mark.gpu
def test_polarity_forces_on_the_device():
    """The numeric known answers of the reference's tests/test_polarity.cu evaluated in a
    kernel (device libm), and device against host evaluation of the same code."""
    run("test_polarity_device", "ALL DEVICE POLARITY TESTS PASSED")


@pytest.mark.gpu
def test_header_level_slab_solver():
    """Solution<Pt, Slab_grid_solver> (include/slab.cuh) as a model program uses it: communicator
    from the environment, slab_init / slab_setup / slab_use_rccl, take_step; an id-indexed functor
    with local index != global id; the callback transport.  One rank (one GPU per box)."""
    run("test_slab_solver", "ALL SLAB SOLVER TESTS PASSED")


@pytest.mark.gpu
def test_vtk_output_at_a_million_cells():
    """Vtk_output (include/vtk.cuh) at config 4's size: byte for byte what one stream insertion
    per number writes (with and without a mask), faster than that, read back by Vtk_input; and a
    model's output loop with the steps in a worker thread (examples/branching.cu:263-280): same
    frames, same final state as the serial loop."""
    run("test_vtk_speed", "ALL VTK SPEED TESTS PASSED")


@pytest.mark.gpu
def test_link_forces_with_cell_ids_beyond_2_to_24():
    """Links::link_forces (reference links.cuh:98-140) on cells whose ids share their low 24 bits
    or equal the former dead key 0xFFFFFF: segmented-sum path (32 key bits) and atomics path
    against a host evaluation."""
    run("test_links_big_ids", "ALL BIG-ID LINK TESTS PASSED")


@pytest.mark.gpu
def test_one_line_declares_a_functor_stateless():
    """YA_STATELESS(Pt, functor) next to a model's functor: Grid_solver at 10 000 cells and
    Tile_solver at 800 pick the several-lanes-per-cell kernels by themselves -- bit-identical
    positions, >= 1.4 x / 2 x faster steps, >= 1e8 cell-updates/s at BASELINE config 2's size."""
    run("test_stateless", "ALL STATELESS TESTS PASSED")


@pytest.mark.gpu
def test_three_parameter_solution_spelling():
    """Solution_n<Pt, n_max, Solver> (north_star's `Solution<Pt, n_max, Solver>`; a class template
    cannot be overloaded on the kind of its parameters, hence the name): default-constructed with the capacity
    as a template argument, the same object and the same steps as Solution<Pt, Solver>{n_max, ...}."""
    run("test_solution_n", "ALL SOLUTION_N TESTS PASSED")


@pytest.mark.gpu
def test_arrays_registered_once_are_kept_in_cube_order():
    """Solution::keep_in_cube_order(every, arrays...): one line in the model instead of a
    renumber call in its loop -- a Property read by id inside the functor, Links, a plain device array and
    cells appended between steps: bit for bit what renumbering by hand every third step gives."""
    run("test_keep_order", "ALL KEEP-ORDER TESTS PASSED")


@pytest.mark.gpu
def test_mesh_header_on_the_device(tmp_path):
    """include/mesh.cuh (reference mesh.cuh:1-462): nearest-partner distances in slices and wavefront shares
    against a host loop bit for bit (empty sets, one workgroup, 70 001 partners, float3 against Po_cell), the
    reference-named kernel at the reference's launch shape, shape comparison between Solutions and against a
    mesh, and a generated torus written, read back, copied, assigned and rotated."""
    from mesh_fixtures import write_torus
    write_torus(tmp_path / "torus.vtk")
    run("test_mesh_device", "ALL MESH TESTS PASSED", args=[str(tmp_path / "torus.vtk")], cwd=tmp_path)


@pytest.mark.gpu
def test_folding_updates_match_the_plain_pipeline():
    """Round 5's update kernels (the centre-of-mass fold inside them, the next stage's right-hand side zeroed for
    the generic forces): steps with and without generic forces in turn, cells appended between steps, Grid and
    Tile solvers -- bit for bit the plain pipeline's states."""
    run("test_folding", "ALL FOLDING TESTS PASSED")


@pytest.mark.gpu
def test_force_launch_trace(tmp_path):
    """tools/micro/force_trace.hip (the force kernel built with -DYA_BITS_TRACE: every workgroup
    stamps its start, end, CU and XCD) through tools/force_trace_summary.py: one workgroup per
    tile, each stamped once, workgroup b on XCD b mod 8 (what xcd_contiguous_tile assumes; on a
    repartitioned GPU this fails and the tile mapping only costs L2 locality, never results), no
    more workgroups resident per CU than its registers hold."""
    import json
    import sys
    exe = os.path.join(ROOT, "tools", "micro", "ab_bin", "force_trace")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "micro/ab_bin/force_trace"], check=True,
                       capture_output=True)
    def traced(*tail):
        stamps = tmp_path / "stamps.csv"
        with open(stamps, "w") as out:
            subprocess.run([exe, "200000", "3", *tail], stdout=out, check=True, timeout=300)
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "force_trace_summary.py"), str(stamps)],
                              capture_output=True, text=True, check=True)
        return json.loads(proc.stdout)

    s = traced("0")   # whole tiles only
    assert s["cells"] == 200000 and s["workgroups"] == 3125
    assert s["block_mod_8_is_the_xcd"]
    assert 1 <= s["resident_workgroups_per_cu_max"] <= 24
    assert s["mean_lifetime"] > 0 and sum(s["in_flight"]) > 0
    # the engine's choice at this size under the opt-in by-plane summation order: as many tiles split in two as fill the chip's 5120 wavefront slots in
    # the launch's one round -- every workgroup still stamped once, the halves of a tile on one XCD
    split = traced()
    assert 5100 <= split["workgroups"] <= 5128 and split["block_mod_8_is_the_xcd"]
    assert split["mean_lifetime"] < s["mean_lifetime"]
