"""BASELINE.json's full size (1 M cells, springs, Grid_solver) through
size-independent properties: the oracle needs ~6 s per step here, so instead of a
bit comparison the run must conserve the centre of mass, leave a consistent grid
(sorted keys, a permutation, ascending ids inside a cube, bounds that match the
counts), match the cube ids recomputed in numpy bit for bit, and repeat bit for
bit (the engine has no run-to-run nondeterminism on this path)."""
import numpy as np
import pytest

from yalla_amd.solution import Solution

pytestmark = pytest.mark.gpu
N, GS = 1_000_000, 64


def run(device, steps):
    with Solution("springs_grid", N, GS, 1.0, lib=device) as s:
        s.random_sphere(0.5, 42)
        X0 = s.h_X[:N].copy()
        s.take_step(0.001, steps)
        return X0, s.positions(), s.old_v(), s.grid()


def test_million_cells_properties(device):
    X0, X, v, (cube_id, point_id, start, end) = run(device, 3)
    assert np.isfinite(X).all() and np.isfinite(v).all()
    radius = np.abs(X0).max()
    # the centre of mass is held fixed (solvers.cuh:241-243)
    com0 = X0.astype(np.float64).mean(axis=0)
    com1 = X.astype(np.float64).mean(axis=0)
    assert np.abs(com1 - com0).max() <= 1e-5 * radius
    # cells moved, but little (dt = 1e-3)
    step = np.linalg.norm(X - X0, axis=1)
    assert 0 < step.max() < 0.5
    # grid of the last build
    assert (np.diff(cube_id) >= 0).all(), "keys not sorted"
    assert np.array_equal(np.sort(point_id), np.arange(N, dtype=np.int32)), "not a permutation"
    same_cube = cube_id[1:] == cube_id[:-1]
    assert (point_id[1:][same_cube] > point_id[:-1][same_cube]).all(), "ids not ascending in a cube"
    counts = np.bincount(cube_id, minlength=GS ** 3)
    occupied = counts > 0
    assert (start[~occupied] == -1).all() and (end[~occupied] == -2).all()
    assert np.array_equal(end[occupied] - start[occupied] + 1, counts[occupied])
    assert np.array_equal(start[occupied], (np.cumsum(counts) - counts)[occupied])


def test_million_cells_cube_ids_match_numpy(device):
    """Build the public Grid on the initial state: ids recomputed in float32."""
    with Solution("springs_grid", N, GS, 1.0, lib=device) as s:
        s.random_sphere(0.5, 42)
        X0 = s.h_X[:N].copy()
        cube_id, point_id, _, _ = s.build_grid(GS, 1.0)
    f = np.float32
    ids = ((np.floor(X0[:, 0] / f(1)) + f(GS // 2)) + (np.floor(X0[:, 1] / f(1)) + f(GS // 2)) * f(GS)
           + ((np.floor(X0[:, 2] / f(1)) + f(GS // 2)) * f(GS)) * f(GS)).astype(np.int32)
    order = np.argsort(ids, kind="stable").astype(np.int32)
    assert np.array_equal(point_id, order)
    assert np.array_equal(cube_id, ids[order])


def test_million_cells_repeatable(device):
    a = run(device, 2)
    b = run(device, 2)
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
