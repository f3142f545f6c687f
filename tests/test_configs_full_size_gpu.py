"""Every BASELINE.json configuration at its STATED size on the device (the oracle needs
seconds per step there), through size-independent properties: finite state, conserved
centre of mass, a consistent grid (sorted keys, a permutation, ascending ids inside a
cube, bounds that match the counts), cube ids recomputed in numpy bit for bit, integer
results (cell counts, neighbour counters) that repeat exactly, and the model's own
signature (sorting sorts, growth grows, the epithelium stays outside).  The bit-exact
comparisons against the oracle run at oracle-affordable sizes in test_parity_gpu.py,
test_growth.py and tests/fuzz_parity.py."""
import numpy as np
import pytest

import branching_case
import growth_case
from yalla_amd.solution import Solution

pytestmark = pytest.mark.gpu


def check_grid(cube_id, point_id, start, end, n, gs):
    cube_id, point_id = cube_id[:n], point_id[:n]
    assert (np.diff(cube_id) >= 0).all(), "keys not sorted"
    assert np.array_equal(np.sort(point_id), np.arange(n, dtype=np.int32)), "not a permutation"
    same_cube = cube_id[1:] == cube_id[:-1]
    assert (point_id[1:][same_cube] > point_id[:-1][same_cube]).all(), "ids not ascending in a cube"
    counts = np.bincount(cube_id, minlength=gs ** 3)
    occupied = counts > 0
    assert (start[~occupied] == -1).all() and (end[~occupied] == -2).all()
    assert np.array_equal(end[occupied] - start[occupied] + 1, counts[occupied])
    assert np.array_equal(start[occupied], (np.cumsum(counts) - counts)[occupied])


def numpy_cube_ids(X, gs, cs=1.0):
    """solvers.cuh:357-360 in binary32, left to right."""
    f = np.float32
    return ((np.floor(X[:, 0] / f(cs)) + f(gs // 2)) + (np.floor(X[:, 1] / f(cs)) + f(gs // 2)) * f(gs)
            + ((np.floor(X[:, 2] / f(cs)) + f(gs // 2)) * f(gs)) * f(gs)).astype(np.int32)


def nearest_neighbour_distance_by_type(X, n):
    """Mean nearest-neighbour distance of the strongly (ids < n/2, sorting.cu:24) and of the
    weakly adhering cells."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(X).query(X, k=2)
    strong = np.arange(n) < n // 2
    return float(d[strong, 1].mean()), float(d[~strong, 1].mean())


def test_config2_sorting_10k_cells_300_steps(device):
    """examples/sorting.cu scaled to 10 000 two-type cells, dt 0.05, 300 steps."""
    n, gs = 10_000, 50
    runs = []
    for _ in range(2):
        with Solution("sorting_grid", n, gs, 1.0, lib=device) as s:
            s.random_sphere(0.5, 42)
            s.set_param("n_cells", n)
            X0 = s.h_X[:n].copy()
            s.take_step(0.05, 300)
            runs.append((s.positions(), s.old_v(), s.grid()))
    (X, v, grid), (X2, v2, _) = runs
    assert np.array_equal(X.view(np.uint32), X2.view(np.uint32)), "not repeatable"
    assert np.array_equal(v.view(np.uint32), v2.view(np.uint32))
    assert np.isfinite(X).all() and np.isfinite(v).all()
    com0, com1 = X0.astype(np.float64).mean(axis=0), X.astype(np.float64).mean(axis=0)
    assert np.abs(com1 - com0).max() <= 1e-4 * np.abs(X0).max()
    check_grid(*grid, n, gs)
    # differential adhesion (sorting.cu:9-28) multiplies the whole pair force, the short-range
    # repulsion included, by 9 / 3 / 1 for strong-strong / mixed / weak-weak pairs: the two
    # types start equally spaced and the strong cells end up visibly further apart
    strong0, weak0 = nearest_neighbour_distance_by_type(X0, n)
    strong1, weak1 = nearest_neighbour_distance_by_type(X, n)
    assert abs(strong0 - weak0) < 0.03 * weak0
    assert strong1 > 1.05 * weak1, "the two cell types did not differentiate"


def test_config3_branching_at_100k_cells(device):
    """examples/branching.cu's model on a 100 000-cell snapshot: one output frame (11 steps of
    dt 0.2) with division frozen, then division switched on."""
    n_0, n_max, gs = 100_000, 140_000, 100
    counts_runs, frozen = [], []
    for _ in range(2):
        s, nbs = branching_case.setup(device, n_0=n_0, n_max=n_max)
        types = s.get_prop("type", n_0)
        assert 0.02 * n_0 < types.sum() < 0.5 * n_0, "expected an epithelial shell around a mesenchyme"
        s.set_param("prolif_rate", 0.0)
        s.take_step(0.2, 11)
        assert s.get_d_n() == n_0
        X = s.positions()
        frozen.append((X.copy(), s.get_prop("mes_nbs", n_0), s.get_prop("epi_nbs", n_0)))
        assert np.isfinite(X).all()
        r = np.linalg.norm(X[:, :3] - X[:, :3].mean(axis=0), axis=1)
        assert r[types == 1].mean() > r[types == 0].mean()
        assert (X[types == 0, 5] == 0).all(), "u only lives on the epithelium"
        check_grid(*s.grid(), n_0, gs)
        s.set_param("prolif_rate", 1.0)
        counts = []
        for _ in range(6):
            s.take_step(0.2)
            counts.append(s.get_d_n())
        assert counts == sorted(counts) and n_0 < counts[-1] <= n_max
        assert np.isfinite(s.positions()).all()
        counts_runs.append(counts)
        s.close()
    # integer results repeat exactly (counters are integer atomics, division is seeded)
    assert counts_runs[0] == counts_runs[1]
    assert np.array_equal(frozen[0][1], frozen[1][1]) and np.array_equal(frozen[0][2], frozen[1][2])
    # every neighbour is counted once per stage: two stages per step (solvers.cuh:236,262)
    total = frozen[0][1] + frozen[0][2]
    assert total.min() >= 0 and 10 < total.mean() < 60


def test_config4_passive_growth_to_a_million_cells(device):
    """examples/passive_growth.cu scaled: 200 cells grow past 10^6 (dynamic n, Po_cell,
    bending_force, per-cell counters updated inside the functor)."""
    target, rate = 1_000_000, 0.03
    n_max = int(target * 1.3)
    gs = 2 * (int((target / 0.64) ** (1 / 3) * 0.75 / 2 * 1.25) + 4)
    seed_state, _ = growth_case.setup(device, "grid", 200, 400)
    X200, types200 = seed_state.positions(), seed_state.get_prop("type", 200)
    seed_state.close()
    trajectories = []
    for _ in range(2):
        with Solution("passive_growth_grid", n_max, gs, 1.0, lib=device) as s:
            s.h_n = 200
            s.h_X[:200] = X200
            s.copy_to_device()
            s.set_prop("type", np.concatenate([types200, np.zeros(n_max - 200, np.int32)]))
            s.set_param("prolif_rate", rate)
            s.set_param("seed", 7)
            counts = [200]
            while counts[-1] < target:
                s.take_step(0.2, 10)
                counts.append(s.get_d_n())
                assert len(counts) < 200, "growth stalled"
            n = counts[-1]
            assert counts == sorted(counts) and n <= n_max
            X = s.positions()
            assert np.isfinite(X).all()
            types = s.get_prop("type", n)
            assert set(np.unique(types)) <= {0, 1} and 0 < types.sum() < n
            r = np.linalg.norm(X[:, :3] - X[:, :3].mean(axis=0), axis=1)
            assert r[types == 1].mean() > r[types == 0].mean(), "the epithelium should stay outside"
            s.set_param("prolif_rate", 0.0)   # one more step without division: its grid holds all n cells
            s.take_step(0.2)
            assert s.get_d_n() == n
            check_grid(*s.grid(), n, gs)
            trajectories.append(counts)
    assert trajectories[0] == trajectories[1], "cell counts must repeat exactly for one seed"


def test_config5_ten_million_cells_on_one_gpu(device):
    """north_star's 10 M-cell system undivided on one MI355X (the 1-GPU point of the
    scaling curve): 2 take_steps, then the invariants and the cube ids in numpy."""
    n, gs = 10_000_000, 130
    with Solution("springs_grid", n, gs, 1.0, lib=device) as s:
        s.random_sphere(0.5, 42)
        X0 = s.h_X[:n].copy()
        cube_id, point_id, _, _ = s.build_grid(gs, 1.0)
        ids = numpy_cube_ids(X0, gs)
        order = np.argsort(ids, kind="stable").astype(np.int32)
        assert np.array_equal(point_id, order)
        assert np.array_equal(cube_id, ids[order])
        del ids, order, cube_id, point_id
        s.take_step(0.001, 2)
        X, v = s.positions(), s.old_v()
        assert np.isfinite(X).all() and np.isfinite(v).all()
        com0, com1 = X0.astype(np.float64).mean(axis=0), X.astype(np.float64).mean(axis=0)
        assert np.abs(com1 - com0).max() <= 1e-5 * np.abs(X0).max()
        step = np.linalg.norm(X - X0, axis=1)
        assert 0 < step.max() < 0.5
        check_grid(*s.grid(), n, gs)


def test_largest_system_the_cube_ids_allow(device):
    """75 M cells: random_sphere(0.5) of radius 122 fills the largest grid binary32 cube ids can
    address exactly (grid_size <= 256, YA_MAX_GRID_SIZE) -- byte offsets pass 2^31, the grid
    holds 1.6e7 cubes.  One take_step, then the size-independent invariants."""
    n, gs = 75_000_000, 250
    with Solution("springs_grid", n, gs, 1.0, lib=device) as s:
        s.random_sphere(0.5, 42)
        com0 = s.h_X[:n].astype(np.float64).mean(axis=0)
        reach0 = np.abs(s.h_X[:n]).max()
        assert reach0 < gs // 2 - 2
        s.take_step(0.001, 1)
        assert s.get_d_n() == n
        X = s.positions()
        assert np.isfinite(X).all()
        assert np.abs(X.astype(np.float64).mean(axis=0) - com0).max() <= 1e-5 * reach0
        cube_id, point_id, start, end = s.grid()
        cube_id = cube_id[:n]
        assert (np.diff(cube_id) >= 0).all(), "keys not sorted"
        counts = np.bincount(cube_id, minlength=gs ** 3)
        assert counts.sum() == n and 5 < counts[counts > 0].mean() < 15
        occupied = counts > 0
        assert np.array_equal(end[:gs ** 3][occupied] - start[:gs ** 3][occupied] + 1, counts[occupied])
