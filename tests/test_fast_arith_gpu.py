"""The fast-arithmetic tier (libyalla_models_fast.so: -DYA_ARITH_FAST -ffp-contract=fast; bare
v_sqrt_f32 / v_rcp_f32, contracted multiply-adds -- include/solvers.cuh ya::exact_sqrt) against
the oracle, for the functor of every BASELINE.json configuration.

Tolerance: north_star's 1e-5 relative on fp32 positions, in LOCK-STEP (every step starts from the
oracle's state, as tests/test_growth.py does for libm-calling functors: a last-ulp difference in a
pair distance is amplified by the dynamics, and near the cut-off it decides whether a pair
interacts at all).  What must stay bit-exact does: cube ids, point ids, cube_start / cube_end
(libyalla_hip.so is shared with the exact tier and always built without contraction), cell
counts, integer per-cell properties.
"""
import numpy as np
import pytest

import branching_case
import growth_case
from yalla_amd import _ffi
from yalla_amd.solution import Solution

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


@pytest.fixture(scope="module")
def fast():
    lib = _ffi.device_lib("fast")
    assert lib.ya_models_arith() == 1 and lib.ya_models_is_device() == 1
    return lib


def lock_step(oracle, fast, model, n, gs, dist, seed, dt, steps, setup=None, cut_off_pairs=0):
    """Both backends from the same state, one step at a time; the device restarts every step from
    the oracle's state.  `cut_off_pairs`: how many cells per step may be moved by a pair within
    rounding of the cut-off (the spring force does not vanish there: such a pair shifts two cells
    by ~0.5 dt; clipped / relu forces are continuous at the cut-off and get no such budget)."""
    with Solution(model, n, gs, 1.0, lib=oracle) as so, Solution(model, n, gs, 1.0, lib=fast) as sd:
        so.set_reduce_order(1)
        for s in (so, sd):
            s.random_sphere(dist, seed)
            if setup:
                setup(s)
        worst = 0.0
        for step in range(steps):
            so.take_step(dt)
            sd.take_step(dt)
            Xo, Xd = so.positions(), sd.positions()
            scale = np.abs(Xo[:, :3]).max()
            diff = np.abs(Xo - Xd).max(axis=1)
            off = int((diff > REL_TOL * scale).sum())
            assert off <= cut_off_pairs, (model, step, off, diff.max())
            assert diff.max() <= 2.0 * dt * max(1.0, np.abs(so.old_v()).max()), (model, step, diff.max())
            worst = max(worst, float(np.sort(diff)[-1 - off] / scale) if off < len(diff) else 0.0)
            sd.h_X[:] = so.h_X
            sd.copy_to_device()
            sd.set_old_v(so.old_v())
        return worst


def test_grid_arrays_stay_bit_exact(oracle, fast):
    """Grid::build on identical positions: integer results identical in both tiers."""
    n = 20000
    with Solution("springs_grid", n, 64, 1.0, lib=oracle) as so, Solution("springs_grid", n, 64, 1.0, lib=fast) as sd:
        for s in (so, sd):
            s.random_sphere(0.5, 42)
        for a, b in zip(so.build_grid(64, 1.0), sd.build_grid(64, 1.0)):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("n,gs", [(800, 50), (20000, 64), (150000, 64)])
def test_config_5_springs_grid(oracle, fast, n, gs):
    """Headline functor (examples/springs.cu:14-21, cut off by the grid), random_sphere(0.5)."""
    worst = lock_step(oracle, fast, "springs_grid", n, gs, 0.5, 42, 0.001, 4, cut_off_pairs=max(4, n // 2000))
    assert worst <= REL_TOL


def test_config_1_springs_tile(oracle, fast):
    worst = lock_step(oracle, fast, "springs_tile", 800, 50, 0.5, 42, 0.001, 5)
    assert worst <= REL_TOL


def test_config_2_sorting(oracle, fast):
    """examples/sorting.cu: differential_adhesion (powf), 10 k two-type cells, dt 0.05."""
    n = 10000
    worst = lock_step(oracle, fast, "sorting_grid", n, 50, 0.5, 42, 0.05, 4,
                      setup=lambda s: s.set_param("n_cells", n))
    assert worst <= REL_TOL


@pytest.mark.parametrize("model", ["clipped_grid", "relu_grid", "relu_po_grid", "relu_cell_grid", "relu_tile"])
def test_other_functors(oracle, fast, model):
    worst = lock_step(oracle, fast, model, 3000, 50, 0.6, 5, 0.1, 4)
    assert worst <= REL_TOL


def restart_from(sd, so, n):
    sd.h_X[:] = so.h_X
    sd.h_n = n
    sd.copy_to_device()
    sd.set_old_v(so.old_v())


def test_config_4_passive_growth(oracle, fast):
    """examples/passive_growth.cu: Po_cell, relu_w_epithelium + bending force, neighbour counters
    updated inside the functor, division: identical cell counts and counters, positions to 1e-5."""
    so, nbs_o = growth_case.setup(oracle)
    sd, nbs_d = growth_case.setup(fast)
    assert np.array_equal(so.get_prop("type", 200), sd.get_prop("type", 200))
    for s in (so, sd):
        s.set_param("prolif_rate", 0.05)
        s.set_param("seed", 77)
    restart_from(sd, so, 200)
    n_o = 200
    for step in range(10):
        so.take_step(0.2)
        sd.take_step(0.2)
        n_o, n_d = so.get_d_n(), sd.get_d_n()
        assert n_o == n_d, f"cell counts differ at step {step}"
        Xo, Xd = so.positions(), sd.positions()
        assert np.abs(Xo - Xd).max() <= REL_TOL * np.abs(Xo[:, :3]).max(), step
        for name in ("type", "mes_nbs", "epi_nbs"):
            assert np.array_equal(so.get_prop(name, n_o), sd.get_prop(name, n_d)), (name, step)
        restart_from(sd, so, n_o)
    assert n_o > 200
    so.close()
    sd.close()


def test_config_3_branching(oracle, fast):
    """examples/branching.cu: 7-float Cell, Turing kinetics, bending, atomicAdd counters (the
    exact tier's test is tests/test_growth.py::test_branching_device_matches_oracle_lockstep)."""
    so, nbs_o = branching_case.setup(oracle)
    sd, nbs_d = branching_case.setup(fast)
    assert np.array_equal(nbs_o, nbs_d)
    assert np.array_equal(so.get_prop("type", 500), sd.get_prop("type", 500))
    restart_from(sd, so, so.get_d_n())
    for step in range(8):
        so.take_step(0.2)
        sd.take_step(0.2)
        n_o, n_d = so.get_d_n(), sd.get_d_n()
        assert n_o == n_d, step
        Xo, Xd = so.positions(), sd.positions()
        for cols in (slice(0, 3), slice(3, 5), slice(5, 7)):
            scale = max(np.abs(Xo[:, cols]).max(), 1e-3)
            assert np.abs(Xo[:, cols] - Xd[:, cols]).max() <= REL_TOL * scale, (step, cols)
        for name in ("type", "mes_nbs", "epi_nbs"):
            assert np.array_equal(so.get_prop(name, n_o), sd.get_prop(name, n_d)), (name, step)
        restart_from(sd, so, n_o)
    so.close()
    sd.close()
