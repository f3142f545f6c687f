"""A THIRD statement of the step itself: `Heun_solver::take_step` with its default friction, written in numpy
float64 straight from the reference (include/solvers.cuh:113-161 euler_step / heun_step / add_rhs, :226-277
take_step, :284-322 compute_tile, :430-463 compute_cube) -- not from include/solvers.cuh of this repository or
oracle/yalla_host.hpp.  tests/test_model_functors_independent.py does this for the model functors with the
velocities at zero; here the velocities are what is looked at: several steps of springs under
friction_w_neighbour, so that every stage averages the neighbours' old velocities, with the centre of mass,
one point, or one point's xy and the centre's z held (the reference's three fix modes, the third with its
quirk: the centre's z is used by the FIRST stage only, the second holds the whole point).  All pairs for
Tile_solver, pairs below the cut-off for Grid_solver.  One case has `link_forces` as the generic force
(include/links.cuh:98-140: -strength r / dist on a, + on b, links with a == b skipped, a pair linked twice pulled
twice).  Both backends must agree with it to binary32 rounding."""
import numpy as np
import pytest

from yalla_amd.solution import Solution

L_0 = 0.5  # examples/springs.cu:8


def spring(r, dist, i, j):  # examples/springs.cu:14-21
    if i == j:
        return np.zeros(3)
    return r * (L_0 - dist) / dist


def friction_w_neighbour(dist, i, j):  # solvers.cuh:26-34
    if i == j:
        return 0.0
    return 1.0 if dist < 1 else 0.0


def link_forces(X, links, strength):  # links.cuh:98-125
    dX = np.zeros_like(X)
    for a, b in links:
        if a == b:
            continue
        r = X[a] - X[b]
        dist = np.sqrt(r.dot(r))
        dX[a] -= strength * r / dist
        dX[b] += strength * r / dist
    return dX


def right_hand_side(X, old_v, cut_off, links=None):
    """gen_forces, pwints, add_rhs: dX = generic forces + sum of pair forces + the friction-weighted mean of the
    neighbours' old velocities."""
    n = len(X)
    dX = np.zeros_like(X)
    gen = link_forces(X, *links) if links is not None else dX  # solvers.cuh:235, :261: before pwints, into d_dX
    for i in range(n):
        F, sum_v, sum_friction = np.zeros(3), np.zeros(3), 0.0
        for j in range(n):
            r = X[i] - X[j]
            dist = np.sqrt(r.dot(r))
            if cut_off is not None and dist >= cut_off:  # compute_cube :450; compute_tile has no such line
                continue
            F += spring(r, dist, i, j)
            friction = friction_w_neighbour(dist, i, j)
            sum_friction += friction
            sum_v += friction * old_v[j]
        dX[i] = gen[i] + F
        if sum_friction > 0:  # add_rhs :146-161
            dX[i] += sum_v / sum_friction
    return dX


def take_step(X, old_v, dt, cut_off, fix, links=None):
    """solvers.cuh:226-277.  fix = ("com",) | ("point", id) | ("point_xy", id)."""
    dX = right_hand_side(X, old_v, cut_off, links)
    if fix[0] == "com":
        fix_dX = dX.mean(axis=0)
    elif fix[0] == "point":
        fix_dX = dX[fix[1]].copy()
    else:  # set_fixed_xy :203-208, take_step :241-250: the centre of mass, x and y overwritten by the point's
        fix_dX = dX.mean(axis=0)
        fix_dX[:2] = dX[fix[1], :2]
    dX = dX - fix_dX  # euler_step :113-125
    X1 = X + dX * dt
    dX1 = right_hand_side(X1, old_v, cut_off, links)
    if fix[0] == "com":
        fix_dX1 = dX1.mean(axis=0)
    else:  # :266-272: `if (fix_com)` -- false after set_fixed_xy too, so the SECOND stage holds the point
        fix_dX1 = dX1[fix[1]].copy()
    dX1 = dX1 - fix_dX1  # heun_step :127-144
    return X + (dX + dX1) * 0.5 * dt, (dX + dX1) * 0.5


def start(n, seed, radius):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, 3))
    X *= radius * rng.random(n)[:, None] ** (1 / 3) / np.linalg.norm(X, axis=1)[:, None]
    return X.astype(np.float32)


def run_backend(lib, model, X0, dt, steps, fix, links=None):
    n = len(X0)
    with Solution(model, n, 16, 1.0, lib=lib) as s:
        s.h_X[:n] = X0
        s.h_n = n
        s.copy_to_device()
        if links is not None:
            s.set_links(np.asarray(links[0], np.int32), links[1])
        if fix[0] == "point":
            s.set_fixed(fix[1])
        elif fix[0] == "point_xy":
            s.set_fixed_xy(fix[1])
        for _ in range(steps):
            s.take_step(dt)
        return s.positions(), s.old_v()[:n]


CASES = [("springs_tile", None, ("com",)), ("springs_grid", 1.0, ("com",)),
         ("springs_grid", 1.0, ("point", 7)), ("springs_tile", None, ("point_xy", 3)),
         ("springs_grid", 1.0, ("point_xy", 11)), ("springs_links_grid", 1.0, ("com",))]


def check(lib, model, cut_off, fix):
    n, dt, steps = 90, 0.05, 4
    # all pairs: every cell pulls on every other, a loose cloud; grid: ~13 neighbours inside the cut-off
    X0 = start(n, 5, 1.4 if cut_off is None else 1.8)
    links = None
    if "links" in model:  # far pairs, a cell with several links, one pair twice, one link from a cell to itself
        rng = np.random.default_rng(9)
        pairs = [(int(a), int(b)) for a, b in rng.integers(0, n, size=(60, 2))] + [(4, 50), (4, 50), (4, 61), (8, 8)]
        links = (pairs, 0.3)
    X, v = X0.astype(np.float64), np.zeros((n, 3))
    for _ in range(steps):
        X, v = take_step(X, v, dt, cut_off, fix, links)
    assert np.abs(v).max() > 0.05 and np.abs(X - X0).max() > 0.02, "hardly anything moved: the case checks nothing"
    if links is not None:
        Xn, _ = take_step(X0.astype(np.float64), np.zeros((n, 3)), dt, cut_off, fix)
        Xl, _ = take_step(X0.astype(np.float64), np.zeros((n, 3)), dt, cut_off, fix, links)
        assert np.abs(Xl - Xn).max() > 5e-3, "the links hardly pulled: the case checks nothing"
    Xb, vb = run_backend(lib, model, X0, dt, steps, fix, links)
    if fix[0] != "com":  # the held point: x, y (and z) where they started
        held = slice(0, 2) if fix[0] == "point_xy" else slice(0, 3)
        assert np.abs(Xb[fix[1], held] - X0[fix[1], held]).max() <= 1e-6
    # (a pair within rounding of the grid's cut-off would differ by dt * 0.5; none in these cases, seeds pinned)
    assert np.abs(Xb - X).max() <= 2e-6 * np.abs(X).max(), np.abs(Xb - X).max()
    assert np.abs(vb - v).max() <= 2e-5 * np.abs(v).max(), np.abs(vb - v).max()


@pytest.mark.parametrize("model,cut_off,fix", CASES)
def test_oracle_against_the_independent_step(oracle, model, cut_off, fix):
    check(oracle, model, cut_off, fix)


@pytest.mark.gpu
@pytest.mark.parametrize("model,cut_off,fix", CASES)
def test_device_against_the_independent_step(device, model, cut_off, fix):
    check(device, model, cut_off, fix)
