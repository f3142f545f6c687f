"""An INDEPENDENT binary32 statement of the reference's step, written in numpy from the reference alone -- first the
grid-force sum, then (further down) compute_tile's sum and whole Heun steps of Tile_solver and Grid_solver -- each held
BIT FOR BIT against the oracle (CPU) and against the HIP engine (GPU tests), no tolerance anywhere.

The grid-force sum
(include/solvers.cuh:430-463 `compute_cube`, :472-483 the stencil table, :349-365 the cube id; examples/springs.cu:14-21
the functor; include/dtypes.cuh:150-217 the operators) -- one running sum per cell over the 27 cubes in d_nhood
order, cells of a cube in ascending id -- held bit for bit against

  * the oracle's default (YA_SUM_REFERENCE) on the CPU, and its second mode (YA_SUM_BY_PLANE) against the same
    statement with the two partial sums;
  * the HIP engine's default and its opt-in order on the GPU.

How a force becomes observable bit for bit: one take_step with dt = 0 and set_fixed(p), p a lone cell far from all
others (its force is exactly 0).  Both stages then see the same positions and old_v = 0, the fixed velocity is
dX[p] = 0, and heun_step leaves old_v = ((F - 0) + (F - 0)) * 0.5 = F exactly (solvers.cuh:113-144).

Arithmetic as the oracle's header states it: r = Xi - Xj, dist = sqrtf(fmaf(z, z, fmaf(y, y, x * x))),
r * (0.5 - dist) componentwise, then `/ dist` = `* float(1. / dist)` (dtypes.cuh:202-208).  fmaf is emulated through
binary64 (the product of two binary32 numbers is exact there)."""
import numpy as np
import pytest

from yalla_amd.solution import Solution

f32 = np.float32
GS = 30


def fma32(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def system(n=260, seed=3):
    rng = np.random.default_rng(seed)
    X = (rng.random((n, 3)) * 4.6 - 2.3).astype(f32)          # ~2.7 cells per unit cube ... dense enough for
    X[: n // 3] = (rng.random((n // 3, 3)) * 2.0 - 1.0).astype(f32)   # ... ~12 in the middle
    X[-1] = (9.5, 8.5, 7.5)                                    # the lone cell that is held fixed
    return X


def reference_forces(X, by_plane, old_v=None, gs=GS, functor="spring"):
    """F[i] exactly as the reference's thread accumulates it (or in the two partial sums of YA_SUM_BY_PLANE).
    With old_v: the stage's whole right-hand side, F + sum_v / sum_friction (solvers.cuh:453-461 and add_rhs
    :146-161) -- sum_friction += friction and sum_v += friction * old_v[k] run in the same loop, the same order."""
    n = len(X)
    cube3 = np.floor(X).astype(np.int64) + gs // 2             # cube_size 1 (solvers.cuh:357-360; exact below 2^24)
    cube = cube3[:, 0] + cube3[:, 1] * gs + cube3[:, 2] * gs * gs
    members = {}
    for i in np.argsort(cube, kind="stable"):                  # stable sort: ascending id inside a cube
        members.setdefault(int(cube[i]), []).append(int(i))
    h = [-1, 0, 1]                                             # solvers.cuh:472-483
    h = h + [h[i % 3] - gs for i in range(3)] + [h[i % 3] + gs for i in range(3)]
    h = h + [h[i % 9] - gs * gs for i in range(9)] + [h[i % 9] + gs * gs for i in range(9)]
    F = np.zeros((n, 3), f32)
    for i in range(n):
        acc = np.zeros(3, f32)
        own = np.zeros(3, f32)
        sv, sv_own = np.zeros(3, f32), np.zeros(3, f32)
        sf, sf_own = f32(0), f32(0)
        for jn, off in enumerate(h):
            if by_plane and jn == 9:
                own, acc = acc, np.zeros(3, f32)
                sv_own, sv = sv, np.zeros(3, f32)
                sf_own, sf = sf, f32(0)
            for k in members.get(int(cube[i]) + off, ()):
                r = X[i] - X[k]                                 # a += -1 * b, componentwise binary32
                d2 = fma32(r[2], r[2], fma32(r[1], r[1], f32(r[0] * r[0])))
                dist = np.sqrt(d2)                              # correctly rounded binary32
                if dist >= f32(1.0):                            # :450
                    continue
                if k == i:                                      # springs.cu:17; friction_w_neighbour: 0 (solvers.cuh:30)
                    continue
                if old_v is not None:                           # friction_w_neighbour: 1 inside dist < 1 (:32)
                    sf = sf + f32(1)
                    sv = sv + f32(1) * old_v[k]
                if functor == "spring":
                    s = f32(0.5) - dist
                    inv = f32(np.float64(1.0) / np.float64(dist))   # `a *= 1. / b` (dtypes.cuh:204-208)
                    acc = acc + (r * s) * inv
                else:   # relu_force, inits.cuh:78-93: scalar float arithmetic, a true division per component
                    Fm = np.maximum(f32(0.8) - dist, f32(0)) * f32(2) - np.maximum(dist - f32(0.8), f32(0))
                    acc = acc + (r * Fm) / dist
        F[i] = own + acc if by_plane else acc
        if old_v is not None:
            if by_plane:
                sv, sf = sv_own + sv, sf_own + sf
            if sf > 0:                                          # add_rhs, :155-159
                F[i] = F[i] + sv / sf
    return F


def forces_of(lib, X, sum_order):
    n = len(X)
    with Solution("springs_grid", n, GS, 1.0, lib=lib) as s:
        s.h_X[:n] = X
        s.h_n = n
        s.copy_to_device()
        if sum_order:
            s.set_param("sum_order", sum_order)
        s.set_fixed(n - 1)
        s.take_step(0.0, 1)
        assert np.array_equal(s.positions().view(np.uint32), X.view(np.uint32))   # dt = 0: nothing moved
        return s.old_v()[:n].copy()


@pytest.mark.parametrize("sum_order", [0, 1])
def test_oracle_sums_as_the_reference_thread_does(oracle, sum_order):
    X = system()
    F = reference_forces(X, by_plane=bool(sum_order))
    got = forces_of(oracle, X, sum_order)
    assert np.abs(F).max() > 1 and (F[-1] == 0).all()
    assert np.array_equal(F.view(np.uint32), got.view(np.uint32))


def rhs_with_friction(lib, X, sum_order):
    """Two steps: a real one (dt = 0.05: positions move, old_v becomes the mean right-hand side), then one with
    dt = 0 whose old_v is the stage's right-hand side F + sum_v / sum_friction of the state in between."""
    n = len(X)
    with Solution("springs_grid", n, GS, 1.0, lib=lib) as s:
        s.h_X[:n] = X
        s.h_n = n
        s.copy_to_device()
        if sum_order:
            s.set_param("sum_order", sum_order)
        s.set_fixed(n - 1)
        s.take_step(0.05, 1)
        X1, v1 = s.positions().copy(), s.old_v()[:n].copy()
        s.take_step(0.0, 1)
        return X1, v1, s.old_v()[:n].copy()


@pytest.mark.parametrize("sum_order", [0, 1])
def test_oracle_friction_sums_in_the_reference_order(oracle, sum_order):
    """The friction mean too: sum_v and sum_friction are accumulated in the same loop as F (solvers.cuh:453-458)."""
    X1, v1, got = rhs_with_friction(oracle, system(), sum_order)
    assert np.abs(v1).max() > 0.1 and (v1[-1] == 0).all()
    want = reference_forces(X1, bool(sum_order), old_v=v1)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("sum_order", [0, 1])
def test_engine_friction_sums_in_the_reference_order(device, sum_order):
    X1, v1, got = rhs_with_friction(device, system(), sum_order)
    want = reference_forces(X1, bool(sum_order), old_v=v1)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_the_two_orders_are_different_roundings_of_the_same_sum(oracle):
    X = system()
    a, b = reference_forces(X, False), reference_forces(X, True)
    assert not np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.abs(a - b).max() <= 4e-6 * np.abs(a).max()


@pytest.mark.gpu
@pytest.mark.parametrize("sum_order", [0, 1])
def test_engine_sums_as_the_reference_thread_does(device, sum_order):
    """The HIP engine's default order IS the reference's (and its opt-in order the by-plane statement), bit for bit
    against numpy -- no oracle in between."""
    X = system()
    F = reference_forces(X, by_plane=bool(sum_order))
    for variant in (-1, 2, 3, 0, 1):   # the engine's choice, bit stream, several lanes per cell, direct, byte FIFO
        n = len(X)
        with Solution("springs_grid", n, GS, 1.0, lib=device) as s:
            s.h_X[:n] = X
            s.h_n = n
            s.copy_to_device()
            s.set_param("force_variant", variant)
            if sum_order:
                s.set_param("sum_order", sum_order)
            s.set_fixed(n - 1)
            s.take_step(0.0, 1)
            got = s.old_v()[:n].copy()
        assert np.array_equal(F.view(np.uint32), got.view(np.uint32)), variant


# ---- Tile_solver: all pairs, j ascending (solvers.cuh:284-322) ---------------------------------------------------
def reference_tile_rhs(X):
    """compute_tile's thread: for j = 0 .. n - 1 in order, no cut-off, the functor called for every pair; then the
    fixed point's right-hand side subtracted (the lone cell is attracted by everybody here: its force is not 0)."""
    n = len(X)
    F = np.zeros((n, 3), f32)
    for i in range(n):
        acc = np.zeros(3, f32)
        for k in range(n):
            if k == i:
                continue
            r = X[i] - X[k]
            dist = np.sqrt(fma32(r[2], r[2], fma32(r[1], r[1], f32(r[0] * r[0]))))
            s = f32(0.5) - dist
            inv = f32(np.float64(1.0) / np.float64(dist))
            acc = acc + (r * s) * inv
        F[i] = acc
    return F - F[-1]          # euler_step / heun_step: dX.xyz -= fix.xyz, fix = dX[fix_point] (solvers.cuh:118-120,250-253)


def tile_rhs_of(lib, X, lanes=None):
    n = len(X)
    with Solution("springs_tile", n, GS, 1.0, lib=lib) as s:
        s.h_X[:n] = X
        s.h_n = n
        s.copy_to_device()
        if lanes is not None:
            s.set_param("tile_lanes", lanes)
        s.set_fixed(n - 1)
        s.take_step(0.0, 1)
        return s.old_v()[:n].copy()


def test_oracle_tile_sums_in_ascending_j(oracle):
    X = system(n=150)
    assert np.array_equal(reference_tile_rhs(X).view(np.uint32), tile_rhs_of(oracle, X).view(np.uint32))


@pytest.mark.gpu
def test_engine_tile_sums_in_ascending_j(device):
    """ya::tile_force (one thread per cell) and ya::tile_force_coop (16 / 64 lanes per cell): the reference's order."""
    X = system(n=150)
    want = reference_tile_rhs(X)
    for lanes in (1, 16, 64, 0):
        assert np.array_equal(want.view(np.uint32), tile_rhs_of(device, X, lanes).view(np.uint32)), lanes


# ---- the whole step, binary32, operation by operation (solvers.cuh:113-161, 226-275, 284-322) ----------------------
def reference_tile_steps(X, steps, dt, p):
    """Heun_solver::take_step with Tile_computer, friction_w_neighbour and set_fixed(p), in numpy binary32: the
    stage's right-hand side F + sum_v / sum_friction (compute_tile j ascending, add_rhs), fix = dX[p], euler_step,
    the second stage on X1 with the SAME old_v, heun_step.  Returns (X, old_v) after `steps` steps."""
    n = len(X)
    X = X.copy()
    old_v = np.zeros((n, 3), f32)
    dt = f32(dt)

    def rhs(Y):
        dX = np.zeros((n, 3), f32)
        for i in range(n):
            F, sv, sf = np.zeros(3, f32), np.zeros(3, f32), f32(0)
            for k in range(n):
                r = Y[i] - Y[k]
                dist = np.sqrt(fma32(r[2], r[2], fma32(r[1], r[1], f32(r[0] * r[0]))))
                if k != i:                                        # springs.cu:17 / solvers.cuh:30
                    F = F + (r * (f32(0.5) - dist)) * f32(np.float64(1.0) / np.float64(dist))
                    if dist < f32(1.0):                           # friction_w_neighbour: 1, else 0 (0 * v adds nothing)
                        sf = sf + f32(1)
                        sv = sv + old_v[k]
            dX[i] = F                                             # d_dX[i] += F on a zero-filled array
            if sf > 0:
                dX[i] = dX[i] + sv / sf                           # add_rhs
        return dX

    for _ in range(steps):
        dX = rhs(X)
        dX = dX - dX[p]                                           # euler_step: d_dX[i].xyz -= fix_dX.xyz
        X1 = X + dX * dt
        dX1 = rhs(X1)
        dX1 = dX1 - dX1[p]                                        # heun_step
        X = X + ((dX + dX1) * f32(0.5)) * dt
        old_v = (dX + dX1) * f32(0.5)
    return X, old_v


def tile_steps_of(lib, X, steps, dt, p, lanes=None):
    n = len(X)
    with Solution("springs_tile", n, GS, 1.0, lib=lib) as s:
        s.h_X[:n] = X
        s.h_n = n
        s.copy_to_device()
        if lanes is not None:
            s.set_param("tile_lanes", lanes)
        s.set_fixed(p)
        s.take_step(dt, steps)
        return s.positions().copy(), s.old_v()[:n].copy()


def test_oracle_takes_the_reference_step_bit_for_bit(oracle):
    """Three whole Heun steps (both stages average the neighbours' old velocities from the second step on)."""
    X = system(n=90)
    Xw, vw = reference_tile_steps(X, 3, 0.05, 7)
    Xo, vo = tile_steps_of(oracle, X, 3, 0.05, 7)
    assert np.abs(Xw - X).max() > 1e-2
    assert np.array_equal(Xw.view(np.uint32), Xo.view(np.uint32))
    assert np.array_equal(vw.view(np.uint32), vo.view(np.uint32))


@pytest.mark.gpu
def test_engine_takes_the_reference_step_bit_for_bit(device):
    X = system(n=90)
    Xw, vw = reference_tile_steps(X, 3, 0.05, 7)
    for lanes in (1, 0):
        Xd, vd = tile_steps_of(device, X, 3, 0.05, 7, lanes)
        assert np.array_equal(Xw.view(np.uint32), Xd.view(np.uint32)), lanes
        assert np.array_equal(vw.view(np.uint32), vd.view(np.uint32)), lanes


def reference_grid_steps(X, steps, dt, p, by_plane):
    """The same step with Grid_computer (solvers.cuh:430-463; the grid rebuilt from X and from X1)."""
    n = len(X)
    X = X.copy()
    old_v = np.zeros((n, 3), f32)
    dt = f32(dt)
    for _ in range(steps):
        dX = reference_forces(X, by_plane, old_v=old_v)
        dX = dX - dX[p]
        X1 = X + dX * dt
        dX1 = reference_forces(X1, by_plane, old_v=old_v)
        dX1 = dX1 - dX1[p]
        X = X + ((dX + dX1) * f32(0.5)) * dt
        old_v = (dX + dX1) * f32(0.5)
    return X, old_v


def grid_steps_of(lib, X, steps, dt, p, sum_order, variant=None):
    n = len(X)
    with Solution("springs_grid", n, GS, 1.0, lib=lib) as s:
        s.h_X[:n] = X
        s.h_n = n
        s.copy_to_device()
        if sum_order:
            s.set_param("sum_order", sum_order)
        if variant is not None:
            s.set_param("force_variant", variant)
        s.set_fixed(p)
        s.take_step(dt, steps)
        return s.positions().copy(), s.old_v()[:n].copy()


@pytest.mark.parametrize("sum_order", [0, 1])
def test_oracle_takes_the_reference_grid_step_bit_for_bit(oracle, sum_order):
    X = system(n=200)
    Xw, vw = reference_grid_steps(X, 3, 0.05, 199, bool(sum_order))
    Xo, vo = grid_steps_of(oracle, X, 3, 0.05, 199, sum_order)
    assert np.abs(Xw - X).max() > 1e-2
    assert np.array_equal(Xw.view(np.uint32), Xo.view(np.uint32))
    assert np.array_equal(vw.view(np.uint32), vo.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("sum_order", [0, 1])
def test_engine_takes_the_reference_grid_step_bit_for_bit(device, sum_order):
    """Three Heun steps on the grid -- two builds per step, the sorted-space second stage, the fused update kernels --
    against numpy: the engine's bits ARE the reference's statement's."""
    X = system(n=200)
    Xw, vw = reference_grid_steps(X, 3, 0.05, 199, bool(sum_order))
    for variant in (-1, 2, 3):
        Xd, vd = grid_steps_of(device, X, 3, 0.05, 199, sum_order, variant)
        assert np.array_equal(Xw.view(np.uint32), Xd.view(np.uint32)), variant
        assert np.array_equal(vw.view(np.uint32), vd.view(np.uint32)), variant


# ---- the golden fixtures themselves, from numpy (set_fixed(): the centre of mass is held) --------------------------
def fold256(lanes):
    """lane[t] += lane[t + s] for s = 128 ... 1 (yalla_amd/csrc/core.hip fold256)."""
    lanes = lanes.copy()
    s2 = 128
    while s2 >= 1:
        lanes[:s2] = lanes[:s2] + lanes[s2:2 * s2]
        s2 //= 2
    return lanes[0]


def tree_sum(v):
    """The engine's documented order for the centre-of-mass sum: B = clamp(ceil(n / 256), 1, 1024) blocks x 256 lanes;
    lane (b, t) sums rows 256 b + t, + 256 B, ... serially from 0; each block folds its lanes by halving; one block
    then sums the B partial sums the same way."""
    n, w = v.shape
    B = min(max((n + 255) // 256, 1), 1024)
    partials = np.zeros((B, w), f32)
    for b in range(B):
        lanes = np.zeros((256, w), f32)
        for t in range(256):
            for i in range(b * 256 + t, n, B * 256):
                lanes[t] = lanes[t] + v[i]
        partials[b] = fold256(lanes)
    lanes = np.zeros((256, w), f32)
    for t in range(256):
        for q in range(t, B, 256):
            lanes[t] = lanes[t] + partials[q]
    return fold256(lanes)


def serial_sum(v):
    acc = np.zeros(v.shape[1], f32)          # thrust::reduce(first, last, Pt{0}) read left to right (solvers.cuh:242)
    for row in v:
        acc = acc + row
    return acc


def golden_steps(X, steps, dt, gs, by_plane, reduce, functor="spring"):
    n = len(X)
    X = X[:, :3].copy()     # (further fields of a wider point get no force from these functors and stay as they are)
    old_v = np.zeros((n, 3), f32)
    dt = f32(dt)
    inv_n = f32(np.float64(1.0) / np.float64(f32(n)))          # Pt / n == Pt * float(1. / n)  (dtypes.cuh:202-208)
    for _ in range(steps):
        dX = reference_forces(X, by_plane, old_v=old_v, gs=gs, functor=functor)
        dX = dX - reduce(dX) * inv_n                            # :241-242, euler_step
        X1 = X + dX * dt
        dX1 = reference_forces(X1, by_plane, old_v=old_v, gs=gs, functor=functor)
        dX1 = dX1 - reduce(dX1) * inv_n                         # :268, heun_step
        X = X + ((dX + dX1) * f32(0.5)) * dt
        old_v = (dX + dX1) * f32(0.5)
    return X, old_v


@pytest.mark.parametrize("name", ["springs_grid_n50", "springs_grid_n800"])
def test_golden_fixture_from_numpy(name):
    """tests/golden/springs_grid_n50.npz / _n800.npz (written by the ORACLE, tests/golden/make_golden.py) recomputed
    from their stored inputs by the numpy statement of the reference: X (reference sum, the engine's centre-of-mass
    tree: one block at 50 cells, four and a second level at 800), X_serial_reduce (all of the reference's defaults)
    and X_by_plane, bit for bit."""
    import os
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    X0 = ref["X0"]
    X, v = golden_steps(X0, 2, 0.001, 50, False, tree_sum)
    assert np.array_equal(X.view(np.uint32), ref["X"].view(np.uint32))
    assert np.array_equal(v.view(np.uint32), ref["old_v"].view(np.uint32))
    Xs, _ = golden_steps(X0, 2, 0.001, 50, False, serial_sum)
    assert np.array_equal(Xs.view(np.uint32), ref["X_serial_reduce"].view(np.uint32))
    Xp, _ = golden_steps(X0, 2, 0.001, 50, True, tree_sum)
    assert np.array_equal(Xp.view(np.uint32), ref["X_by_plane"].view(np.uint32))


def test_golden_fixture_of_five_float_points_from_numpy():
    """tests/golden/relu_po_grid_n250.npz: relu_force (inits.cuh:78-93) on Po_cell {x, y, z, theta, phi}, five steps
    of dt 0.1 -- positions and velocities from numpy, the polarity fields untouched."""
    import os
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "relu_po_grid_n250.npz"))
    X0 = ref["X0"]
    X, v = golden_steps(X0, 5, 0.1, 50, False, tree_sum, functor="relu")
    assert np.array_equal(X.view(np.uint32), ref["X"][:, :3].copy().view(np.uint32))
    assert np.array_equal(v.view(np.uint32), ref["old_v"].view(np.uint32))
    assert np.array_equal(ref["X"][:, 3:].view(np.uint32), X0[:, 3:].view(np.uint32))
    Xp, _ = golden_steps(X0, 5, 0.1, 50, True, tree_sum, functor="relu")
    assert np.array_equal(Xp.view(np.uint32), ref["X_by_plane"][:, :3].copy().view(np.uint32))
