"""Parity of the HIP engine with the CPU oracle on identical seeded inputs.

Integer results (cube ids, point ids, cube start/end) must be bit-exact.
Positions: the north_star tolerance is 1e-5 relative on fp32 positions; for
functors that only use + - * / sqrt the engine and the oracle evaluate the same
IEEE binary32 expressions in the same order, so those cases are additionally
required to match BIT FOR BIT (oracle in tree-reduction mode, DESIGN.md
"deterministic COM reduction").
"""
import numpy as np
import pytest

from yalla_amd.solution import Solution

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # BASELINE.json north_star: "within 1e-5 relative on fp32 positions"


def run_both(oracle, device, model, n, gs, cs, dist, seed, dt, steps, tree=True, setup=None, sum_order=0):
    """sum_order: 0 = the reference's one running sum per cell (the default of both libraries), 1 = the
    engine's opt-in own-plane | other-planes order (Grid_computer::sum_order), selected in BOTH."""
    out = []
    for lib in (oracle, device):
        with Solution(model, n, gs, cs, lib=lib) as s:
            if lib is oracle and tree:
                assert s.set_reduce_order(1) == 0
            if sum_order:
                assert s.set_param("sum_order", sum_order) == 0
            s.random_sphere(dist, seed)
            if setup:
                setup(s)
            s.take_step(dt, steps)
            X = s.positions()
            v = s.old_v()[:n]
            g = s.grid() if "grid" in model else None
            out.append((X, v, g))
    return out


def assert_close_positions(a, b):
    scale = np.abs(a).max()
    assert np.abs(a - b).max() <= REL_TOL * scale


@pytest.mark.parametrize("n,gs,steps", [(50, 50, 3), (800, 50, 3), (4096, 50, 2), (20000, 64, 2),
                                        (150000, 64, 1)])
def test_springs_grid_bit_exact(oracle, device, n, gs, steps):
    (Xo, vo, go), (Xd, vd, gd) = run_both(
        oracle, device, "springs_grid", n, gs, 1.0, 0.5, 42, 0.001, steps)
    for name, a, b in zip(("cube_id", "point_id", "cube_start", "cube_end"), go, gd):
        if name in ("cube_id", "point_id"):
            a, b = a[:n], b[:n]
        assert np.array_equal(a, b), f"{name} differs"
    assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32)), "positions not bit-identical"
    assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32)), "old_v not bit-identical"


@pytest.mark.parametrize("n,steps", [(4, 5), (50, 3), (800, 2), (1500, 1)])
def test_springs_tile_bit_exact(oracle, device, n, steps):
    (Xo, vo, _), (Xd, vd, _) = run_both(
        oracle, device, "springs_tile", n, 50, 1.0, 0.5, 42, 0.001, steps)
    assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32))
    assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32))


@pytest.mark.parametrize("model", ["clipped_grid", "relu_grid", "relu_po_grid", "relu_tile",
                                   "relu_po_tile", "clipped_tile", "fading_grid", "fading_tile"])
def test_other_functors_bit_exact(oracle, device, model):
    n = 1000
    (Xo, vo, _), (Xd, vd, _) = run_both(oracle, device, model, n, 50, 1.0, 0.6, 5, 0.1, 3)
    assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32))
    assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32))


# (force_variant, coop_lanes, stage_v_max[, tail_tiles]): grid_force_direct, grid_force, grid_force_bits with
# old_v from global memory and from LDS, grid_force_coop with 16, 8 and 4 lanes per cell; and, only under
# sum_order 1 (own plane | other planes), grid_force_bits with the last 40 / 8 tiles (or all tiles) of every launch
# as half tiles that meet through memory
FORCE_KERNELS = [(0, 0, 0), (1, 0, 0), (2, 0, 0), (2, 0, 1 << 30), (3, 16, 0), (3, 8, 0), (3, 4, 0)]
HALF_TILE_KERNELS = [(2, 0, 0, 40), (2, 0, 1 << 30, 8), (2, 0, 0, 1 << 20)]   # the last: EVERY tile as halves
# (kernel, sum_order): every kernel in the reference's order, every kernel and the half tiles by plane
KERNELS_AND_ORDERS = [(k, 0) for k in FORCE_KERNELS] + [(k, 1) for k in FORCE_KERNELS + HALF_TILE_KERNELS]


def select_kernel(s, kernel, sum_order=0):
    variant, lanes, stage_v_max = kernel[:3]
    assert s.set_param("sum_order", sum_order) == 0
    s.set_param("force_variant", variant)
    s.set_param("coop_lanes", lanes)
    s.set_param("stage_v_max", stage_v_max)
    s.set_param("tail_tiles", kernel[3] if len(kernel) > 3 else 0)


def test_all_force_kernels_agree(oracle, device):
    """grid_force_bits (bit-stream hit list, the default; old_v from global memory or LDS),
    grid_force (byte FIFO), grid_force_direct (the reference's structure) and grid_force_coop
    (several lanes per cell) are the same sums in the same order: bit-identical to each other
    and to the oracle -- in the reference's summation order (the default) and, with half tiles too,
    in the opt-in by-plane order, each against the oracle's restatement of that order."""
    n = 30000
    res = {0: [], 1: []}
    for kernel, order in KERNELS_AND_ORDERS:
        (Xo, vo, _), (Xd, vd, _) = run_both(
            oracle, device, "springs_grid", n, 50, 1.0, 0.5, 4, 0.001, 2, sum_order=order,
            setup=lambda s: select_kernel(s, kernel, order) if s.lib is device else None)
        assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32)), (kernel, order)
        assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32)), (kernel, order)
        res[order].append(Xd)
    for order in (0, 1):
        for X in res[order][1:]:
            assert np.array_equal(res[order][0].view(np.uint32), X.view(np.uint32))
    # the two orders are different roundings of the same sums: not the same bits, far inside 1e-5
    assert not np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    assert_close_positions(res[0][0], res[1][0])


def test_tail_of_half_tiles_at_its_default_size(oracle, device):
    """450 000 cells are 7032 tiles: with sum_order 1 the launch ends with 768 tiles as pairs of half-tile
    workgroups (Grid_computer::forces).  The by-plane oracle's bits, and the bits of a launch of whole tiles only."""
    n = 450000
    (Xo, vo, _), (Xd, vd, _) = run_both(oracle, device, "springs_grid", n, 64, 1.0, 0.5, 11, 0.001, 1, sum_order=1)
    assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32))
    assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32))
    with Solution("springs_grid", n, 64, 1.0, lib=device) as s:
        s.random_sphere(0.5, 11)
        assert s.set_param("sum_order", 1) == 0
        s.set_param("tail_tiles", 0)
        s.take_step(0.001, 1)
        assert np.array_equal(s.positions().view(np.uint32), Xd.view(np.uint32))


def test_tail_exchange_over_many_launches(device):
    """The half tiles' hand-over through memory (relaxed device-scope stores, a ticket, relaxed loads) under
    load: 10^6 cells, 300 steps = 600 launches x 768 split tiles x 64 cells x 7 sums.  One stale or torn value
    anywhere would be amplified by the dynamics (friction with the neighbours) into a different trajectory:
    the positions after 300 steps are bit for bit those of launches made of whole tiles only."""
    n, out = 1_000_000, []
    for tail in (-1, 0):
        with Solution("springs_grid", n, 64, 1.0, lib=device) as s:
            s.random_sphere(0.5, 5)
            assert s.set_param("sum_order", 1) == 0
            s.set_param("tail_tiles", tail)
            s.take_step(0.001, 300)
            out.append(s.positions())
    assert np.abs(out[0]).max() < 40 and np.isfinite(out[0]).all()
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


def test_cooperative_kernel_picks_its_lanes_from_n(oracle, device):
    """force_variant 3 with the lanes per cell left to ya::coop::lanes_for: 16, 8 and 4 lanes and, above
    1.2 * 10^5 cells (7 * 10^4 under sum_order 1), the one-lane kernel -- under sum_order 1 made of half tiles
    altogether while they are all resident at once (10^5, 1.6 * 10^5 cells), of whole tiles beyond (2 * 10^5)
    -- always the bits of the oracle in the same order."""
    for order in (0, 1):
        for n, gs in ((9000, 40), (30000, 50), (60000, 50), (100000, 64), (160000, 64), (200000, 64)):
            (Xo, vo, _), (Xd, vd, _) = run_both(
                oracle, device, "springs_grid", n, gs, 1.0, 0.5, 4, 0.001, 1, sum_order=order,
                setup=lambda s: s.set_param("force_variant", 3) if s.lib is device else None)
            assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32)), (n, order)
            assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32)), (n, order)


@pytest.mark.parametrize("model,n,dist,cube", [
    ("springs_grid", 20000, 0.12, 1.0),    # ~700 cells per cube: rows of hundreds of candidates,
    ("relu_po_grid", 6000, 0.1, 1.0),      #   the bit stream's stretch-by-stretch path, staging in chunks
    ("springs_grid", 3000, 0.5, 2.5),      # cut-off 2.5: ~150 candidates per row
    ("relu_cell_grid", 4000, 0.5, 1.0),    # 32-byte entries
])
def test_force_kernels_agree_on_dense_rows(device, model, n, dist, cube):
    """The kernels against each other where a lane's row exceeds one pass of the bit stream,
    the plane exceeds the staging capacity and a cell's hits exceed its list: bit-identical."""
    res = {0: [], 1: []}
    for kernel, order in KERNELS_AND_ORDERS:
        with Solution(model, n, 50, cube, lib=device) as s:
            s.random_sphere(dist, 5)
            select_kernel(s, kernel, order)
            s.take_step(0.0005, 2)
            res[order].append((s.positions(), s.old_v()))
    for order in (0, 1):
        for X, v in res[order][1:]:
            assert np.array_equal(res[order][0][0].view(np.uint32), X.view(np.uint32))
            assert np.array_equal(res[order][0][1].view(np.uint32), v.view(np.uint32))


def test_both_second_stage_pipelines_agree(oracle, device):
    """The second Heun stage built from the cube-sorted cells (default) and from d_X1
    (the reference's structure) are the same arithmetic: bit-identical positions,
    velocities and grid arrays, for a 16-byte and a 24-byte point entry."""
    for model, n, dt in (("springs_grid", 60000, 0.001), ("relu_po_grid", 20000, 0.1), ("springs_grid", 300000, 0.001)):
        res = []
        # (2 = the sorted pipeline with ya_reduce_mean's own second launch instead of the update kernels
        # folding the partial sums themselves, round 5: the same tree, the same bits)
        for sorted_pipeline in (0, 1, 2):
            with Solution(model, n, 64, 1.0, lib=device) as s:
                s.random_sphere(0.5, 11)
                s.set_param("force_variant", 2)   # (one kernel for every size: the comparison is about the pipelines)
                s.set_param("sorted_pipeline", sorted_pipeline)
                s.take_step(dt, 3)
                res.append((s.positions(), s.old_v(), s.grid()))
        Xa, va, ga = res[0]
        for Xb, vb, gb in res[1:]:
            assert np.array_equal(Xa.view(np.uint32), Xb.view(np.uint32)), model
            assert np.array_equal(va.view(np.uint32), vb.view(np.uint32)), model
            for a, b in zip(ga, gb):
                assert np.array_equal(a, b), model


def test_randomised_configurations_bit_exact(oracle, device):
    """A fixed slice of tests/fuzz_parity.py's generator: sizes around the wavefront /
    workgroup boundaries, sparse to very dense systems, cube sizes, functors, point types."""
    import fuzz_parity
    for seed in range(1000, 1040):
        case = fuzz_parity.draw(seed)
        assert fuzz_parity.run_case(oracle, device, case), case


def test_device_kernels_really_ran(device):
    """Guards against a silent fallback: the HIP force kernel launches are
    counted by the engine's own event profiler (2 per take_step)."""
    with Solution("springs_grid", 2000, 50, 1.0, lib=device) as s:
        assert device.ya_models_is_device() == 1
        s.random_sphere(0.5, 1)
        s.profile(True)
        s.take_step(0.001, 3)
        ms, launches = s.profile_read()
        assert launches == 6 and ms > 0


def test_serial_reduce_within_tolerance(oracle, device):
    """Against the plain left-to-right COM sum the match is tolerance-only."""
    n = 4096
    (Xo, _, _), (Xd, _, _) = run_both(
        oracle, device, "springs_grid", n, 50, 1.0, 0.5, 42, 0.001, 3, tree=False)
    assert_close_positions(Xo, Xd)


@pytest.mark.parametrize("solver", ["tile", "grid"])
def test_sorting_lockstep_within_tolerance(oracle, device, solver):
    """differential_adhesion calls powf (glibc vs ocml differ in the last ulp) and
    the overlapping start is stiff, so trajectories are compared in lock-step:
    every step starts from the oracle's state and must agree to 1e-5 relative."""
    n = 2000 if solver == "grid" else 600
    with Solution(f"sorting_{solver}", n, 50, 1.0, lib=oracle) as o, \
            Solution(f"sorting_{solver}", n, 50, 1.0, lib=device) as d:
        for s in (o, d):
            s.set_param("n_cells", n)
            s.random_sphere(0.5, 42)
        o.set_reduce_order(1)
        for step in range(6):
            o.take_step(0.05)
            d.take_step(0.05)
            Xo, Xd = o.positions(), d.positions()
            assert_close_positions(Xo, Xd)
            scale = np.abs(o.old_v()).max()
            assert np.abs(o.old_v() - d.old_v()).max() <= 1e-4 * scale
            d.h_X[:] = o.h_X
            d.copy_to_device()
            d.set_old_v(o.old_v())


def test_fixed_point_modes(oracle, device):
    n = 500
    for mode in ("point", "xy"):
        def setup(s, mode=mode):
            (s.set_fixed if mode == "point" else s.set_fixed_xy)(17)
        (Xo, vo, _), (Xd, vd, _) = run_both(
            oracle, device, "clipped_grid", n, 50, 1.0, 0.6, 9, 0.1, 3, setup=setup)
        assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32)), mode


def test_links_parity(oracle, device):
    """Atomic accumulation order is unspecified on the device: tolerance."""
    n = 2000
    rng = np.random.default_rng(3)
    pairs = rng.integers(0, n, size=(3000, 2))
    (Xo, _, _), (Xd, _, _) = run_both(
        oracle, device, "springs_links_grid", n, 50, 1.0, 0.5, 42, 0.01, 3,
        setup=lambda s: s.set_links(pairs, 0.2))
    assert_close_positions(Xo, Xd)


def test_links_segmented_sum_matches_atomics_and_oracle(oracle, device):
    """link_forces' atomics-free path (pairs sorted by cell, one sum per cell) against the
    one-thread-per-link atomics kernel and the oracle: random links, inert links (a == b), a
    hub cell with more links than the 256-entry segment cut, unused slots beyond *d_n."""
    n = 3000
    rng = np.random.default_rng(5)
    pairs = rng.integers(0, n, size=(5000, 2))
    pairs[::50, 1] = pairs[::50, 0]            # inert links
    pairs[1000:1700, 0] = 17                   # a hub
    res = {}
    try:
        for name, lib, seg_min in (("oracle", oracle, None), ("atomics", device, 1 << 30), ("segmented", device, 1)):
            with Solution("springs_links_grid", n, 50, 1.0, lib=lib) as s:
                if lib is oracle:
                    s.set_reduce_order(1)
                else:
                    s.set_param("links_segmented_min", seg_min)
                s.random_sphere(0.5, 42)
                s.set_links(pairs, 0.2)
                s.take_step(0.01, 3)
                res[name] = s.positions()
    finally:
        with Solution("springs_links_grid", 8, 50, 1.0, lib=device) as s:
            s.set_param("links_segmented_min", 1000000)
    assert_close_positions(res["oracle"], res["atomics"])
    assert_close_positions(res["oracle"], res["segmented"])
    # without a hub every cell's sum has a fixed order: the run repeats bit for bit
    # (the atomics' order, and so their rounding, need not)
    plain = rng.integers(0, n, size=(5000, 2))
    runs = []
    for _ in range(2):
        with Solution("springs_links_grid", n, 50, 1.0, lib=device) as s:
            s.set_param("links_segmented_min", 1)
            s.random_sphere(0.5, 42)
            s.set_links(plain, 0.2)
            s.take_step(0.01, 3)
            runs.append(s.positions())
            s.set_param("links_segmented_min", 1000000)
    assert np.array_equal(runs[0].view(np.uint32), runs[1].view(np.uint32))


def test_dynamic_n(oracle, device):
    """h_n < n_max: only the first n points take part (solvers.cuh:229)."""
    out = []
    for lib in (oracle, device):
        with Solution("springs_grid", 3000, 50, 1.0, lib=lib) as s:
            if lib is oracle:
                s.set_reduce_order(1)
            s.h_n = 1234
            s.random_sphere(0.5, 8)
            s.take_step(0.001, 2)
            assert s.get_d_n() == 1234
            out.append(s.positions())
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


def test_growing_and_shrinking_n(oracle, device):
    """Cells appended (n grows) and removed (n shrinks) between steps, as model
    kernels do through *d_n: the grid's visit order must stay a permutation."""
    out = []
    for lib in (oracle, device):
        with Solution("springs_grid", 4000, 50, 1.0, lib=lib) as s:
            if lib is oracle:
                s.set_reduce_order(1)
            s.random_sphere(0.5, 21)      # all 4000 positions exist on the host
            snaps = []
            for n in (1000, 2500, 4000, 700, 3999):
                s.copy_to_host()
                s.h_n = n
                s.copy_to_device()
                s.take_step(0.001, 2)
                snaps.append(s.positions())
                cube_id, point_id, start, end = s.grid()
                snaps.append(point_id[:n].copy())
            out.append(snaps)
    for a, b in zip(*out):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_pair_distance_sqrt_is_correctly_rounded_for_every_float(device):
    """ya::exact_sqrt (the engine's sqrt for pair distances) against sqrtf for ALL
    non-negative binary32 bit patterns, zero, denormals and infinity included."""
    assert device.ya_check_sqrt(0x00000000, 0x7F800000) == 0
    # negative arguments and NaNs cannot arise from a sum of squares; still equal in class
    assert device.ya_check_sqrt(0x7F800001, 0x7F800100) == 0


def test_point_division_reciprocal_is_correctly_rounded_for_every_float(device):
    """ya::reciprocal (the factor of `Pt / float`, reference dtypes.cuh:202-217) against
    the IEEE 1.0f / x for ALL binary32 bit patterns of either sign."""
    assert device.ya_check_reciprocal(0x00000000, 0x7FFFFFFF) == 0
    assert device.ya_check_reciprocal(0x80000000, 0xFFFFFFFF) == 0


@pytest.mark.parametrize("dist,n", [(0.2, 6000), (0.12, 3000), (1.4, 3000)])
def test_dense_and_sparse_systems_bit_exact(oracle, device, dist, n):
    """Rarely taken paths of grid_force: rows longer than 64 candidates (re-anchoring),
    FIFO overflow drains, planes that need several LDS chunks (dense: ~150-700 cells
    per cube), and nearly empty stencils (sparse)."""
    (Xo, vo, go), (Xd, vd, gd) = run_both(
        oracle, device, "clipped_grid", n, 50, 1.0, dist, 11, 0.0005, 2)
    for a, b in zip(go, gd):
        assert np.array_equal(a[:n] if len(a) == n else a, b[:n] if len(b) == n else b)
    assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32))
    assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32))


def test_dense_po_cell_bit_exact(oracle, device):
    """The same stress with 24-byte entries (Po_cell: smaller LDS chunks)."""
    n = 4000
    (Xo, vo, _), (Xd, vd, _) = run_both(oracle, device, "relu_po_grid", n, 50, 1.0, 0.2, 3, 0.001, 2)
    assert np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32))
    assert np.array_equal(vo.view(np.uint32), vd.view(np.uint32))


def test_graph_replay_is_the_same_step(oracle, device):
    """Heun_solver::graph_steps: the step replayed as one hipGraph (captured the second time
    the same step is asked for) gives bit-identical positions, velocities and grid arrays to
    plain launches and to the oracle -- also across a change of dt (new capture), of the
    fixed point, and for a 24-byte point."""
    for model, n, dts in (("springs_grid", 20000, (0.001, 0.001, 0.001, 0.001, 0.002, 0.002, 0.002)),
                          ("relu_po_grid", 6000, (0.05,) * 6)):
        res = []
        for mode in ("oracle", 0, 1, -1, "coop"):   # "coop": the replayed step with several lanes per cell
            lib = oracle if mode == "oracle" else device
            with Solution(model, n, 50, 1.0, lib=lib) as s:
                if mode == "oracle":
                    s.set_reduce_order(1)
                elif mode == "coop":
                    s.set_param("graph", 1)
                    s.set_param("force_variant", 3)
                else:
                    s.set_param("graph", mode)
                s.random_sphere(0.5, 9)
                for k, dt in enumerate(dts):
                    if k == 3:
                        s.set_fixed(7)
                    s.take_step(dt, 1)
                res.append((s.positions(), s.old_v(), s.grid()))
        for X, v, g in res[1:]:
            assert np.array_equal(res[0][0].view(np.uint32), X.view(np.uint32)), model
            assert np.array_equal(res[0][1].view(np.uint32), v.view(np.uint32)), model
            for a, b in zip(res[0][2], g):
                assert np.array_equal(a[:n] if len(a) == n else a, b[:n] if len(b) == n else b)


@pytest.mark.parametrize("model,dt", [("springs_tile", 0.001), ("clipped_tile", 0.01), ("relu_po_tile", 0.1),
                                      ("oscillator_tile", 0.01)])
def test_many_lanes_per_cell_tile_kernel_bit_exact(oracle, device, model, dt):
    """Tile_computer::lanes_per_cell = 16 / 64 (ya::tile_force_coop: that many lanes evaluate a
    cell's pairs, one lane per component sums them in ascending j): bit-identical to the oracle
    and to the one-thread-per-cell kernel, for 3-, 4- and 5-float points, around the workgroup
    and tile boundaries."""
    for n in (1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 129, 511, 512, 513, 800, 1500):
        res = []
        for lib, lanes in ((oracle, None), (device, 1), (device, 16), (device, 64)):
            with Solution(model, n, lib=lib) as s:
                if lib is oracle:
                    s.set_reduce_order(1)
                else:
                    s.set_param("tile_lanes", lanes)
                s.random_sphere(0.6, 21)
                if model == "oscillator_tile":
                    s.h_X[:n, 3] = np.linspace(0, 1, n, dtype=np.float32)
                    s.copy_to_device()
                s.take_step(dt, 2)
                res.append((s.positions(), s.old_v()[:n]))
        for X, v in res[1:]:
            assert np.array_equal(res[0][0].view(np.uint32), X.view(np.uint32)), (model, n)
            assert np.array_equal(res[0][1].view(np.uint32), v.view(np.uint32)), (model, n)


def test_empty_and_tiny_systems(oracle, device):
    """n = 0 (take_step returns after reading *d_n, solvers.cuh:229-230), one cell (no partner: only
    the centre-of-mass fix acts, and it cancels the cell's own velocity), two, none again, five --
    on both solvers, the device's grid arrays staying a permutation throughout."""
    for model in ("springs_grid", "springs_tile"):
        out = []
        for lib in (oracle, device):
            with Solution(model, 64, 20, 1.0, lib=lib) as s:
                if lib is oracle:
                    s.set_reduce_order(1)
                s.random_sphere(0.5, 5)
                snaps = []
                for n in (0, 1, 2, 0, 5):
                    s.copy_to_host()
                    s.h_n = n
                    s.copy_to_device()
                    s.take_step(0.01, 3)
                    assert s.get_d_n() == n
                    snaps.append(s.positions().copy())
                out.append(snaps)
        for a_, b_ in zip(*out):
            assert a_.shape == b_.shape and np.array_equal(a_.view(np.uint32), b_.view(np.uint32)), model
