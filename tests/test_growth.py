"""BASELINE config 4: growth with dynamic n, per-cell neighbour counters updated
inside the functor, bending_force between epithelial cells."""
import numpy as np
import pytest

import branching_case
import growth_case


def test_growth_on_oracle(oracle):
    s, nbs = growth_case.setup(oracle)
    types = s.get_prop("type", 200)
    assert 0 < types.sum() < 200, "expected both cell types"
    # the functor counts every neighbour once per stage (2 stages per step)
    assert 10 < nbs.mean() < 30
    counts = growth_case.grow(s, 25)
    assert counts[-1] > 200 and counts == sorted(counts)
    assert counts[-1] <= s.n_max
    X = s.positions()
    assert np.isfinite(X).all()
    # daughters inherit the mother's type; epithelium stays further out on average
    t = s.get_prop("type", counts[-1])
    r = np.linalg.norm(X[:, :3] - X[:, :3].mean(axis=0), axis=1)
    assert r[t == 1].mean() > r[t == 0].mean()
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["grid", "tile"])
def test_growth_device_matches_oracle(oracle, device, solver):
    """Same cell counts at every step (bit-exact integer result) and positions
    within tolerance (sinf/cosf/acosf/powf differ between ocml and glibc)."""
    so, nbs_o = growth_case.setup(oracle, solver)
    sd, nbs_d = growth_case.setup(device, solver)
    assert np.array_equal(nbs_o, nbs_d), "neighbour counters differ"
    assert np.array_equal(so.get_prop("type", 200), sd.get_prop("type", 200))
    # Lock-step: sinf/cosf/acosf/powf differ in the last ulp between ocml and glibc
    # and the dynamics amplify that, so every step starts from the oracle's state
    # and must agree to 1e-5 relative (positions) with IDENTICAL cell counts.
    for s in (so, sd):
        s.set_param("prolif_rate", 0.05)
        s.set_param("seed", 77)
    n_before = 200
    for step in range(12):
        so.take_step(0.2)
        sd.take_step(0.2)
        n_o, n_d = so.get_d_n(), sd.get_d_n()
        assert n_o == n_d, f"cell counts differ at step {step}"
        Xo, Xd = so.positions(), sd.positions()
        scale = np.abs(Xo[:, :3]).max()
        assert np.abs(Xo - Xd).max() <= 1e-5 * scale, step
        for name in ("type", "mes_nbs", "epi_nbs"):
            assert np.array_equal(so.get_prop(name, n_o), sd.get_prop(name, n_d)), (name, step)
        sd.h_X[:] = so.h_X
        sd.h_n = n_o
        sd.copy_to_device()
        sd.set_old_v(so.old_v())
        n_before = n_o
    assert n_before > 200, "nothing divided"
    so.close()
    sd.close()


def test_branching_on_oracle(oracle):
    s, nbs = branching_case.setup(oracle)
    t = s.get_prop("type", 500)
    assert 0 < t.sum() < 500
    n = [s.get_d_n()]
    for _ in range(15):
        s.take_step(0.2)
        n.append(s.get_d_n())
    assert n[-1] > 500 and n == sorted(n)
    X = s.positions()
    assert np.isfinite(X).all()
    tt = s.get_prop("type", n[-1])
    assert (X[tt == 0, 5] == 0).all(), "u only lives on the epithelium"
    assert (X[tt == 1, 6] != 0).any()
    s.close()


@pytest.mark.gpu
def test_branching_device_matches_oracle_lockstep(oracle, device):
    """Config 3: 7-float points, atomics inside the functor, division before the
    step.  Identical cell counts / types / counters, positions and morphogens to
    1e-5 relative per step (libm differs between ocml and glibc)."""
    so, nbs_o = branching_case.setup(oracle)
    sd, nbs_d = branching_case.setup(device)
    assert np.array_equal(nbs_o, nbs_d)
    assert np.array_equal(so.get_prop("type", 500), sd.get_prop("type", 500))
    grown = False
    for step in range(12):
        so.take_step(0.2)
        sd.take_step(0.2)
        n_o, n_d = so.get_d_n(), sd.get_d_n()
        assert n_o == n_d, f"cell counts differ at step {step}"
        grown = grown or n_o > 500
        Xo, Xd = so.positions(), sd.positions()
        for cols in (slice(0, 3), slice(3, 5), slice(5, 7)):
            scale = max(np.abs(Xo[:, cols]).max(), 1e-3)
            assert np.abs(Xo[:, cols] - Xd[:, cols]).max() <= 1e-5 * scale, (step, cols)
        for name in ("type", "mes_nbs", "epi_nbs"):
            assert np.array_equal(so.get_prop(name, n_o), sd.get_prop(name, n_d)), (name, step)
        sd.h_X[:] = so.h_X
        sd.h_n = n_o
        sd.copy_to_device()
        sd.set_old_v(so.old_v())
    assert grown
    so.close()
    sd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [4, 8, 16])
def test_branching_with_several_lanes_per_cell(device, lanes):
    """Config 3's functor counts neighbours with atomicAdd (branching.cu:105-107), so it may be
    called for one cell from several lanes at once: grid_force_coop (force_variant 3) gives the
    default kernel's positions, polarities, morphogens, cell counts, types and counters bit for
    bit, through cell division (n grows between steps)."""
    runs = []
    for variant in (2, 3):
        s, _ = branching_case.setup(device, n_0=3000, n_max=6000)
        s.set_param("force_variant", variant)
        s.set_param("coop_lanes", lanes if variant == 3 else 0)
        s.set_param("prolif_rate", 1.0)
        counts = []
        for _ in range(8):
            s.take_step(0.2)
            counts.append(s.get_d_n())
        n = counts[-1]
        runs.append((counts, s.positions(), s.get_prop("type", n), s.get_prop("mes_nbs", n), s.get_prop("epi_nbs", n)))
        s.close()
    (counts_a, Xa, ta, ma, ea), (counts_b, Xb, tb, mb, eb) = runs
    assert counts_a == counts_b and counts_a[-1] > 3000
    assert np.array_equal(Xa.view(np.uint32), Xb.view(np.uint32))
    assert np.array_equal(ta, tb) and np.array_equal(ma, mb) and np.array_equal(ea, eb)


# ---- Solution::renumber (opt-in, not in the reference): cells renumbered in cube order -------------
def _state(s, n):
    return s.positions(), s.old_v()[:n].copy(), {k: s.get_prop(k, n) for k in ("type", "mes_nbs", "epi_nbs")}


def test_renumber_is_a_permutation_into_cube_order_oracle(oracle):
    """renumber(type, mes_nbs, epi_nbs) on the grown system: the same multiset of cell records
    {X, old_v, type, counters}, now stored cube by cube (what a fresh grid build would sort them to),
    ascending old id inside a cube; a second call changes nothing."""
    s, _ = growth_case.setup(oracle, "grid", n_max=4000)
    growth_case.grow(s, 20)
    n = s.get_d_n()
    X0, v0, p0 = _state(s, n)
    s.set_param("renumber_now", 1)
    X1, v1, p1 = _state(s, n)
    # the permutation, recovered from the (distinct) positions
    key0 = {tuple(x): i for i, x in enumerate(X0.view(np.uint32).tolist())}
    order = np.array([key0[tuple(x)] for x in X1.view(np.uint32).tolist()])
    assert sorted(order.tolist()) == list(range(n))
    assert np.array_equal(v1.view(np.uint32), v0[order].view(np.uint32))
    for k in p0:
        assert np.array_equal(p1[k], p0[k][order]), k
    # cube order: cube id as the grid computes it (grid_size 50, cube_size 1), non-decreasing, ids ascending inside
    cube = ((np.floor(X1[:, 0]) + 25) + (np.floor(X1[:, 1]) + 25) * 50 + (np.floor(X1[:, 2]) + 25) * 2500).astype(int)
    assert (np.diff(cube) >= 0).all()
    same = np.diff(cube) == 0
    assert (np.diff(order)[same] > 0).all()
    assert len(np.unique(cube)) < n, "test too sparse: no cube holds two cells"
    s.set_param("renumber_now", 1)
    X2, v2, p2 = _state(s, n)
    assert np.array_equal(X2.view(np.uint32), X1.view(np.uint32)) and np.array_equal(p2["type"], p1["type"])
    s.close()


def test_renumbered_model_keeps_running_oracle(oracle):
    """A model that renumbers every 4th step (set_param renumber_every: what its loop would do) keeps
    its population, its two cell types and the epithelium outside; daughters are appended behind
    the renumbered cells as before."""
    s, _ = growth_case.setup(oracle, "grid", n_max=4000)
    s.set_param("renumber_every", 4)
    counts = growth_case.grow(s, 25)
    assert counts[-1] > 200 and counts == sorted(counts)
    n = counts[-1]
    X = s.positions()
    t = s.get_prop("type", n)
    assert np.isfinite(X).all() and 0 < t.sum() < n
    r = np.linalg.norm(X[:, :3] - X[:, :3].mean(axis=0), axis=1)
    assert r[t == 1].mean() > r[t == 0].mean()
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["growth", "branching"])
def test_renumbering_device_matches_oracle_lockstep(oracle, device, model):
    """The same model loop with a renumbering every 3rd step on both backends, in lock-step: identical
    cell counts, types and neighbour counters (so the permutations agree), positions to 1e-5."""
    case = growth_case if model == "growth" else branching_case
    so, _ = case.setup(oracle)
    sd, _ = case.setup(device)
    n0 = so.get_d_n()
    for s in (so, sd):
        s.set_param("renumber_every", 3)
        if model == "growth":
            s.set_param("prolif_rate", 0.05)
            s.set_param("seed", 77)
    for step in range(12):
        so.take_step(0.2)
        sd.take_step(0.2)
        n_o, n_d = so.get_d_n(), sd.get_d_n()
        assert n_o == n_d, f"cell counts differ at step {step}"
        Xo, Xd = so.positions(), sd.positions()
        scale = np.abs(Xo[:, :3]).max()
        assert np.abs(Xo[:, :3] - Xd[:, :3]).max() <= 1e-5 * scale, step
        for name in ("type", "mes_nbs", "epi_nbs"):
            assert np.array_equal(so.get_prop(name, n_o), sd.get_prop(name, n_d)), (name, step)
        sd.h_X[:] = so.h_X
        sd.h_n = n_o
        sd.copy_to_device()
        sd.set_old_v(so.old_v())
    assert so.get_d_n() > n0, "nothing divided"
    so.close()
    sd.close()


@pytest.mark.gpu
def test_renumber_moves_links_and_leaves_forces_alone(device):
    """springs + links: renumbering permutes the cells and renames the links' endpoints; the next steps
    then give every cell (found again by its position) the position the unrenumbered run gives it, to
    rounding -- sums are accumulated in another order where cells have since changed cubes."""
    from yalla_amd.solution import Solution
    n = 20000
    runs = []
    for renumber in (False, True):
        with Solution("springs_links_grid", n, 50, 1.0, lib=device) as s:
            s.random_sphere(0.7, 11)
            rng = np.random.default_rng(5)
            a = rng.integers(0, n, 3000)
            links = np.stack([a, (a + rng.integers(1, 50, 3000)) % n], axis=1).astype(np.int32)
            s.set_links(links, 0.2)
            s.take_step(0.002, 2)
            before = s.positions()
            if renumber:
                s.set_param("renumber_now", 1)
                after = s.positions()
                key = {tuple(x): i for i, x in enumerate(before.view(np.uint32).tolist())}
                order = np.array([key[tuple(x)] for x in after.view(np.uint32).tolist()])
            else:
                order = np.arange(n)
            s.take_step(0.002, 3)
            X = np.empty((n, 3), np.float32)
            X[order] = s.positions()       # back to the original numbering
            runs.append(X)
    assert np.abs(runs[0] - runs[1]).max() <= 1e-5 * np.abs(runs[0]).max()
    assert not np.array_equal(runs[0], np.zeros_like(runs[0]))


@pytest.mark.parametrize("model", ["sorting_grid", "push_grid", "clipped_push_grid"])
def test_models_whose_physics_depend_on_ids_refuse_renumbering_oracle(oracle, model):
    """sorting (a cell's type is `i < n / 2`) and the push models (the generic force pushes cell 1): new ids
    would be other physics, so the harness offers renumbering neither every k steps nor on request."""
    from yalla_amd.solution import Solution, YallaError
    with Solution(model, 200, 50, 1.0, lib=oracle) as s:
        s.random_sphere(0.7, 3)
        before = s.positions().copy()
        for name in ("renumber_now", "renumber_every"):
            with pytest.raises(YallaError):
                s.set_param(name, 1)
        assert np.array_equal(s.positions(), before)
