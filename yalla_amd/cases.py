"""The BASELINE.json example models as reusable set-ups (host-side sequencing of the model
programs examples/passive_growth.cu and examples/branching.cu of the reference, on the named
models of the harness): used by the tests, by bench.py --model passive_growth_grid /
branching_grid and by the tools that grow a system to its stated size."""
import numpy as np

from .solution import Solution

MESENCHYME, EPITHELIUM = 0, 1


# ---- config 4: examples/passive_growth.cu ----------------------------------------------------
def growth_setup(lib, solver="grid", n_0=200, n_max=3000, seed=5):
    s = Solution(f"passive_growth_{solver}", n_max, 50, 1.0, lib=lib)
    if lib.ya_models_is_device() == 0:
        s.set_reduce_order(1)
    # relaxed_sphere(0.75): random_sphere(0.6), relax with relu_force, scale by
    # 0.75 / 0.8 (inits.cuh:95-125; fewer relaxation steps than the reference's 1000)
    with Solution(f"relu_po_{solver}", n_0, 50, 1.0, lib=lib) as relax:
        if lib.ya_models_is_device() == 0:
            relax.set_reduce_order(1)
        relax.random_sphere(0.6, seed)
        relax.take_step(0.1, 300)
        X0 = relax.positions()
    s.h_n = n_0
    s.h_X[:] = 0
    s.h_X[:n_0, :3] = X0[:, :3] * np.float32(0.75 / 0.8)
    s.copy_to_device()
    s.set_prop("type", np.zeros(n_max, np.int32))
    # find the epithelium: passive_growth.cu:120-138
    s.set_prop("mes_nbs", np.zeros(n_max, np.int32))
    s.set_param("reset_nbs", 0)
    s.take_step(0.2)
    s.set_param("reset_nbs", 1)
    s.copy_to_host()
    nbs = s.get_prop("mes_nbs", n_0)
    types = np.zeros(n_max, np.int32)
    X = s.h_X
    for i in range(n_0):
        if nbs[i] < 12 * 2:                    # *2 for the 2nd order solver
            types[i] = EPITHELIUM
            dist = np.sqrt(np.float32(X[i, 0] * X[i, 0] + X[i, 1] * X[i, 1] + X[i, 2] * X[i, 2]))
            X[i, 3] = np.arccos(np.float32(X[i, 2] / dist))
            X[i, 4] = np.arctan2(X[i, 1], X[i, 0])
        else:
            X[i, 3] = 0
            X[i, 4] = 0
    s.copy_to_device()
    s.set_prop("type", types)
    return s, nbs


def growth_grow(s, steps, rate=0.05, dt=0.2):
    s.set_param("prolif_rate", rate)
    s.set_param("seed", 77)
    counts = []
    for _ in range(steps):
        s.take_step(dt)
        counts.append(s.get_d_n())
    return counts


# ---- config 3: examples/branching.cu ---------------------------------------------------------
def branching_setup(lib, n_0=500, n_max=4000, seed=9):
    with Solution("relu_cell_grid", n_0, 100, 1.0, lib=lib) as relax:   # relaxed_sphere(0.75)
        if lib.ya_models_is_device() == 0:
            relax.set_reduce_order(1)
        relax.random_sphere(0.6, seed)
        relax.take_step(0.1, 300)
        X0 = relax.positions()
    s = Solution("branching_grid", n_max, 100, 1.0, lib=lib)            # branching.cu:176
    if lib.ya_models_is_device() == 0:
        s.set_reduce_order(1)
    s.h_n = n_0
    s.h_X[:] = 0
    s.h_X[:n_0, :3] = X0[:, :3] * np.float32(0.75 / 0.8)
    s.copy_to_device()
    s.set_prop("type", np.zeros(n_max, np.int32))
    # find the epithelium: branching.cu:231-250 (a step of dt = 0 just counts)
    s.set_prop("mes_nbs", np.zeros(n_max, np.int32))
    s.set_param("reset_nbs", 0)
    s.take_step(0.0)
    s.set_param("reset_nbs", 1)
    s.copy_to_host()
    nbs = s.get_prop("mes_nbs", n_0)
    types = np.zeros(n_max, np.int32)
    rng = np.random.default_rng(1)
    X = s.h_X
    for i in range(n_0):
        if nbs[i] < 20:
            types[i] = EPITHELIUM
            dist = np.sqrt(np.float32(X[i, 0] * X[i, 0] + X[i, 1] * X[i, 1] + X[i, 2] * X[i, 2]))
            X[i, 3] = np.arccos(np.float32(X[i, 2] / dist))
            X[i, 4] = np.arctan2(X[i, 1], X[i, 0])
            X[i, 5] = rng.random() / 5 - 0.1
            X[i, 6] = rng.random() / 5 - 0.1
    s.copy_to_device()
    s.set_prop("type", types)
    s.set_param("prolif_rate", 1.0)   # switches division on; the rule has its own rates
    s.set_param("seed", 123)
    return s, nbs


# ---- the configurations at their stated sizes (bench.py, tools/) ------------------------------
def grid_size_for_growth(target):
    """Grid that holds a relaxed sphere of `target` cells at spacing ~0.75 with room to spare
    (passive growth ends at R ~ 43.5 for 10^6 cells: examples/passive_growth.cu scaled)."""
    return 2 * (int((target / 0.64) ** (1 / 3) * 0.75 / 2 * 1.25) + 4)


def config4_state(lib, target=1_000_000, rate=0.03, seed=7):
    """BASELINE config 4: passive growth from 200 cells until >= target cells (dynamic d_n,
    proliferation by the model's own kernel), returned as a state dict with division frozen."""
    import time
    n_max = int(target * 1.3)
    gs = grid_size_for_growth(target)
    seed_state, _ = growth_setup(lib, "grid", 200, 400)
    X200, types200 = seed_state.positions(), seed_state.get_prop("type", 200)
    seed_state.close()
    with Solution("passive_growth_grid", n_max, gs, 1.0, lib=lib) as s:
        s.h_n = 200
        s.h_X[:200] = X200
        s.copy_to_device()
        s.set_prop("type", np.concatenate([types200, np.zeros(n_max - 200, np.int32)]))
        s.set_param("prolif_rate", rate)
        s.set_param("seed", seed)
        t0 = time.perf_counter()
        steps = 0
        while s.get_d_n() < target:
            s.take_step(0.2, 10)
            steps += 10
        s.synchronize()
        seconds = time.perf_counter() - t0
        n = s.get_d_n()
        return {"model": "passive_growth_grid", "n": n, "n_max": n_max, "grid_size": gs, "dt": 0.2,
                "X": s.positions(), "old_v": s.old_v()[:n].copy(), "type": s.get_prop("type", n),
                "growth_steps": steps, "growth_seconds": seconds}


def config3_state(lib, n=100_000):
    """BASELINE config 3: the branching model's cell type and functor on an n-cell relaxed sphere
    with its epithelium found as the model program does, division frozen."""
    n_max = int(n * 1.4)
    s, _ = branching_setup(lib, n_0=n, n_max=n_max)
    state = {"model": "branching_grid", "n": n, "n_max": n_max, "grid_size": 100, "dt": 0.2,
             "X": s.positions(), "old_v": s.old_v()[:n].copy(), "type": s.get_prop("type", n),
             "growth_steps": 0, "growth_seconds": 0.0}
    s.close()
    return state


def from_state(state, lib):
    """A Solution of the state's model holding the state, division switched off."""
    n, n_max = int(state["n"]), int(state["n_max"])
    s = Solution(str(state["model"]), n_max, int(state["grid_size"]), 1.0, lib=lib)
    if lib.ya_models_is_device() == 0:
        s.set_reduce_order(1)
    s.h_n = n
    s.h_X[:] = 0
    s.h_X[:n] = state["X"]
    s.copy_to_device()
    v = np.zeros((n_max, 3), np.float32)
    v[:n] = state["old_v"]
    s.set_old_v(v)
    types = np.zeros(n_max, np.int32)
    types[:n] = state["type"]
    s.set_prop("type", types)
    s.set_prop("mes_nbs", np.zeros(n_max, np.int32))
    if str(state["model"]).startswith("branching"):
        s.set_prop("epi_nbs", np.zeros(n_max, np.int32))
    s.set_param("reset_nbs", 1)
    s.set_param("prolif_rate", 0.0)
    return s


def save_state(state, path):
    np.savez(path, **{k: np.asarray(v) for k, v in state.items()})


def load_state(path):
    with np.load(path, allow_pickle=False) as f:
        return {k: (f[k].item() if f[k].ndim == 0 else f[k]) for k in f.files}
