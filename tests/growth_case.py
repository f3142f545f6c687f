"""BASELINE config 4 (examples/passive_growth.cu) as a reusable test case:
mesenchyme enveloped by a polarized epithelium, growing by cell division."""
import numpy as np

from yalla_amd.solution import Solution

MESENCHYME, EPITHELIUM = 0, 1


def setup(lib, solver="grid", n_0=200, n_max=3000, seed=5):
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


def grow(s, steps, rate=0.05, dt=0.2):
    s.set_param("prolif_rate", rate)
    s.set_param("seed", 77)
    counts = []
    for _ in range(steps):
        s.take_step(dt)
        counts.append(s.get_d_n())
    return counts
