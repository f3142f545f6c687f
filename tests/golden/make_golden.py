#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no stored vectors for the step path (its tests are analytic)
and cannot run in this image (CUDA), so these fixtures pin the *oracle* (and
through it the HIP engine) against regressions: inputs are regenerated from the
seed by random_sphere (glibc rand(), deterministic), expected outputs are the
oracle's.  Run from the repo root:  python tests/golden/make_golden.py

What is stored per case:
  X, old_v          the oracle with the REFERENCE's grid summation order (one running sum per cell,
                    ref solvers.cuh:437-459; Grid_computer::sum_order = YA_SUM_REFERENCE, the default of the
                    oracle AND of the HIP engine) and the engine's centre-of-mass tree (YA_REDUCE_TREE)
  X_serial_reduce   the same with the serial centre-of-mass sum: the oracle's all-defaults reading of the reference
  X_by_plane        grid cases: the engine's OPT-IN own-plane | other-planes order (YA_SUM_BY_PLANE), tree COM

Three of the fixtures are ALSO recomputed from their stored inputs by an independent numpy binary32 statement of the
reference (tests/test_reference_statement_numpy.py: springs_grid_n50, springs_grid_n800, relu_po_grid_n250 -- X, old_v,
X_serial_reduce, X_by_plane bit for bit), so those are pinned by more than the oracle that wrote them.

clipped_grid_n4096_100steps pins nothing physical: springs + friction_w_neighbour are chaotic over 100 steps (the
tree and the serial COM order alone move every cell by ~1.4, 148 % of the system's extent; so do the two grid
summation orders).  It is a BIT-REGRESSION fixture only -- same order, same bits -- which is why the cross-order
comparisons in tests/test_golden.py skip runs of more than 10 steps.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import build_oracle  # noqa: E402
from yalla_amd import _ffi  # noqa: E402
from yalla_amd.solution import Solution  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name -> (model, n, grid_size, cube_size, dist, seed, dt, steps, extras)
CASES = {
    "springs_grid_n50": ("springs_grid", 50, 50, 1.0, 0.5, 42, 0.001, 2, {}),
    "springs_grid_n800": ("springs_grid", 800, 50, 1.0, 0.5, 42, 0.001, 2, {}),
    "springs_grid_n4096": ("springs_grid", 4096, 50, 1.0, 0.5, 42, 0.001, 2, {}),
    "springs_tile_n800": ("springs_tile", 800, 50, 1.0, 0.5, 42, 0.001, 2, {}),
    "clipped_grid_n4096_100steps": ("clipped_grid", 4096, 50, 1.0, 0.5, 42, 0.001, 100, {}),
    "relu_po_grid_n250": ("relu_po_grid", 250, 50, 1.0, 0.6, 7, 0.1, 5, {}),
    "sorting_grid_n100": ("sorting_grid", 100, 50, 1.0, 0.5, 42, 0.05, 10, {"n_cells": 100}),
    "links_square_of_four": ("links_tile", 4, 50, 1.0, 0.0, 0, 0.1, 10, {"links": [(0, 1), (1, 2), (2, 3), (3, 0)]}),
}


def run(lib, case, tree, sum_order=0):
    model, n, gs, cs, dist, seed, dt, steps, extras = CASES[case]
    with Solution(model, n, gs, cs, lib=lib) as s:
        if lib.ya_models_is_device() == 0:
            s.set_reduce_order(1 if tree else 0)
        if sum_order:
            s.set_param("sum_order", sum_order)
        if "links" in extras:
            s.h_X[:] = [(1, 1, 0), (1, -1, 0), (-1, -1, 0), (-1, 1, 0)]
            s.copy_to_device()
            s.set_links(extras["links"])
        else:
            s.random_sphere(dist, seed)
        if "n_cells" in extras:
            s.set_param("n_cells", extras["n_cells"])
        out = {"X0": s.h_X[:n].copy()}
        if "grid" in model:
            # grid of the initial state (first build), before any step
            cid, pid, start, end = s.build_grid(gs, cs)
            out["cube_id"], out["point_id"] = cid[:n].copy(), pid[:n].copy()
            occupied = np.nonzero(start >= 0)[0].astype(np.int32)
            out["occupied_cubes"] = occupied
            out["cube_start"], out["cube_end"] = start[occupied].copy(), end[occupied].copy()
        s.take_step(dt, steps)
        out["X"] = s.positions()
        out["old_v"] = s.old_v()[:n].copy()
    return out


def main():
    lib = _ffi.bind(build_oracle())
    for case in CASES:
        tree = run(lib, case, True)
        serial = run(lib, case, False)
        tree["X_serial_reduce"] = serial["X"]
        if "grid" in CASES[case][0]:
            tree["X_by_plane"] = run(lib, case, True, sum_order=1)["X"]
        np.savez_compressed(os.path.join(OUT, case + ".npz"), **tree)
        print(case, {k: v.shape for k, v in tree.items()})


if __name__ == "__main__":
    main()
