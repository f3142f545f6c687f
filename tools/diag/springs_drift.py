"""How the headline workload changes while it runs: the springs system (rest length 0.5, force cut off at 1)
started from random_sphere(0.5) contracts, so the pairs inside the cut-off per cell -- the work of a
cell-update -- grow with the step count.  springs_drift.py [cells] > profiles/rNN_springs_drift.json"""
import json
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yalla_amd.solution import Solution  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gs = 64 if n == 1_000_000 else int(2 * ((n / 0.64) ** (1 / 3) * 0.25 + 4))
rows = []
with Solution("springs_grid", n, gs, 1.0) as sim:
    sim.random_sphere(0.5, 42)
    done = 0
    for upto in (0, 3, 23, 43, 63, 103, 203):  # (soon after, the collapsing system throws a cell out of any grid)
        if upto > done:
            sim.take_step(0.001, upto - done)
            done = upto
        X = sim.positions()[:, :3].astype(np.float64)
        tree = cKDTree(X)
        pairs = tree.count_neighbors(tree, 1.0) - n          # ordered pairs i != j inside the cut-off
        t0 = time.perf_counter()
        sim.take_step(0.001, 4)
        sim.positions()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        done += 4
        rows.append({"steps_taken": upto, "pairs_inside_cutoff_per_cell": round(pairs / n, 2),
                     "radius_of_gyration": round(float(np.sqrt((X ** 2).sum(1).mean())), 3),
                     "ms_per_step_next_4_steps_incl_copy": round(ms, 3)})
        print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps({"cells": n, "model": "springs_grid, dt 0.001, random_sphere(0.5) seed 42", "rows": rows}, indent=1))
