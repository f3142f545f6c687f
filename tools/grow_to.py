#!/usr/bin/env python3
"""BASELINE config 4 at scale: passive growth (examples/passive_growth.cu scaled)
from 200 cells to --target cells on one MI355X, timing the last --timed steps."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import growth_case
from yalla_amd import device_lib

ap = argparse.ArgumentParser()
ap.add_argument("--target", type=int, default=1_000_000)
ap.add_argument("--rate", type=float, default=0.03)
ap.add_argument("--timed", type=int, default=20)
a = ap.parse_args()
lib = device_lib()
n_max = int(a.target * 1.3)
s, _ = growth_case.setup(lib, "grid", 200, n_max)
s.grid_size  # gs fixed at 50 in growth_case.setup: rebuild with a larger grid
state = s.positions(); types = s.get_prop("type", 200); s.close()
from yalla_amd.solution import Solution
gs = 2 * (int((a.target / 0.64) ** (1 / 3) * 0.75 / 2 * 1.25) + 4)
s = Solution("passive_growth_grid", n_max, gs, 1.0, lib=lib)
s.h_n = 200; s.h_X[:200] = state; s.copy_to_device()
s.set_prop("type", np.concatenate([types, np.zeros(n_max - 200, np.int32)]))
s.set_param("prolif_rate", a.rate); s.set_param("seed", 7)
t0 = time.perf_counter(); steps = 0
while s.get_d_n() < a.target:
    s.take_step(0.2, 10); steps += 10
    if steps % 200 == 0: print(steps, s.get_d_n(), file=sys.stderr)
s.synchronize(); grow_s = time.perf_counter() - t0
n = s.get_d_n()
s.set_param("prolif_rate", 0.0)
s.take_step(0.2, 3); s.synchronize()
t0 = time.perf_counter(); s.take_step(0.2, a.timed); s.synchronize(); dt = time.perf_counter() - t0
X = s.positions()
print(json.dumps({"workload": "passive_growth (Po_cell, relu_w_epithelium + bending_force, reset_nbs)",
                  "n_final": n, "grid_size": gs, "growth_steps": steps, "growth_seconds": grow_s,
                  "cell_updates_per_s_at_n_final": n * a.timed / dt, "ms_per_step": dt / a.timed * 1e3,
                  "finite": bool(np.isfinite(X).all()), "extent": float(np.abs(X[:, :3]).max())}))
