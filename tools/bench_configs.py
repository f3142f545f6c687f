#!/usr/bin/env python3
"""One throughput line per BASELINE.json configuration at its stated size (run on the GPU box):

    python tools/bench_configs.py > profiles/r02_configs.json

cfg 1  springs, Tile_solver, 500 and 800 cells, 100 steps (device) + the serial host loop (oracle)
cfg 2  sorting, 10 000 two-type cells, Grid_solver, dt 0.05, 300 steps
cfg 3  branching model (7-float Cell, links-free snapshot), 100 000 cells, division frozen
cfg 4  passive growth 200 -> 10^6 Po_cell cells, then 20 timed steps
cfg 5  springs, Grid_solver, 1 M and 10 M cells (bench.py's own lines)
One cell-update = one cell advanced by one take_step."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import branching_case
import growth_case
from yalla_amd import _ffi
from yalla_amd.solution import Solution

dev = _ffi.device_lib()
out = []


def timed(sim, dt, steps, warm=3):
    sim.take_step(dt, warm)
    sim.synchronize()
    t0 = time.perf_counter()
    sim.take_step(dt, steps)
    sim.synchronize()
    return time.perf_counter() - t0


# cfg 1
oracle = _ffi.bind(os.path.join(ROOT, "oracle", "_build", "liboracle_models.so"))
for n in (500, 800):
    for name, lib, lanes in (("device, one thread per cell (reference semantics)", dev, 1),
                             ("device, 64 lanes per cell (opt-in, bit-identical)", dev, 64),
                             ("serial host loop (oracle, 1 thread)", oracle, None)):
        with Solution("springs_tile", n, lib=lib) as s:
            if lanes is not None:
                s.set_param("tile_lanes", lanes)
            s.random_sphere(0.5, 42)
            el = timed(s, 0.001, 100, warm=3 if lib is dev else 0)
            out.append({"config": 1, "workload": f"springs, Tile_solver, {n} cells, 100 steps", "backend": name,
                        "cell_updates_per_s": n * 100 / el, "ms_per_step": el / 100 * 1e3})


def bench_py(*args):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", *args],
                          capture_output=True, text=True, check=True)
    return json.loads(proc.stdout.strip().splitlines()[-1])


# cfg 2
line = bench_py("--model", "sorting_grid", "--cells-total", "10000", "--dt", "0.05", "--steps", "300")
out.append({"config": 2, "workload": "sorting (differential_adhesion), Grid_solver, 10 000 cells, dt 0.05, 300 steps",
            "cell_updates_per_s": line["value"], "ms_per_step": line["ms_per_step"],
            "force_launch_us": line["roofline"]["avg_launch_us"]})

line = bench_py("--model", "sorting_grid", "--cells-total", "10000", "--dt", "0.05", "--steps", "300",
                "--force-variant", "3")
out.append({"config": 2, "workload": "the same with force_variant = 3 (grid_force_coop, 16 lanes per cell; opt-in, bit-identical)",
            "cell_updates_per_s": line["value"], "ms_per_step": line["ms_per_step"],
            "force_launch_us": line["roofline"]["avg_launch_us"]})

# cfg 3 (its functor counts neighbours with atomicAdd, branching.cu:105-107: several lanes per cell are allowed)
for variant, label in ((2, ""), (3, "; force_variant = 3 (grid_force_coop, 4 lanes per cell; opt-in)")):
    s, _ = branching_case.setup(dev, n_0=100_000, n_max=140_000)
    s.set_param("prolif_rate", 0.0)
    s.set_param("force_variant", variant)
    el = timed(s, 0.2, 22)
    out.append({"config": 3, "workload": "branching model (Cell, epi_turing_mes_noturing + reset_nbs), Grid_solver gs 100, "
                                         "100 000-cell snapshot, division frozen, 22 steps of dt 0.2" + label,
                "cell_updates_per_s": 100_000 * 22 / el, "ms_per_step": el / 22 * 1e3})
    s.close()

# cfg 4
target = 1_000_000
n_max = int(target * 1.3)
gs = 2 * (int((target / 0.64) ** (1 / 3) * 0.75 / 2 * 1.25) + 4)
seed_state, _ = growth_case.setup(dev, "grid", 200, 400)
X200, types200 = seed_state.positions(), seed_state.get_prop("type", 200)
seed_state.close()
with Solution("passive_growth_grid", n_max, gs, 1.0, lib=dev) as s:
    s.h_n = 200
    s.h_X[:200] = X200
    s.copy_to_device()
    s.set_prop("type", np.concatenate([types200, np.zeros(n_max - 200, np.int32)]))
    s.set_param("prolif_rate", 0.03)
    s.set_param("seed", 7)
    t0 = time.perf_counter()
    steps = 0
    while s.get_d_n() < target:
        s.take_step(0.2, 10)
        steps += 10
    s.synchronize()
    grow_s = time.perf_counter() - t0
    n = s.get_d_n()
    s.set_param("prolif_rate", 0.0)
    el = timed(s, 0.2, 20)
    out.append({"config": 4, "workload": "passive growth (Po_cell, relu_w_epithelium + bending_force + reset_nbs), "
                                         f"200 -> {n} cells in {steps} steps, then 20 timed steps",
                "growth_seconds": grow_s, "n_final": n, "grid_size": gs,
                "cell_updates_per_s": n * 20 / el, "ms_per_step": el / 20 * 1e3})

# cfg 5
for cells in (1_000_000, 10_000_000):
    line = bench_py("--cells-total", str(cells))
    out.append({"config": 5, "workload": f"springs, Grid_solver, {cells} cells, 1 GPU (bench.py)",
                "cell_updates_per_s": line["value"], "ms_per_step": line["ms_per_step"],
                "force_launch_us": line["roofline"]["avg_launch_us"], "grid_size": line["config"]["grid_size"]})

json.dump(out, sys.stdout, indent=1)
print()
