"""Is a force launch over the TOP cap of a ball dearer than one over the bottom cap?  (The last slab of the
10 M-cell rehearsal's force launches took 7-20 % longer than the first slab's.)  The two caps of the
10 M-cell random_sphere as systems of their own, the top cap also mirrored (z -> -z), stepped undivided."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yalla_amd.solution import Solution

n = 10_000_000
with Solution("springs_tile", n) as s:
    s.random_sphere(0.5, 42)
    X = s.h_X[:n].copy()
z = X[:, 2]
q = np.quantile(z, [0.125, 0.875])
caps = {"bottom": X[z < q[0]], "top": X[z >= q[1]]}
caps["top_mirrored"] = caps["top"] * np.array([1, 1, -1], np.float32)
caps["bottom_mirrored"] = caps["bottom"] * np.array([1, 1, -1], np.float32)
out = {}
for name, cells in caps.items():
    m = len(cells)
    with Solution("springs_grid", m, 130, 1.0) as s:
        s.h_X[:m] = cells
        s.h_n = m
        s.copy_to_device()
        s.take_step(0.001, 3)
        s.synchronize()
        s.profile(True, every=1)
        t0 = time.perf_counter()
        s.take_step(0.001, 24)
        s.synchronize()
        ms = (time.perf_counter() - t0) / 24 * 1e3
        force_ms, launches = s.profile_read()
        out[name] = {"cells": m, "ms_per_step": round(ms, 4), "force_launch_us": round(force_ms / launches * 1e3, 1)}
        print(name, out[name], file=sys.stderr, flush=True)
print(json.dumps(out))
