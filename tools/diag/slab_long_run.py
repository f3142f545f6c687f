"""A long run of the decomposition against the undivided system with the model that has no jumps at the cut-off
(fading_grid): slab_long_run.py [cells] [slabs] [steps] [dt] [migrate_every].  Prints the largest difference over
all cells at the end, how far the cells travelled and how many changed owner."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_slab
from yalla_amd import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dt = float(sys.argv[4]) if len(sys.argv) > 4 else 0.02
every = int(sys.argv[5]) if len(sys.argv) > 5 else 8
device = _ffi.device_lib()
gs = int(2 * ((n / 0.64) ** (1 / 3) * 0.25 + 6))
t0 = time.time()
X0, Xref = test_slab.reference_run(device, n, gs, 0.5, 3, dt, steps, model="fading_grid")
t1 = time.time()
try:
    X, moved = test_slab.slab_run(device, X0, world, gs, dt, steps, "hip", every, model="fading_grid")
except Exception as err:
    print("the slabs stopped:", getattr(err, "all_codes", None), str(err)[:120])
    raise SystemExit(1)
t2 = time.time()
diff = np.abs(X - Xref).max(axis=1)
scale = np.abs(Xref).max()
print(json.dumps({"cells": n, "slabs": world, "steps": steps, "dt": dt, "migrate_every": every, "grid_size": gs,
                  "changed_owner": int(moved), "largest_travel": float(np.abs(Xref - X0).max()),
                  "extent": float(scale), "max_diff": float(diff.max()), "max_diff_rel": float(diff.max() / scale),
                  "cells_beyond_1e-5": int((diff > 1e-5 * scale).sum()),
                  "undivided_s": round(t1 - t0, 1), "slabs_s": round(t2 - t1, 1)}))
