"""Bit-exact parity THROUGH the clumping of the springs system: device and oracle stepped side by side for
hundreds of steps while the pairs inside the cut-off per cell grow by an order of magnitude (dense rows, chunked
staging and the many-candidates passes of the force kernels all get exercised).
clumping_parity.py [cells] [steps] > profiles/rNN_clumping_parity.json"""
import json
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_oracle  # noqa: E402
from yalla_amd import _ffi  # noqa: E402
from yalla_amd.solution import Solution  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
oracle, device = _ffi.bind(build_oracle()), _ffi.device_lib()
gs = 2 * (int((n / 0.64) ** (1 / 3) * 0.25) + 30)   # room: the collapsing system throws cells outwards
rows = []
with Solution("springs_grid", n, gs, 1.0, lib=oracle) as a, Solution("springs_grid", n, gs, 1.0, lib=device) as b:
    a.set_reduce_order(1)
    for s in (a, b):
        s.random_sphere(0.5, 9)
    done = 0
    while done < steps:
        for s in (a, b):
            s.take_step(0.001, 50)
        done += 50
        Xa, Xb = a.positions(), b.positions()
        same = bool(np.array_equal(Xa.view(np.uint32), Xb.view(np.uint32)) and
                    np.array_equal(a.old_v()[:n].view(np.uint32), b.old_v()[:n].view(np.uint32)))
        tree = cKDTree(Xa[:, :3].astype(np.float64))
        pairs = (tree.count_neighbors(tree, 1.0) - n) / n
        rows.append({"steps": done, "pairs_inside_cutoff_per_cell": round(float(pairs), 1), "bit_identical": same})
        print(rows[-1], file=sys.stderr, flush=True)
        if not same:
            break
print(json.dumps({"cells": n, "model": "springs_grid, dt 0.001, random_sphere(0.5) seed 9", "rows": rows,
                  "all_bit_identical": all(r["bit_identical"] for r in rows)}, indent=1))
