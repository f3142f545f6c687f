"""One case of tests/fuzz_slab.py looked at pair by pair: slab_case.py seed [n world steps dt migrate_every [grid_size sphere_seed]]
-- which cells differ from the undivided run, in which step first, and WHY: the pair that interacts in one run
and not in the other, with its distance in both (tests/slab_explain.py, the oracle's pair trace).  Runs on the
CPU (oracle backend); without the five numbers the case is drawn from the seed as tests/fuzz_slab.py draws it."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slab_explain
from yalla_amd import _ffi

seed = int(sys.argv[1])
gs, sphere_seed = 50, 3   # what tests/fuzz_slab.py runs its cases with
if len(sys.argv) >= 7:
    n, world, steps, dt, every = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
    if len(sys.argv) >= 9:
        gs, sphere_seed = int(sys.argv[7]), int(sys.argv[8])
else:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(500, 60000)); world = int(rng.integers(1, 7)); steps = int(rng.integers(1, 13))
    dt = float(rng.choice([0.001, 0.004])); every = int(rng.choice([1, 2, 4]))
print(dict(n=n, world=world, steps=steps, dt=dt, migrate_every=every, seed=seed, grid_size=gs, sphere_seed=sphere_seed,
           backend="oracle"), flush=True)
oracle = _ffi.bind(os.path.join(ROOT, "oracle", "_build", "liboracle_models.so"))
# SLAB_CASE_TOL: the relative distance from the undivided run at which a cell is looked at (default 1e-5, the
# suite's criterion).  A flipped pair moves two cells by 0.5 dt: in a big system with a small step that is BELOW
# 1e-5 of the system's extent (10 M cells, dt 0.001: 5e-4 against 6.2e-4), the cell crosses the criterion only
# steps later and the pair's distances have drifted apart by then -- look closer (1e-6) to catch the flip itself.
tol = float(os.environ.get("SLAB_CASE_TOL", "1e-5"))
report = slab_explain.explain(oracle, n, world, steps, dt, every, gs=gs, seed=sphere_seed, tol=tol,
                              log=lambda *a: print(*a, flush=True))
pairs = {tuple(sorted((f["cell"], p["partner"]))) for f in report["flips"] for p in f["pairs"]}
print(f"{report['cells_beyond_tol']} cells beyond {tol:g}: {len(report['flips'])} in {len(pairs)} pairs at the cut-off "
      f"{sorted(pairs)}, {len(report['followers'])} followers, {len(report['unexplained'])} UNEXPLAINED")
print(json.dumps({k: report[k] for k in ("n", "world", "steps", "dt", "migrate_every", "cells_beyond_tol")} |
                 {"pairs_at_cut_off": sorted(pairs), "followers": len(report["followers"]),
                  "unexplained": [u["cell"] for u in report["unexplained"]]}))
sys.exit(1 if report["unexplained"] else 0)
