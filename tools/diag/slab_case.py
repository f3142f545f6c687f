"""One case of tests/fuzz_slab.py looked at closely: slab_case.py seed  -- which cells differ from the undivided
run, where they sit relative to the cuts, and whether each of them has a partner at the force's cut-off (the
signature of a pair that interacts in one run and not in the other) in the step where it first differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy.spatial import cKDTree
import test_slab
from yalla_amd import _ffi

seed = int(sys.argv[1])
device = _ffi.device_lib()
rng = np.random.default_rng(seed)
n = int(rng.integers(500, 60000)); world = int(rng.integers(1, 7)); steps = int(rng.integers(1, 13))
dt = float(rng.choice([0.001, 0.004])); every = int(rng.choice([1, 2, 4]))
print(dict(n=n, world=world, steps=steps, dt=dt, migrate_every=every, seed=seed))
first_off = {}
prev_ref = None
for k in range(1, steps + 1):
    X0, Xref = test_slab.reference_run(device, n, 50, 0.5, 3, dt, k)
    X, moved = test_slab.slab_run(device, X0, world, 50, dt, k, "hip", every)
    bounds = test_slab.slab_mod.slab_bounds(X0[:, 2], world)
    diff = np.abs(X - Xref).max(axis=1)
    scale = np.abs(Xref).max()
    off = np.nonzero(diff > 1e-5 * scale)[0]
    new = [i for i in off if i not in first_off]
    before = X0 if prev_ref is None else prev_ref
    tree = cKDTree(before[:, :3].astype(np.float64))
    for i in new:
        first_off[i] = k
        d, j = tree.query(before[i, :3].astype(np.float64), k=40)
        near_cut = [(int(jj), float(dd)) for dd, jj in zip(d, j) if abs(dd - 1.0) < 2e-5]
        print(f"step {k}: cell {i} diff {diff[i]:.2e} z {before[i,2]:.3f} nearest cut {np.abs(bounds[1:-1] - before[i,2]).min():.3f} "
              f"partners within 2e-5 of the cut-off (positions one step earlier): {near_cut}")
    prev_ref = Xref
    print(f"after {k} steps: {len(off)} cells beyond 1e-5, max {diff.max():.2e}", flush=True)
