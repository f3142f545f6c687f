"""slab_long_run.py in chunks: after every `chunk` steps the slabs' positions against the undivided system's and
the guard's books (selections so far, those the guard asked for, moved / predicted maxima).  Beside them a CONTROL:
a second undivided system whose start differs by one rounding (every coordinate moved to a neighbouring binary32
with probability 1/2) -- how fast two runs of the same system drift apart by themselves.  A many-cell system is
chaotic: only while the control stays small does "the slabs agree with the undivided system" mean anything."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_slab
from yalla_amd import _ffi, slab as slab_mod
from yalla_amd.solution import Solution

n = int(sys.argv[1]); world = int(sys.argv[2]); steps = int(sys.argv[3]); dt = float(sys.argv[4])
every = int(sys.argv[5]); chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 10
model = sys.argv[7] if len(sys.argv) > 7 else "fading_grid"
device = _ffi.device_lib()
gs = int(2 * ((n / 0.64) ** (1 / 3) * 0.25 + 6))
with Solution(model, n, gs, 1.0, lib=device) as whole:
    whole.random_sphere(0.5, 3)
    X0 = whole.h_X[:n].copy()
    twin = Solution(model, n, gs, 1.0, lib=device)
    rng = np.random.default_rng(1)
    nudged = np.where(rng.random(X0.shape) < 0.5, np.nextafter(X0, np.float32(np.inf)), X0).astype(np.float32)
    twin.h_X[:n] = nudged
    twin.h_n = n
    twin.copy_to_device()
    plan = slab_mod.slab_plan(X0, world, 1.0, device)
    slabs = [slab_mod.Slab(model, X0, r, world, gs, lib=device, plan=plan) for r in range(world)]
    done = 0
    while done < steps:
        k = min(chunk, steps - done)
        whole.take_step(dt, k)
        Xref = whole.positions()
        twin.take_step(dt, k)
        control = np.abs(twin.positions() - Xref).max(axis=1)
        try:
            slab_mod.run_slabs(slabs, dt, k, every, device_memory=True)
        except Exception as err:
            print("stopped in steps", done, "..", done + k, getattr(err, "all_codes", None))
            for s in slabs:
                print(" rank", s.rank, "info", s.info(), "guard", s.guard_state(), "own", s.n_own(), "local", s.n_local())
            break
        done += k
        X = np.zeros_like(X0)
        for s in slabs:
            gid, Xr = s.own_cells()
            X[gid] = Xr
        diff = np.abs(X - Xref).max(axis=1)
        worst = int(diff.argmax())
        print(json.dumps({"steps": done, "control_max_diff": float(control.max()),
                          "control_beyond_1e-5": int((control > 1e-5 * np.abs(Xref).max()).sum()), "max_diff": float(diff.max()), "beyond_1e-5": int((diff > 1e-5 * np.abs(Xref).max()).sum()),
                          "worst_cell": worst, "worst_z": float(Xref[worst, 2]), "speed_max_per_step": float(np.abs(Xref - X0).max() / done),
                          "selections": [s.info()[0] for s in slabs][:3], "guard_asked": [s.info()[1] for s in slabs][:3],
                          "guard_moved_pred": [tuple(round(v, 4) for v in s.guard_state()) for s in slabs][:3]}), flush=True)
