"""Why a cell of a z-slab run differs from the undivided run (test infrastructure; runs on the oracle
backend, whose pair trace -- oracle/yalla_host.hpp, Pair_trace -- writes down every candidate pair of chosen
cells with the stage it was tested in).

Both runs compute the same arithmetic except for the order in which the centre-of-mass sum is associated, so
positions agree to ~1e-7 relative -- until a pair whose distance is within that of the force's cut-off
(`dist >= cube_size -> skip`, reference solvers.cuh:450) interacts in one run and not in the other.  For the
springs model, whose force does not vanish at the cut-off, the two cells then jump by ~dt / 2 (a FLIP); one
or two steps later the cells that average those two cells' velocities (friction_w_neighbour) follow at ~1e-4
(FOLLOWERS).  Anything else -- a divergent cell with identical hit sets and no divergent partner -- would be a
fault of the decomposition (UNEXPLAINED), e.g. a lost neighbour across a cut.

    explain(oracle_lib, n, world, steps, dt, migrate_every) -> report (dict), text lines
"""
import ctypes as C
import threading

import numpy as np

from yalla_amd import slab as slab_mod
from yalla_amd.solution import Solution


# The order of the centre-of-mass sums: 1 = the device's tree (B blocks x 256 lanes, folded by halving; the
# oracle's YA_REDUCE_TREE) -- with it the oracle's runs, undivided and in slabs, are the device's bit for bit,
# so what is explained here is what the device did; 0 = the reference-like serial sum.
REDUCE_ORDER = 1


def _trace(lib, ids, margin, run):
    """run() with the pair trace armed for `ids`; returns {(call, i): {j: dist}}."""
    lib.ya_oracle_trace_begin.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_float]
    lib.ya_oracle_trace_read.argtypes = [C.POINTER(C.c_int), C.c_int]
    arr = np.ascontiguousarray(ids, dtype=np.int32)
    assert lib.ya_oracle_trace_begin(arr.ctypes.data_as(C.POINTER(C.c_int)), len(arr), float(margin)) == 0
    try:
        run()
        count = lib.ya_oracle_trace_read(None, 0)
        rows = np.zeros((max(count, 1), 4), np.int32)
        assert lib.ya_oracle_trace_read(rows.ctypes.data_as(C.POINTER(C.c_int)), count) == count
    finally:
        lib.ya_oracle_trace_end()
    out = {}
    dist = rows[:count, 3].copy().view(np.float32)
    for (call, i, j), d in zip(rows[:count, :3].tolist(), dist.tolist()):
        out.setdefault((call, i), {})[j] = d
    return out


def undivided_snapshots(lib, n, gs, dist, seed, dt, steps, model):
    """X0 and the positions after every step of the undivided system."""
    with Solution(model, n, gs, 1.0, lib=lib) as s:
        s.set_reduce_order(REDUCE_ORDER)
        s.random_sphere(dist, seed)
        if model.startswith("sorting"):
            s.set_param("n_cells", n)
        X0 = s.h_X[:n].copy()
        snaps = []
        for _ in range(steps):
            s.take_step(dt, 1)
            snaps.append(s.positions().copy())
    return X0, snaps


def slab_snapshots(lib, X0, world, gs, dt, steps, migrate_every, model):
    """The positions by global id after every step of the same system in `world` slabs (a host thread per slab,
    yalla_amd.slab.run_slabs' schedule: migration every `migrate_every`-th step and after the last one)."""
    plan = slab_mod.slab_plan(X0, world, 1.0, lib)
    slabs = [slab_mod.Slab(model, X0, r, world, gs, lib=lib, plan=plan) for r in range(world)]
    for s in slabs:
        s.sim.set_reduce_order(REDUCE_ORDER)
        if model.startswith("sorting"):
            s.sim.set_param("n_cells", len(X0))
    shared = slab_mod.ThreadTransport.Shared(world, False)
    for r, s in enumerate(slabs):
        s.use(transport=slab_mod.ThreadTransport(shared, r))
    snaps = [np.full_like(X0, np.nan) for _ in range(steps)]
    errors = []

    def work(s):
        try:
            for k in range(steps):
                s.step(dt, migrate=(k + 1) % migrate_every == 0 or k == steps - 1)
                gid, Xr = s.own_cells()     # every cell has exactly one owner: the threads write disjoint rows
                snaps[k][gid] = Xr
        except Exception as err:
            errors.append(err)
            shared.barrier.abort()

    threads = [threading.Thread(target=work, args=(s,)) for s in slabs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s in slabs:
        s.close()
    if errors:
        raise errors[0]
    assert not any(np.isnan(x).any() for x in snaps), "a cell was lost"
    return snaps


def explain(lib, n, world, steps, dt, migrate_every, model="springs_grid", gs=50, cut_off=1.0, tol=1e-5, log=None,
            dist=0.5, seed=3):
    """Follows one slab case step by step: ONE undivided and ONE decomposed run, positions compared after
    every step; then both runs again with every candidate pair of the divergent cells written down, and each
    cell looked at in the step that first leaves it further than `tol` relative from the undivided run."""
    say = log or (lambda *a: None)
    assert lib.ya_models_is_device() == 0, "the pair trace lives in the oracle"
    X0, ref_snaps = undivided_snapshots(lib, n, gs, dist, seed, dt, steps, model)
    dec_snaps = slab_snapshots(lib, X0, world, gs, dt, steps, migrate_every, model)
    first_off, diff_at = {}, {}
    for k in range(1, steps + 1):
        diff = np.abs(dec_snaps[k - 1] - ref_snaps[k - 1]).max(axis=1)
        scale = np.abs(ref_snaps[k - 1]).max()
        off = np.nonzero(diff > tol * scale)[0]
        for i in off.tolist():
            if i not in first_off:
                first_off[i], diff_at[i] = k, float(diff[i])
        say(f"after {k} steps: {len(off)} cells beyond {tol:g} relative, max {diff.max():.2e}")
    del ref_snaps, dec_snaps
    flips, followers, unexplained = [], [], []
    if first_off:
        cells = sorted(first_off)
        ref = _trace(lib, cells, 1e-3, lambda: undivided_snapshots(lib, n, gs, dist, seed, dt, steps, model))
        dec = _trace(lib, cells, 1e-3, lambda: slab_snapshots(lib, X0, world, gs, dt, steps, migrate_every, model))
    for i in sorted(first_off, key=lambda c: (first_off[c], c)):
        k = first_off[i]
        found = []
        for call in (2 * k - 2, 2 * k - 1):   # the two stages of step k
            a, b = ref.get((call, i), {}), dec.get((call, i), {})
            hits_a = {j for j, d in a.items() if d < cut_off}
            hits_b = {j for j, d in b.items() if d < cut_off}
            for j in sorted(hits_a ^ hits_b):
                found.append({"stage": call - (2 * k - 2) + 1, "partner": j, "dist_undivided": a.get(j),
                              "dist_slabs": b.get(j)})
        if found:
            flips.append({"step": k, "cell": i, "diff": diff_at[i], "pairs": found})
            for f in found:
                say(f"step {k}: cell {i} diff {diff_at[i]:.2e}  FLIP in stage {f['stage']}: partner {f['partner']} at "
                    f"{f['dist_undivided']!r} undivided, {f['dist_slabs']!r} in slabs (cut-off {cut_off:g})")
            continue
        # same hit sets in both stages: whose velocities does it average?
        last = ref.get((2 * k - 1, i), {})
        earlier = sorted(j for j, d in last.items() if d < cut_off and first_off.get(j, k) < k)
        if earlier:
            followers.append({"step": k, "cell": i, "diff": diff_at[i],
                              "partners_off_earlier": [(j, first_off[j]) for j in earlier]})
            say(f"step {k}: cell {i} diff {diff_at[i]:.2e}  FOLLOWER: same hit sets in both runs; interacts with "
                + ", ".join(f"{j} (off since step {first_off[j]})" for j in earlier))
        else:
            unexplained.append({"step": k, "cell": i, "diff": diff_at[i]})
            say(f"step {k}: cell {i} diff {diff_at[i]:.2e}  UNEXPLAINED: same hit sets, no partner that differed earlier")
    return {"n": n, "world": world, "steps": steps, "dt": dt, "migrate_every": migrate_every, "model": model,
            "cells_beyond_tol": len(first_off), "flips": flips, "followers": followers, "unexplained": unexplained}
