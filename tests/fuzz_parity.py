#!/usr/bin/env python3
"""Randomised bit-exact parity sweep, device engine vs CPU oracle (test infrastructure;
run on a GPU box: python tests/fuzz_parity.py [cases] [first_seed]).  Draws system size,
density, cube size, grid size, functor, point type, time step and step count; for every case
positions, old velocities and the four grid arrays must match bit for bit.  The pytest
suite runs a short fixed slice of the same generator (test_parity_gpu.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from yalla_amd.solution import Solution

MODELS = ["springs_grid", "clipped_grid", "relu_grid", "relu_po_grid",   # + - * / sqrt only
          "springs_tile", "clipped_tile", "relu_tile", "relu_po_tile"]


def draw(seed):
    rng = np.random.default_rng(seed)
    model = MODELS[rng.integers(len(MODELS))]
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 257, 300, 1000, 3000, 7000, 20000]))
    if rng.random() < 0.5:
        n = int(rng.integers(1, 9000))
    if model.endswith("_tile"):
        n = min(n, int(rng.integers(1, 1500)))  # all pairs on the CPU
    dist = float(rng.choice([0.08, 0.15, 0.3, 0.5, 0.75, 1.2, 2.5]))
    cs = float(rng.choice([0.5, 1.0, 1.0, 1.7]))
    radius = (n / 0.64) ** (1 / 3) * dist / 2
    # room to move: dense spring systems overshoot by tens of units within a few steps
    gs = int(2 * (int((radius + 30) / cs) + 2) + rng.integers(0, 3))
    while gs > 256:  # YA_MAX_GRID_SIZE: binary32 cube ids are exact only up to 256^3 cubes
        cs *= 2
        gs = int(2 * (int((radius + 30) / cs) + 2))
    # very dense systems push hard (hundreds of overlapping neighbours): small steps
    dt = 1e-4 if dist < 0.3 else float(rng.choice([0.001, 0.01] if dist < 0.75 else [0.001, 0.01, 0.05]))
    steps = int(rng.integers(1, 4))
    # force kernel of the device engine: mostly the default bit-stream kernel, sometimes the
    # byte-FIFO one or the direct one (drawn last: the earlier draws keep their round-1 values)
    variant = int(rng.choice([2, 2, 2, 1, 0])) if model.endswith("_grid") else 2
    # then (drawn after everything else again): a third of the grid cases through grid_force_coop
    # with 4, 8 or 16 lanes per cell, and grid_force_bits with old_v staged in LDS or not
    lanes = int(rng.choice([0, 0, 0, 0, 4, 8, 16, 16]))
    stage_v = bool(rng.random() < 0.5)
    if lanes and model.endswith("_grid"):
        variant = 3
    # round 3 (drawn after everything else again): a third of the grid cases leave the choice to the
    # engine (force_variant -1: the models' functors are declared stateless, so small systems go to
    # grid_force_coop with the lanes chosen from n); Tile_solver cases always do (lanes_per_cell 0)
    if model.endswith("_grid") and rng.random() < 0.33:
        variant, lanes = -1, 0
    # round 5 (drawn after everything else): grid_force_bits with the last 1 ... 20 tiles of its launches as
    # half tiles in half of the cases (the kernel keeps a list of fewer than four tails whole)
    tail = int(rng.integers(1, 21)) if rng.random() < 0.5 else 0
    # ... and (drawn after that) a third of the cases each: that draw, the engine's own choice (-1: at these sizes
    # every tile as halves), every tile as halves by the knob
    tail = [tail, -1, 1 << 20][int(rng.integers(0, 3))]
    # round 6 (drawn last): the summation order of the grid force, in BOTH libraries -- the reference's one
    # running sum (Grid_computer::sum_order 0, the default) in two thirds of the cases, by plane (1: the only
    # order in which the drawn tail of half tiles exists) in the others
    sum_order = int(rng.random() < 0.34)
    return dict(model=model, n=n, gs=max(gs, 8), cs=cs, dist=dist, seed=int(seed), dt=dt, steps=steps,
                variant=variant, lanes=lanes if variant == 3 else 0, stage_v=stage_v, tail=tail, sum_order=sum_order)


def run_case(oracle, device, c):
    out = []
    for lib in (oracle, device):
        with Solution(c["model"], c["n"], c["gs"], c["cs"], lib=lib) as s:
            if c["model"].endswith("_grid"):
                assert s.set_param("sum_order", c.get("sum_order", 0)) == 0
            if lib is oracle:
                assert s.set_reduce_order(1) == 0
            elif c["model"].endswith("_grid"):
                s.set_param("force_variant", c.get("variant", 2))
                s.set_param("coop_lanes", c.get("lanes", 0))
                s.set_param("stage_v_max", 1 << 30 if c.get("stage_v") else 0)
                s.set_param("tail_tiles", c.get("tail", 0))
            s.random_sphere(c["dist"], c["seed"])
            if lib is oracle and c["model"].endswith("_grid"):
                # a drawn system may blow up (dense start, large dt) and leave the grid: the oracle
                # does not check, the engine aborts as the reference's D_ASSERT does
                # (solvers.cuh:361-362) -- such a case is no parity case
                half = c["gs"] // 2 * c["cs"]
                for _ in range(c["steps"]):
                    s.take_step(c["dt"], 1)
                    X = s.positions()[:, :3]
                    if not np.isfinite(X).all() or np.abs(X).max() >= half - 2 * c["cs"]:
                        return None
            else:
                s.take_step(c["dt"], c["steps"])
            out.append((s.positions(), s.old_v()[:c["n"]],
                        s.grid() if c["model"].endswith("_grid") else ()))
    (Xo, vo, go), (Xd, vd, gd) = out
    ok = np.array_equal(Xo.view(np.uint32), Xd.view(np.uint32)) and \
        np.array_equal(vo.view(np.uint32), vd.view(np.uint32))
    for name, a, b in zip(("cube_id", "point_id", "cube_start", "cube_end"), go, gd):
        if name in ("cube_id", "point_id"):
            a, b = a[:c["n"]], b[:c["n"]]
        ok = ok and np.array_equal(a, b)
    return ok


if __name__ == "__main__":
    from yalla_amd import _ffi
    from conftest import build_oracle
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    oracle, device = _ffi.bind(build_oracle()), _ffi.device_lib()
    import json
    bad = skipped = 0
    log = open(os.environ["FUZZ_LOG"], "w") if os.environ.get("FUZZ_LOG") else None  # one JSON line per case
    for seed in range(first, first + cases):
        c = draw(seed)
        if os.environ.get("FUZZ_VERBOSE"):
            print(c, flush=True)
        ok = run_case(oracle, device, c)
        if ok is None:
            skipped += 1
            if log:
                log.write(json.dumps(dict(c, skipped="left the grid on the oracle")) + "\n")
            continue
        if log:
            log.write(json.dumps(dict(c, bit_exact=bool(ok))) + "\n")
            log.flush()
        if not ok:
            bad += 1
            print("MISMATCH", c, flush=True)
    print(f"{cases - bad - skipped} of {cases - skipped} cases bit-exact ({skipped} drawn systems left their grid: skipped)")
    if log:
        log.write(json.dumps({"cases": cases, "first_seed": first, "skipped": skipped,
                              "bit_exact": cases - bad - skipped}) + "\n")
        log.close()
    sys.exit(1 if bad else 0)
