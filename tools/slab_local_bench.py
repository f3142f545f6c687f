#!/usr/bin/env python3
"""W slabs of one n-cell system on ONE GPU (LocalComm): the device work of the z-slab path,
ghost layers included, without RCCL.  Prints ms per step summed over the W slabs, next
to the undivided system's."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from yalla_amd import device_lib, slab as slab_mod
from yalla_amd.solution import Solution

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=1_000_000)
ap.add_argument("--world", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--migrate-every", type=int, default=4)
a = ap.parse_args()
lib = device_lib()
n = a.cells
radius = (n / 0.64) ** (1 / 3) * 0.25
gs = max(2 * (int(radius) + 3), 8)
with Solution("springs_grid", n, gs, 1.0, lib=lib) as s:
    s.random_sphere(0.5, 42)
    X0 = s.h_X[:n].copy()
    s.take_step(0.001, a.warmup); s.synchronize()
    t0 = time.perf_counter(); s.take_step(0.001, a.steps); s.synchronize()
    plain = (time.perf_counter() - t0) / a.steps * 1e3
bounds = slab_mod.slab_bounds(X0[:, 2], a.world)
slabs = [slab_mod.Slab("springs_grid", X0, r, a.world, bounds, gs, lib=lib, device="hip")
         for r in range(a.world)]
comm = slab_mod.LocalComm()
k = [0]
def advance(steps):
    for _ in range(steps):
        k[0] += 1
        slab_mod.step(slabs, comm, 0.001, migrate=k[0] % a.migrate_every == 0)
advance(a.warmup); slabs[0].sim.synchronize()
t0 = time.perf_counter(); advance(a.steps); slabs[0].sim.synchronize()
sl = (time.perf_counter() - t0) / a.steps * 1e3
print(json.dumps({"cells": n, "world": a.world, "plain_ms_per_step": plain, "slabs_ms_per_step_total": sl,
                  "n_local": [s.n_local for s in slabs], "n_own": [s.n_own() for s in slabs]}))
