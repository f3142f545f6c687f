#!/usr/bin/env python3
"""Config 1 (springs, Tile_solver) on the GPU box: python tools/tile_cfg1.py [n lanes]  (all pairs of
n cells, Tile_computer::lanes_per_cell = lanes; without arguments a small table)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yalla_amd import _ffi
from yalla_amd.solution import Solution

dev = _ffi.device_lib()
cases = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else \
    [(n, lanes) for n in (500, 800, 2000) for lanes in (1, 16, 64)]
for n, lanes in cases:
    with Solution("springs_tile", n, lib=dev) as s:
        s.set_param("tile_lanes", lanes)
        s.random_sphere(0.5, 42)
        s.take_step(0.001, 3)
        s.synchronize()
        t0 = time.perf_counter()
        s.take_step(0.001, 100)
        s.synchronize()
        el = time.perf_counter() - t0
        print(n, lanes, "%.3g c-u/s" % (n * 100 / el), "%.1f us/step" % (el / 100 * 1e6))
