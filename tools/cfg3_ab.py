#!/usr/bin/env python3
"""Config 3's model (branching: 7-float cells, bending_force, Turing kinetics, atomic neighbour
counters) frozen at several sizes, default kernel against grid_force_coop (force_variant 3) with the
lanes per cell chosen from n or forced: python tools/cfg3_ab.py [lanes ...]   (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import branching_case
from yalla_amd import _ffi

dev = _ffi.device_lib()
forced = [int(a) for a in sys.argv[1:]]
for n0 in (10_000, 30_000, 100_000, 200_000):
    for variant, lanes in [(2, 0), (3, 0)] + [(3, l) for l in forced]:
        s, _ = branching_case.setup(dev, n_0=n0, n_max=int(n0 * 1.4))
        s.set_param("prolif_rate", 0.0)
        s.set_param("force_variant", variant)
        s.set_param("coop_lanes", lanes)
        s.take_step(0.2, 3)
        s.synchronize()
        t0 = time.perf_counter()
        s.take_step(0.2, 22)
        s.synchronize()
        el = time.perf_counter() - t0
        print("branching", n0, "variant", variant, "lanes", lanes or "auto", "%.3g c-u/s %.1f us/step" % (n0 * 22 / el, el / 22 * 1e6))
        s.close()
