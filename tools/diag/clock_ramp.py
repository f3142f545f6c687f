"""Is the headline's timed region (20 steps = 12 ms, started 3 steps after the host generated the system) timing
a clock ramp?  VERDICT r05 weak 8: the force launch takes 246 us in the headline pass and 207 us in the sustained
pass on one box.

One process, one 10^6-cell springs system at dt = 0 (every step the same work).  After `idle` seconds of an idle
device (host sleep), take_steps one chunk of 2 at a time; every chunk's wall time (synchronised) and the shader clock
(ya_shader_clock_mhz, 100 us probe after the chunk) are recorded against the time since the first launch.

    clock_ramp.py [cells] > profiles/r06_clock_ramp.json"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yalla_amd import _ffi  # noqa: E402
from yalla_amd.solution import Solution  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gs = 64 if n == 1_000_000 else int(2 * ((n / 0.64) ** (1 / 3) * 0.25 + 4))
core = C.CDLL(_ffi.CORE_LIB, mode=C.RTLD_LOCAL)
core.ya_shader_clock_mhz.argtypes = [C.c_double, C.POINTER(C.c_double)]


def clock():
    mhz = C.c_double()
    return mhz.value if core.ya_shader_clock_mhz(100.0, C.byref(mhz)) == 0 else None


runs = []
with Solution("springs_grid", n, gs, 1.0) as sim:
    sim.random_sphere(0.5, 42)
    sim.take_step(0.0, 5)
    sim.synchronize()
    for idle in (0.0, 0.003, 0.03, 0.3, 2.0, 0.3):
        time.sleep(idle)
        series = []
        start = time.perf_counter()
        while time.perf_counter() - start < 0.6:
            t0 = time.perf_counter()
            sim.take_step(0.0, 2)
            sim.synchronize()
            t1 = time.perf_counter()
            series.append((round((t0 - start) * 1e3, 2), round((t1 - t0) / 2 * 1e3, 4)))
        mhz = clock()
        # summary: the mean ms per step in windows after the first launch
        def window(lo, hi):
            v = [ms for t, ms in series if lo <= t < hi]
            return round(sum(v) / len(v), 4) if v else None
        runs.append({"idle_before_s": idle,
                     "ms_per_step_0_5ms": window(0, 5), "ms_per_step_5_15ms": window(5, 15),
                     "ms_per_step_15_50ms": window(15, 50), "ms_per_step_50_150ms": window(50, 150),
                     "ms_per_step_150_300ms": window(150, 300), "ms_per_step_300_600ms": window(300, 600),
                     "clock_mhz_at_end": mhz, "first_20_chunks": series[:20]})
        print(runs[-1], file=sys.stderr, flush=True)
    # the same with the probe beside every chunk (does the probe itself see the ramp?)
    time.sleep(2.0)
    ramp = []
    start = time.perf_counter()
    while time.perf_counter() - start < 0.3:
        sim.take_step(0.0, 2)
        ramp.append((round((time.perf_counter() - start) * 1e3, 2), clock()))
    runs.append({"idle_before_s": 2.0, "clock_mhz_series_first_40": ramp[:40], "clock_mhz_last": ramp[-1]})
print(json.dumps({"cells": n, "what": "take_step(dt = 0) in chunks of 2 after an idle period: ms per step against the "
                                      "time since the first launch (synchronised per chunk)", "runs": runs}, indent=1))
