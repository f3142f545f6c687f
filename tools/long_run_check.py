#!/usr/bin/env python3
"""Resource check over a long run (GPU box): 3 * 10^5 take_steps of small systems on both solvers,
host RSS and free device memory before and after -- a per-step leak of a few hundred bytes shows."""
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yalla_amd.solution import Solution


def snapshot():
    free, _ = torch.cuda.mem_get_info()
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, free / 2 ** 20


for model, n in (("springs_grid", 2000), ("springs_tile", 300), ("sorting_grid", 2000)):
    with Solution(model, n, 50, 1.0) as s:
        s.random_sphere(0.8, 3)
        if model.startswith("sorting"):
            s.set_param("n_cells", n)
        s.take_step(0.0005, 2000)
        s.synchronize()
        rss0, free0 = snapshot()
        t0 = time.perf_counter()
        for _ in range(10):
            s.take_step(0.0005, 10000)
        s.synchronize()
        rss1, free1 = snapshot()
        print(f"{model} {n} cells: 100000 steps in {time.perf_counter() - t0:.1f} s, host RSS {rss0:.0f} -> {rss1:.0f} MiB, "
              f"free device memory {free0:.0f} -> {free1:.0f} MiB")
        assert rss1 - rss0 < 8 and free0 - free1 < 8, "resources grew over 10^5 steps"
print("LONG RUN OK")
