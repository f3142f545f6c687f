#!/usr/bin/env python3
"""Summary of one grid_force_bits launch traced by tools/micro/force_trace.hip (a -DYA_BITS_TRACE
build stamps s_memtime at every workgroup's entry and exit, HW_ID and XCC_ID):

    force_trace_summary.py stamps.csv[.gz] [more ...] > profiles/rNN_force_trace.json

s_memtime counters differ from CU to CU, so every CU's stamps are taken relative to the first
workgroup that CU started.  Prints one JSON object per file: workgroups per CU, resident workgroups
per CU, the number of workgroups in flight over twenty equal time bins, lifetimes by start decile."""
import gzip
import json
import sys

import numpy as np


def summary(path):
    opener = gzip.open if path.endswith(".gz") else open
    lines = opener(path, "rt").read().split("\n")
    head = dict(zip(lines[0].split()[1::2], lines[0].split()[2::2]))
    d = np.array([list(map(int, l.split(","))) for l in lines[2:] if l])
    block, t0, t1, hw, xcc, tile = d.T
    xcd, cu, se = xcc & 0xf, (hw >> 8) & 0xf, (hw >> 13) & 0x7
    key = xcd * 1000 + se * 100 + cu
    start = np.zeros_like(t0)
    for k in np.unique(key):
        start[key == k] = t0[key == k].min()
    s, e = t0 - start, t1 - start
    life = e - s
    cus = np.unique(key)
    per_cu = np.array([np.sum(key == k) for k in cus])
    resident = [int(max(np.sum((s[key == k] <= t) & (e[key == k] > t))
                        for t in np.linspace(0, e[key == k].max(), 80))) for k in cus]
    last_end = np.array([e[key == k].max() for k in cus])
    edges = np.linspace(0, e.max(), 21)
    in_flight = [int(np.sum((s <= (a + b) / 2) & (e > (a + b) / 2))) for a, b in zip(edges[:-1], edges[1:])]
    order = np.argsort(s, kind="stable")
    deciles = []
    for q in range(10):
        idx = order[q * len(order) // 10:(q + 1) * len(order) // 10]
        deciles.append({"mean_start": int(s[idx].mean()), "mean_lifetime": int(life[idx].mean()),
                        "p5": int(np.percentile(life[idx], 5)), "p95": int(np.percentile(life[idx], 95))})
    slots = int(np.max(resident)) * len(cus)
    return {
        "file": path.split("/")[-1], "cells": int(head.get("cells", 0)), "workgroups": int(len(block)),
        "launch_us_hip_events": float(head.get("launch_us", 0)),
        "block_mod_8_is_the_xcd": bool(np.all(block % 8 == xcd)),
        "compute_units": int(len(cus)), "workgroups_per_cu_min_median_max":
            [int(per_cu.min()), int(np.median(per_cu)), int(per_cu.max())],
        "resident_workgroups_per_cu_max": int(np.max(resident)), "wavefront_slots": slots,
        "cycles_unit": "s_memtime ticks, per CU from that CU's first workgroup",
        "last_end_per_cu_min_median_max": [int(last_end.min()), int(np.median(last_end)), int(last_end.max())],
        "mean_lifetime": int(life.mean()),
        "span_if_always_full": int(life.sum() / slots),
        "in_flight_over_20_bins_of": int(e.max() / 20), "in_flight": in_flight,
        "lifetime_by_start_decile": deciles,
        "per_xcd_sum_of_lifetimes_M": [round(float(life[xcd == x].sum()) / 1e6, 1) for x in range(8)],
    }


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(json.dumps(summary(p)))
