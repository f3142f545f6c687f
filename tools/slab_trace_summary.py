#!/usr/bin/env python3
"""Device-busy time per slab and step from a kernel trace of tools/slab_rehearsal run with
YALLA_REHEARSAL_MARKERS=1 (a marker kernel `slab_takes_the_gpu<r>` whenever slab r takes the GPU):

    python tools/slab_trace_summary.py <k_kernel_trace.csv> <steps incl. warm-up> [skip_steps]

Per slab: the union of its kernels' intervals (overlapping launches counted once), split into the
force kernel and everything else, per step; and the same for the undivided system that the
program steps first.  This is the compute side of a rank's step WITHOUT host launch gaps and
without the rehearsal's device synchronisations -- the lower bracket of the projection, next to
the program's own wall-clock figures (the upper one)."""
import collections
import csv
import json
import re
import sys


def union(intervals):
    total, end = 0, -1
    for a, b in sorted(intervals):
        if a > end:
            total += b - a
            end = b
        elif b > end:
            total += b - end
            end = b
    return total


def main(path, steps, skip=0):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    first_marker = next(i for i, r in enumerate(rows) if "slab_takes_the_gpu" in r[2])
    whole = rows[:first_marker]
    last_raw = max(i for i, r in enumerate(whole) if "heun_step_raw" in r[2])
    whole = whole[: last_raw + 1]
    n_whole_steps = sum("heun_step_raw" in r[2] for r in whole)
    per_rank = collections.defaultdict(list)
    rank = None
    for a, b, name in rows[first_marker:]:
        m = re.search(r"slab_takes_the_gpu<(\d+)>", name) or re.search(r"slab_takes_the_gpuILi(\d+)E", name)
        if m:
            rank = int(m.group(1))
            continue
        if "slab_hands_the_gpu_back" in name:
            rank = "transport"   # the rehearsal's own copies between the slabs' buffers: nobody's work
            continue
        per_rank[rank].append((a, b, name))
    out = {"undivided_device_ms_per_step": union([(a, b) for a, b, _ in whole]) / n_whole_steps / 1e6,
           "undivided_steps": n_whole_steps, "slabs": []}
    worst = worst_kernels = 0
    transport = per_rank.pop("transport", [])
    out["rehearsal_transport_ms_per_step_all_slabs"] = sum(b - a for a, b, _ in transport) / steps / 1e6
    for r in sorted(per_rank):
        ks = per_rank[r]
        busy = union([(a, b) for a, b, _ in ks]) / steps / 1e6
        # without the copy / fill kernels: nearly all of them are the REHEARSAL's transport -- a message as a
        # device-to-device copy, the all-reduce through the host -- which RCCL replaces on a stream of its own;
        # the slab's own few (two counts per selection, a zeroed guard) go with them (< 2 us per step)
        busy_kernels = union([(a, b) for a, b, n in ks if "copyBuffer" not in n and "fillBuffer" not in n]) / steps / 1e6
        force = union([(a, b) for a, b, n in ks if "grid_force" in n]) / steps / 1e6
        copies = sum(b - a for a, b, n in ks if "copyBuffer" in n or "fillBuffer" in n) / steps / 1e6
        by_kernel = collections.defaultdict(lambda: [0, 0])
        for a, b, n in ks:
            short = re.sub(r"^.*?(k_\w+|grid_force_\w+?|\w+_step\w*|ghosts_into_sorted|copyBuffer|fillBuffer\w*)\b.*$", r"\1",
                           re.sub(r"I.*$", "", n.replace("_ZN2ya15", "").replace("void ", "")) if n.startswith("_ZN2ya15")
                           else n)
            by_kernel[short][0] += b - a
            by_kernel[short][1] += 1
        # the force launches by themselves: boundary and interior launch of a stage start a few microseconds
        # apart (the shorter one of such a pair is the boundary launch)
        fl = sorted((a, b) for a, b, n in ks if "grid_force" in n)
        pairs = [(fl[k], fl[k + 1]) for k in range(0, len(fl) - 1, 2) if fl[k + 1][0] - fl[k][0] < 50_000]
        first = [x[1] - x[0] for x, y in pairs]
        second = [y[1] - y[0] for x, y in pairs]
        spans = [max(x[1], y[1]) - x[0] for x, y in pairs]
        mean = lambda v: round(sum(v) / max(len(v), 1) / 1e3, 1)
        out["slabs"].append({"rank": r, "device_ms_per_step": busy, "device_ms_per_step_without_copies": busy_kernels,
                             "force_ms_per_step": force,
                             "force_stage_us": {"first_launch": mean(first), "second_launch": mean(second),
                                                "span": mean(spans), "stages": len(pairs),
                                                "span_by_stage": [round(v / 1e3) for v in spans]},
                             "copy_fill_ms_per_step": copies, "launches_per_step": len(ks) / steps,
                             "kernels_us_per_step_and_launches_per_step": {
                                 k: [round(v[0] / steps / 1e3, 1), round(v[1] / steps, 2)]
                                 for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][0])}})
        worst = max(worst, busy)
        worst_kernels = max(worst_kernels, busy_kernels)
    # (SLAB_TIMELINE_RANK=r: one stretch of that slab's launches -- start, duration, kernel -- on stderr)
    import os
    if os.environ.get("SLAB_TIMELINE_RANK"):
        per_rank["transport"] = transport
        for which in os.environ["SLAB_TIMELINE_RANK"].split(","):
            ks = per_rank[which if which == "transport" else int(which)]
            lo = len(ks) // 2
            print(f"---- rank {which}", file=sys.stderr)
            for a, b, n in ks[lo:lo + 140]:
                print(f"{(a - ks[lo][0]) / 1e3:10.1f} {(b - a) / 1e3:8.1f}  {n[:70]}", file=sys.stderr)
    out["slowest_slab_device_ms_per_step"] = worst
    out["projected_speedup_device_time_only"] = out["undivided_device_ms_per_step"] / worst
    out["slowest_slab_device_ms_per_step_without_copies"] = worst_kernels
    out["projected_speedup_device_time_only_without_copies"] = out["undivided_device_ms_per_step"] / worst_kernels
    out["note"] = ("device-busy time only (no host launch gaps, no RCCL latency, no xGMI transfer time; the rehearsal's "
                   "own copies between the slabs' buffers -- its stand-in for RCCL -- run after the marker "
                   "slab_hands_the_gpu_back and are attributed to no slab; a slab's own copy kernels, a few per "
                   "selection of the mirrored cells, are in its device time and listed under copy_fill): the "
                   "optimistic bracket of the projection")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0)
