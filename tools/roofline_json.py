#!/usr/bin/env python3
"""profiles/r02_counters.json from the rocprofv3 PMC passes of `python3 bench.py --no-cpu-baseline
--steps 10 --warmup 2` (tools/gpu_profile_round.sh): per-launch means for the dominant kernel.

    python tools/roofline_json.py gpurun_out/<tag> profiles/r02_counters.json

Definitions (every input is a raw counter kept in profiles/r02_pmc_1M_springs_grid.txt):
  kernel_cycles       GRBM_GUI_ACTIVE / 8                 (the counter sums the 8 XCDs)
  valu_insts_per_wave SQ_INSTS_VALU / SQ_WAVES
  valu_issue_frac     SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * kernel_cycles)   (2 = ideal wave64 issue on SIMD-32)
  valu_rate_frac      the same with the SUSTAINED non-packed rate measured on this chip by
                      tools/micro/halfwave.hip (profiles/r02_valu_rate_probe.json: 2.78 cycles per
                      wave64 fma at 2.4 GHz) -- how close the kernel is to the VALU rate the chip delivers
  lanes_active_frac   SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  lds_conflict_frac   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  lds_busy_frac       SQ_LDS_IDX_ACTIVE / (256 CUs * kernel_cycles)
  wait_frac           SQ_WAIT_ANY / SQ_WAVE_CYCLES         (share of wave time parked at s_waitcnt / barriers)
FETCH_SIZE / WRITE_SIZE are KiB per launch from their own passes; bench.py doubles FETCH_SIZE as
MI355X_MICROARCH.md prescribes."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "grid_force_bits"
SUSTAINED_CYCLES_PER_VALU = 2.78


def kernel_source_sha():
    h = hashlib.sha256()
    for rel in ("include/solvers.cuh", "include/dtypes.cuh", "yalla_amd/csrc/core.hip",
                "yalla_amd/csrc/model_functors.h", "yalla_amd/csrc/Makefile"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main(src, dst):
    agg = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    cycles = m["GRBM_GUI_ACTIVE"] / 8
    rec = {
        "kernel": "ya::grid_force_bits<float3, spring, friction_w_neighbour>",
        "launches_averaged": len(agg["SQ_WAVES"]),
        "FETCH_SIZE_KiB": m["FETCH_SIZE"],
        "WRITE_SIZE_KiB": m["WRITE_SIZE"],
        "kernel_cycles": cycles,
        "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
        "salu_insts_per_wave": m["SQ_INSTS_SALU"] / m["SQ_WAVES"],
        "lds_insts_per_wave": m["SQ_INSTS_LDS"] / m["SQ_WAVES"],
        "valu_issue_frac": m["SQ_INSTS_VALU"] * 2 / (1024 * cycles),
        "valu_rate_frac": m["SQ_INSTS_VALU"] * SUSTAINED_CYCLES_PER_VALU / (1024 * cycles),
        "lanes_active_frac": m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]),
        "lds_conflict_frac": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"],
        "lds_busy_frac": m["SQ_LDS_IDX_ACTIVE"] / (256 * cycles),
        "wait_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        "kernel_sha": kernel_source_sha(),
        "head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                               text=True).stdout.strip() or None,
        "command": "rocprofv3 --kernel-trace --pmc <pass> -- python3 bench.py --no-cpu-baseline --steps 10 "
                   "--warmup 2 (tools/gpu_profile_round.sh; FETCH_SIZE and WRITE_SIZE in passes of their own)",
    }
    json.dump({"grid_force_1M_springs": rec}, open(dst, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
