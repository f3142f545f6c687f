#!/usr/bin/env python3
"""One entry of profiles/r06_counters.json from the rocprofv3 PMC passes of a bench.py command
(tools/gpu_profile_round.sh): per-launch means for the dominant kernel (grid_force_bits).

    python tools/roofline_json.py gpurun_out/<tag> <out.json> <key> "<bench args>" [commit]

(tools/gpu_profile_round.sh runs it on the GPU box; tools/merge_counters.py folds the entries into
profiles/r06_counters.json)

Definitions (every input is a raw counter kept in profiles/r06_pmc_<key>.txt):
  kernel_cycles       GRBM_GUI_ACTIVE / 8                 (the counter sums the 8 XCDs)
  clock_ghz           kernel_cycles / the kernel's average duration in the same pass
  valu_insts_per_wave SQ_INSTS_VALU / SQ_WAVES
  valu_issue_frac     SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * kernel_cycles)   (2 = ideal wave64 issue on SIMD-32)
  lanes_active_frac   SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  fp32_lane_util      valu_issue_frac * lanes_active_frac: share of the fp32 lanes' issue slots doing work
  valu_ns_per_inst    kernel duration * 1024 SIMDs / SQ_INSTS_VALU: wall time per wave64 VALU instruction and SIMD
  valu_rate_frac      probe_ns_per_inst / valu_ns_per_inst, where the probe figure is what tools/micro/valu_probe.hip
                      measured IN SHADER CYCLES (s_memtime) with 8 wavefronts per SIMD on the VOP2 mix of the distance
                      test (profiles/r03_valu_probe.jsonl: 2.23 cycles at the 2.21 GHz the chip held = 1.01 ns; a pure
                      v_fma_f32 stream: 3.29 cycles at 2.04 GHz = 1.61 ns).  The chip is power-limited under dense VALU
                      issue, so the sustained rate is quoted in ns with its measured clock, not derived from 2.4 GHz.
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
KERNEL = "grid_force"  # grid_force_bits, or grid_force_coop where the model declared its functor stateless
PROBE_VOP2_MIX_NS = 1.012   # profiles/r03_valu_probe.jsonl: distance_test_mix, 8 waves per SIMD
PROBE_VOP2_MIX_CYCLES, PROBE_VOP2_MIX_GHZ = 2.23, 2.207
PROBE_FMA_NS, PROBE_FMA_CYCLES, PROBE_FMA_GHZ = 1.613, 3.29, 2.037


def kernel_source_sha():
    h = hashlib.sha256()
    for rel in ("include/solvers.cuh", "include/dtypes.cuh", "yalla_amd/csrc/core.hip",
                "yalla_amd/csrc/model_functors.h", "yalla_amd/csrc/Makefile"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main(src, dst, key, bench_args, head=None):
    agg = collections.defaultdict(list)
    names = collections.Counter()
    durations = []
    for path in glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names[r["Kernel_Name"]] += 1
    for path in glob.glob(os.path.join(src, "pmc*", "*kernel_trace.csv")):
        if "GRBM_GUI_ACTIVE" not in open(path.replace("kernel_trace", "counter_collection")).read():
            continue
        for r in csv.DictReader(open(path)):
            if KERNEL in r["Kernel_Name"]:
                durations.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    cycles = m["GRBM_GUI_ACTIVE"] / 8
    duration_ns = sum(durations) / len(durations) if durations else None
    issue = m["SQ_INSTS_VALU"] * 2 / (1024 * cycles)
    lanes = m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"])
    ns_per_inst = duration_ns * 1024 / m["SQ_INSTS_VALU"] if duration_ns else None
    rec = {
        "kernel": names.most_common(1)[0][0][:160],
        "bench_args": bench_args,
        "launches_averaged": len(agg["SQ_WAVES"]),
        "FETCH_SIZE_KiB": m["FETCH_SIZE"],
        "WRITE_SIZE_KiB": m["WRITE_SIZE"],
        "kernel_cycles": cycles,
        "duration_us_in_counter_pass": duration_ns / 1e3 if duration_ns else None,
        "clock_ghz": cycles / duration_ns if duration_ns else None,
        "waves": m["SQ_WAVES"],
        "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
        "salu_insts_per_wave": m["SQ_INSTS_SALU"] / m["SQ_WAVES"],
        "lds_insts_per_wave": m["SQ_INSTS_LDS"] / m["SQ_WAVES"],
        "valu_issue_frac": issue,
        "lanes_active_frac": lanes,
        "fp32_lane_util": issue * lanes,
        "valu_ns_per_inst": ns_per_inst,
        "valu_rate_frac": PROBE_VOP2_MIX_NS / ns_per_inst if ns_per_inst else None,
        "valu_rate_probe": {"vop2_mix_ns_per_inst": PROBE_VOP2_MIX_NS, "vop2_mix_cycles": PROBE_VOP2_MIX_CYCLES,
                            "vop2_mix_clock_ghz": PROBE_VOP2_MIX_GHZ, "v_fma_f32_ns_per_inst": PROBE_FMA_NS,
                            "v_fma_f32_cycles": PROBE_FMA_CYCLES, "v_fma_f32_clock_ghz": PROBE_FMA_GHZ,
                            "source": "profiles/r03_valu_probe.jsonl (tools/micro/valu_probe.hip, s_memtime, 8 waves per SIMD)"},
        "lds_conflict_frac": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"],
        "lds_busy_frac": m["SQ_LDS_IDX_ACTIVE"] / (256 * cycles),
        "wait_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        "kernel_sha": kernel_source_sha(),
        "head": head or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                       text=True).stdout.strip() or None,
        "command": "rocprofv3 --kernel-trace --pmc <pass> -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 "
                   + bench_args + " (tools/gpu_profile_round.sh; FETCH_SIZE and WRITE_SIZE in passes of their own)",
    }
    try:
        every = json.load(open(dst))
    except (OSError, ValueError):
        every = {}
    every[key] = rec
    json.dump(every, open(dst, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "",
         sys.argv[5] if len(sys.argv) > 5 else None)
