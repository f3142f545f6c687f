#!/usr/bin/env python3
"""bench.py -- cell-updates/s of Solution<float3, Grid_solver>::take_step<spring>
on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one take_step (two force evaluations + two grid builds + the Heun
update) over every cell of a random_sphere(0.5, seed 42) system with the
`spring` functor of examples/springs.cu clipped by the grid cut-off
(cube_size 1), friction_w_neighbour, dt = 0.001.  One cell-update = one cell
advanced by one take_step.

  N = 1   the 1 M-cell configuration the metric is quoted on (the headline line;
          --cells-total 10000000 gives the 1-GPU 10 M point).
  N > 1   north_star's multi-GPU configuration: ONE 10 M-cell system cut into N
          z-slabs, one slab per GPU, ghost layers exchanged point-to-point with
          RCCL send/recv each Heun stage ("scaling": "strong": total work fixed).
          The base of that strong-scaling curve is NOT the N = 1 line (a 1 M-cell
          system: 1.5e9 c-u/s) but the same 10 M-cell system on one GPU (1.9e9):
          rank 0 measures it on its own GPU after the timed region and the line
          carries it as `one_gpu_same_system` with `speedup_vs_one_gpu_same_system`.
          Started either by torch.distributed.run (RANK / WORLD_SIZE in the
          environment) or by this script itself: without WORLD_SIZE it starts the
          N rank processes (before anything touches a GPU) and relays rank 0's line.

Inputs are resident in HBM before the timed region; nothing inside it touches the
host except take_step's own 4-byte read of n (and, on the slab path, one 8-byte
read of the two ghost counts per step).  Prints ONE JSON line on rank 0 (see
DESIGN.md "Measurement").
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_VALU_PEAK_TFLOPS = 157.3
CUS, SIMDS_PER_CU = 256, 4
MULTI_GPU_CELLS = 10_000_000   # north_star: "10 M cells ... across 8 x MI355X"

# The named models of the harness that bench.py can time: floats per point, the
# kernel the roofline object describes, and the workload label.
MODELS = {
    "springs_grid": (3, "ya::grid_force_bits<float3, spring, friction_w_neighbour>",
                     "Solution<float3, Grid_solver>::take_step<spring> (examples/springs.cu functor, "
                     "friction_w_neighbour)"),
    "springs_links_grid": (3, "ya::grid_force_bits<float3, spring, friction_w_neighbour>",
                           "Solution<float3, Grid_solver>::take_step<spring> + link_forces"),
    "clipped_grid": (3, "ya::grid_force_bits<float3, clipped_spring, friction_w_neighbour>",
                     "Solution<float3, Grid_solver>::take_step<clipped_spring> (tests/test_solvers.cu)"),
    "relu_grid": (3, "ya::grid_force_bits<float3, relu_force, friction_w_neighbour>",
                  "Solution<float3, Grid_solver>::take_step<relu_force> (inits.cuh)"),
    "sorting_grid": (3, "ya::grid_force_bits<float3, differential_adhesion, friction_w_neighbour>",
                     "Solution<float3, Grid_solver>::take_step<differential_adhesion> (examples/sorting.cu)"),
    "relu_po_grid": (5, "ya::grid_force_bits<Po_cell, relu_force, friction_w_neighbour>",
                     "Solution<Po_cell, Grid_solver>::take_step<relu_force>"),
    "relu_cell_grid": (7, "ya::grid_force_bits<Cell, relu_force, friction_w_neighbour>",
                       "Solution<Cell, Grid_solver>::take_step<relu_force> (examples/branching.cu point type)"),
    "springs_tile": (3, "ya::tile_force<float3, spring, friction_w_neighbour>",
                     "Solution<float3, Tile_solver>::take_step<spring> (examples/springs.cu)"),
    # BASELINE configs 4 and 3 at their stated sizes (yalla_amd/cases.py grows / sets them up, or
    # --state loads a saved system); division frozen during the timed steps
    "passive_growth_grid": (5, "ya::grid_force_bits<Po_cell, relu_w_epithelium, friction_w_neighbour>",
                            "Solution<Po_cell, Grid_solver>::take_step<relu_w_epithelium> + reset_nbs "
                            "(examples/passive_growth.cu grown from 200 cells)"),
    "branching_grid": (7, "ya::grid_force_bits<Cell, epi_turing_mes_noturing, friction_w_neighbour>",
                       "Solution<Cell, Grid_solver>::take_step<epi_turing_mes_noturing> + reset_nbs "
                       "(examples/branching.cu, 7-float cells)"),
}
STATE_MODELS = ("passive_growth_grid", "branching_grid")
# models bench.py can cut into z-slabs: float3 points, nothing indexed by cell id but the cell itself
# (sorting_grid's functor reads the GLOBAL id: `i < n_cells / 2`, examples/sorting.cu:24)
SLAB_MODELS = ("springs_grid", "clipped_grid", "relu_grid", "sorting_grid")


STATELESS_MODELS = ("springs_grid", "clipped_grid", "sorting_grid", "relu_grid", "relu_po_grid", "relu_cell_grid",
                    "branching_grid", "springs_links_grid")   # YA_STATELESS in yalla_amd/csrc/model_functors.h


def force_kernel_label(default_name, variant, n, model="", sum_order=0):
    """The force kernel a launch of n cells goes to (Grid_computer::forces, ya::coop::lanes_for)."""
    if variant < 0:
        variant = 3 if model in STATELESS_MODELS else 2
    if variant == 2 or "grid_force_bits" not in default_name:
        return default_name
    if variant == 3:
        lanes = 16 if n <= 15000 else 8 if n <= 40000 else 4 if n <= (70000 if sum_order else 120000) else 1
        if lanes == 1:
            return default_name
        return default_name.replace("grid_force_bits<", "grid_force_coop<").replace(">", f", {lanes} lanes per cell>")
    return default_name.replace("grid_force_bits", {0: "grid_force_direct", 1: "grid_force"}.get(variant, f"force_variant_{variant}"))


def force_bytes_per_cell(n_floats):
    """Algorithmic bytes of one force launch per cell (SURVEY.md §8(d) force row):
    r P + V + 2I, w P with P = 4 n_floats, V = 12, I = 4."""
    return 2 * 4 * n_floats + 12 + 2 * 4


def step_bytes_per_cell(n_floats):
    """Whole take_step: 13 P + 3 V + 18 I; 264 B for float3."""
    return 13 * 4 * n_floats + 108


def gather_model_bytes_per_cell_update(n_floats, dist):
    """SURVEY.md §8(d)'s second figure: what the reference's kernel structure requests,
    2 stages x [27 rho (P + V + I) + 27 * 2I] + streaming ~= 2 * 27 rho (P + 16) + 450,
    rho = cells per unit cube of random_sphere(dist) = 0.64 * 6 / (pi dist^3)."""
    rho = 0.64 * 6.0 / (math.pi * dist ** 3)
    return 2 * 27 * rho * (4 * n_floats + 16) + 450


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells-total", type=int, default=0,
                    help="cells of the whole system (default: 1 M on one GPU, 10 M on several)")
    ap.add_argument("--cells", type=int, default=0, help="same as --cells-total (kept for scripts)")
    ap.add_argument("--grid-size", type=int, default=0, help="0 = smallest that fits")
    ap.add_argument("--dist", type=float, default=0.5, help="random_sphere spacing")
    ap.add_argument("--cpu-steps", type=int, default=4, help="oracle steps for cpu_baseline (4 steps of 1 M cells: ~12 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="N > 1: skip rank 0's untimed run of the same whole system on one GPU "
                         "(the base of the strong-scaling curve)")
    ap.add_argument("--model", default="springs_grid",
                    help="named model of the harness (default: the headline springs_grid)")
    ap.add_argument("--dt", type=float, default=None, help="default: 0.001 (springs), the model's own 0.2 for configs 3 / 4")
    ap.add_argument("--state", default="", help="configs 3 / 4: .npz written by tools/make_state.py instead of setting "
                                                "the system up here (rocprofv3 runs then see the steady state only)")
    ap.add_argument("--migrate-every", type=int, default=8,
                    help="slab path: hand over cells that left their slab (and re-select the mirrored cells) every "
                         "this many steps; cells must not drift further than an eighth of a cube in between")
    ap.add_argument("--slab", action="store_true",
                    help="use the z-slab path (ghost exchange + all-reduce) even on 1 GPU")
    ap.add_argument("--tail-tiles", type=int, default=-1,
                    help="grid_force_bits under --sum-order 1: the last this-many tiles of a launch as two half-tile "
                         "workgroups each (-1 = the engine's choice: 768 for launches of 6144 tiles or more, 0 = none); "
                         "an A/B knob, results do not depend on it")
    ap.add_argument("--sum-order", type=int, default=0, choices=[0, 1],
                    help="Grid_computer::sum_order: 0 = the reference's one running sum per cell (default, "
                         "bit-comparable with the oracle's default), 1 = own z-plane | other planes (opt-in; the only "
                         "order in which --tail-tiles / half-tile workgroups exist)")
    ap.add_argument("--preheat-ms", type=float, default=400.0,
                    help="before the warm-up steps: this many milliseconds of dt = 0 steps on a SCRATCH copy of the "
                         "system (never on the system that is timed), so that the timed region does not start on an "
                         "idle GPU's clock ramp; 0 = none.  Reported as preheat_ms")
    ap.add_argument("--no-tail-ab-line", action="store_true",
                    help="headline run: skip the extra untimed pass with --sum-order 1 (reported as tail_ab)")
    ap.add_argument("--force-variant", type=int, default=-1,
                    help="-1 = the engine's choice (default: grid_force_bits, or grid_force_coop below ~1.5e5 cells "
                         "when the model declared its functors stateless), 2 = grid_force_bits always, "
                         "1 = grid_force (byte FIFO), 0 = grid_force_direct (A/B), 3 = grid_force_coop")
    ap.add_argument("--arith", default="exact", choices=["exact", "fast"],
                    help="arithmetic tier: exact (default; libyalla_models.so: IEEE binary32 statement by "
                         "statement, bit-comparable with the oracle) or fast (libyalla_models_fast.so: "
                         "contracted multiply-adds, bare v_sqrt_f32 / v_rcp_f32; within 1e-5 of exact)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 / --slab: nccl (= RCCL), or gloo with "
                         "YALLA_BENCH_DEVICE=0 to rehearse the N-rank path on one GPU (RCCL refuses "
                         "two ranks per GPU; messages are then staged through the host)")
    ap.add_argument("--time-every", type=int, default=5,
                    help="attach HIP events to every this-many-th force-kernel launch (odd: both stages)")
    ap.add_argument("--sorted-pipeline", type=int, default=1,
                    help="1 = second Heun stage built from the sorted cells (default), 0 = from d_X1 (A/B)")
    ap.add_argument("--link-order", choices=["by-first", "shuffled", "by-min"], default="by-first",
                    help="springs_links_grid: order of the link array (default: as generated, by first cell)")
    ap.add_argument("--links-per-cell", type=int, default=3,
                    help="springs_links_grid: links from every cell to this many nearest neighbours "
                         "(protrusion-like load for Links::link_forces)")
    ap.add_argument("--tile-lanes", type=int, default=1,
                    help="Tile_solver models: Tile_computer::lanes_per_cell (1 = reference semantics, 16 = tile_force_coop)")
    ap.add_argument("--renumber-every", type=int, default=0,
                    help="configs 3 / 4: the model calls Solution::renumber(type, mes_nbs, epi_nbs) -- new cell ids in "
                         "cube order, opt-in, not in the reference -- once before the warm-up and after every this-many-th "
                         "step (0 = never: ids stay in birth order as in the reference)")
    ap.add_argument("--sustained", action="store_true",
                    help="a SUSTAINED figure instead of the 20-step headline: --steps (default 3000, > 1 s) take_steps of "
                         "the same system with dt = 0, i.e. every step recomputes the same state (forces, both grid "
                         "builds, reductions, updates: all the work, none of the drift of the clumping springs system), "
                         "with the shader clock sampled beside it (ya_shader_clock_mhz) so that throttling shows")
    ap.add_argument("--no-sustained-line", action="store_true",
                    help="headline run: skip the extra pass, outside the timed region, of --sustained-steps take_steps "
                         "with dt = 0 (reported as `sustained` beside the headline value)")
    ap.add_argument("--sustained-steps", type=int, default=3000)
    ap.add_argument("--no-fast-tier-line", action="store_true",
                    help="headline run: skip the extra, untimed-by-the-contract pass on the fast-arithmetic build "
                         "(reported as fast_arith_tier beside the headline value)")
    ap.add_argument("--graph", type=int, default=0,
                    help="Heun_solver::graph_steps: 1 = replay the step as a hipGraph, -1 = below 400 k "
                         "cells only, 0 = plain launches (default)")
    args = ap.parse_args(argv)
    if args.sustained:
        if args.steps == 20:
            args.steps = 3000
        if args.dt is None:
            args.dt = 0.0
    if args.model not in MODELS:
        ap.error(f"--model must be one of {sorted(MODELS)}")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def kernel_source_sha():
    """Identifies the force kernel's source: the counter file under profiles/ records the
    value it was measured at, so a stale file is detected on the GPU box (no .git there)."""
    h = hashlib.sha256()
    for rel in ("include/solvers.cuh", "include/dtypes.cuh", "yalla_amd/csrc/core.hip",
                "yalla_amd/csrc/model_functors.h", "yalla_amd/csrc/Makefile"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def counters_key(args, n_total):
    """Key of this workload in profiles/r06_counters.json (None: no counters kept for it)."""
    if args.slab or args.gpus > 1 or args.force_variant not in (-1, 2) or args.sum_order != 0:
        return None   # (the counters on file were taken on the default kernel choice and summation order)
    tier = "" if args.arith == "exact" else "_fast"
    if args.model == "springs_grid" and args.dist == 0.5 and n_total in (1_000_000, 10_000_000):
        return ("springs_1M" if n_total == 1_000_000 else "springs_10M") + tier
    if args.model == "sorting_grid" and n_total == 10_000:
        return "cfg2_sorting_10k" + tier
    renumbered = "_renumbered" if args.renumber_every > 0 else ""
    if renumbered and args.model not in STATE_MODELS:
        return None
    if args.model == "branching_grid":
        return "cfg3_branching_100k" + renumbered + tier
    if args.model == "passive_growth_grid":
        return "cfg4_passive_growth_1M" + renumbered + tier
    return None


def measured_counters(kernel_key):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 passes of
    this same command (profiles/r06_counters.json, written by tools/roofline_json.py):
    HBM-side traffic (FETCH_SIZE and WRITE_SIZE from separate --pmc passes, KiB, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950) and the
    ceilings that bind this kernel.  Returns ({}, None) when there is no record."""
    path = os.path.join(ROOT, "profiles", "r06_counters.json")
    try:
        with open(path) as f:
            rec = json.load(f)[kernel_key]
    except (OSError, KeyError, ValueError, TypeError):
        return {}, None
    out = {k: rec.get(k) for k in ("valu_issue_frac", "lanes_active_frac", "fp32_lane_util", "valu_ns_per_inst",
                                   "valu_rate_frac", "valu_rate_probe", "clock_ghz", "lds_conflict_frac",
                                   "lds_busy_frac", "wait_frac", "valu_insts_per_wave")}
    out["traffic"] = (2 * rec["FETCH_SIZE_KiB"] + rec["WRITE_SIZE_KiB"]) * 1024.0
    head = {"commit": rec.get("head"), "kernel_sha": rec.get("kernel_sha")}
    if rec.get("kernel_sha") != kernel_source_sha():
        # measured on an older kernel: do not pass the numbers off as this build's
        sys.stderr.write("bench.py: profiles/r06_counters.json[%s] was measured on another kernel source "
                         "(%s != %s); traffic and PMC fractions omitted\n" % (kernel_key, rec.get("kernel_sha"), kernel_source_sha()))
        return {"stale_counters": True}, head
    return out, head


def springs_drift_note(warmup, steps, dist):
    """The springs workload is not stationary: what the committed measurement
    (tools/diag/springs_drift.py -> profiles/r04_springs_drift.json) says about it, or a pointer to the
    script if that file is missing or empty."""
    path = os.path.join(ROOT, "profiles", "r04_springs_drift.json")
    try:
        with open(path) as f:
            rows = json.load(f)["rows"]
        pairs = ", ".join("%g after %d steps" % (r["pairs_inside_cutoff_per_cell"], r["steps_taken"]) for r in rows[:6])
        evidence = "pairs inside the cut-off per cell: %s (profiles/r04_springs_drift.json)" % pairs
    except (OSError, ValueError, KeyError, TypeError):
        evidence = "no measurement on file: run tools/diag/springs_drift.py"
    return ("value is for steps %d..%d of a run started from random_sphere(%g); the springs system clumps as it "
            "runs and a cell-update costs proportionally more -- %s" % (warmup + 1, warmup + steps, dist, evidence))


def grid_size_for(n, dist):
    """Smallest even grid that keeps a random_sphere(dist) of n cells, radius
    (n/0.64)^(1/3) dist/2, two cubes inside the border (cube_size 1)."""
    radius = (n / 0.64) ** (1.0 / 3) * dist / 2
    gs = 2 * (int(radius) + 3)
    return max(gs, 8)


def cpu_baseline(model, n, gs, dist, dt, steps, state=None, renumber_every=0):
    """The oracle (oracle/, a plain C++ port of the reference's algorithm, one
    thread) timed on the same workload: `steps` take_steps of the same system."""
    from yalla_amd import _ffi, cases
    from yalla_amd.solution import Solution

    path = os.path.join(ROOT, "oracle", "_build", "liboracle_models.so")
    lib = _ffi.bind(path)
    with (cases.from_state(state, lib) if state is not None else Solution(model, n, gs, 1.0, lib=lib)) as s:
        if state is None:
            s.random_sphere(dist, 42)
        if model.startswith("sorting"):
            s.set_param("n_cells", n)
        if renumber_every > 0:
            s.set_param("renumber_now", 1)
            s.set_param("renumber_every", renumber_every)
        t0 = time.perf_counter()
        s.take_step(dt, steps)
        elapsed = time.perf_counter() - t0
    return {
        "value": n * steps / elapsed,
        "unit": "cell-updates/s",
        "cores": 1,
        "host_cores": os.cpu_count(),
        "kind": "port",
        "sample": f"{steps} take_steps of the same {n}-cell {model} system, dt {dt:g} "
                  "(oracle/yalla_host.hpp, g++ -O3, 1 thread)",
    }


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N rank processes
    ourselves (children of this process, which never touches a GPU), relay rank 0's
    JSON line, fail if any rank fails."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr))
    # rank 0's stdout is one JSON line (small: no pipe can fill up); poll all ranks so that one
    # that dies takes the others -- blocked in a collective -- down with it instead of hanging
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        time.sleep(0.2)
    if failed:
        time.sleep(2.0)  # let the others report their own error first
        for p in procs:
            if p.poll() is None:
                p.kill()   # the exact processes we started
    line = procs[0].stdout.read()
    codes = [p.wait() for p in procs]
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    if any(codes):
        sys.exit(f"bench.py: rank exit codes {codes}")


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)

    # RCCL prints a version banner on stdout; the contract is ONE JSON line there.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "YALLA_BENCH_DEVICE" in os.environ:  # testing only: several ranks on one GPU
        local_rank = int(os.environ["YALLA_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("YALLA_BENCH_TEST_HANG_RANK") == str(rank):  # testing only: a rank stuck in a collective
        time.sleep(600)
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}\n")
        args.gpus = world
    slab_path = world > 1 or args.slab
    n_floats, kernel_name, workload = MODELS[args.model]
    if slab_path:
        ignored = []
        if args.model not in SLAB_MODELS:
            sys.exit("bench.py: the z-slab path runs %s (float3 models without per-cell model arrays or links: "
                     "those would have to be replicated / migrated by the model); drop --model or run on one GPU"
                     % ", ".join(SLAB_MODELS))
        if args.sorted_pipeline != 1:
            ignored.append("--sorted-pipeline")
        if args.graph == 1:
            ignored.append("--graph")
        if ignored:
            sys.exit(f"bench.py: {', '.join(ignored)} not available on the z-slab path")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants GPU {local_rank}, but only {torch.cuda.device_count()} "
                 "are visible (one process per GPU)")
    torch.cuda.set_device(local_rank)
    native_rccl = slab_path and args.backend == "nccl"
    use_torch_dist = slab_path and not native_rccl
    if use_torch_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from yalla_amd import _ffi
    from yalla_amd.solution import Solution as _Solution

    engine = _ffi.device_lib(args.arith)

    def Solution(*a, **kw):   # every system of this run on the chosen arithmetic tier
        kw.setdefault("lib", engine)
        return _Solution(*a, **kw)

    n_total = args.cells_total or args.cells or (1_000_000 if world == 1 else MULTI_GPU_CELLS)
    gs = args.grid_size or grid_size_for(n_total, args.dist)
    dt = args.dt if args.dt is not None else (0.2 if args.model in STATE_MODELS else 0.001)
    state = None
    if args.model in STATE_MODELS:
        # configs 4 / 3: the system at its stated size, set up (grown) here or loaded
        from yalla_amd import cases
        if args.state:
            state = cases.load_state(args.state)
            assert str(state["model"]) == args.model, "the saved state is another model's"
        elif args.model == "passive_growth_grid":
            state = cases.config4_state(engine, args.cells_total or args.cells or 1_000_000)
        else:
            state = cases.config3_state(engine, args.cells_total or args.cells or 100_000)
        n_total, gs = int(state["n"]), int(state["grid_size"])

    native_comm = None
    if native_rccl:
        from yalla_amd import slab as slab_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL communicator of libyalla_hip.so; its unique id travels through the rendezvous
        # store the launcher already serves on MASTER_PORT (the engine's own TCP hand-over on
        # MASTER_PORT + 1 if that store cannot be reached)
        try:
            native_comm = slab_mod.NativeComm.over_store(rank, world)
        except (RuntimeError, ValueError, OSError) as err:
            print(f"bench.py: rank {rank}: no rendezvous store ({err}); id over TCP", file=sys.stderr)
            native_comm = slab_mod.NativeComm(port_offset=1)
        assert (native_comm.rank, native_comm.world) == (rank, world)

    # What RCCL itself says about the job (ya_comm_info -> ncclCommCount / ncclCommUserRank /
    # ncclCommCuDevice + the PCI location of that device), gathered over the communicator: the line
    # carries it as "rccl", and a job whose communicator does not span --gpus ranks on as many
    # distinct GPUs stops here instead of reporting a number for something else.
    rccl_facts = None
    if native_comm is not None and world > 1:
        gathered = native_comm.gather_info()
        devices = [g["pci_bus_id"] for g in gathered]
        rccl_facts = {"ranks": gathered[0]["ranks"], "user_ranks": [g["rank"] for g in gathered],
                      "devices": devices, "hip_devices": [g["device"] for g in gathered],
                      "how": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice of every rank's communicator, "
                             "all-gathered over it (libyalla_hip.so: ya_comm_info)"}
        problems = []
        if any(g["ranks"] != args.gpus for g in gathered) or len(gathered) != args.gpus:
            problems.append(f"ncclCommCount says {[g['ranks'] for g in gathered]}, --gpus is {args.gpus}")
        if [g["rank"] for g in gathered] != list(range(world)):
            problems.append(f"ncclCommUserRank says {[g['rank'] for g in gathered]}")
        if len(set(devices)) != world and "YALLA_BENCH_DEVICE" not in os.environ:
            problems.append(f"ranks share a GPU: {devices}")
        if problems:
            sys.exit("bench.py: the RCCL communicator is not what --gpus promises: " + "; ".join(problems))
    elif slab_path and world > 1:
        rccl_facts = {"ranks": None, "devices": None,
                      "how": f"rehearsal mode: messages over {args.backend}, no RCCL communicator"}

    def barrier():
        torch.cuda.synchronize()
        if native_comm is not None:
            native_comm.barrier()
        elif world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_over_ranks(value, take_max=False):
        if native_comm is not None:
            return native_comm.allreduce_host([float(value)], take_max)[0]
        if world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX if take_max else dist.ReduceOp.SUM)
        return float(t.item())

    n_links = 0
    link_pairs = None

    def make_sim(lib=None, sum_order=None):
        """The undivided system of this run, ready for its first step (called again for the untimed
        passes that repeat the timed steps: force-kernel events, the fast tier)."""
        nonlocal n_links, link_pairs
        sum_order = args.sum_order if sum_order is None else sum_order
        if state is not None:
            sim = cases.from_state(state, lib or engine)
        else:
            sim = Solution(args.model, n_total, gs, 1.0, lib=lib or engine)
            sim.random_sphere(args.dist, 42)
        if "grid" in args.model:
            sim.set_param("force_variant", args.force_variant)
            if sum_order:
                sim.set_param("sum_order", sum_order)
            if args.tail_tiles != -1:
                sim.set_param("tail_tiles", args.tail_tiles)
            sim.set_param("sorted_pipeline", args.sorted_pipeline)
            if args.graph != 0:
                sim.set_param("graph", args.graph)
        if args.model.startswith("sorting"):
            sim.set_param("n_cells", n_total)
        if args.renumber_every > 0:
            if args.model not in STATE_MODELS and args.model not in ("springs_grid", "clipped_grid", "relu_grid"):
                sys.exit("bench.py: --renumber-every is for the models whose harness hands every id-indexed array "
                         "over (passive_growth_grid, branching_grid) or that have none (springs_grid, clipped_grid, relu_grid)")
            sim.set_param("renumber_now", 1)
            sim.set_param("renumber_every", args.renumber_every)
        if args.model.endswith("_tile") and args.tile_lanes != 1:
            sim.set_param("tile_lanes", args.tile_lanes)
        if args.model == "springs_links_grid" and args.links_per_cell > 0:
            if link_pairs is None:
                # every cell linked to its k nearest neighbours (as the protrusions of
                # examples/intercalation.cu:32-68 link nearby cells), fixed for the run
                import numpy as np
                from scipy.spatial import cKDTree
                X = sim.h_X[:n_total].copy()
                _, idx = cKDTree(X).query(X, k=args.links_per_cell + 1)
                pairs = np.stack([np.repeat(np.arange(n_total), args.links_per_cell),
                                  idx[:, 1:].reshape(-1)], axis=1).astype(np.int32)
                if args.link_order == "shuffled":     # as a model's protrusions are: no order at all
                    pairs = pairs[np.random.default_rng(7).permutation(len(pairs))]
                elif args.link_order == "by-min":     # sorted by the smaller cell id of the pair
                    pairs = pairs[np.argsort(pairs.min(axis=1), kind="stable")]
                link_pairs = pairs
            sim.set_links(link_pairs, 0.2)
            n_links = len(link_pairs)
        return sim

    if not slab_path:
        sim = make_sim()

        def advance(k):
            sim.take_step(dt, k)
    else:
        # ONE system of n_total cells, cut into z-slabs, one slab per GPU, ghost
        # layers exchanged point-to-point over xGMI each stage (yalla_amd/slab.py).
        # Every rank regenerates the same initial state from the seed (host-side
        # glibc rand(), as inits.cuh does) and keeps its own slab.
        from yalla_amd import slab as slab_mod

        with Solution("springs_tile", n_total) as whole:
            whole.random_sphere(args.dist, 42)
            X0 = whole.h_X[:n_total].copy()
        my_slab = slab_mod.Slab(args.model, X0, rank, world, gs, cube_size=1.0, lib=engine,
                                global_ids=args.model == "sorting_grid")  # the others only compare i with j
        del X0
        sim = my_slab.sim
        sim.set_param("force_variant", args.force_variant)
        if args.sum_order:
            sim.set_param("sum_order", args.sum_order)
        if args.tail_tiles != -1:
            sim.set_param("tail_tiles", args.tail_tiles)
        if args.model == "sorting_grid":
            sim.set_param("n_cells", n_total)   # types split at the GLOBAL id n / 2
        if native_rccl:
            my_slab.use(comm=native_comm)
        else:
            my_slab.use(transport=slab_mod.CallbackTransport(device_memory=True))

        step_no = [0]

        def slab_step(migrate):
            my_slab.step(dt, migrate)

        def advance(k):
            # cells move ~1e-2 per step here; between migrations they may drift an eighth of a cube
            for _ in range(k):
                step_no[0] += 1
                slab_step(step_no[0] % args.migrate_every == 0)

    def preheat(make_scratch):
        """dt = 0 steps on a SCRATCH system for --preheat-ms, so that what follows does not start on the clock
        ramp of an idle GPU (the device sat idle while the host generated the system).  Returns the scratch
        system (closed by the caller AFTER its pass: freeing syncs the device) and what was done."""
        if args.preheat_ms <= 0 or args.sustained:
            return None, {"ms": 0.0, "steps": 0}
        scratch = make_scratch()
        scratch.take_step(0.0, 2)   # (allocations, first launches)
        scratch.synchronize()
        t = time.perf_counter()
        steps = 0
        while (time.perf_counter() - t) * 1e3 < args.preheat_ms:
            scratch.take_step(0.0, 25)
            steps += 25
            scratch.synchronize()
        return scratch, {"ms": (time.perf_counter() - t) * 1e3, "steps": steps}

    def slab_scratch():
        n_s = max(n_total // world, 1000)
        scratch = Solution("springs_grid", n_s, grid_size_for(n_s, args.dist), 1.0)
        scratch.random_sphere(args.dist, 42)
        return scratch

    def headline_scratch(sum_order=None):
        """The scratch copy of the headline run steps the SAME positions with the clipped_spring functor (the same
        force for every pair inside the cut-off, another template instantiation): rocprofv3's per-kernel averages of
        `grid_force_bits<float3, spring, ...>` then hold the warm-up and measured launches only, not the preheat's."""
        scratch = Solution("clipped_grid", n_total, gs, 1.0)
        scratch.random_sphere(args.dist, 42)
        scratch.set_param("force_variant", args.force_variant)
        if (args.sum_order if sum_order is None else sum_order):
            scratch.set_param("sum_order", 1)
        return scratch

    springs_headline = args.model == "springs_grid" and state is None and not slab_path
    scratch, preheat_facts = preheat(slab_scratch if slab_path else headline_scratch if springs_headline else make_sim)
    advance(args.warmup)
    barrier()
    # The engine replays small systems' steps as a hipGraph (Heun_solver::graph_steps), which
    # per-launch events would switch off.
    graph_mode = (not slab_path and "grid" in args.model and
                  (args.graph == 1 or (args.graph == -1 and n_total < 400_000)))
    clock_samples = []
    sampler = None
    stop_sampling = None

    def start_clock_sampler(every_s=0.05):
        """The shader clock beside a run: a wavefront of its own on a stream of its own, every 50 ms (2 ms beside
        the headline's 12 ms events pass; ya_shader_clock_mhz probes the device that is current for the calling
        thread)."""
        import ctypes as C
        import threading
        core = C.CDLL(_ffi.CORE_LIB, mode=C.RTLD_LOCAL)
        core.ya_shader_clock_mhz.argtypes = [C.c_double, C.POINTER(C.c_double)]
        stop = threading.Event()
        samples = []

        def sample_clock():
            torch.cuda.set_device(local_rank)   # a new thread starts on device 0
            mhz = C.c_double()
            while not stop.is_set():
                if core.ya_shader_clock_mhz(200.0, C.byref(mhz)) == 0:
                    samples.append((time.perf_counter(), mhz.value))
                stop.wait(every_s)

        thread = threading.Thread(target=sample_clock, daemon=True)
        thread.start()
        return thread, stop, samples

    def clock_summary(samples):
        cs = sorted(s[1] if isinstance(s, tuple) else s for s in samples)
        return {"samples": len(cs), "min": cs[0] if cs else None, "median": cs[len(cs) // 2] if cs else None,
                "max": cs[-1] if cs else None,
                "how": "ya_shader_clock_mhz: s_memtime against the 100 MHz wall clock, one wavefront beside the run, "
                       "every 50 ms (2 ms beside the headline's events pass)"}

    if args.sustained and rank == 0:
        sampler, stop_sampling, clock_samples = start_clock_sampler()
    # ---- the timed region: K steps, nothing else (no events, no host work but take_step's own) ----
    t0 = time.perf_counter()
    advance(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if sampler is not None:
        stop_sampling.set()
        sampler.join()
    # The force kernel's launch time: HIP events attached to the kernel's own dispatch on the stream it
    # is launched on -- in a SECOND, untimed pass.  On one GPU that pass repeats the timed region on a
    # fresh copy of the system (same seed, same warm-up, the same K steps: the launches measured are
    # the launches that were timed; events on every launch cost ~24 us of stream time per step, which
    # is why they are not inside the timed region any more).  On the z-slab path and in graph mode it
    # continues from the state the timed region left.
    if scratch is not None:
        scratch.close()
    headline_clock = None

    def events_pass_on_fresh_copy(sum_order=None):
        """The timed region repeated on a fresh copy of the system -- same seed, same preheat, same warm-up, the
        same K steps -- with events on every force launch and the shader clock sampled beside it.  Events and
        sampler are switched on BEFORE the preheat (their first use allocates: events, a stream, pinned memory), so
        that nothing but the K steps happens between the warm-up and the end of the pass; only what falls into the
        K steps is reported."""
        nonlocal sim
        sim.close()
        sim = make_sim(sum_order=sum_order)
        thread, stop, samples = start_clock_sampler(0.002)
        sim.profile(True, every=1)
        scratch, _ = preheat((lambda: headline_scratch(sum_order)) if springs_headline else (lambda: make_sim(sum_order=sum_order)))
        advance(args.warmup)
        barrier()
        sim.profile_read()   # (the warm-up's launches: dropped)
        t1 = time.perf_counter()
        advance(args.steps)
        barrier()
        t2 = time.perf_counter()
        stop.set()
        thread.join()
        ms, count = sim.profile_read()
        sim.profile(False)
        if scratch is not None:
            scratch.close()
        return ms, count, t2 - t1, clock_summary([mhz for t, mhz in samples if t1 <= t <= t2])

    tail_ab = None
    if not slab_path and not graph_mode and not args.sustained:
        force_ms, launches, events_seconds, headline_clock = events_pass_on_fresh_copy()
        events_pass = "the timed steps repeated on a fresh copy of the system, events on every force launch"
        if (rank == 0 and world == 1 and args.model in STATELESS_MODELS and "grid" in args.model and args.sum_order == 0
                and args.force_variant in (-1, 2) and state is None and not args.no_tail_ab_line):
            # A/B the driver can see: the same pass with Grid_computer::sum_order = YA_SUM_BY_PLANE, i.e. with the
            # tail of half-tile workgroups the engine then chooses (round 5's default; opt-in since round 6)
            ab_ms, ab_launches, ab_seconds, _ = events_pass_on_fresh_copy(sum_order=1)
            tail_ab = {
                "force_us_reference_order_whole_tiles": force_ms / max(launches, 1) * 1e3,
                "force_us_by_plane_half_tile_tail": ab_ms / max(ab_launches, 1) * 1e3,
                "ms_per_step_reference_order": events_seconds / args.steps * 1e3,
                "ms_per_step_by_plane": ab_seconds / args.steps * 1e3,
                "tail_tiles": args.tail_tiles,
                "what": "the timed steps repeated twice on fresh copies with events on every force launch: "
                        "--sum-order 0 (the default: the reference's one running sum per cell, whole tiles) and "
                        "--sum-order 1 (own plane | other planes, the last tiles of a launch as half-tile workgroups: "
                        "768 of 15625 at 10^6 cells); the ms_per_step figures include ~24 us of event overhead per step"}
    else:
        sim.profile(True, every=1 if graph_mode else args.time_every)
        advance(min(args.steps, 10 if graph_mode or slab_path else 20))
        barrier()
        events_pass = "steps after the timed region, from the state it left"
        force_ms, launches = sim.profile_read()
        sim.profile(False)
    if not slab_path:
        assert sim.get_d_n() == n_total
        n_force = n_total
    else:
        slab_step(True)  # untimed: settle ownership, then count
        n_force = my_slab.n_own()   # cells a force launch computes (ghost cells get none)
        assert int(reduce_over_ranks(n_force)) == n_total, "cells were lost or duplicated in migration"

    elapsed = reduce_over_ranks(elapsed, take_max=True)

    # The base of the strong-scaling curve: the SAME whole system, undivided, on one GPU (rank
    # 0's), same step counts, measured here and now -- outside the timed region.
    one_gpu = None
    if slab_path and world > 1 and not args.no_one_gpu_reference:
        if rank == 0:
            my_slab.close()
            with Solution(args.model, n_total, gs, 1.0) as whole:
                whole.random_sphere(args.dist, 42)
                whole.set_param("force_variant", args.force_variant)
                if args.model == "sorting_grid":
                    whole.set_param("n_cells", n_total)
                whole.take_step(dt, args.warmup)
                whole.synchronize()
                t1 = time.perf_counter()
                whole.take_step(dt, args.steps)
                whole.synchronize()
                one_gpu = n_total * args.steps / (time.perf_counter() - t1)
        barrier()

    if rank == 0:
        value = n_total * args.steps / elapsed
        force_s = force_ms / 1e3 / max(launches, 1)
        split_force = slab_path and world > 1
        if split_force:
            # a slab's stage is TWO force launches (tiles next to the faces, then all others on a stream
            # of their own), each timed by itself and each computing only its share of the own cells:
            # the stage's force time is at most their sum (they overlap), which is what is priced here
            force_s *= 2
        force_bytes = force_bytes_per_cell(n_floats)
        achieved = n_force * force_bytes / force_s / 1e9 if launches else None
        key = counters_key(args, n_total)
        counters, counters_head = measured_counters(key) if key else ({}, None)
        gather_bytes = gather_model_bytes_per_cell_update(n_floats, args.dist)
        out = {
            "metric": "cell-updates/sec at 1M cells (Grid_solver)",
            "value": value,
            "unit": "cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            # before the W warm-up steps: dt = 0 steps on a scratch copy (never the timed system), see --preheat-ms
            "preheat_ms": preheat_facts["ms"],
            "preheat_steps_on_scratch_copy": preheat_facts["steps"],
            # N > 1: one 10 M-cell system over N GPUs (total work fixed).  N = 1 is the 1 M-cell
            # headline and NOT the base of that curve: see one_gpu_same_system on the N > 1 lines.
            "scaling": "strong",
            "scaling_note": ("single-GPU headline (1 M cells); the N > 1 lines run ONE %d-cell system and carry "
                             "their own 1-GPU base (one_gpu_same_system)" % MULTI_GPU_CELLS) if world == 1 else
                            "strong scaling of one %d-cell system; base = one_gpu_same_system, not the N = 1 line" % n_total,
            "vs_baseline": None,
            # the springs workload is not stationary: see DESIGN.md section 5
            **({"workload_note": springs_drift_note(args.warmup, args.steps, args.dist)}
               if args.model.startswith("springs") and state is None else {}),
            "dtype": "f32",
            "data": ("synthetic: random_sphere(%g) seed 42, glibc rand()" % args.dist) if state is None else
                    ("synthetic: the model's own set-up (yalla_amd/cases.py)" +
                     (", grown from 200 cells in %d steps" % int(state["growth_steps"]) if int(state["growth_steps"]) else "")),
            "config": {
                "workload": f"{workload}, dt {dt:g}",
                "model": args.model,
                "cells_per_gpu": n_total // world,
                "total_cells": n_total,
                "grid_size": gs,
                "cube_size": 1.0,
                "step_replayed_as_hipgraph": bool(graph_mode),
                "force_variant": args.force_variant,
                "sum_order": "reference (one running sum per cell)" if args.sum_order == 0 else "by plane (opt-in)",
                "arith": args.arith,
                "links": n_links if not slab_path else 0,
                "renumber_every": args.renumber_every,
                "parallelism": "1 GPU" if world == 1 else
                               f"{world} z-slabs of one {n_total}-cell system, step sequenced in C++ (ya_slab_step: "
                               "mirrored ghost cells, one message of right-hand sides per neighbour and stage beside the "
                               "interior tiles' forces), messages via "
                               + ("RCCL send/recv (libyalla_hip.so's communicator)" if native_rccl else
                                  f"{args.backend} send/recv staged through the host (rehearsal mode)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": force_kernel_label(kernel_name, args.force_variant, n_total // world, args.model, args.sum_order),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None,
                "traffic": counters.get("traffic"),
                "traffic_head": counters_head,
                "bytes_per_launch": n_force * force_bytes,
                "avg_launch_us": force_s * 1e6,
                **({"avg_launch_note": "sum of a stage's two launches (boundary tiles + all others, overlapping): an "
                                       "upper bound of the stage's force time, so `achieved` is a lower bound"}
                   if split_force else {}),
                "timed_launches": launches,
                "events_pass": events_pass,
                "launches": (4 if split_force else 2) * args.steps,
                "whole_step_achieved_GBs": step_bytes_per_cell(n_floats) * value / world / 1e9,
                # SURVEY.md §8(d)'s second, explicitly labelled figure: bytes the reference's
                # kernel structure requests (served here from LDS, not from HBM)
                "gather_model_GBs": gather_bytes * value / world / 1e9,
                "gather_model_bytes_per_cell_update": gather_bytes,
                # the ceilings that do bind this kernel (PMC passes under profiles/, same command):
                # valu_issue_frac at the ideal 2 cycles per wave64 VALU; fp32_lane_util = that times the
                # share of lanes active; valu_rate_frac against the rate tools/micro/valu_probe.hip
                # measured in shader cycles (valu_rate_probe holds it with the clock the chip ran at)
                "counters_key": key,
                "valu_issue_frac": counters.get("valu_issue_frac"),
                "lanes_active_frac": counters.get("lanes_active_frac"),
                "fp32_lane_util": counters.get("fp32_lane_util"),
                "valu_ns_per_inst": counters.get("valu_ns_per_inst"),
                "valu_rate_frac": counters.get("valu_rate_frac"),
                "valu_rate_probe": counters.get("valu_rate_probe"),
                "kernel_clock_ghz": counters.get("clock_ghz"),
                "lds_conflict_frac": counters.get("lds_conflict_frac"),
                "lds_busy_frac": counters.get("lds_busy_frac"),
                "wait_frac": counters.get("wait_frac"),
                "valu_insts_per_wave": counters.get("valu_insts_per_wave"),
                "fp32_valu_peak_TFLOPs": FP32_VALU_PEAK_TFLOPS,
            },
        }
        if args.sustained:
            out["sustained"] = {
                "seconds": elapsed,
                "what": "dt = %g: every take_step does all its work on the same state (no drift of the workload)" % dt,
                "shader_clock_mhz": clock_summary(clock_samples),
            }
        if headline_clock is not None:
            out["headline_clock_mhz"] = headline_clock   # sampled beside the events pass (a repeat of the timed steps)
        if tail_ab is not None:
            out["tail_ab"] = tail_ab
        if rccl_facts is not None:
            out["rccl"] = rccl_facts
        if world > 1:
            out["one_gpu_same_system"] = one_gpu
            out["speedup_vs_one_gpu_same_system"] = value / one_gpu if one_gpu else None
        if counters.get("stale_counters"):
            out["roofline"]["stale_counters"] = True
        headline = (world == 1 and not args.slab and not args.sustained and state is None
                    and args.model == "springs_grid" and args.arith == "exact" and args.renumber_every == 0)
        if headline and not args.no_sustained_line:
            # Beside the headline, never instead of it, and outside its timed region: the figure that does not
            # drift.  --sustained-steps take_steps with dt = 0 on a fresh copy of the system -- every step
            # recomputes the same rho ~ 9.8 state: both grid builds, both force launches, reductions, updates --
            # with the shader clock sampled beside the run, then a few more steps with events for the launch time.
            sim.close()
            sim = make_sim()
            sim.take_step(0.0, args.warmup)
            sim.synchronize()
            thread, stop, samples = start_clock_sampler()
            t1 = time.perf_counter()
            sim.take_step(0.0, args.sustained_steps)
            sim.synchronize()
            sustained_s = time.perf_counter() - t1
            stop.set()
            thread.join()
            sim.profile(True, every=1)
            sim.take_step(0.0, 20)
            sim.synchronize()
            s_ms, s_launches = sim.profile_read()
            sim.profile(False)
            out["sustained"] = {
                "value": n_total * args.sustained_steps / sustained_s, "unit": "cell-updates/s",
                "ms_per_step": sustained_s / args.sustained_steps * 1e3, "steps": args.sustained_steps,
                "seconds": sustained_s, "force_us": s_ms / max(s_launches, 1) * 1e3,
                "shader_clock_mhz": clock_summary(samples),
                "what": "dt = 0 on a fresh copy of the system: every take_step does all its work on the same state "
                        "(40.2 pairs inside the cut-off per cell, no clumping), long enough for the clock to settle; "
                        "`value` above is steps %d..%d of the moving system" % (args.warmup + 1, args.warmup + args.steps)}
        if headline and not args.no_fast_tier_line:
            # beside the headline, never instead of it: the same steps on the fast-arithmetic build of the same
            # sources (what nvcc's default contraction and norm3df give the reference's own CUDA build)
            sim.close()
            with make_sim(lib=_ffi.device_lib("fast")) as fast:
                scratch, _ = preheat(headline_scratch)   # (as the headline: not on an idle GPU's clock ramp)
                fast.take_step(dt, args.warmup)
                fast.synchronize()
                t1 = time.perf_counter()
                fast.take_step(dt, args.steps)
                fast.synchronize()
                fast_s = time.perf_counter() - t1
                if scratch is not None:
                    scratch.close()
            out["fast_arith_tier"] = {
                "value": n_total * args.steps / fast_s, "unit": "cell-updates/s", "ms_per_step": fast_s / args.steps * 1e3,
                "what": "libyalla_models_fast.so: the same sources with -DYA_ARITH_FAST -ffp-contract=fast (bare v_sqrt_f32 / "
                        "v_rcp_f32, contracted multiply-adds); positions within 1e-5 relative of the exact tier in lock-step, "
                        "cube ids and cell counts bit-exact (tests/test_fast_arith_gpu.py).  Not the headline: `value` above "
                        "is the exact tier, bit-comparable with the CPU oracle"}
        if world == 1 and not args.slab and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, n_total, gs, args.dist, dt, args.cpu_steps, state,
                                               args.renumber_every)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if native_comm is not None:
        native_comm.close()
    if use_torch_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
