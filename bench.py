#!/usr/bin/env python3
"""bench.py -- cell-updates/s of Solution<float3, Grid_solver>::take_step<spring>
on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one take_step (two force evaluations + two grid builds + the Heun
update) over every cell of a random_sphere(0.5, seed 42) system with the
`spring` functor of examples/springs.cu clipped by the grid cut-off
(cube_size 1), friction_w_neighbour, dt = 0.001.  One cell-update = one cell
advanced by one take_step.  N = 1 runs the 1 M-cell configuration the metric is
quoted on.  Inputs are resident in HBM before the timed region; nothing inside
it touches the host except take_step's own 4-byte read of n.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# Algorithmic bytes, float3 points: P = 12, V = 12, I = 4.
FORCE_BYTES_PER_CELL = 12 + 12 + 2 * 4 + 12   # force kernel: r P + V + 2I, w P
STEP_BYTES_PER_CELL = 13 * 12 + 108           # whole take_step: 13 P + 3 V + 18 I = 264


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=1_000_000, help="cells per GPU")
    ap.add_argument("--grid-size", type=int, default=0, help="0 = smallest that fits")
    ap.add_argument("--dist", type=float, default=0.5, help="random_sphere spacing")
    ap.add_argument("--cpu-steps", type=int, default=2, help="oracle steps for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="springs_grid",
                    help="named model of the harness (default: the headline springs_grid)")
    ap.add_argument("--dt", type=float, default=0.001)
    ap.add_argument("--migrate-every", type=int, default=16,
                    help="slab path: hand over cells that left their slab every this many steps")
    ap.add_argument("--slab", action="store_true",
                    help="use the z-slab path (ghost exchange + all-reduce) even on 1 GPU")
    ap.add_argument("--force-variant", type=int, default=1,
                    help="1 = LDS-staged grid_force (default), 0 = grid_force_direct (A/B)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 / --slab: nccl (= RCCL), or gloo with "
                         "YALLA_BENCH_DEVICE=0 to rehearse the N-rank path on one GPU (RCCL refuses "
                         "two ranks per GPU; messages are then staged through the host)")
    ap.add_argument("--time-every", type=int, default=5,
                    help="attach HIP events to every this-many-th force-kernel launch (odd: both stages)")
    ap.add_argument("--sorted-pipeline", type=int, default=1,
                    help="1 = second Heun stage built from the sorted cells (default), 0 = from d_X1 (A/B)")
    return ap.parse_args()


def measured_traffic(kernel_key):
    """HBM-side bytes per launch of the dominant kernel from the committed
    rocprofv3 PMC passes of this same command (profiles/r01_traffic.json; FETCH_SIZE
    and WRITE_SIZE collected in separate --pmc passes, in KiB, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)[kernel_key]
        return (2 * rec["FETCH_SIZE_KiB"] + rec["WRITE_SIZE_KiB"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def grid_size_for(n, dist):
    """Smallest even grid that keeps a random_sphere(dist) of n cells, radius
    (n/0.64)^(1/3) dist/2, two cubes inside the border (cube_size 1)."""
    radius = (n / 0.64) ** (1.0 / 3) * dist / 2
    gs = 2 * (int(radius) + 3)
    return max(gs, 8)


def cpu_baseline(n, gs, dist, steps):
    """The oracle (oracle/, a plain C++ port of the reference's algorithm, one
    thread) timed on the same workload: `steps` take_steps of the same system."""
    from yalla_amd import _ffi
    from yalla_amd.solution import Solution

    path = os.path.join(ROOT, "oracle", "_build", "liboracle_models.so")
    lib = _ffi.bind(path)
    with Solution("springs_grid", n, gs, 1.0, lib=lib) as s:
        s.random_sphere(dist, 42)
        t0 = time.perf_counter()
        s.take_step(0.001, steps)
        dt = time.perf_counter() - t0
    return {
        "value": n * steps / dt,
        "unit": "cell-updates/s",
        "cores": 1,
        "host_cores": os.cpu_count(),
        "kind": "port",
        "sample": f"{steps} take_steps of the same {n}-cell system (oracle/yalla_host.hpp, g++ -O3, 1 thread)",
    }


def main():
    args = parse()
    # RCCL prints a version banner on stdout; the contract is ONE JSON line there.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "YALLA_BENCH_DEVICE" in os.environ:  # testing only: several ranks on one GPU
        local_rank = int(os.environ["YALLA_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.slab:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from yalla_amd.solution import Solution

    n = args.cells            # cells per GPU (weak scaling: the system grows with N)
    n_total = n * world
    gs = args.grid_size or grid_size_for(n_total, args.dist)
    dt = args.dt

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1 and not args.slab:
        sim = Solution(args.model, n, gs, 1.0)
        sim.random_sphere(args.dist, 42)
        if "grid" in args.model:
            sim.set_param("force_variant", args.force_variant)
            sim.set_param("sorted_pipeline", args.sorted_pipeline)
        if args.model.startswith("sorting"):
            sim.set_param("n_cells", n)

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
        bounds = slab_mod.slab_bounds(X0[:, 2], world)
        my_slab = slab_mod.Slab("springs_grid", X0, rank, world, bounds, gs, cube_size=1.0,
                                device=f"cuda:{local_rank}")
        del X0
        sim = my_slab.sim
        sim.set_param("force_variant", args.force_variant)
        comm = slab_mod.DistComm()

        step_no = [0]

        def advance(k):
            # cells move ~1e-2 per step here; the ghost layer tolerates 0.25 of stray
            for _ in range(k):
                step_no[0] += 1
                slab_mod.step([my_slab], comm, dt, migrate=step_no[0] % args.migrate_every == 0)

    advance(args.warmup)
    barrier()
    # HIP events on every 5th launch of the force kernel (both stages alternate):
    # a timed launch costs a few microseconds of stream time, see DESIGN.md section 6
    sim.profile(True, every=args.time_every)
    t0 = time.perf_counter()
    advance(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    force_ms, launches = sim.profile_read()
    sim.profile(False)
    if world == 1 and not args.slab:
        assert sim.get_d_n() == n
        n_force = n
    else:
        slab_mod.step([my_slab], comm, dt, migrate=True)  # untimed: settle ownership, then count
        n_force = my_slab.n_own()   # cells a force launch computes (ghost cells get none)
        counts = torch.tensor([n_force], dtype=torch.int64,
                              device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(counts)
        assert int(counts.item()) == n_total, "cells were lost or duplicated in migration"

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_cells = n * world
        value = total_cells * args.steps / elapsed
        force_s = force_ms / 1e3 / max(launches, 1)
        achieved = n_force * FORCE_BYTES_PER_CELL / force_s / 1e9
        out = {
            "metric": "cell-updates/sec at 1M cells (Grid_solver)",
            "value": value,
            "unit": "cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: random_sphere(%g) seed 42, glibc rand()" % args.dist,
            "config": {
                "workload": "Solution<float3, Grid_solver>::take_step<spring> (examples/springs.cu "
                            "functor, friction_w_neighbour), dt 0.001",
                "cells_per_gpu": n,
                "total_cells": total_cells,
                "grid_size": gs,
                "cube_size": 1.0,
                "parallelism": "1 GPU" if world == 1 else
                               f"{world} z-slabs of one {n_total}-cell system, ghost exchange via "
                               + ("RCCL send/recv" if args.backend == "nccl" else
                                  f"{args.backend} send/recv staged through the host (rehearsal mode)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ya::grid_force<float3, spring, friction_w_neighbour>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic("grid_force_1M_springs") if world == 1 and n == 1_000_000 else None,
                "bytes_per_launch": n_force * FORCE_BYTES_PER_CELL,
                "avg_launch_us": force_s * 1e6,
                "timed_launches": launches,
                "launches": 2 * args.steps,
                "whole_step_achieved_GBs": STEP_BYTES_PER_CELL * value / world / 1e9,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, gs, args.dist, args.cpu_steps)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1 or args.slab:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
