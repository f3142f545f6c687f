"""bench.py contract pieces: what can be checked without a GPU, and (marked gpu) the
self-launched N-rank run the driver starts as `python bench.py --gpus N`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_bench_refuses_to_run_without_gpu():
    proc = subprocess.run([sys.executable, BENCH, "--steps", "1"],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0
    assert "needs a GPU" in (proc.stderr + proc.stdout)


def test_bench_starts_its_own_ranks():
    """No launcher, no WORLD_SIZE: --gpus 2 must start two rank processes itself (here
    both refuse for want of a GPU, and the parent reports both exit codes)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"],
                          capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode != 0
    assert proc.stderr.count("needs a GPU") == 2
    assert "rank exit codes [1, 1]" in proc.stderr
    assert proc.stdout.strip() == ""


def test_a_dead_rank_takes_the_others_down():
    """One rank fails while the other is stuck (as in a collective its peer never joins): the
    parent must end the stuck one and report, not hang until the driver's limit."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["YALLA_BENCH_TEST_HANG_RANK"] = "0"
    t0 = time.time()
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"],
                          capture_output=True, text=True, timeout=300, env=env)
    assert time.time() - t0 < 120
    assert proc.returncode != 0 and "rank exit codes [-9, 1]" in proc.stderr
    assert proc.stdout.strip() == ""


def test_grid_size_keeps_the_sphere_inside():
    sys.path.insert(0, ROOT)
    import bench
    for n, dist in ((1_000_000, 0.5), (8_000_000, 0.5), (10_000_000, 0.5), (1_000_000, 0.75)):
        gs = bench.grid_size_for(n, dist)
        radius = (n / 0.64) ** (1 / 3) * dist / 2
        assert gs // 2 - radius >= 2 and gs % 2 == 0
        assert gs <= 256, "cube ids are computed in binary32: exact only up to 256^3"


def test_defaults_follow_north_star():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.MULTI_GPU_CELLS == 10_000_000
    assert bench.step_bytes_per_cell(3) == 264 and bench.force_bytes_per_cell(3) == 44
    assert bench.step_bytes_per_cell(5) == 368 and bench.step_bytes_per_cell(7) == 472
    # ~16 KB (rho 9.8) and ~5 KB (rho 2.9) for float3
    assert 14e3 < bench.gather_model_bytes_per_cell_update(3, 0.5) < 17e3
    assert 4e3 < bench.gather_model_bytes_per_cell_update(3, 0.75) < 6e3
    a = bench.parse(["--gpus", "4"])
    assert a.gpus == 4 and a.cells_total == 0 and a.tail_tiles == -1   # the tail of half tiles: the engine's choice
    # which force kernel a launch goes to (labels of the bench line): several lanes per cell up to 7e4 cells
    name = "ya::grid_force_bits<float3, spring, friction_w_neighbour>"
    assert "16 lanes" in bench.force_kernel_label(name, -1, 10_000, "springs_grid")
    assert "4 lanes" in bench.force_kernel_label(name, -1, 70_000, "springs_grid")
    assert "4 lanes" in bench.force_kernel_label(name, -1, 100_000, "springs_grid")       # default summation order
    assert bench.force_kernel_label(name, -1, 100_000, "springs_grid", 1) == name          # by plane: half tiles instead
    assert bench.force_kernel_label(name, -1, 130_000, "springs_grid") == name


@pytest.mark.gpu
def test_self_launched_two_ranks_on_one_gpu(device):
    """`python bench.py --gpus 2` exactly as the driver starts it (no torchrun), both ranks
    on GPU 0 with gloo transport (RCCL refuses two ranks per GPU): ONE JSON line, n_gpus 2,
    strong scaling, the whole system accounted for."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["YALLA_BENCH_DEVICE"] = "0"
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "4",
                           "--warmup", "1", "--cells-total", "300000", "--migrate-every", "2"],
                          capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["total_cells"] == 300000 and out["config"]["cells_per_gpu"] == 150000
    assert out["value"] > 0 and out["unit"] == "cell-updates/s"
    assert out["roofline"]["achieved"] > 0 and "cpu_baseline" not in out
    # the base of the strong-scaling curve travels with the line: the same system on one GPU
    assert out["one_gpu_same_system"] > 0
    assert abs(out["speedup_vs_one_gpu_same_system"] - out["value"] / out["one_gpu_same_system"]) < 1e-9
    # no RCCL communicator in rehearsal mode, and the line says so
    assert out["rccl"]["ranks"] is None and "rehearsal" in out["rccl"]["how"]


@pytest.mark.gpu
def test_headline_line_carries_the_sustained_figure_and_untimed_events(device):
    """The default run's extras: `sustained` measured outside the timed region
    (dt = 0, here shortened), the force kernel's events taken in a second pass over the same steps."""
    proc = subprocess.run([sys.executable, BENCH, "--cells-total", "200000", "--steps", "5", "--warmup", "2",
                           "--sustained-steps", "200", "--no-cpu-baseline", "--no-fast-tier-line"],
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.strip()][-1])
    s = out["sustained"]
    assert s["steps"] == 200 and s["value"] > 0 and s["force_us"] > 0 and s["ms_per_step"] > 0
    assert s["shader_clock_mhz"]["samples"] >= 1 and 500 < s["shader_clock_mhz"]["median"] < 3000
    r = out["roofline"]
    assert r["timed_launches"] == 2 * 5 and "fresh copy" in r["events_pass"]
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


@pytest.mark.gpu
def test_slab_path_runs_the_id_indexed_sorting_model(device):
    """bench.py's z-slab path is not springs-only (missing 3): sorting_grid, whose functor reads the
    GLOBAL id of both cells, in two slabs on one GPU (gloo rehearsal transport)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["YALLA_BENCH_DEVICE"] = "0"
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--model", "sorting_grid",
                           "--steps", "4", "--warmup", "1", "--cells-total", "60000", "--dt", "0.002",
                           "--migrate-every", "2"], capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.strip()][-1])
    assert out["n_gpus"] == 2 and out["config"]["model"] == "sorting_grid" and out["value"] > 0
    assert out["one_gpu_same_system"] > 0
    refused = subprocess.run([sys.executable, BENCH, "--slab", "--model", "passive_growth_grid", "--steps", "1"],
                             capture_output=True, text=True, timeout=300)
    assert refused.returncode != 0 and "z-slab path runs" in refused.stderr
