"""z-slab decomposition (yalla_amd/slab.py): a system cut into slabs must evolve
like the undivided system.  CPU: the orchestration on the oracle backend, in
process (LocalComm) and across two gloo ranks (DistComm).  GPU: the device path
with several slabs on one GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from yalla_amd import slab as slab_mod
from yalla_amd.solution import Solution

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_run(lib, n, gs, dist, seed, dt, steps, tree=False):
    with Solution("springs_grid", n, gs, 1.0, lib=lib) as s:
        if lib.ya_models_is_device() == 0:
            s.set_reduce_order(1 if tree else 0)
        s.random_sphere(dist, seed)
        X0 = s.h_X[:n].copy()
        s.take_step(dt, steps)
        return X0, s.positions()


def slab_run(lib, X0, world, gs, dt, steps, device="cpu", migrate_every=1):
    bounds = slab_mod.slab_bounds(X0[:, 2], world)
    slabs = [slab_mod.Slab("springs_grid", X0, r, world, bounds, gs, lib=lib, device=device)
             for r in range(world)]
    comm = slab_mod.LocalComm()
    moved = 0
    owners0 = [set(s.own_cells()[0].tolist()) for s in slabs]
    for k in range(steps):
        slab_mod.step(slabs, comm, dt, migrate=(k + 1) % migrate_every == 0 or k == steps - 1)
    X = np.zeros_like(X0)
    seen = np.zeros(len(X0), bool)
    for r, s in enumerate(slabs):
        gid, Xr = s.own_cells()
        assert not seen[gid].any(), "a cell is owned twice"
        seen[gid] = True
        X[gid] = Xr
        moved += len(set(gid.tolist()) - owners0[r])
        s.close()
    assert seen.all(), "a cell was lost"
    return X, moved


def check(lib, n, world, steps, dt, device="cpu", migrate_every=1):
    X0, Xref = reference_run(lib, n, 50, 0.5, 3, dt, steps)
    X, moved = slab_run(lib, X0, world, 50, dt, steps, device, migrate_every)
    scale = np.abs(Xref).max()
    assert np.abs(X - Xref).max() <= 1e-5 * scale
    return moved


@pytest.mark.parametrize("world,n", [(1, 3000), (2, 3000), (3, 3000), (5, 9000)])
def test_slabs_match_undivided_system_oracle(oracle, world, n):
    check(oracle, n, world, 4, 0.002)


def test_cells_migrate_between_slabs_oracle(oracle):
    moved = check(oracle, 3000, 4, 12, 0.004)
    assert moved > 0, "test too gentle: nothing crossed a slab face"


def test_postponed_migration_oracle(oracle):
    """Migrating every 4th step only: strays stay inside the ghost margin."""
    moved = check(oracle, 3000, 3, 12, 0.002, migrate_every=4)
    assert moved >= 0


def test_slabs_thinner_than_the_ghost_layer_are_refused(oracle):
    X0, _ = reference_run(oracle, 300, 50, 0.5, 3, 0.001, 0)
    bounds = slab_mod.slab_bounds(X0[:, 2], 6)
    with pytest.raises(slab_mod.YallaError):
        slab_mod.Slab("springs_grid", X0, 2, 6, bounds, 50, lib=oracle)


def test_two_gloo_ranks_match_undivided_system(oracle, tmp_path):
    """DistComm over gloo, world_size 2, one process per rank (oracle backend)."""
    out = tmp_path / "slab_gloo.npz"
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(ROOT, "tests", "slab_worker.py"), str(out)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    got = np.load(out)
    X0, Xref = reference_run(oracle, 3000, 50, 0.5, 3, 0.003, 6)
    assert np.array_equal(got["X0"], X0)
    scale = np.abs(Xref).max()
    assert np.abs(got["X"] - Xref).max() <= 1e-5 * scale


@pytest.mark.gpu
def test_two_ranks_sharing_one_gpu_match_undivided_system(device, tmp_path):
    """The one-process-per-rank path on the device: DistComm, torch device buffers handed to
    the engine, world_size 2.  RCCL refuses two ranks on one GPU, so the transport is gloo
    with the messages staged through host memory; everything else is what bench.py runs."""
    out = tmp_path / "slab_gloo_device.npz"
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29612",
           os.path.join(ROOT, "tests", "slab_worker.py"), str(out), "device"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    got = np.load(out)
    X0, Xref = reference_run(device, 40000, 50, 0.5, 3, 0.003, 6)
    assert np.array_equal(got["X0"], X0)
    scale = np.abs(Xref).max()
    assert np.abs(got["X"] - Xref).max() <= 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_slabs_match_undivided_system_device(device, world):
    moved = check(device, 40000, world, 6, 0.004, device="hip")
    assert moved >= 0


@pytest.mark.gpu
def test_postponed_migration_device(device):
    """Cells that left their slab are handed over every 4th step only (what bench.py does)."""
    moved = check(device, 40000, 3, 12, 0.004, device="hip", migrate_every=4)
    assert moved >= 0
